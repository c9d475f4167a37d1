// Row-wise / gather kernels of the Point-SAM hot path for gfx950 (wave64): LayerNorm (+residual, +GELU), SwiGLU
// gate with inner LayerNorm, per-group max-pool, positional encodings, token assembly, 3-NN feature interpolation
// and the small prompt-token attentions of the two-way decoder.  All memory-bound: one wave per row, lane-consecutive
// (coalesced) accesses, shuffle reductions.
#include "common.h"
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// y = act(LayerNorm(x (+ res)))          nn.LayerNorm / apex FusedLayerNorm (pc_sam/utils/torch_utils.py:28-38)
// One wave per row; rows of <= 2816 columns are held in registers, longer rows are re-read (L2-resident).
// ------------------------------------------------------------------------------------------------
template <int NREG>  // NREG*64 >= cols for the register path; NREG == 0 -> streaming path
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ res, int64_t ldr,
                                                        const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y,
                                                        int64_t ldy, int64_t rows, int cols, float eps, int act, float* __restrict__ rscale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= rows) return;
    float amax = 0.f;
    const float* xr = x + row * ldx;
    const float* rr = res ? res + row * ldr : nullptr;
    float* yr = y + row * ldy;
    const float inv = 1.0f / (float)cols;
    if (NREG > 0) {
        float v[NREG > 0 ? NREG : 1];
#pragma unroll
        for (int i = 0; i < NREG; ++i) {   // unconditional loads (clamped index), see layernorm_v4_kernel
            const int c = i * 64 + lane;
            v[i] = xr[c < cols ? c : cols - 1];
        }
        if (rr) {
#pragma unroll
            for (int i = 0; i < NREG; ++i) {
                const int c = i * 64 + lane;
                v[i] += rr[c < cols ? c : cols - 1];
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            if (i * 64 + lane >= cols) v[i] = 0.f;
            s += v[i];
        }
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            const int c = i * 64 + lane;
            const float d = c < cols ? v[i] - mean : 0.f;
            q += d * d;
        }
        const float r = 1.0f / sqrtf(wave_sum(q) * inv + eps);
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            const int c = i * 64 + lane;
            if (c < cols) {
                float o = (v[i] - mean) * r * w[c] + b[c];
                if (act == 1) o = gelu_erf(o);
                yr[c] = o;
                amax = fmaxf(amax, fabsf(o));
            }
        }
    } else {
        float s = 0.f;
        for (int c = lane; c < cols; c += 64) { float t = xr[c]; if (rr) t += rr[c]; s += t; }
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
        for (int c = lane; c < cols; c += 64) { float t = xr[c]; if (rr) t += rr[c]; const float d = t - mean; q += d * d; }
        const float r = 1.0f / sqrtf(wave_sum(q) * inv + eps);
        for (int c = lane; c < cols; c += 64) {
            float t = xr[c]; if (rr) t += rr[c];
            float o = (t - mean) * r * w[c] + b[c];
            if (act == 1) o = gelu_erf(o);
            yr[c] = o;
            amax = fmaxf(amax, fabsf(o));
        }
    }
    if (rscale) { amax = wave_max(amax); if (lane == 0) rscale[row] = f16_row_scale(amax); }
}

// float4 variant: NV4*256 >= cols; rows must be 16-byte aligned with ld >= round_up(cols, 4) (the last, partial float4 of
// a row is read whole -- still inside the row's stride -- masked in the statistics and written back element-wise).
template <int NV4>
__global__ __launch_bounds__(256) void layernorm_v4_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ res, int64_t ldr,
                                                           const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ y,
                                                           int64_t ldy, int64_t rows, int cols, float eps, int act, float* __restrict__ rscale, int pack,
                                                           float* __restrict__ rbound, float bc2, float bc1, float bc0) {
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= rows) return;
    float amax = 0.f;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * ldx);
    const f32x4* rr = res ? reinterpret_cast<const f32x4*>(res + row * ldr) : nullptr;
    float* yrow = y + row * ldy;
    const int c4n = (cols + 3) >> 2;
    const float inv = 1.0f / (float)cols;
    // Loads are UNCONDITIONAL (index clamped to the last float4 of the row, value masked afterwards): a load under an
    // exec-mask branch makes the compiler wait vmcnt(0) at the join, which serialised the 11..16 loads of a row
    // (one full memory round trip each).
    f32x4 v[NV4];
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = i * 64 + lane;
        v[i] = xr[c < c4n ? c : c4n - 1];
    }
    if (rr) {
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c = i * 64 + lane;
            v[i] += rr[c < c4n ? c : c4n - 1];
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = i * 64 + lane;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (c * 4 + e >= cols) v[i][e] = 0.f;
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) * inv;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = i * 64 + lane;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = (c * 4 + e < cols) ? v[i][e] - mean : 0.f;
            q += d * d;
        }
    }
    const float r = 1.0f / sqrtf(wave_sum(q) * inv + eps);
    const int cfull = cols >> 2;    // float4 groups entirely inside the row (cols >= 256 on this path, so cfull >= 64)
    if (pack) {
        // g8-packed output for the f16x3p GEMM (gemm_f16x3p.hip): the outputs stay in registers until the row maximum (hence the row
        // scale) is known; then every group of 8 columns is written as [hi x8 | lo x8] fp16.  A lane holds 4 columns (hi: 8 B, lo: 8 B),
        // its neighbour lane ^ 1 the other 4 of the group: the even lane collects the 16-byte hi chunk, the odd lane the lo chunk (one
        // exchange), so consecutive lanes still write consecutive 16-byte chunks.  Columns cols .. round_up(cols, 32) - 1 (the GEMM's
        // K padding) are written as zeros when the row stride has room for them.
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c = i * 64 + lane;
            const int cl = c < cfull ? c : cfull - 1;
            f32x4 w4 = *reinterpret_cast<const f32x4*>(w + cl * 4), b4 = *reinterpret_cast<const f32x4*>(b + cl * 4);
            if (c == cfull && c * 4 < cols) {          // partial last group: element-wise weights
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int ce = c * 4 + e < cols ? c * 4 + e : cols - 1; w4[e] = w[ce]; b4[e] = b[ce]; }
            }
            f32x4 o = (v[i] - mean) * r * w4 + b4;
            if (act == 1) { o[0] = gelu_erf(o[0]); o[1] = gelu_erf(o[1]); o[2] = gelu_erf(o[2]); o[3] = gelu_erf(o[3]); }
#pragma unroll
            for (int e = 0; e < 4; ++e) if (c * 4 + e >= cols) o[e] = 0.f;
            v[i] = o;
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
        }
        if (rbound) {      // bc2 t^2 + bc1 t + bc0 with t = ||output row||_2: what the consuming GEMM's epilogue bounds ITS output rows with (psam_gemm_fuse_t.out_bound)
            float n2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV4; ++i) n2 += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
            const float t = sqrtf(wave_sum(n2)) * 1.0001f;
            if (lane == 0) rbound[row] = (bc2 * t + bc1) * t + bc0;
        }
        amax = wave_max(amax);
        const float sc = f16_row_scale(amax);
        if (lane == 0) rscale[row] = sc;
        typedef unsigned ln_u32x4 __attribute__((ext_vector_type(4)));
        const int kp = (cols + 31) & ~31;
        const int klim = kp <= ldy ? kp : (cols + 7) & ~7;      // columns to write (multiple of 8)
        const bool odd = lane & 1;
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c = i * 64 + lane;
            unsigned h0, l0, h1, l1;
            psam_split2_f16(v[i][0], v[i][1], sc, h0, l0);
            psam_split2_f16(v[i][2], v[i][3], sc, h1, l1);
            // even lane keeps its hi and receives the neighbour's hi; odd lane keeps its lo and receives the neighbour's lo
            const unsigned r0 = __shfl_xor(odd ? h0 : l0, 1, 64), r1 = __shfl_xor(odd ? h1 : l1, 1, 64);
            if (c * 4 < klim) *reinterpret_cast<ln_u32x4*>(yrow + c * 4) = odd ? ln_u32x4{r0, r1, l0, l1} : ln_u32x4{h0, h1, r0, r1};
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = i * 64 + lane;
        const int cl = c < cfull ? c : cfull - 1;    // unconditional (clamped) weight loads: no wait per group
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(w + cl * 4), b4 = *reinterpret_cast<const f32x4*>(b + cl * 4);
        if (c * 4 + 3 < cols) {
            f32x4 o = (v[i] - mean) * r * w4 + b4;
            if (act == 1) { o[0] = gelu_erf(o[0]); o[1] = gelu_erf(o[1]); o[2] = gelu_erf(o[2]); o[3] = gelu_erf(o[3]); }
            *reinterpret_cast<f32x4*>(yrow + c * 4) = o;
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
        } else if (c * 4 < cols) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (c * 4 + e < cols) {
                    float o = (v[i][e] - mean) * r * w[c * 4 + e] + b[c * 4 + e];
                    if (act == 1) o = gelu_erf(o);
                    yrow[c * 4 + e] = o;
                    amax = fmaxf(amax, fabsf(o));
                }
            }
        }
    }
    if (rscale) { amax = wave_max(amax); if (lane == 0) rscale[row] = f16_row_scale(amax); }
}

// row_scale (optional, [rows]): the f16x3 GEMM's power-of-two row scale of the OUTPUT rows (psam_row_scale_f16 fused in).
// pack != 0: y receives the g8-packed form of the row-scaled output (psam_pack_rows_f16x2_g8 fused in; needs row_scale and the
// float4 path: 256 <= cols <= 4096, 32-byte aligned output rows) -- the A operand of psam_gemm_f16x3p.
// row_bound (optional, packed output only, [rows]): c2 t^2 + c1 t + c0 with t = the L2 norm of the output row (times 1.0001) -- the a-priori
// bound of the rows of a GEMM that consumes this output, e.g. |W_n . h + b_n| <= ||W_n|| t + |b_n| (psam_gemm_fuse_t.out_bound).
PSAM_API int32_t psam_layernorm_ex2(const float* x, int64_t ldx, const float* res, int64_t ldr, const float* w, const float* b, float* y,
                                    int64_t ldy, int64_t rows, int32_t cols, float eps, int32_t act, float* row_scale, int32_t pack,
                                    float* row_bound, float c2, float c1, float c0, hipStream_t stream) {
    PSAM_REQUIRE(x && w && b && y, PSAM_EINVAL, "psam_layernorm: null pointer");
    PSAM_REQUIRE(!row_bound || pack, PSAM_EINVAL, "psam_layernorm: row_bound comes with the packed output");
    PSAM_REQUIRE(rows > 0 && cols > 0, PSAM_EINVAL, "psam_layernorm: bad shape");
    PSAM_REQUIRE(act == 0 || act == 1, PSAM_EINVAL, "psam_layernorm: act must be 0 or 1 (GELU)");
    const dim3 grid((unsigned)psam_cdiv(rows, 4)), block(256);
    const int64_t c4 = ((int64_t)cols + 3) & ~(int64_t)3;
    const bool vec = cols >= 256 && cols <= 4096 && ((ldx | ldy | (res ? ldr : 0)) & 3) == 0 && ldx >= c4 && ldy >= c4 && (!res || ldr >= c4) &&
                     (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res | (uintptr_t)w | (uintptr_t)b) & 15) == 0;
    PSAM_REQUIRE(!pack || (vec && row_scale && (ldy & 7) == 0 && ((uintptr_t)y & 31) == 0 && ldy >= ((cols + 7) & ~7)), PSAM_EINVAL,
                 "psam_layernorm: packed output needs row_scale, the float4 path (256 <= cols <= 4096, aligned) and 32-byte aligned output rows");
    if (vec) {
#define LNV_LAUNCH(R) hipLaunchKernelGGL(layernorm_v4_kernel<R>, grid, block, 0, stream, x, ldx, res, ldr, w, b, y, ldy, rows, cols, eps, act, row_scale, pack, row_bound, c2, c1, c0)
        const int span = pack ? ((cols + 31) & ~31) : cols;     // the packed form also writes the zero padding up to the 32-k slab
        if (span <= 256) LNV_LAUNCH(1);
        else if (span <= 512) LNV_LAUNCH(2);
        else if (span <= 1024) LNV_LAUNCH(4);
        else if (span <= 2048) LNV_LAUNCH(8);
        else LNV_LAUNCH(16);
        if ((psam_ablate_repeat() & 2) && x != y && res != y) {
            if (span <= 256) LNV_LAUNCH(1);
            else if (span <= 512) LNV_LAUNCH(2);
            else if (span <= 1024) LNV_LAUNCH(4);
            else if (span <= 2048) LNV_LAUNCH(8);
            else LNV_LAUNCH(16);
        }
#undef LNV_LAUNCH
        return psam_launch_status("psam_layernorm: launch failed");
    }
#define LN_LAUNCH(R) hipLaunchKernelGGL(layernorm_kernel<R>, grid, block, 0, stream, x, ldx, res, ldr, w, b, y, ldy, rows, cols, eps, act, row_scale)
    if (cols <= 128) LN_LAUNCH(2);
    else if (cols <= 256) LN_LAUNCH(4);
    else if (cols <= 512) LN_LAUNCH(8);
    else if (cols <= 1024) LN_LAUNCH(16);
    else if (cols <= 2816) LN_LAUNCH(44);
    else LN_LAUNCH(0);
#undef LN_LAUNCH
    return psam_launch_status("psam_layernorm: launch failed");
}

PSAM_API int32_t psam_layernorm_ex(const float* x, int64_t ldx, const float* res, int64_t ldr, const float* w, const float* b, float* y,
                                   int64_t ldy, int64_t rows, int32_t cols, float eps, int32_t act, float* row_scale, int32_t pack,
                                   hipStream_t stream) {
    return psam_layernorm_ex2(x, ldx, res, ldr, w, b, y, ldy, rows, cols, eps, act, row_scale, pack, nullptr, 0.f, 0.f, 0.f, stream);
}

PSAM_API int32_t psam_layernorm(const float* x, int64_t ldx, const float* res, int64_t ldr, const float* w, const float* b, float* y,
                                int64_t ldy, int64_t rows, int32_t cols, float eps, int32_t act, hipStream_t stream) {
    return psam_layernorm_ex(x, ldx, res, ldr, w, b, y, ldy, rows, cols, eps, act, nullptr, 0, stream);
}

PSAM_API int32_t psam_layernorm_rs(const float* x, int64_t ldx, const float* res, int64_t ldr, const float* w, const float* b, float* y,
                                   int64_t ldy, int64_t rows, int32_t cols, float eps, int32_t act, float* row_scale, hipStream_t stream) {
    return psam_layernorm_ex(x, ldx, res, ldr, w, b, y, ldy, rows, cols, eps, act, row_scale, 0, stream);
}

// ------------------------------------------------------------------------------------------------
// timm SwiGLU with inner norm (eva02 blocks): out = LayerNorm_H(SiLU(g) * x); g = gx[:, 0:H], x = gx[:, xoff:xoff+H].
// out has leading dimension ldo >= H; columns [H, ldo) are written as zeros (K padding of the following fc2 GEMM).
// ------------------------------------------------------------------------------------------------
template <int NREG>  // NREG*64 >= H: the gated row lives in registers (one read of gx, one write of out)
__global__ __launch_bounds__(256) void swiglu_ln_kernel(const float* __restrict__ gx, int64_t ldg, int xoff, const float* __restrict__ w,
                                                        const float* __restrict__ b, float* __restrict__ out, int64_t ldo, int64_t rows, int H,
                                                        float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (row >= rows) return;
    const float* g = gx + row * ldg;
    const float* x = g + xoff;
    float* o = out + row * ldo;
    const float inv = 1.0f / (float)H;
    if (NREG > 0) {
        float u[NREG > 0 ? NREG : 1];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            const int c = i * 64 + lane;
            const int cc = c < H ? c : H - 1;        // unconditional loads (clamped), see layernorm_v4_kernel
            const float gv = g[cc], xv = x[cc];
            u[i] = c < H ? silu(gv) * xv : 0.f;
            s += u[i];
        }
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            const float d = (i * 64 + lane) < H ? u[i] - mean : 0.f;
            q += d * d;
        }
        const float r = 1.0f / sqrtf(wave_sum(q) * inv + eps);
#pragma unroll
        for (int i = 0; i < NREG; ++i) {
            const int c = i * 64 + lane;
            if (c < H) o[c] = (u[i] - mean) * r * w[c] + b[c];
            else if (c < ldo) o[c] = 0.f;
        }
        for (int c = NREG * 64 + lane; c < ldo; c += 64) o[c] = 0.f;
    } else {
        float s = 0.f;
        for (int c = lane; c < H; c += 64) {
            const float u = silu(g[c]) * x[c];
            o[c] = u;  // same lane re-reads its own columns below
            s += u;
        }
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
        for (int c = lane; c < H; c += 64) { const float d = o[c] - mean; q += d * d; }
        const float r = 1.0f / sqrtf(wave_sum(q) * inv + eps);
        for (int c = lane; c < H; c += 64) o[c] = (o[c] - mean) * r * w[c] + b[c];
        for (int c = H + lane; c < ldo; c += 64) o[c] = 0.f;
    }
}

PSAM_API int32_t psam_swiglu_ln(const float* gx, int64_t ldg, int32_t xoff, const float* w, const float* b, float* out, int64_t ldo,
                                int64_t rows, int32_t H, float eps, hipStream_t stream) {
    PSAM_REQUIRE(gx && w && b && out, PSAM_EINVAL, "psam_swiglu_ln: null pointer");
    PSAM_REQUIRE(rows > 0 && H > 0 && xoff >= H && ldo >= H && ldg >= xoff + H, PSAM_EINVAL, "psam_swiglu_ln: bad shape");
    const dim3 grid((unsigned)psam_cdiv(rows, 4)), block(256);
#define SG_LAUNCH(R) hipLaunchKernelGGL(swiglu_ln_kernel<R>, grid, block, 0, stream, gx, ldg, xoff, w, b, out, ldo, rows, H, eps)
    if (H <= 8 * 64) SG_LAUNCH(8);
    else if (H <= 32 * 64) SG_LAUNCH(32);
    else if (H <= 44 * 64) SG_LAUNCH(44);
    else SG_LAUNCH(0);
#undef SG_LAUNCH
    return psam_launch_status("psam_swiglu_ln: launch failed");
}

// ------------------------------------------------------------------------------------------------
// Max-pool over the K members of every group: x [groups*K, C] -> y [groups, C]     (common.py:502,505)
// ------------------------------------------------------------------------------------------------
__global__ void group_max_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int64_t groups, int K, int C) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= groups * C) return;
    const int64_t g = t / C;
    const int c = (int)(t % C);
    const float* p = x + g * K * ldx + c;
    float m = p[0];
    for (int k = 1; k < K; ++k) m = fmaxf(m, p[(int64_t)k * ldx]);
    y[g * ldy + c] = m;
}

PSAM_API int32_t psam_group_max(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t groups, int32_t K, int32_t C, hipStream_t stream) {
    PSAM_REQUIRE(x && y, PSAM_EINVAL, "psam_group_max: null pointer");
    PSAM_REQUIRE(groups > 0 && K > 0 && C > 0, PSAM_EINVAL, "psam_group_max: bad shape");
    hipLaunchKernelGGL(group_max_kernel, dim3((unsigned)psam_cdiv(groups * C, 256)), dim3(256), 0, stream, x, ldx, y, ldy, groups, K, C);
    return psam_launch_status("psam_group_max: launch failed");
}

// ------------------------------------------------------------------------------------------------
// Voronoi variant (PointCloudSAMNN): per-point features relative to the point's nearest centre, and the max-pool of point rows into their cells.
//   mode 0, NNGrouper.forward (common.py:203-211):          out[b*N + n] = [ (xyz - c) / max(|xyz - c|, 1e-8) : 3 | |xyz - c| : 1 | feats[b, n, :C] ]
//   mode 1, MaskEncoderNN.forward (prompt_encoder.py:281-287): out[z*N + n] = [ logit[z, n] | xyz - c : 3 | |xyz - c| : 1 ],  z = b * rep + r
// rows are ldo floats apart and zero-padded behind the last channel (ldo % 4 == 0: the K of the Linear that follows).
// ------------------------------------------------------------------------------------------------
__global__ void nn_group_feats_kernel(const float* __restrict__ xyz, const float* __restrict__ centers, const int64_t* __restrict__ nn_idx,
                                      const float* __restrict__ feats, const float* __restrict__ logits, int B, int rep, int N, int G, int C, int mode,
                                      float* __restrict__ out, int ldo) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)B * rep * N) return;
    const int n = (int)(t % N);
    const int z = (int)(t / N), b = z / rep;
    const float* p = xyz + ((int64_t)b * N + n) * 3;
    const float* c = centers + ((int64_t)b * G + nn_idx[(int64_t)b * N + n]) * 3;
    const float dx = p[0] - c[0], dy = p[1] - c[1], dz = p[2] - c[2];
    const float dist = sqrtf((dx * dx + dy * dy) + dz * dz);
    float* o = out + t * ldo;
    int w = 0;
    if (mode == 0) {
        const float inv = fmaxf(dist, 1e-8f);
        o[0] = dx / inv; o[1] = dy / inv; o[2] = dz / inv; o[3] = dist;
        for (int k = 0; k < C; ++k) o[4 + k] = feats[((int64_t)b * N + n) * C + k];
        w = 4 + C;
    } else {
        o[0] = logits[(int64_t)z * N + n]; o[1] = dx; o[2] = dy; o[3] = dz; o[4] = dist;
        w = 5;
    }
    for (int k = w; k < ldo; ++k) o[k] = 0.f;
}

PSAM_API int32_t psam_nn_group_feats(const float* xyz, const float* centers, const int64_t* nn_idx, const float* feats, const float* logits, int32_t B,
                                     int32_t rep, int32_t N, int32_t G, int32_t C, int32_t mode, float* out, int64_t ldo, hipStream_t stream) {
    PSAM_REQUIRE(xyz && centers && nn_idx && out && B > 0 && rep > 0 && N > 0 && G > 0, PSAM_EINVAL, "psam_nn_group_feats: bad argument");
    PSAM_REQUIRE((mode == 0 && (C == 0 || feats) && rep == 1 && ldo >= 4 + C) || (mode == 1 && logits && ldo >= 5), PSAM_EINVAL,
                 "psam_nn_group_feats: mode 0 needs feats (rep 1), mode 1 needs logits; ldo must hold the row");
    hipLaunchKernelGGL(nn_group_feats_kernel, dim3((unsigned)psam_cdiv((int64_t)B * rep * N, 256)), dim3(256), 0, stream, xyz, centers, nn_idx, feats, logits,
                       B, rep, N, G, C, mode, out, (int)ldo);
    return psam_launch_status("psam_nn_group_feats: launch failed");
}

// out[dest(r), :] = max over the rows r that map to it of x[r, :]; dest(r) = idx[(r / rows_per_set / idx_rep) * rows_per_set + r % rows_per_set]
// + (r / rows_per_set) * set_stride.  The maximum is exact and order-independent, so integer atomics on the float pattern give the same
// bits in any arrival order: non-negative values by signed max, negative ones by unsigned min.  include_self != 0: the destination starts at
// 0 and takes part (torch.scatter_reduce(zeros, ..., "amax"), prompt_encoder.py:291-297); otherwise it starts at -inf and destinations that
// received nothing end as 0 (scatter_reduce_(..., include_self=False) into zeros, pc_encoder.py:190-193).
__global__ void scatter_fill_kernel(float* __restrict__ out, int64_t n, float v) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = v;
}
__global__ void scatter_amax_kernel(const float* __restrict__ x, int64_t ldx, const int64_t* __restrict__ idx, int64_t rows, int C, int64_t rows_per_set,
                                    int64_t set_stride, int idx_rep, float* __restrict__ out, int64_t out_rows) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * C) return;
    const int64_t r = t / C;
    const int c = (int)(t % C);
    const int64_t set = r / rows_per_set, within = r - set * rows_per_set;
    const int64_t dest = idx[(set / idx_rep) * rows_per_set + within] + set * set_stride;
    if (dest < 0 || dest >= out_rows) return;      // a bad index / set stride from the caller never writes outside `out`
    const float raw = x[r * ldx + c];
    float* o = out + dest * C + c;
    if (raw != raw) {      // NaN propagates (torch scatter_reduce "amax"): the canonical positive quiet NaN is the largest signed pattern any input can set
        atomicMax(reinterpret_cast<int*>(o), 0x7fc00000);
        return;
    }
    const float v = raw == 0.f ? 0.f : raw;      // -0.0 -> +0.0: as an integer pattern -0.0 is INT_MIN and would lose against every negative value
    // a destination already holding that NaN keeps it: signed max cannot go below it, and the unsigned min of a negative value (>= 0x80000000) neither
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(o), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned*>(o), __float_as_uint(v));
}
__global__ void scatter_fix_kernel(float* __restrict__ out, int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n && out[t] == -INFINITY) out[t] = 0.f;
}

PSAM_API int32_t psam_scatter_amax(const float* x, int64_t ldx, const int64_t* idx, int64_t rows, int32_t C, int64_t rows_per_set, int64_t set_stride,
                                   int32_t idx_rep, float* out, int64_t out_rows, int32_t include_self, hipStream_t stream) {
    PSAM_REQUIRE(x && idx && out && rows > 0 && C > 0 && rows_per_set > 0 && rows % rows_per_set == 0 && idx_rep > 0 && out_rows > 0 && ldx >= C, PSAM_EINVAL,
                 "psam_scatter_amax: bad argument");
    const int64_t n = out_rows * C;
    hipLaunchKernelGGL(scatter_fill_kernel, dim3((unsigned)psam_cdiv(n, 256)), dim3(256), 0, stream, out, n, include_self ? 0.f : -INFINITY);
    hipLaunchKernelGGL(scatter_amax_kernel, dim3((unsigned)psam_cdiv(rows * C, 256)), dim3(256), 0, stream, x, ldx, idx, rows, C, rows_per_set, set_stride, idx_rep, out, out_rows);
    if (!include_self) hipLaunchKernelGGL(scatter_fix_kernel, dim3((unsigned)psam_cdiv(n, 256)), dim3(256), 0, stream, out, n);
    return psam_launch_status("psam_scatter_amax: launch failed");
}

// ------------------------------------------------------------------------------------------------
// pos_embed first layer: y[r, 0:128] = GELU(W[128,3] @ centers[r] + bias)                (pc_encoder.py:102-104)
// ------------------------------------------------------------------------------------------------
__global__ void pos_l1_kernel(const float* __restrict__ c, const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ y,
                              int64_t rows) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * 128) return;
    const int64_t r = t >> 7;
    const int j = (int)(t & 127);
    const float* p = c + r * 3;
    y[t] = gelu_erf(fmaf(W[j * 3 + 2], p[2], fmaf(W[j * 3 + 1], p[1], fmaf(W[j * 3], p[0], bias[j]))));
}

PSAM_API int32_t psam_pos_l1(const float* centers, const float* W, const float* bias, float* y, int64_t rows, hipStream_t stream) {
    PSAM_REQUIRE(centers && W && bias && y && rows > 0, PSAM_EINVAL, "psam_pos_l1: bad argument");
    hipLaunchKernelGGL(pos_l1_kernel, dim3((unsigned)psam_cdiv(rows * 128, 256)), dim3(256), 0, stream, centers, W, bias, y, rows);
    return psam_launch_status("psam_pos_l1: launch failed");
}

// ------------------------------------------------------------------------------------------------
// Random-Fourier positional encoding (+ optional label embedding):                  (prompt_encoder.py:27-48,63-77)
//   e = [sin(2*pi*x@Gm), cos(2*pi*x@Gm)]  (+ emb0 if label==0, + emb1 if label==1)
// Row r is written to out + (r / rows_per_batch) * batch_stride + (r % rows_per_batch) * 2F (token assembly).
// *flag is OR-ed with 1 if any coordinate is outside [-1-1e-6, 1+1e-6] (the reference raises ValueError).
// ------------------------------------------------------------------------------------------------
__global__ void fourier_pe_kernel(const float* __restrict__ x, const float* __restrict__ Gm, int F, const int64_t* __restrict__ labels,
                                  const float* __restrict__ emb0, const float* __restrict__ emb1, float* __restrict__ out, int64_t rows,
                                  int rows_per_batch, int64_t batch_stride, int* __restrict__ flag) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * F) return;
    const int64_t r = t / F;
    const int j = (int)(t % F);
    const float px = x[r * 3], py = x[r * 3 + 1], pz = x[r * 3 + 2];
    if (j == 0 && flag) {
        const float lo = -1.0f - 1e-6f, hi = 1.0f + 1e-6f;
        if (px < lo || py < lo || pz < lo || px > hi || py > hi || pz > hi) atomicOr(flag, 1);
    }
    // (x @ Gm) accumulated in the order of a 3-term dot product, then * 2*pi in fp32 like the reference
    float v = px * Gm[j];
    v = fmaf(py, Gm[F + j], v);
    v = fmaf(pz, Gm[2 * F + j], v);
    v = 6.283185307179586f * v;
    float s = sinf(v), c = cosf(v);
    if (labels) {
        const int64_t lab = labels[r];
        if (lab == 0) { s += emb0[j]; c += emb0[F + j]; }
        else if (lab == 1) { s += emb1[j]; c += emb1[F + j]; }
    }
    float* o = out + (r / rows_per_batch) * batch_stride + (r % rows_per_batch) * (2 * F);
    o[j] = s;
    o[F + j] = c;
}

PSAM_API int32_t psam_fourier_pe(const float* coords, const float* gauss, int32_t F, const int64_t* labels, const float* emb0, const float* emb1,
                                 float* out, int64_t rows, int32_t rows_per_batch, int64_t batch_stride, int32_t* flag, hipStream_t stream) {
    PSAM_REQUIRE(coords && gauss && out && rows > 0 && F > 0 && rows_per_batch > 0, PSAM_EINVAL, "psam_fourier_pe: bad argument");
    PSAM_REQUIRE(!labels || (emb0 && emb1), PSAM_EINVAL, "psam_fourier_pe: labels need both embeddings");
    hipLaunchKernelGGL(fourier_pe_kernel, dim3((unsigned)psam_cdiv(rows * F, 256)), dim3(256), 0, stream, coords, gauss, F, labels, emb0, emb1,
                       out, rows, rows_per_batch, batch_stride, flag);
    return psam_launch_status("psam_fourier_pe: launch failed");
}

// ------------------------------------------------------------------------------------------------
// out[z, r, :] = a[(z / rep), r, :] + (b ? b[z*sb + r*ldb + :] : 0)      rows R, cols C (C % 4 == 0)
// Covers: src = repeat_interleave(pc_embeddings) + dense (mask_decoder.py:136-139), q = queries + query_pe,
// k = keys + key_pe (transformer.py:153-170), and plain broadcast copies (b == null).
// sb == 0 && ldb == 0 broadcasts one row vector (no_mask_embed, prompt_encoder.py:119-122).
// ------------------------------------------------------------------------------------------------
__global__ void add_bcast_kernel(const float* __restrict__ a, int64_t sa, int rep, const float* __restrict__ b, int64_t sb, int64_t ldb,
                                 float* __restrict__ out, int64_t so, int64_t Z, int64_t R, int C4) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= Z * R * C4) return;
    const int c = (int)(t % C4);
    const int64_t r = (t / C4) % R;
    const int64_t z = t / ((int64_t)C4 * R);
    f32x4 v = reinterpret_cast<const f32x4*>(a + (z / rep) * sa + r * C4 * 4)[c];
    if (b) v += reinterpret_cast<const f32x4*>(b + z * sb + r * ldb)[c];
    reinterpret_cast<f32x4*>(out + z * so + r * C4 * 4)[c] = v;
}

PSAM_API int32_t psam_add_bcast(const float* a, int64_t sa, int32_t rep, const float* b, int64_t sb, int64_t ldb, float* out, int64_t so,
                                int64_t Z, int64_t R, int32_t C, hipStream_t stream) {
    PSAM_REQUIRE(a && out && Z > 0 && R > 0 && C > 0 && rep > 0, PSAM_EINVAL, "psam_add_bcast: bad argument");
    PSAM_REQUIRE((C & 3) == 0 && (sa & 3) == 0 && (sb & 3) == 0 && (ldb & 3) == 0 && (so & 3) == 0 && ((uintptr_t)a & 15) == 0 &&
                     ((uintptr_t)out & 15) == 0 && ((uintptr_t)b & 15) == 0,
                 PSAM_EALIGN, "psam_add_bcast: needs C, strides multiples of 4 and 16-byte aligned pointers");
    hipLaunchKernelGGL(add_bcast_kernel, dim3((unsigned)psam_cdiv(Z * R * (C / 4), 256)), dim3(256), 0, stream, a, sa, rep, b, sb, ldb, out, so,
                       Z, R, C / 4);
    return psam_launch_status("psam_add_bcast: launch failed");
}

// ------------------------------------------------------------------------------------------------
// interpolate_features (common.py:258-274): out[z, n, :] = sum_k w3[b,n,k] * src[z, idx3[b,n,k], :], b = z / rep.
// One wave per point, float4 per lane (C == 256).
// ln_g != NULL (C == 256: the wave holds the whole row): LayerNorm(gamma, beta, eps) and the activation are applied to the interpolated
// row before it is written.  This is how the decoder's upscaling MLP starts (mask_decoder.py:53-59,146-160): its first Linear is applied
// to the G patch rows BEFORE the interpolation -- interpolation is an affine combination (weights sum to 1), so
// Linear(interp(x)) = interp(Linear(x)) -- and the LayerNorm + GELU that follow it ride on the interpolation kernel: the [N, 256]
// GEMM and the separate LayerNorm pass over [N, 256] disappear.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void interp3_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx3, const float* __restrict__ w3,
                                                      float* __restrict__ out, int rep, int64_t Z, int N, int G, int C, float* __restrict__ scale_out,
                                                      const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps, int act) {
    const int lane = threadIdx.x & 63;
    const int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wv >= Z * N) return;
    const int64_t z = wv / N, n = wv % N, b = z / rep;
    const int64_t o = (b * N + n) * 3;
    const int64_t i0 = idx3[o], i1 = idx3[o + 1], i2 = idx3[o + 2];
    const float w0 = w3[o], w1 = w3[o + 1], w2 = w3[o + 2];
    const float* s = src + z * (int64_t)G * C;
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(s + i0 * C + c);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(s + i1 * C + c);
        const f32x4 cc = *reinterpret_cast<const f32x4*>(s + i2 * C + c);
        // same association as (x*w).sum(-2): ((a*w0) + b*w1) + c*w2
        f32x4 v = a * w0;
        v = v + bb * w1;
        v = v + cc * w2;
        if (ln_g) {      // C == 256 (host-checked): two-pass LayerNorm over the row held by this wave, then the activation
            const float mean = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 256.0f);
            const f32x4 d = v - mean;
            const float rstd = 1.0f / sqrtf(wave_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 256.0f) + ln_eps);
            v = d * rstd * *reinterpret_cast<const f32x4*>(ln_g + c) + *reinterpret_cast<const f32x4*>(ln_b + c);
            if (act == 1) v = f32x4{gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])};
            else if (act == 2) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
        }
        if (scale_out) {      // C == 256 (host-checked): the whole row is in this wave -> g8-packed row + its scale (gemm_f16x3p.hip)
            const float sc = f16_row_scale(wave_max(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])))));
            if (lane == 0) scale_out[z * N + n] = sc;
            unsigned h0, l0, h1, l1;
            psam_split2_f16(v[0], v[1], sc, h0, l0);
            psam_split2_f16(v[2], v[3], sc, h1, l1);
            const bool odd = lane & 1;
            const unsigned r0 = __shfl_xor(odd ? h0 : l0, 1, 64), r1 = __shfl_xor(odd ? h1 : l1, 1, 64);
            typedef unsigned ip_u32x4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<ip_u32x4*>(out + (z * N + n) * C + c) = odd ? ip_u32x4{r0, r1, l0, l1} : ip_u32x4{h0, h1, r0, r1};
        } else {
            *reinterpret_cast<f32x4*>(out + (z * N + n) * C + c) = v;
        }
    }
}

// C == 256: R rows per wave, every load of all R rows issued before the first use.  With one row per wave the kernel is a chain of two
// memory round trips (indices, then the three source rows) and ~10 cross-lane steps, and its time is that latency times the number
// of occupancy rounds (262144 rows / 8192 resident waves = 32 rounds: 160 us); R independent chains per wave divide the rounds by R.
template <int R>
__global__ __launch_bounds__(256) void interp3_c256_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx3, const float* __restrict__ w3,
                                                           float* __restrict__ out, int rep, int64_t Z, int N, int G, float* __restrict__ scale_out,
                                                           const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps, int act) {
    constexpr int C = 256;
    const int lane = threadIdx.x & 63, c = lane * 4;
    const int64_t total = Z * N, row0 = ((((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6)) * R;
    if (row0 >= total) return;
    int64_t rows[R];
    const float* p0[R]; const float* p1[R]; const float* p2[R];
    float w0[R], w1[R], w2[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        rows[r] = row0 + r < total ? row0 + r : total - 1;      // clamped duplicates are computed and not stored
        const int64_t z = rows[r] / N, n = rows[r] % N, b = z / rep, o = (b * N + n) * 3;
        const float* s = src + z * (int64_t)G * C + c;
        p0[r] = s + idx3[o] * C; p1[r] = s + idx3[o + 1] * C; p2[r] = s + idx3[o + 2] * C;
        w0[r] = w3[o]; w1[r] = w3[o + 1]; w2[r] = w3[o + 2];
        // each weight in a register of its own: hipcc otherwise keeps (w0, w1) as a 64-bit pair and multiplies with `v_pk_mul_f32 .. op_sel:[1,0]`
        // (broadcast of the HIGH register), a form that returns wrong lanes 48-63 beside another stream's GEMM workgroups (point_sam_amd/isa_lint.py)
        asm volatile("" : "+v"(w0[r]), "+v"(w1[r]), "+v"(w2[r]));
    }
    f32x4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p0[r]), bb = *reinterpret_cast<const f32x4*>(p1[r]), cc = *reinterpret_cast<const f32x4*>(p2[r]);
        v[r] = a * w0[r];      // same association as (x*w).sum(-2): ((a*w0) + b*w1) + c*w2
        v[r] = v[r] + bb * w1[r];
        v[r] = v[r] + cc * w2[r];
    }
    if (ln_g) {
        const f32x4 gg = *reinterpret_cast<const f32x4*>(ln_g + c), be = *reinterpret_cast<const f32x4*>(ln_b + c);
        float mean[R], q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) mean[r] = wave_sum((v[r][0] + v[r][1]) + (v[r][2] + v[r][3])) * (1.0f / 256.0f);
#pragma unroll
        for (int r = 0; r < R; ++r) { v[r] = v[r] - mean[r]; q[r] = wave_sum((v[r][0] * v[r][0] + v[r][1] * v[r][1]) + (v[r][2] * v[r][2] + v[r][3] * v[r][3])); }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float rstd = 1.0f / sqrtf(q[r] * (1.0f / 256.0f) + ln_eps);
            v[r] = v[r] * rstd * gg + be;
            if (act == 1) v[r] = f32x4{gelu_erf(v[r][0]), gelu_erf(v[r][1]), gelu_erf(v[r][2]), gelu_erf(v[r][3])};
            else if (act == 2) v[r] = f32x4{fmaxf(v[r][0], 0.f), fmaxf(v[r][1], 0.f), fmaxf(v[r][2], 0.f), fmaxf(v[r][3], 0.f)};
        }
    }
    if (scale_out) {
        float sc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) sc[r] = f16_row_scale(wave_max(fmaxf(fmaxf(fabsf(v[r][0]), fabsf(v[r][1])), fmaxf(fabsf(v[r][2]), fabsf(v[r][3])))));
#pragma unroll
        for (int r = 0; r < R; ++r) {
            unsigned h0, l0, h1, l1;
            psam_split2_f16(v[r][0], v[r][1], sc[r], h0, l0);
            psam_split2_f16(v[r][2], v[r][3], sc[r], h1, l1);
            const bool odd = lane & 1;
            const unsigned r0 = __shfl_xor(odd ? h0 : l0, 1, 64), r1 = __shfl_xor(odd ? h1 : l1, 1, 64);
            typedef unsigned ip_u32x4 __attribute__((ext_vector_type(4)));
            if (row0 + r < total) {
                if (lane == 0) scale_out[rows[r]] = sc[r];
                *reinterpret_cast<ip_u32x4*>(out + rows[r] * C + c) = odd ? ip_u32x4{r0, r1, l0, l1} : ip_u32x4{h0, h1, r0, r1};
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (row0 + r < total) *reinterpret_cast<f32x4*>(out + rows[r] * C + c) = v[r];
    }
}

// scale_out [Z*N] != NULL (C == 256 only): out receives the g8-packed rows (A operand of psam_gemm_f16x3p) and scale_out their scales.
// ln_gamma / ln_beta != NULL (C == 256 only): LayerNorm(ln_eps) + activation `act` (PSAM_ACT_NONE / GELU / RELU) of every interpolated row.
PSAM_API int32_t psam_interp3_ex(const float* src, const int64_t* idx3, const float* w3, float* out, int32_t rep, int64_t Z, int32_t N, int32_t G,
                                 int32_t C, float* scale_out, const float* ln_gamma, const float* ln_beta, float ln_eps, int32_t act,
                                 hipStream_t stream) {
    PSAM_REQUIRE(src && idx3 && w3 && out && rep > 0 && Z > 0 && N > 0 && G > 0 && C > 0, PSAM_EINVAL, "psam_interp3: bad argument");
    PSAM_REQUIRE((C & 3) == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)out & 15) == 0, PSAM_EALIGN, "psam_interp3: C % 4 and 16B alignment");
    PSAM_REQUIRE(!scale_out || (C == 256 && ((uintptr_t)out & 31) == 0), PSAM_EINVAL, "psam_interp3: packed output needs C == 256 and 32-byte aligned rows");
    PSAM_REQUIRE((ln_gamma == nullptr) == (ln_beta == nullptr) && (!ln_gamma || (C == 256 && (((uintptr_t)ln_gamma | (uintptr_t)ln_beta) & 15) == 0)) &&
                 act >= 0 && act <= 2 && (ln_gamma || act == 0), PSAM_EINVAL, "psam_interp3: row LayerNorm needs gamma and beta, C == 256, act in {none, gelu, relu}");
    if (C == 256) {
        constexpr int R = 4;
        hipLaunchKernelGGL(interp3_c256_kernel<R>, dim3((unsigned)psam_cdiv(Z * N, 4 * R)), dim3(256), 0, stream, src, idx3, w3, out, rep, Z, N, G, scale_out,
                           ln_gamma, ln_beta, ln_eps, act);
        return psam_launch_status("psam_interp3: launch failed");
    }
    hipLaunchKernelGGL(interp3_kernel, dim3((unsigned)psam_cdiv(Z * N, 4)), dim3(256), 0, stream, src, idx3, w3, out, rep, Z, N, G, C, scale_out, ln_gamma,
                       ln_beta, ln_eps, act);
    return psam_launch_status("psam_interp3: launch failed");
}

PSAM_API int32_t psam_interp3(const float* src, const int64_t* idx3, const float* w3, float* out, int32_t rep, int64_t Z, int32_t N, int32_t G,
                              int32_t C, hipStream_t stream) {
    return psam_interp3_ex(src, idx3, w3, out, rep, Z, N, G, C, nullptr, nullptr, nullptr, 0.f, 0, stream);
}

// ------------------------------------------------------------------------------------------------
// Small multi-head attention for the decoder's token-sized problems (transformer.py:214-236): softmax(q k^T / sqrt(hd)) v.
// q/k/v/out are [Z, L, H*hd] with explicit row strides.  Two shapes occur (7 output tokens against 512 patch tokens, both ways):
//  * many keys (attention_small_kernel): one wave per (batch, head, query).  Keys are scored one per lane (float4 loads), the
//    probabilities parked in LDS; for the output the wave is split into 64 / hd4 key groups x hd4 float4 channels, each lane sums
//    its group's keys and the groups are added with fixed-order shuffles.  (The first version gave every output channel to one
//    lane -- 16 active lanes walking all 512 keys: 52 us for the token -> image attention.)
//  * few keys (Lk <= 16, attention_fewkeys_kernel): one THREAD per (batch, head, query): the scores, the softmax and the output
//    stay in registers; consecutive threads are consecutive queries of one (batch, head), so every k / v address is wave-uniform.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attention_small_kernel(const float* __restrict__ q, int64_t ldq, int64_t sq, const float* __restrict__ k,
                                                              int64_t ldk, int64_t sk, const float* __restrict__ v, int64_t ldv, int64_t sv,
                                                              float* __restrict__ out, int64_t ldo, int64_t so, int64_t Z, int H, int Lq, int Lk,
                                                              int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) float s_p[];  // [4 waves][Lk]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wv = (int64_t)blockIdx.x * 4 + wave;
    if (wv >= Z * H * Lq) return;
    const int qi = (int)(wv % Lq);
    const int hh = (int)((wv / Lq) % H);
    const int64_t z = wv / ((int64_t)Lq * H);
    const float* qp = q + z * sq + (int64_t)qi * ldq + hh * hd;
    const float* kp = k + z * sk + hh * hd;
    const float* vp = v + z * sv + hh * hd;
    float* p = s_p + wave * Lk;
    const bool vec = (hd & 3) == 0 && hd <= 64 && (64 % (hd >> 2)) == 0 && ((ldq | ldk | ldv | ldo | sq | sk | sv | so) & 3) == 0 &&
                     ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0);      // wave-uniform
    float m = -INFINITY;
    for (int j = lane; j < Lk; j += 64) {
        const float* kr = kp + (int64_t)j * ldk;
        float s = 0.f;
        if (vec) {
            for (int d = 0; d < hd; d += 4) {
                const float4 a = *reinterpret_cast<const float4*>(qp + d), b = *reinterpret_cast<const float4*>(kr + d);
                s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
            }
        } else {
            for (int d = 0; d < hd; ++d) s = fmaf(qp[d], kr[d], s);
        }
        s *= scale;
        p[j] = s;
        m = fmaxf(m, s);
    }
    m = wave_max(m);
    float l = 0.f;
    for (int j = lane; j < Lk; j += 64) { const float e = __expf(p[j] - m); p[j] = e; l += e; }
    l = wave_sum(l);
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const float invl = 1.0f / l;
    if (vec) {
        const int hd4 = hd >> 2, ng = 64 / hd4, c4 = lane % hd4, grp = lane / hd4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = grp; j < Lk; j += ng) {
            const float pj = p[j];
            const float4 vv = *reinterpret_cast<const float4*>(vp + (int64_t)j * ldv + c4 * 4);
            acc.x = fmaf(pj, vv.x, acc.x); acc.y = fmaf(pj, vv.y, acc.y); acc.z = fmaf(pj, vv.z, acc.z); acc.w = fmaf(pj, vv.w, acc.w);
        }
        for (int o = 32; o >= hd4; o >>= 1) {      // add the key groups (lanes hd4 apart), fixed order
            acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64); acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
        }
        if (grp == 0)
            *reinterpret_cast<float4*>(out + z * so + (int64_t)qi * ldo + hh * hd + c4 * 4) = make_float4(acc.x * invl, acc.y * invl, acc.z * invl, acc.w * invl);
        return;
    }
    for (int d = lane; d < hd; d += 64) {
        float acc = 0.f;
        for (int j = 0; j < Lk; ++j) acc = fmaf(p[j], vp[(int64_t)j * ldv + d], acc);
        out[z * so + (int64_t)qi * ldo + hh * hd + d] = acc * invl;
    }
}

// Few queries against many keys (the decoder's 7-10 tokens x 8 heads against 512 patch tokens: 72 waves in attention_small_kernel, each walking all
// 512 keys twice -- 19.6 us): one WORKGROUP per (batch, head, query), wave w takes the keys [w * per, (w + 1) * per); every wave runs the softmax and the
// value sum of its keys relative to its own maximum (the code of attention_small_kernel), the four (max, sum, value) partials meet in LDS and wave 0
// combines them in fixed order.  Head dims with hd % 4 == 0, hd <= 64, 64 % (hd / 4) == 0, 16-byte aligned rows.
__global__ __launch_bounds__(256) void attention_small_split_kernel(const float* __restrict__ q, int64_t ldq, int64_t sq, const float* __restrict__ k,
                                                                    int64_t ldk, int64_t sk, const float* __restrict__ v, int64_t ldv, int64_t sv,
                                                                    float* __restrict__ out, int64_t ldo, int64_t so, int H, int Lq, int Lk, int hd, float scale) {
    extern __shared__ __attribute__((aligned(16))) float s_p[];  // [Lk] probabilities, then [4][2 + 64] partials
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t wv = blockIdx.x;
    const int qi = (int)(wv % Lq);
    const int hh = (int)((wv / Lq) % H);
    const int64_t z = wv / ((int64_t)Lq * H);
    const float* qp = q + z * sq + (int64_t)qi * ldq + hh * hd;
    const float* kp = k + z * sk + hh * hd;
    const float* vp = v + z * sv + hh * hd;
    float* part = s_p + ((Lk + 3) & ~3) + wave * 68;
    const int per = (Lk + 3) >> 2, j0 = wave * per, j1 = j0 + per < Lk ? j0 + per : Lk;
    float m = -INFINITY;
    for (int j = j0 + lane; j < j1; j += 64) {
        const float* kr = kp + (int64_t)j * ldk;
        float s = 0.f;
        for (int d = 0; d < hd; d += 4) {
            const float4 a = *reinterpret_cast<const float4*>(qp + d), b = *reinterpret_cast<const float4*>(kr + d);
            s = fmaf(a.x, b.x, s); s = fmaf(a.y, b.y, s); s = fmaf(a.z, b.z, s); s = fmaf(a.w, b.w, s);
        }
        s *= scale;
        s_p[j] = s;
        m = fmaxf(m, s);
    }
    m = wave_max(m);
    float l = 0.f;
    for (int j = j0 + lane; j < j1; j += 64) { const float e = __expf(s_p[j] - m); s_p[j] = e; l += e; }
    l = wave_sum(l);
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    const int hd4 = hd >> 2, ng = 64 / hd4, c4 = lane % hd4, grp = lane / hd4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = j0 + grp; j < j1; j += ng) {
        const float pj = s_p[j];
        const float4 vv = *reinterpret_cast<const float4*>(vp + (int64_t)j * ldv + c4 * 4);
        acc.x = fmaf(pj, vv.x, acc.x); acc.y = fmaf(pj, vv.y, acc.y); acc.z = fmaf(pj, vv.z, acc.z); acc.w = fmaf(pj, vv.w, acc.w);
    }
    for (int o = 32; o >= hd4; o >>= 1) {      // add the key groups (lanes hd4 apart), fixed order
        acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64); acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
    }
    if (lane == 0) { part[0] = m; part[1] = l; }
    if (grp == 0) *reinterpret_cast<float4*>(part + 4 + c4 * 4) = acc;
    __syncthreads();
    if (wave != 0 || lane >= hd4) return;
    const float* p0 = s_p + ((Lk + 3) & ~3);
    float mm = fmaxf(fmaxf(p0[0], p0[68]), fmaxf(p0[136], p0[204]));
    float lt = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float f = __expf(p0[w * 68] - mm);      // a wave without keys: exp(-inf) = 0
        const float4 a = *reinterpret_cast<const float4*>(p0 + w * 68 + 4 + lane * 4);
        lt = fmaf(f, p0[w * 68 + 1], lt);
        o.x = fmaf(f, a.x, o.x); o.y = fmaf(f, a.y, o.y); o.z = fmaf(f, a.z, o.z); o.w = fmaf(f, a.w, o.w);
    }
    const float invl = 1.0f / lt;
    *reinterpret_cast<float4*>(out + z * so + (int64_t)qi * ldo + hh * hd + lane * 4) = make_float4(o.x * invl, o.y * invl, o.z * invl, o.w * invl);
}

template <int HD4>   // head dim / 4
__global__ __launch_bounds__(256) void attention_fewkeys_kernel(const float* __restrict__ q, int64_t ldq, int64_t sq, const float* __restrict__ k,
                                                                int64_t ldk, int64_t sk, const float* __restrict__ v, int64_t ldv, int64_t sv,
                                                                float* __restrict__ out, int64_t ldo, int64_t so, int H, int Lq, int Lk, float scale) {
    // grid: (ceil(Lq / 256), H, Z)
    const int qi = blockIdx.x * 256 + threadIdx.x, hh = blockIdx.y;
    const int64_t z = blockIdx.z;
    if (qi >= Lq) return;
    const float* qp = q + z * sq + (int64_t)qi * ldq + hh * (HD4 * 4);
    const float* kp = k + z * sk + hh * (HD4 * 4);
    const float* vp = v + z * sv + hh * (HD4 * 4);
    float4 qr[HD4];
#pragma unroll
    for (int d = 0; d < HD4; ++d) qr[d] = *reinterpret_cast<const float4*>(qp + 4 * d);
    float sc[16];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        sc[j] = -INFINITY;
        if (j < Lk) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD4; ++d) {
                const float4 b = *reinterpret_cast<const float4*>(kp + (int64_t)j * ldk + 4 * d);
                s = fmaf(qr[d].x, b.x, s); s = fmaf(qr[d].y, b.y, s); s = fmaf(qr[d].z, b.z, s); s = fmaf(qr[d].w, b.w, s);
            }
            sc[j] = s * scale;
            m = fmaxf(m, sc[j]);
        }
    }
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { sc[j] = j < Lk ? __expf(sc[j] - m) : 0.f; l += sc[j]; }
    const float invl = 1.0f / l;
    float4 acc[HD4];
#pragma unroll
    for (int d = 0; d < HD4; ++d) acc[d] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < Lk) {
#pragma unroll
            for (int d = 0; d < HD4; ++d) {
                const float4 vv = *reinterpret_cast<const float4*>(vp + (int64_t)j * ldv + 4 * d);
                acc[d].x = fmaf(sc[j], vv.x, acc[d].x); acc[d].y = fmaf(sc[j], vv.y, acc[d].y);
                acc[d].z = fmaf(sc[j], vv.z, acc[d].z); acc[d].w = fmaf(sc[j], vv.w, acc[d].w);
            }
        }
    float* op = out + z * so + (int64_t)qi * ldo + hh * (HD4 * 4);
#pragma unroll
    for (int d = 0; d < HD4; ++d) *reinterpret_cast<float4*>(op + 4 * d) = make_float4(acc[d].x * invl, acc[d].y * invl, acc[d].z * invl, acc[d].w * invl);
}

static int g_attn_small_split = 1;      // test / A-B hook: 0 = always one wave per query
PSAM_API void psam_attention_small_force_split(int32_t on) { g_attn_small_split = on; }
PSAM_API int32_t psam_attention_small(const float* q, int64_t ldq, int64_t sq, const float* k, int64_t ldk, int64_t sk, const float* v,
                                      int64_t ldv, int64_t sv, float* out, int64_t ldo, int64_t so, int64_t Z, int32_t H, int32_t Lq, int32_t Lk,
                                      int32_t hd, float scale, hipStream_t stream) {
    PSAM_REQUIRE(q && k && v && out && Z > 0 && H > 0 && Lq > 0 && Lk > 0 && hd > 0, PSAM_EINVAL, "psam_attention_small: bad argument");
    PSAM_REQUIRE((size_t)Lk * 16 <= 128 * 1024, PSAM_EINVAL, "psam_attention_small: Lk too large");
    const bool aligned = ((ldq | ldk | ldv | ldo | sq | sk | sv | so) & 3) == 0 && ((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0);
    if (Lk <= 16 && Lq >= 64 && aligned && (hd == 16 || hd == 32) && Z <= 65535 && H <= 65535) {
        const dim3 grid((unsigned)psam_cdiv(Lq, 256), H, (unsigned)Z);
        if (hd == 16) hipLaunchKernelGGL(attention_fewkeys_kernel<4>, grid, dim3(256), 0, stream, q, ldq, sq, k, ldk, sk, v, ldv, sv, out, ldo, so, H, Lq, Lk, scale);
        else hipLaunchKernelGGL(attention_fewkeys_kernel<8>, grid, dim3(256), 0, stream, q, ldq, sq, k, ldk, sk, v, ldv, sv, out, ldo, so, H, Lq, Lk, scale);
        return psam_launch_status("psam_attention_small: launch failed");
    }
    const int64_t waves = Z * H * Lq;
    if (g_attn_small_split && Lk >= 128 && waves <= 2048 && aligned && (hd & 3) == 0 && hd <= 64 && (64 % (hd >> 2)) == 0) {      // few queries, many keys: a workgroup per query
        hipLaunchKernelGGL(attention_small_split_kernel, dim3((unsigned)waves), dim3(256), (size_t)(((Lk + 3) & ~3) + 4 * 68) * 4, stream, q, ldq, sq, k, ldk, sk, v, ldv,
                           sv, out, ldo, so, H, Lq, Lk, hd, scale);
        return psam_launch_status("psam_attention_small: launch failed");
    }
    hipLaunchKernelGGL(attention_small_kernel, dim3((unsigned)psam_cdiv(waves, 4)), dim3(256), (size_t)Lk * 16, stream, q, ldq, sq, k, ldk, sk, v,
                       ldv, sv, out, ldo, so, Z, H, Lq, Lk, hd, scale);
    return psam_launch_status("psam_attention_small: launch failed");
}

// ------------------------------------------------------------------------------------------------
// Three-layer ReLU MLP on a handful of rows: the decoder's hyper-networks and IoU head (mask_decoder.py:171-180,189-211), 256-wide
// layers applied to one token row per prompt.  As GEMMs these were nine + three launches of a one-workgroup kernel (~10 us each, all
// latency); here one workgroup per (prompt row, MLP) walks the three layers with the activations in LDS.  A wave takes 32 output rows
// of W at a time: lane l multiplies its float4 of the input (k = 4l .. 4l+3 of each 256-wide chunk) with the same float4 of every
// row -- 32 independent, fully coalesced 1 KiB loads in flight -- and the 64 per-lane partial sums of each output are added through a
// wave-private LDS transpose (lane o sums row o: fixed order).  (A thread-per-output version over transposed weights was a chain of
// dependent loads: 59 us per launch.)  MLP m of a stack reads x + z*ldx + m*sx and uses the m-th [out][in] matrix of each weight.
// ------------------------------------------------------------------------------------------------
constexpr int MLP3_MAXD = 1024, MLP3_OB = 16, MLP3_NW = 16;      // 16 waves x 16 outputs: a 256-wide layer is ONE round of 16 loads per lane (round 5; before: four waves x 32 outputs, two rounds per layer, 256 registers + 54 spilled)
__device__ __forceinline__ void mlp3_body(const float* __restrict__ x, int64_t ldx, int64_t sx, const float* __restrict__ w1,
                                          const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                          const float* __restrict__ w3, const float* __restrict__ b3, float* __restrict__ out, int64_t ldo,
                                          int64_t so, int din, int dh, int dout, int z, int m) {
    __shared__ __attribute__((aligned(16))) float s_a[MLP3_MAXD], s_b[MLP3_MAXD];
    __shared__ float s_red[MLP3_NW][MLP3_OB][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xp = x + (int64_t)z * ldx + (int64_t)m * sx;
    for (int i = tid; i < MLP3_MAXD; i += 64 * MLP3_NW) s_a[i] = i < din ? xp[i] : 0.f;
    __syncthreads();
    auto layer = [&](const float* __restrict__ src, float* __restrict__ dst, const float* __restrict__ W, const float* __restrict__ b, int ni, int no,
                     bool relu) {
        for (int ob = wave * MLP3_OB; ob < no; ob += MLP3_NW * MLP3_OB) {
            float part[MLP3_OB];
#pragma unroll
            for (int o = 0; o < MLP3_OB; ++o) part[o] = 0.f;
            // buffer loads: the row offset is a scalar (soffset), the lane's k offset the one address register of all MLP3_OB loads.  The scalar offset
            // is NOT covered by the resource's bounds check, so rows past `no` are clamped to the last row (their sums are never stored) and k past `ni`
            // to the row's last chunk (it meets a zero of the input: LDS holds zeros beyond ni) -- no load leaves the matrix.
            typedef unsigned m3_u32x4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, no * ni * 4, 0x00020000);
            const int obu = __builtin_amdgcn_readfirstlane(ob);
            for (int kc = 0; kc < ni; kc += 256) {
                const int k = kc + 4 * lane, kw = k < ni ? k : ni - 4;
                const f32x4 xv = *reinterpret_cast<const f32x4*>(src + (k < MLP3_MAXD ? k : 0));      // zero beyond ni (ni % 4 == 0)
                // every input element in a register of its own: hipcc otherwise multiplies pairs with `v_pk_fma_f32 .. op_sel:[0,0,1]`, a form that
                // returns wrong lanes 48-63 beside another stream's GEMM workgroups (point_sam_amd/isa_lint.py)
                float x0 = xv[0], x1 = xv[1], x2 = xv[2], x3 = xv[3];
                asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
#pragma unroll
                for (int o = 0; o < MLP3_OB; ++o) {
                    const int row = obu + o < no ? obu + o : no - 1;
                    const f32x4 wv = __builtin_bit_cast(f32x4, (m3_u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, kw * 4, row * ni * 4, 0));
                    part[o] = fmaf(x3, wv[3], fmaf(x2, wv[2], fmaf(x1, wv[1], fmaf(x0, wv[0], part[o]))));      // one chain per output (16 independent chains): nothing to pair
                }
            }
#pragma unroll
            for (int o = 0; o < MLP3_OB; ++o) {      // lane pairs first, then 32 values per output through LDS
                const float pr = part[o] + dpp_f32<DPP_XOR1>(part[o]);
                if (!(lane & 1)) s_red[wave][o][lane >> 1] = pr;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane < MLP3_OB && ob + lane < no) {
                float y = b[ob + lane];
                for (int l = 0; l < 32; ++l) y += s_red[wave][lane][l];
                if (relu) y = fmaxf(y, 0.f);
                dst[ob + lane] = y;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
    layer(s_a, s_b, w1 + (int64_t)m * dh * din, b1 + (int64_t)m * dh, din, dh, true);
    __syncthreads();
    for (int i = dh + tid; i < MLP3_MAXD; i += 64 * MLP3_NW) s_b[i] = 0.f;      // (the next layer reads whole float4 chunks)
    __syncthreads();
    layer(s_b, s_a, w2 + (int64_t)m * dh * dh, b2 + (int64_t)m * dh, dh, dh, true);
    __syncthreads();
    for (int i = dh + tid; i < MLP3_MAXD; i += 64 * MLP3_NW) s_a[i] = 0.f;
    __syncthreads();
    layer(s_a, out + (int64_t)z * ldo + (int64_t)m * so, w3 + (int64_t)m * dout * dh, b3 + (int64_t)m * dout, dh, dout, false);
}
__global__ __launch_bounds__(64 * MLP3_NW) void mlp3_kernel(const float* __restrict__ x, int64_t ldx, int64_t sx, const float* __restrict__ w1,
                                                   const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                   const float* __restrict__ w3, const float* __restrict__ b3, float* __restrict__ out, int64_t ldo,
                                                   int64_t so, int din, int dh, int dout) {
    mlp3_body(x, ldx, sx, w1, b1, w2, b2, w3, b3, out, ldo, so, din, dh, dout, blockIdx.x, blockIdx.y);
}
// Two MLP stacks over rows of the same matrix in one launch (grid.y = M_a + M_b): the decoder's hyper-networks (mask tokens) and its IoU head (token 0)
// both read the transformer's token output (mask_decoder.py:167-182) -- two single-wave-of-workgroups launches of ~20 us, all latency, side by side.
__global__ __launch_bounds__(64 * MLP3_NW) void mlp3_pair_kernel(const psam_mlp3_args_t a, const psam_mlp3_args_t b) {
    const int m = blockIdx.y;
    if (m < a.M) mlp3_body(a.x, a.ldx, a.sx, a.w1, a.b1, a.w2, a.b2, a.w3, a.b3, a.out, a.ldo, a.so, a.din, a.dh, a.dout, blockIdx.x, m);
    else mlp3_body(b.x, b.ldx, b.sx, b.w1, b.b1, b.w2, b.b2, b.w3, b.b3, b.out, b.ldo, b.so, b.din, b.dh, b.dout, blockIdx.x, m - a.M);
}

// x rows [Z] (row stride ldx; MLP m reads at + m * sx), stacked weights in the reference's [out, in] layout w1 [M, dh, din], w2 [M, dh, dh],
// w3 [M, dout, dh], biases [M, dh], [M, dh], [M, dout]; out rows [Z] (stride ldo; MLP m writes dout values at + m * so).  din, dh % 4 == 0.
PSAM_API int32_t psam_mlp3(const float* x, int64_t ldx, int64_t sx, const float* w1, const float* b1, const float* w2, const float* b2,
                           const float* w3, const float* b3, float* out, int64_t ldo, int64_t so, int32_t Z, int32_t M, int32_t din, int32_t dh,
                           int32_t dout, hipStream_t stream) {
    PSAM_REQUIRE(x && w1 && b1 && w2 && b2 && w3 && b3 && out, PSAM_EINVAL, "psam_mlp3: null pointer");
    PSAM_REQUIRE(Z > 0 && M > 0 && M <= 65535 && din > 0 && dh > 0 && dout > 0 && din <= MLP3_MAXD && dh <= MLP3_MAXD && (din & 3) == 0 && (dh & 3) == 0,
                 PSAM_EINVAL, "psam_mlp3: bad shape (din, dh <= 1024 and multiples of 4)");
    PSAM_REQUIRE((((uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)w3) & 15) == 0, PSAM_EALIGN, "psam_mlp3: weights must be 16-byte aligned");
    hipLaunchKernelGGL(mlp3_kernel, dim3((unsigned)Z, (unsigned)M), dim3(64 * MLP3_NW), 0, stream, x, ldx, sx, w1, b1, w2, b2, w3, b3, out, ldo, so, din, dh, dout);
    return psam_launch_status("psam_mlp3: launch failed");
}

static int32_t mlp3_check(const psam_mlp3_args_t& a) {
    PSAM_REQUIRE(a.x && a.w1 && a.b1 && a.w2 && a.b2 && a.w3 && a.b3 && a.out, PSAM_EINVAL, "psam_mlp3_pair: null pointer");
    PSAM_REQUIRE(a.M > 0 && a.M <= 32767 && a.din > 0 && a.dh > 0 && a.dout > 0 && a.din <= MLP3_MAXD && a.dh <= MLP3_MAXD && (a.din & 3) == 0 && (a.dh & 3) == 0,
                 PSAM_EINVAL, "psam_mlp3_pair: bad shape (din, dh <= 1024 and multiples of 4)");
    PSAM_REQUIRE((((uintptr_t)a.w1 | (uintptr_t)a.w2 | (uintptr_t)a.w3) & 15) == 0, PSAM_EALIGN, "psam_mlp3_pair: weights must be 16-byte aligned");
    return PSAM_OK;
}
// psam_mlp3 of stack a and of stack b over the same Z rows in one launch (the same arithmetic per output: the same bits as two psam_mlp3 calls).
PSAM_API int32_t psam_mlp3_pair(const psam_mlp3_args_t* a, const psam_mlp3_args_t* b, int32_t Z, hipStream_t stream) {
    PSAM_REQUIRE(a && b && Z > 0, PSAM_EINVAL, "psam_mlp3_pair: bad argument");
    int32_t rc = mlp3_check(*a);
    if (!rc) rc = mlp3_check(*b);
    if (rc) return rc;
    hipLaunchKernelGGL(mlp3_pair_kernel, dim3((unsigned)Z, (unsigned)(a->M + b->M)), dim3(64 * MLP3_NW), 0, stream, *a, *b);
    return psam_launch_status("psam_mlp3_pair: launch failed");
}

// out[i] = ((p0[i] + p1[i]) + p2[i]) + ...: the partial planes of a GEMM's hyper-product epilogue (psam_gemm_fuse_t.hyper), fixed order.
__global__ __launch_bounds__(256) void sum_planes_kernel(const float* __restrict__ parts, int P, int64_t pstride, int64_t count4, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count4) return;
    f32x4 a = *reinterpret_cast<const f32x4*>(parts + 4 * i);
    for (int q = 1; q < P; ++q) a = a + *reinterpret_cast<const f32x4*>(parts + q * pstride + 4 * i);
    *reinterpret_cast<f32x4*>(out + 4 * i) = a;
}

PSAM_API int32_t psam_sum_planes(const float* parts, int32_t P, int64_t pstride, int64_t count, float* out, hipStream_t stream) {
    PSAM_REQUIRE(parts && out && P > 0 && count > 0, PSAM_EINVAL, "psam_sum_planes: bad argument");
    PSAM_REQUIRE((count & 3) == 0 && (pstride & 3) == 0 && (((uintptr_t)parts | (uintptr_t)out) & 15) == 0, PSAM_EALIGN, "psam_sum_planes: count, stride % 4 and 16B alignment");
    hipLaunchKernelGGL(sum_planes_kernel, dim3((unsigned)psam_cdiv(count / 4, 256)), dim3(256), 0, stream, parts, P, pstride, count / 4, out);
    return psam_launch_status("psam_sum_planes: launch failed");
}
