// Token side of one TwoWayAttentionBlock (pc_sam/model/transformer.py:103-176) -- or of the final token -> image attention
// (transformer.py:91-99) -- in ONE launch.
//
// The decoder's output tokens are a handful of rows (1 IoU + 4 mask tokens + the prompts, per prompt set: R = Z * T <= 64 rows of 256
// channels).  As separate launches a layer's token side is ~22 dependent kernels of 5-9 us each (q/k/v/out projections, three
// attentions' worth of them, the MLP, three LayerNorms, the query_pe additions): all latency.  Here a small team of workgroups walks the
// stages with the activations in a [64-row] workspace that never leaves L2:
//   self-attention:  q, k = (queries + pe) W^T, v = queries W^T | attention over the T tokens | out_proj (+ queries) | LayerNorm
//   token -> image:  q = (queries + pe) W^T | attention over the G patch tokens (their k / v projections come from the image-side GEMMs)
//                    | out_proj + queries | LayerNorm
//   MLP:             ReLU(queries W1^T) | W2^T + queries | LayerNorm
//   for image -> token: k = (queries + pe) W^T, v = queries W^T   (consumed by the image-side attention that follows)
// Every stage splits its output columns (16 per workgroup and step, all rows at once: the arithmetic of linear_skinny_kernel, exact fp32
// products on v_mfma_f32_16x16x4_f32) or its (prompt, head, query) attention rows over the team; stages are separated by a counter
// barrier (agent-scope atomics), and what one stage hands to the next is written and read with device-coherent (sc1) accesses.  The team is
// NWG workgroups placed on ONE XCD (the grid is 8 x NWG, only ids = 0 mod 8 take part: profiles/r03/r03_fabric_probe.txt, 1.0-1.7 us per
// hand-over there against 1.3-2.1 us spread out); placement is a speed matter only.
// The arithmetic follows the kernels this replaces (csrc/gemm.hip linear_skinny_kernel, csrc/rowops.hip attention_small_kernel and
// layernorm_v4_kernel) operation for operation.
// MEASURED (profiles/r03/r03_twoway.txt): parity-green, but SLOWER than the launches it replaces (two-way transformer at cfg #2: 0.56 ms with 64
// workgroups against 0.48 ms) -- each of ~11 stage boundaries per layer costs a fabric round trip plus a coherent first load, and the separate
// launches spread every projection over N / 16 workgroups with their operands hot in L2.  A variant with one workgroup per prompt set (tokens
// in LDS, no cross-CU barrier) was slower still (0.78 ms).  The model therefore keeps the separate launches (PointCloudSAM.fuse_tokens =
// False); this entry point stays as the C-ABI way to run a layer's token side in one call.
#include <cstdlib>
#include "common.h"


namespace {
constexpr int TW_E = 256, TW_IX = 128, TW_CH = 8;
typedef float tw_f32x4 __attribute__((ext_vector_type(4)));

struct TwDev {
    psam_twoway_tokens_t a;
    float *wq, *wk, *wv, *wa, *wy, *wm;
    unsigned* bar;
};

// Activations that one stage writes and the next reads travel through device-coherent accesses (sc1: stores write through, loads do not
// hit a stale line of this CU's L1 / this XCD's L2), so the barrier needs no bulk cache maintenance: an agent-scope release / acquire
// fence pair writes back and invalidates whole caches -- it cost ~20 us per barrier here and threw the layer's weights out of L2 each time.
constexpr int TW_SC1 = 16;      // cache-policy bit of the raw buffer builtins: sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tw_rsrc(const void* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7fffffff, 0x00020000); }
template <bool COH> __device__ __forceinline__ tw_f32x4 tw_ld4(__amdgpu_buffer_rsrc_t r, int off) {
    return __builtin_bit_cast(tw_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, COH ? TW_SC1 : 0));
}
__device__ __forceinline__ float tw_ld1(__amdgpu_buffer_rsrc_t r, int off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, TW_SC1)); }
__device__ __forceinline__ void tw_st4(__amdgpu_buffer_rsrc_t r, int off, tw_f32x4 v) {
    typedef unsigned tw_u32x4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(tw_u32x4, v), r, off, 0, TW_SC1);
}
__device__ __forceinline__ void tw_st1(__amdgpu_buffer_rsrc_t r, int off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, TW_SC1); }

template <int NWG>
__device__ __forceinline__ void tw_barrier(unsigned* ctr, unsigned& phase) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's (write-through) stage outputs are acknowledged
    __syncthreads();
    ++phase;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = phase * NWG;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// y [M, N] = act((x (+ xadd)) W^T + bias) (+ res): this workgroup's 16-column blocks cb = wg, wg + NWG, ...; wave w holds rows 16 w .. 16 w + 15.
// x, res, y: activations of this launch (coherent accesses); xadd (the token embeddings), W, bias: inputs of the launch.
template <int NWG>
__device__ __forceinline__ void tw_linear(const float* x, int ldx, const float* xadd, const float* __restrict__ W, int64_t ldw,
                                          const float* __restrict__ bias, const float* res, int ldr, float* y, int ldy, int M, int N, int K,
                                          int act, int wg, int first) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int row = wave * 16 + r;
    const bool rok = row < M;
    const __amdgpu_buffer_rsrc_t rx = tw_rsrc(x), ra = tw_rsrc(xadd ? xadd : x), rr = tw_rsrc(res ? res : x), ry = tw_rsrc(y);
    const int xoff = ((rok ? row : 0) * ldx + 4 * g) * 4;
    for (int cb = (wg + first) % NWG; cb * 16 < N; cb += NWG) {
        const int col = cb * 16 + r;
        const bool cok = col < N;
        const float* wp = W + (int64_t)(cok ? col : 0) * ldw + 4 * g;
        tw_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        tw_f32x4 xa[TW_CH], wb[TW_CH], xn[TW_CH], wn[TW_CH];
        auto load = [&](tw_f32x4 (&xr)[TW_CH], tw_f32x4 (&wr)[TW_CH], int kc) {
#pragma unroll
            for (int s = 0; s < TW_CH; ++s) {
                const int k = kc + 16 * s;
                const bool in = k < K;
                xr[s] = (in && rok) ? tw_ld4<true>(rx, xoff + k * 4) : tw_f32x4{0.f, 0.f, 0.f, 0.f};
                if (xadd && in && rok) xr[s] += tw_ld4<false>(ra, xoff + k * 4);
                wr[s] = (in && cok) ? *reinterpret_cast<const tw_f32x4*>(wp + k) : tw_f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        load(xa, wb, 0);
        for (int kc = 0; kc < K; kc += 16 * TW_CH) {
            const bool more = kc + 16 * TW_CH < K;
            if (more) load(xn, wn, kc + 16 * TW_CH);
#pragma unroll
            for (int s = 0; s < TW_CH; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s][e], wb[s][e], acc, 0, 0, 0);
            if (more) {
#pragma unroll
                for (int s = 0; s < TW_CH; ++s) { xa[s] = xn[s]; wb[s] = wn[s]; }
            }
        }
        if (cok) {
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int orow = wave * 16 + 4 * g + v;
                if (orow < M) {
                    float val = acc[v] + bv;
                    if (act == 2) val = fmaxf(val, 0.f);
                    if (res) val += tw_ld1(rr, (orow * ldr + col) * 4);
                    tw_st1(ry, (orow * ldy + col) * 4, val);
                }
            }
        }
    }
}

// softmax(q k^T scale) v, one wave per (prompt set, head, query): the arithmetic of attention_small_kernel's float4 path (hd % 4 == 0, hd <= 64).
// q, out: activations of this launch; k, v: activations (KVC, the self-attention) or inputs of the launch (the patch tokens' projections).
template <int NWG, bool KVC>
__device__ __forceinline__ void tw_attention(const float* q, int ldq, int sq, const float* k, int64_t ldk, int64_t sk, const float* v, int64_t ldv,
                                             int64_t sv, float* out, int ldo, int so, int Z, int H, int Lq, int Lk, int hd, float scale, float* s_p,
                                             int wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* p = s_p + wave * Lk;
    const int total = Z * H * Lq;
    const __amdgpu_buffer_rsrc_t rq = tw_rsrc(q), ro = tw_rsrc(out);
    for (int wv = wg * 4 + wave; wv < total; wv += NWG * 4) {
        const int qi = wv % Lq, hh = (wv / Lq) % H, z = wv / (Lq * H);
        const int qoff = (z * sq + qi * ldq + hh * hd) * 4;
        const float* kp = k + (int64_t)z * sk + hh * hd;
        const float* vp = v + (int64_t)z * sv + hh * hd;
        const __amdgpu_buffer_rsrc_t rk = tw_rsrc(kp), rv = tw_rsrc(vp);
        float m = -INFINITY;
        for (int j = lane; j < Lk; j += 64) {
            float s = 0.f;
            for (int d = 0; d < hd; d += 4) {
                const tw_f32x4 a = tw_ld4<true>(rq, qoff + d * 4), b = tw_ld4<KVC>(rk, (int)((int64_t)j * ldk + d) * 4);
                s = fmaf(a[0], b[0], s); s = fmaf(a[1], b[1], s); s = fmaf(a[2], b[2], s); s = fmaf(a[3], b[3], s);
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
        const int hd4 = hd >> 2, ng = 64 / hd4, c4 = lane % hd4, grp = lane / hd4;
        tw_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = grp; j < Lk; j += ng) {
            const float pj = p[j];
            const tw_f32x4 vv = tw_ld4<KVC>(rv, (int)((int64_t)j * ldv + c4 * 4) * 4);
            acc[0] = fmaf(pj, vv[0], acc[0]); acc[1] = fmaf(pj, vv[1], acc[1]); acc[2] = fmaf(pj, vv[2], acc[2]); acc[3] = fmaf(pj, vv[3], acc[3]);
        }
        for (int o = 32; o >= hd4; o >>= 1) {
            acc[0] += __shfl_xor(acc[0], o, 64); acc[1] += __shfl_xor(acc[1], o, 64); acc[2] += __shfl_xor(acc[2], o, 64); acc[3] += __shfl_xor(acc[3], o, 64);
        }
        if (grp == 0) tw_st4(ro, (z * so + qi * ldo + hh * hd + c4 * 4) * 4, acc * invl);
        __builtin_amdgcn_wave_barrier();      // every lane is done with p before the next (z, head, query) overwrites it
        __threadfence_block();
    }
}

// out[row] = LayerNorm_256(x[row]) * g + b, one wave per row, a float4 per lane (layernorm_v4_kernel<1>'s arithmetic)
template <int NWG>
__device__ __forceinline__ void tw_layernorm(const float* x, const float* __restrict__ gam, const float* __restrict__ bet, float* out, int R, float eps, int wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv = 1.0f / (float)TW_E;
    const __amdgpu_buffer_rsrc_t rx = tw_rsrc(x), ro = tw_rsrc(out);
    for (int row = wg * 4 + wave; row < R; row += NWG * 4) {
        const tw_f32x4 v = tw_ld4<true>(rx, (row * TW_E + lane * 4) * 4);
        const float s = (v[0] + v[1]) + (v[2] + v[3]);
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
        const float r = 1.0f / sqrtf(wave_sum(q) * inv + eps);
        const tw_f32x4 w4 = *reinterpret_cast<const tw_f32x4*>(gam + lane * 4), b4 = *reinterpret_cast<const tw_f32x4*>(bet + lane * 4);
        tw_st4(ro, (row * TW_E + lane * 4) * 4, (v - mean) * r * w4 + b4);
    }
}

template <int NWG>
__global__ __launch_bounds__(256) void twoway_tokens_kernel(const TwDev d) {
    if ((blockIdx.x & 7) != 0) return;      // ids are dealt to the XCDs round-robin: the team sits on one XCD
    const int wg = blockIdx.x >> 3;
    extern __shared__ __attribute__((aligned(16))) float s_p[];      // [4 waves][max(T, G)] attention probabilities
    const psam_twoway_tokens_t& a = d.a;
    const int R = a.Z * a.T, T = a.T, H = a.heads;
    unsigned phase = 0;
    if (a.mode == 0) {
        const float* xadd = a.skip_pe ? nullptr : a.pe;
        tw_linear<NWG>(a.queries, TW_E, xadd, a.sq_w, TW_E, a.sq_b, nullptr, 0, d.wq, TW_E, R, TW_E, TW_E, 0, wg, 0);
        tw_linear<NWG>(a.queries, TW_E, xadd, a.sk_w, TW_E, a.sk_b, nullptr, 0, d.wk, TW_E, R, TW_E, TW_E, 0, wg, NWG > 16 ? 16 : 0);
        tw_linear<NWG>(a.queries, TW_E, nullptr, a.sv_w, TW_E, a.sv_b, nullptr, 0, d.wv, TW_E, R, TW_E, TW_E, 0, wg, NWG > 32 ? 32 : 0);
        tw_barrier<NWG>(d.bar, phase);
        tw_attention<NWG, true>(d.wq, TW_E, T * TW_E, d.wk, TW_E, (int64_t)T * TW_E, d.wv, TW_E, (int64_t)T * TW_E, d.wa, TW_E, T * TW_E, a.Z, H, T, T, TW_E / H,
                                1.0f / sqrtf((float)(TW_E / H)), s_p, wg);
        tw_barrier<NWG>(d.bar, phase);
        tw_linear<NWG>(d.wa, TW_E, nullptr, a.so_w, TW_E, a.so_b, a.skip_pe ? nullptr : a.queries, TW_E, d.wy, TW_E, R, TW_E, TW_E, 0, wg, 0);
        tw_barrier<NWG>(d.bar, phase);
        tw_layernorm<NWG>(d.wy, a.n1_g, a.n1_b, a.queries, R, a.eps, wg);
        tw_barrier<NWG>(d.bar, phase);
    }
    // token -> image
    tw_linear<NWG>(a.queries, TW_E, a.pe, a.cq_w, TW_E, a.cq_b, nullptr, 0, d.wq, TW_IX, R, TW_IX, TW_E, 0, wg, 0);
    tw_barrier<NWG>(d.bar, phase);
    tw_attention<NWG, false>(d.wq, TW_IX, T * TW_IX, a.kimg, a.ldk, a.sk, a.vimg, a.ldv, a.sv, d.wa, TW_IX, T * TW_IX, a.Z, H, T, a.G, TW_IX / H,
                             1.0f / sqrtf((float)(TW_IX / H)), s_p, wg);
    tw_barrier<NWG>(d.bar, phase);
    tw_linear<NWG>(d.wa, TW_IX, nullptr, a.co_w, TW_IX, a.co_b, a.queries, TW_E, d.wy, TW_E, R, TW_E, TW_IX, 0, wg, 0);
    tw_barrier<NWG>(d.bar, phase);
    tw_layernorm<NWG>(d.wy, a.n2_g, a.n2_b, a.queries, R, a.eps, wg);
    if (a.mode != 0) return;
    tw_barrier<NWG>(d.bar, phase);
    // MLP
    tw_linear<NWG>(a.queries, TW_E, nullptr, a.m1_w, TW_E, a.m1_b, nullptr, 0, d.wm, a.mlp, R, a.mlp, TW_E, 2, wg, 0);
    tw_barrier<NWG>(d.bar, phase);
    tw_linear<NWG>(d.wm, a.mlp, nullptr, a.m2_w, a.mlp, a.m2_b, a.queries, TW_E, d.wy, TW_E, R, TW_E, a.mlp, 0, wg, 0);
    tw_barrier<NWG>(d.bar, phase);
    tw_layernorm<NWG>(d.wy, a.n3_g, a.n3_b, a.queries, R, a.eps, wg);
    tw_barrier<NWG>(d.bar, phase);
    // k / v of the image -> token attention (eight 16-column blocks each: the second one starts at another part of the team)
    tw_linear<NWG>(a.queries, TW_E, a.pe, a.ik_w, TW_E, a.ik_b, nullptr, 0, a.ktok, TW_IX, R, TW_IX, TW_E, 0, wg, 0);
    tw_linear<NWG>(a.queries, TW_E, nullptr, a.iv_w, TW_E, a.iv_b, nullptr, 0, a.vtok, TW_IX, R, TW_IX, TW_E, 0, wg, 8);
}

}  // namespace

// floats of workspace for a layer with `mlp` hidden units: q, k, v, attention output, pre-LayerNorm rows [64, 256] each, the MLP's hidden
// rows [64, mlp], and the barrier counter
PSAM_API int64_t psam_twoway_tokens_ws_floats(int32_t mlp) { return (int64_t)64 * (5 * TW_E + (mlp > 0 ? mlp : 0)) + 64; }

// Token side of one two-way layer (mode 0) or of the final token -> image attention (mode 1): see the head of this file.
// Replaces, per layer, transformer.py:144-169 for the tokens (self_attn, norm1, cross_attn_token_to_image, norm2, mlp, norm3) and the k / v
// projections of cross_attn_image_to_token (:172-175); embedding_dim 256, attention_downsample_rate 2, Z * T <= 64 rows, T <= 64.
PSAM_API int32_t psam_twoway_tokens(const psam_twoway_tokens_t* args, hipStream_t stream) {
    PSAM_REQUIRE(args, PSAM_EINVAL, "psam_twoway_tokens: null argument block");
    const psam_twoway_tokens_t& a = *args;
    PSAM_REQUIRE(a.Z > 0 && a.T > 0 && a.G > 0 && (int64_t)a.Z * a.T <= 64 && a.heads > 0, PSAM_EINVAL, "psam_twoway_tokens: need Z * T <= 64 token rows");
    PSAM_REQUIRE(TW_E % a.heads == 0 && TW_IX % a.heads == 0 && ((TW_E / a.heads) & 3) == 0 && ((TW_IX / a.heads) & 3) == 0 && TW_E / a.heads <= 64 &&
                     64 % (TW_E / a.heads / 4) == 0 && 64 % (TW_IX / a.heads / 4) == 0,
                 PSAM_EINVAL, "psam_twoway_tokens: head count must divide 256 and 128 into multiples of 4 channels");
    PSAM_REQUIRE(a.mode == 0 || a.mode == 1, PSAM_EINVAL, "psam_twoway_tokens: mode is 0 (layer) or 1 (final attention)");
    PSAM_REQUIRE(a.queries && a.pe && a.kimg && a.vimg && a.cq_w && a.co_w && a.n2_g && a.n2_b && a.ws, PSAM_EINVAL, "psam_twoway_tokens: null pointer");
    PSAM_REQUIRE(a.mode == 1 || (a.sq_w && a.sk_w && a.sv_w && a.so_w && a.n1_g && a.n1_b && a.m1_w && a.m2_w && a.n3_g && a.n3_b && a.ik_w && a.iv_w && a.ktok &&
                                 a.vtok && a.mlp > 0 && (a.mlp & 15) == 0),
                 PSAM_EINVAL, "psam_twoway_tokens: a full layer needs every weight, ktok / vtok and mlp % 16 == 0");
    PSAM_REQUIRE(a.ws_floats >= psam_twoway_tokens_ws_floats(a.mode == 0 ? a.mlp : 0), PSAM_EWORKSPACE, "psam_twoway_tokens: workspace too small");
    PSAM_REQUIRE(((a.ldk | a.ldv | a.sk | a.sv) & 3) == 0 && a.ldk >= TW_IX && a.ldv >= TW_IX, PSAM_EALIGN, "psam_twoway_tokens: k / v strides must be multiples of 4 floats");
    const uintptr_t al = (uintptr_t)a.queries | (uintptr_t)a.pe | (uintptr_t)a.kimg | (uintptr_t)a.vimg | (uintptr_t)a.ws | (uintptr_t)a.ktok | (uintptr_t)a.vtok |
                         (uintptr_t)a.sq_w | (uintptr_t)a.sk_w | (uintptr_t)a.sv_w | (uintptr_t)a.so_w | (uintptr_t)a.cq_w | (uintptr_t)a.co_w | (uintptr_t)a.m1_w |
                         (uintptr_t)a.m2_w | (uintptr_t)a.ik_w | (uintptr_t)a.iv_w | (uintptr_t)a.n1_g | (uintptr_t)a.n1_b | (uintptr_t)a.n2_g | (uintptr_t)a.n2_b |
                         (uintptr_t)a.n3_g | (uintptr_t)a.n3_b;
    PSAM_REQUIRE((al & 15) == 0, PSAM_EALIGN, "psam_twoway_tokens: 16-byte aligned pointers");
    const size_t lds = (size_t)4 * (a.G > a.T ? a.G : a.T) * sizeof(float);
    PSAM_REQUIRE(lds <= 64 * 1024, PSAM_EINVAL, "psam_twoway_tokens: too many keys for the probability rows in LDS");
    TwDev d;
    d.a = a;
    d.wq = a.ws; d.wk = d.wq + 64 * TW_E; d.wv = d.wk + 64 * TW_E; d.wa = d.wv + 64 * TW_E; d.wy = d.wa + 64 * TW_E; d.wm = d.wy + 64 * TW_E;
    d.bar = reinterpret_cast<unsigned*>(a.ws + psam_twoway_tokens_ws_floats(a.mode == 0 ? a.mlp : 0) - 64);
    if (hipMemsetAsync(d.bar, 0, 64 * sizeof(float), stream) != hipSuccess) { psam_set_error("psam_twoway_tokens: cannot reset the barrier counter"); return PSAM_EINVAL; }
    static int nwg = 0;      // team size (tuning hook, environment, read once): 16 | 32 | 64 workgroups
    if (!nwg) { const char* e = getenv("PSAM_TW_NWG"); nwg = e ? atoi(e) : 64; if (nwg != 16 && nwg != 32) nwg = 64; }
    if (nwg == 16) hipLaunchKernelGGL(twoway_tokens_kernel<16>, dim3(8 * 16), dim3(256), lds, stream, d);
    else if (nwg == 32) hipLaunchKernelGGL(twoway_tokens_kernel<32>, dim3(8 * 32), dim3(256), lds, stream, d);
    else hipLaunchKernelGGL(twoway_tokens_kernel<64>, dim3(8 * 64), dim3(256), lds, stream, d);
    return psam_launch_status("psam_twoway_tokens: launch failed");
}
