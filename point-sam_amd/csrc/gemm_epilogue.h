// Shared GEMM epilogue: accumulator tiles -> C through a wave-private LDS transpose, so that global memory sees full
// 128..256-byte row segments (float4 per lane) instead of the MFMA layout's 4-byte column-strided scatter.
//
// Why: measured with per-workgroup cycle counters on the bf16x6 kernel (128x128 tile, K = 1024), the direct epilogue --
// 64 global_store_dword per lane, each wave instruction touching 2 rows x 128 B, plus equally scattered residual loads --
// took 37 k cycles per tile against 128 k for the whole K loop.  Here a wave stages 32 rows x (TN*32) columns at a time in
// LDS (row stride +4 floats: conflict-free for the b32 writes and the b128 reads), reads them back as float4 with 16 (or
// 8) lanes per row, applies bias / per-group row bias / activation / residual with vector loads, and stores float4.
// C/D layout of v_mfma_f32_32x32x*: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma once
#include "common.h"

typedef float ep_f32x16 __attribute__((ext_vector_type(16)));
typedef float ep_f32x4 __attribute__((ext_vector_type(4)));

template <int TN>
constexpr int gemm_epilogue_lds_floats_per_wave() { return 32 * (TN * 32 + 4); }

// 1/s for a power-of-two scale s in [2^-125, 2^126] (exact): exponent field 254 - E
__device__ __forceinline__ float inv_pow2(float s) { return __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(unsigned, s)); }

// ArgsT needs: M, N, act, alpha, bias, rowbias, ldrb, rowgroup, ldc, ldr.   C / R already offset for the batch.
// SCALED (gemm_f16x3.hip): ArgsT also has scaleA[M], scaleW[N]; the accumulator is multiplied by 1/(scaleA[row] scaleW[col]).
template <int TM, int TN, bool SCALED = false, typename ArgsT>
__device__ __noinline__ void gemm_store_tile_general(const ArgsT& p, ep_f32x16 (&acc)[TM][TN], float* __restrict__ lw, int row_base, int col_base,
                                                int lane, float* __restrict__ C, const float* __restrict__ R) {
    const int r32 = lane & 31, h = lane >> 5;
    const bool swiglu = p.act == 3;
    // staged width: TN*32 columns, or TN*16 gated columns for the SwiGLU pairing (tile 2q = g, tile 2q+1 = x)
    const int W = swiglu ? TN * 16 : TN * 32;
    constexpr int LD = TN * 32 + 4;
    const int c4n = W >> 2;                 // float4 per staged row: 16, 8 or 4
    const int rpp = 64 / c4n;               // rows per read-back pass
    const int out_col_base = swiglu ? col_base / 2 : col_base;
    const int n_out = swiglu ? p.N / 2 : p.N;
    const bool vec_ok = ((p.ldc & 3) == 0) && (((uintptr_t)C & 15) == 0) && (!R || (((p.ldr & 3) == 0) && (((uintptr_t)R & 15) == 0))) &&
                        (!p.rowbias || (((p.ldrb & 3) == 0) && (((uintptr_t)p.rowbias & 15) == 0))) && (!p.bias || (((uintptr_t)p.bias & 15) == 0));
    float cmul[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        cmul[j] = swiglu ? 1.f : p.alpha;
        if constexpr (SCALED) {
            const int col = col_base + j * 32 + r32;
            cmul[j] *= col < p.N ? inv_pow2(p.scaleW[col]) : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if constexpr (SCALED) {   // un-scale in place (exact: powers of two)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float rm = row < p.M ? inv_pow2(p.scaleA[row]) : 0.f;
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j][r] *= rm;
            }
        }
        // ---- stage this 32-row stripe
        if (swiglu) {
            if constexpr (TN % 2 == 0) {
#pragma unroll
                for (int q = 0; q < TN / 2; ++q) {
                    const int colg = col_base + 2 * q * 32 + r32;
                    const float bg = (p.bias && colg < p.N) ? p.bias[colg] : 0.f;
                    const float bx = (p.bias && colg + 32 < p.N) ? p.bias[colg + 32] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        lw[((r & 3) + 8 * (r >> 2) + 4 * h) * LD + q * 32 + r32] =
                            silu(acc[i][2 * q][r] * cmul[2 * q] + bg) * (acc[i][2 * q + 1][r] * cmul[2 * q + 1] + bx);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) lw[((r & 3) + 8 * (r >> 2) + 4 * h) * LD + j * 32 + r32] = acc[i][j][r] * cmul[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- read back row-major, finish, store
        const int rl0 = lane / c4n, c4 = lane % c4n;
        for (int rl = rl0; rl < 32; rl += rpp) {
            const int row = row_base + i * 32 + rl;
            const int col = out_col_base + c4 * 4;
            if (row >= p.M || col >= n_out) continue;
            ep_f32x4 v = *reinterpret_cast<const ep_f32x4*>(lw + rl * LD + c4 * 4);
            if (vec_ok && col + 3 < n_out) {
                if (!swiglu) {
                    if (p.bias) v += *reinterpret_cast<const ep_f32x4*>(p.bias + col);
                    if (p.rowbias) v += *reinterpret_cast<const ep_f32x4*>(p.rowbias + (int64_t)(row / p.rowgroup) * p.ldrb + col);
                    if (p.act == 1) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
                    else if (p.act == 2) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    if (R) v += *reinterpret_cast<const ep_f32x4*>(R + (int64_t)row * p.ldr + col);
                }
                *reinterpret_cast<ep_f32x4*>(C + (int64_t)row * p.ldc + col) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (col + e >= n_out) break;
                    float o = v[e];
                    if (!swiglu) {
                        if (p.bias) o += p.bias[col + e];
                        if (p.rowbias) o += p.rowbias[(int64_t)(row / p.rowgroup) * p.ldrb + col + e];
                        if (p.act == 1) o = gelu_erf(o);
                        else if (p.act == 2) o = fmaxf(o, 0.f);
                        if (R) o += R[(int64_t)row * p.ldr + col + e];
                    }
                    C[(int64_t)row * p.ldc + col + e] = o;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the stripe is consumed before the next one overwrites it
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

__device__ __forceinline__ ep_f32x4 ep_load4(const float* q) { return *reinterpret_cast<const ep_f32x4*>(q); }
__device__ __forceinline__ ep_f32x4 ep_act4(ep_f32x4 v, int act) {
    if (act == 1) { v[0] = gelu_erf(v[0]); v[1] = gelu_erf(v[1]); v[2] = gelu_erf(v[2]); v[3] = gelu_erf(v[3]); }
    else if (act == 2) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    return v;
}

// Interior tiles (whole wave tile inside M x N, every operand float4-addressable): no per-element checks; the staging is a pure
// transposition and ALL arithmetic happens at read-back, where the loads of a 32-row stripe (residual, row bias, row scale)
// are issued together before the first use -- the general path's dependent load chains cost ~10 us per tile.
template <int TM, int TN, bool SCALED, typename ArgsT>
__device__ __forceinline__ void gemm_store_tile_fast(const ArgsT& p, ep_f32x16 (&acc)[TM][TN], float* __restrict__ lw, int row_base, int col_base,
                                                     int lane, float* __restrict__ C, const float* __restrict__ R) {
    const int r32 = lane & 31, h = lane >> 5;
    constexpr int LD = TN * 32 + 4;
    const bool swiglu = (TN % 2 == 0) && p.act == 3;
    // read-back geometry: non-gated: TN*8 float4 per row; gated: TN*4 float4 of output per row
    const int c4n = swiglu ? TN * 4 : TN * 8, rpp = 64 / c4n;
    const int rl0 = lane / c4n, c4 = lane % c4n;
    // staged column of this lane's (first) float4, packed column in N, output column
    const int scol = swiglu ? ((c4 * 4) >> 5) * 64 + ((c4 * 4) & 31) : c4 * 4;
    const int pcol = col_base + scol;
    const int ocol = swiglu ? (col_base >> 1) + c4 * 4 : pcol;
    ep_f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0, m0 = {p.alpha, p.alpha, p.alpha, p.alpha}, m1;
    if (swiglu) m0 = ep_f32x4{1.f, 1.f, 1.f, 1.f};
    m1 = m0;
    if (p.bias) { b0 = ep_load4(p.bias + pcol); if (swiglu) b1 = ep_load4(p.bias + pcol + 32); }
    if constexpr (SCALED) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { m0[e] *= inv_pow2(p.scaleW[pcol + e]); if (swiglu) m1[e] *= inv_pow2(p.scaleW[pcol + 32 + e]); }
    }
    constexpr int NPMAX = 8, NB = 4;   // passes per stripe: 32 / rpp = 4 or 8
    const int np = 32 / rpp;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) lw[((r & 3) + 8 * (r >> 2) + 4 * h) * LD + j * 32 + r32] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q0 = 0; q0 < NPMAX; q0 += NB) {     // batches of NB passes: loads first, then arithmetic and stores
            if (q0 < np) {
                ep_f32x4 res[NB], rb[NB];
                float rs[NB];
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const int row = row_base + i * 32 + (q0 + q) * rpp + rl0;
                    if (R) res[q] = ep_load4(R + (int64_t)row * p.ldr + ocol);
                    if (p.rowbias) rb[q] = ep_load4(p.rowbias + (int64_t)(row / p.rowgroup) * p.ldrb + pcol);
                    if constexpr (SCALED) rs[q] = inv_pow2(p.scaleA[row]);
                }
#pragma unroll
                for (int q = 0; q < NB; ++q) {
                    const int rl = (q0 + q) * rpp + rl0;
                    const int row = row_base + i * 32 + rl;
                    ep_f32x4 v = ep_load4(lw + rl * LD + scol);
                    if constexpr (SCALED) v *= rs[q];
                    v = v * m0 + b0;
                    if (swiglu) {
                        ep_f32x4 x = ep_load4(lw + rl * LD + scol + 32);
                        if constexpr (SCALED) x *= rs[q];
                        x = x * m1 + b1;
                        v = ep_f32x4{silu(v[0]) * x[0], silu(v[1]) * x[1], silu(v[2]) * x[2], silu(v[3]) * x[3]};
                    } else {
                        if (p.rowbias) v += rb[q];
                        v = ep_act4(v, p.act);
                        if (R) v += res[q];
                    }
                    *reinterpret_cast<ep_f32x4*>(C + (int64_t)row * p.ldc + ocol) = v;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();   // the stripe is consumed before the next one overwrites it
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int TM, int TN, bool SCALED = false, typename ArgsT>
__device__ __forceinline__ void gemm_store_tile(const ArgsT& p, ep_f32x16 (&acc)[TM][TN], float* __restrict__ lw, int row_base, int col_base,
                                                int lane, float* __restrict__ C, const float* __restrict__ R) {
    const bool vec_ok = ((p.ldc & 3) == 0) && (((uintptr_t)C & 15) == 0) && (!R || (((p.ldr & 3) == 0) && (((uintptr_t)R & 15) == 0))) &&
                        (!p.rowbias || (((p.ldrb & 3) == 0) && (((uintptr_t)p.rowbias & 15) == 0))) && (!p.bias || (((uintptr_t)p.bias & 15) == 0));
    if (vec_ok && row_base + TM * 32 <= p.M && col_base + TN * 32 <= p.N) gemm_store_tile_fast<TM, TN, SCALED>(p, acc, lw, row_base, col_base, lane, C, R);
    else {   // edge tiles: out-of-line, on a COPY of the accumulators (only the copy's address escapes; acc stays in registers)
        ep_f32x16 tmp[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) tmp[i][j] = acc[i][j];
        gemm_store_tile_general<TM, TN, SCALED>(p, tmp, lw, row_base, col_base, lane, C, R);
    }
}
