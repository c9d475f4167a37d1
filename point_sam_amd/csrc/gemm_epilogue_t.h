// Register-only GEMM epilogue for TRANSPOSED accumulator tiles (round 4).
//
// The packed-operand kernels may issue their MFMAs with the operands swapped -- D = W_frag x A_frag^T instead of A_frag x W_frag^T: same
// products, same k order, same bits per element -- and the C/D layout of v_mfma_f32_32x32x16 (col = lane & 31, row = (reg & 3) + 8 (reg >> 2)
// + 4 (lane >> 5)) then puts ONE OUTPUT ROW in a lane: acc[i][j][r] = C[row_base + 32 i + (lane & 31)][col_base + 32 j + 8 (r >> 2) + 4 h + (r & 3)],
// h = lane >> 5.  Every lane owns four runs of four consecutive output columns per 32x32 tile, so
//   * the per-row operands (1 / scaleA, folded-LayerNorm mean / rstd, the packed output's bound) are lane-local scalars: no lane shuffles;
//   * a run is one float4: un-scale, bias, activation / SwiGLU gate, residual and the fp32 store happen in registers, 16 bytes per lane;
//   * the g8-packed output needs 8 consecutive columns per lane: ONE half-wave exchange (v_permlane32_swap: lanes 32-63 of the first operand
//     swap with lanes 0-31 of the second) per packed register turns runs (g, h = 0 | 1) into column groups 8 (2 g' + h) .. + 7;
//   * the LayerNorm partials of a gated row (32 columns) are in-lane sums plus three half-wave exchanges.
// gemm_epilogue.h's LDS transposition (ds_write x 16 per tile, rolled pass loop of LDS reads + ds_bpermute row scalars + parked rows) is not
// used at all: 11-12 us of a 75 us 256x256 tile there (profiles/r03/r03_epi_ablation.txt).  The arithmetic per element -- and the summation order of the row statistics -- is the same as in
// gemm_epilogue.h, so both epilogues give the same bits (tests/test_gpu_kernels.py::test_gemm_f16x3_register_epilogue_bitwise).
//
// Options covered (what the encoder / decoder hot path launches): bias, alpha, GELU / ReLU, SwiGLU gate (act == 3, column tiles (2q, 2q+1)),
// residual, folded LayerNorm (ln_c / ln_mean / ln_rstd), packed output (pack_out with out_k1 / out_k2 or out_bound), row statistics (stats).
// hyper products of 64-column wave tiles (TN == 2; N / 64 partial planes) with or without storing the activation (no_store).
// rowbias (a bias row per group of rows).
// Not covered (the host keeps gemm_epilogue.h for those launches): group maximum, full-row LayerNorm / full-row hyper products.
// Store pattern: a wave instruction writes 32 rows x 2 x 16 B (32 B contiguous per row for fp32 outputs, 2 x 16 B 32 B apart for packed
// ones); the four instructions of a tile complete each row's 128-byte line in L2.
#pragma once
#include "gemm_epilogue.h"

typedef unsigned ept_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ept_u32x2 __attribute__((ext_vector_type(2)));

// Scheduling fence between the tiles of an epilogue: without it the pre-RA scheduler hoists every tile's constant / residual loads to the top
// (latency hiding the epilogue does not need at two waves per SIMD) and the 128-register accumulator tiles of the ping-pong kernels spill.
__device__ __forceinline__ void ept_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
}
// lanes 32-63 of a <-> lanes 0-31 of b
__device__ __forceinline__ void ept_swap32(unsigned& a, unsigned& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const ept_u32x2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
#endif
}
__device__ __forceinline__ void ept_swap32f(float& a, float& b) {
    unsigned x = __builtin_bit_cast(unsigned, a), y = __builtin_bit_cast(unsigned, b);
    ept_swap32(x, y);
    a = __builtin_bit_cast(float, x); b = __builtin_bit_cast(float, y);
}
// s[g] = this lane's partial over run g (columns 8 g + 4 h .. + 3 of a 32-column segment): returns, in every lane, the segment's total in the
// order ((c0 + c1) + (c2 + c3)) + ((c4 + c5) + (c6 + c7)) over the eight float4 partials c = 2 g + h -- the xor-1 / xor-2 / xor-4 butterfly of
// gemm_epilogue.h's ep_group_sum<8> over lanes c4 = 0 .. 7.
__device__ __forceinline__ float ept_segment_sum(float s0, float s1, float s2, float s3) {
    ept_swap32f(s0, s1);            // s0 = {lo: c0, hi: c2}, s1 = {lo: c1, hi: c3}
    ept_swap32f(s2, s3);            // s2 = {lo: c4, hi: c6}, s3 = {lo: c5, hi: c7}
    float p = s0 + s1, q = s2 + s3; // p = {c0 + c1 | c2 + c3}, q = {c4 + c5 | c6 + c7}
    ept_swap32f(p, q);              // p = {c01 | c45}, q = {c23 | c67}
    float u = p + q, w = u;         // u = {c0123 | c4567}
    ept_swap32f(u, w);              // u = {c0123 | c0123}, w = {c4567 | c4567}
    return u + w;
}

// F >= 0: the option set of the launch as EP_* bits (gemm_epilogue.h), every option test folds at compile time; F < 0: run-time tests.
// FENCE = false (the persistent kernel's 64-accumulator wave tile, which has the registers): no scheduling fences between the tiles -- the scheduler
// may issue every tile's constant and residual loads at the top, one memory latency for the whole epilogue instead of one per tile.
template <int TM, int TN, int F, typename ArgsT, bool FENCE = true>
__device__ __forceinline__ void gemm_store_tile_t_impl(const ArgsT& p, ep_f32x16 (&acc)[TM][TN], int row_base, int col_base, int lane, float* __restrict__ C,
                                                       const float* __restrict__ R, bool interior) {
#pragma clang fp contract(off)
    const int r32 = lane & 31, h = lane >> 5;
    const bool swiglu = (TN % 2 == 0) && (F < 0 ? p.act == 3 : bool(F & EP_SWIGLU));
    const float alpha = swiglu ? 1.f : p.alpha;
    const int o_act = F < 0 ? p.act : ((F & EP_GELU) ? 1 : ((F & EP_RELU) ? 2 : 0));
    const bool o_res = !swiglu && (F < 0 ? R != nullptr : bool(F & EP_RES));
    const bool o_rowbias = !swiglu && (F < 0 ? p.rowbias != nullptr : bool(F & EP_ROWBIAS));      // bias row per group of `rowgroup` rows: the lane's own row picks it
    if (!interior) {
        // ---- edge tiles: every element bounds-checked, plain options only (the host guarantees interior tiles for the fused extras)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = row_base + i * 32 + r32;
            const float rm = row < p.M ? inv_pow2(p.scaleA[row]) : 1.f;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (swiglu && (j & 1)) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pc = col_base + j * 32 + 8 * (r >> 2) + 4 * h + (r & 3);            // column in N (of g when gated)
                    if (row >= p.M || pc >= p.N) continue;
                    float v = (acc[i][j][r] * rm) * (alpha * inv_pow2(p.scaleW[pc])) + (p.bias ? p.bias[pc] : 0.f);
                    int oc = pc;
                    if (swiglu) {
                        if constexpr (TN % 2 == 0) {
                            const float x = pc + 32 < p.N ? (acc[i][j + 1][r] * rm) * inv_pow2(p.scaleW[pc + 32]) + (p.bias ? p.bias[pc + 32] : 0.f) : 0.f;
                            v = silu(v) * x;
                            oc = (col_base >> 1) + (j >> 1) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                        }
                    } else {
                        if (o_rowbias) v += p.rowbias[(int64_t)(row / p.rowgroup) * p.ldrb + pc];
                        v = ep_act(v, o_act);
                        if (o_res) v += R[(int64_t)row * p.ldr + pc];
                    }
                    C[(int64_t)row * p.ldc + oc] = v;
                }
            }
        }
        return;
    }
    const bool o_lnc = F < 0 ? p.ln_c != nullptr : bool(F & EP_LNC), o_pack = F < 0 ? p.pack_out != 0 : bool(F & EP_PACK);
    const bool o_bnd = F < 0 ? (o_pack && !o_lnc && p.out_bound != nullptr) : bool(F & EP_BND), o_stats = swiglu && (F < 0 ? p.stats != nullptr : bool(F & EP_STATS));
    // hyper-network products (mask_decoder.py:171-176): masks[z, c, n] = sum_e hyper[z, c, e] out[z * hyper_rows + n, e] over this wave's TN * 32 columns -> plane
    // col_base / (TN * 32) of the partial sums (psam_sum_planes adds the planes in a fixed order).  A row lives in a lane pair (lane, lane + 32): in-lane
    // fused multiply-adds in column order of the lane's runs, ONE half-wave exchange per product.  hyper_rows % 32 == 0: one z per 32-row tile.
    const bool o_hyper = !swiglu && (F < 0 ? p.hyper != nullptr : bool(F & EP_HYPER)), o_nostore = F < 0 ? p.no_store != 0 : bool(F & EP_NOSTORE);
    float dot[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) dot[i][cc] = 0.f;
    // ---- per-row operands: the lane's own rows
    float rsq[TM], lmean[TM], lrstd[TM], so[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = row_base + i * 32 + r32;
        rsq[i] = inv_pow2(p.scaleA[row]);
        lmean[i] = 0.f; lrstd[i] = 1.f; so[i] = 1.f;
        if (o_lnc) { lmean[i] = p.ln_mean[row]; lrstd[i] = p.ln_rstd[row]; }
        if (o_pack) {
            const float bnd = p.out_k1 * rsq[i] + p.out_k2;
            so[i] = f16_row_scale(o_bnd ? p.out_bound[row] : (swiglu ? bnd * bnd : bnd));
            if (h == 0 && col_base == 0) p.out_scale[row] = so[i];
        }
    }
#pragma unroll
    for (int j0 = 0; j0 < TN; ++j0) {
        if (swiglu && (j0 & 1)) continue;
        const int j = j0;
        if (FENCE) ept_fence();
        const int pcol = col_base + j * 32 + 4 * h;                 // run g starts at pcol + 8 g
        const int ocol = swiglu ? (col_base >> 1) + (j >> 1) * 32 + 4 * h : pcol;
        // column constants of the tile (and of its partner tile when gated)
        ep_f32x4 m[4], b[4], mx[4], bx[4], lc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const ep_f32x4 sw = ep_load4(p.scaleW + pcol + 8 * g);
            b[g] = p.bias ? ep_load4(p.bias + pcol + 8 * g) : ep_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) m[g][e] = alpha * inv_pow2(sw[e]);
            mx[g] = m[g]; bx[g] = b[g]; lc[g] = b[g];
            if (swiglu) {
                const ep_f32x4 sx = ep_load4(p.scaleW + pcol + 32 + 8 * g);
                bx[g] = p.bias ? ep_load4(p.bias + pcol + 32 + 8 * g) : ep_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) mx[g][e] = inv_pow2(sx[e]);
            }
            if (o_lnc) lc[g] = ep_load4(p.ln_c + pcol + 8 * g);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (FENCE && i > 0) ept_fence();
            const int row = row_base + i * 32 + r32;
            ep_f32x4 res[4];
            if (o_res) {
#pragma unroll
                for (int g = 0; g < 4; ++g) res[g] = ep_load4(R + (int64_t)row * p.ldr + ocol + 8 * g);
            }
            ep_f32x4 rbv[4];
            if (o_rowbias) {
                const float* rbp = p.rowbias + (int64_t)(row / p.rowgroup) * p.ldrb + pcol;
#pragma unroll
                for (int g = 0; g < 4; ++g) rbv[g] = ep_load4(rbp + 8 * g);
            }
            ep_f32x4 v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = acc[i][j][4 * g + e] * rsq[i];                       // exact: a power of two
                    if (o_lnc) a = __builtin_fmaf(lrstd[i], __builtin_fmaf(m[g][e], a, -(lc[g][e] * lmean[i])), b[g][e]);
                    else a = __builtin_fmaf(a, m[g][e], b[g][e]);
                    if (swiglu) {
                        if constexpr (TN % 2 == 0) {
                            const float x = __builtin_fmaf(acc[i][j + 1][4 * g + e] * rsq[i], mx[g][e], bx[g][e]);
                            a = silu(a) * x;
                        }
                    } else {
                        if (o_rowbias) a += rbv[g][e];
                        a = ep_act(a, o_act);
                    }
                    v[g][e] = a;
                }
            }
            if (o_stats) {      // LayerNorm partials of this row over the 32 gated columns of the tile pair (valid ones: < stat_cols)
                const int seg0 = (col_base >> 1) + (j >> 1) * 32;
                const int nv = p.stat_cols - seg0 < 32 ? (p.stat_cols - seg0 > 0 ? p.stat_cols - seg0 : 0) : 32;
                float s[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    s[g] = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[g] += (ocol + 8 * g + e < p.stat_cols) ? v[g][e] : 0.f;
                }
                const float sm = ept_segment_sum(s[0], s[1], s[2], s[3]);
                const float mean = nv > 0 ? sm / (float)nv : 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float d[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) d[e] = (ocol + 8 * g + e < p.stat_cols) ? v[g][e] - mean : 0.f;
                    // the order hipcc gives gemm_epilogue.h's `m2 += d * d` chain (contraction on): d1^2 first, then fused adds of d0^2, d2^2, d3^2
                    s[g] = __builtin_fmaf(d[3], d[3], __builtin_fmaf(d[2], d[2], __builtin_fmaf(d[0], d[0], d[1] * d[1])));
                }
                const float m2 = ept_segment_sum(s[0], s[1], s[2], s[3]);
                if (h == 0) {
                    float* st = p.stats + ((int64_t)row * p.stat_segs + seg0 / 32) * 2;
                    st[0] = mean; st[1] = m2;
                }
            }
            if (o_res) {
#pragma unroll
                for (int g = 0; g < 4; ++g) v[g] += res[g];
            }
            if (o_hyper) {
                const int z = (row_base + i * 32) / p.hyper_rows;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
                    if (cc < p.hyper_c) {
                        const float* hb = p.hyper + ((int64_t)z * p.hyper_c + cc) * p.N + pcol;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const ep_f32x4 hv = ep_load4(hb + 8 * g);
#pragma unroll
                            for (int e = 0; e < 4; ++e) dot[i][cc] = __builtin_fmaf(v[g][e], hv[e], dot[i][cc]);
                        }
                    }
            }
            if (o_nostore) continue;
            if (o_pack) {
                // runs (2 g', 2 g' + 1) of the two half-waves -> column groups 8 (2 g' + h) .. + 7 of this lane: [hi x 8 | lo x 8] = 32 contiguous bytes
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    unsigned xh[2], xl[2], yh[2], yl[2];
                    psam_split2_f16(v[2 * gp][0], v[2 * gp][1], so[i], xh[0], xl[0]);
                    psam_split2_f16(v[2 * gp][2], v[2 * gp][3], so[i], xh[1], xl[1]);
                    psam_split2_f16(v[2 * gp + 1][0], v[2 * gp + 1][1], so[i], yh[0], yl[0]);
                    psam_split2_f16(v[2 * gp + 1][2], v[2 * gp + 1][3], so[i], yh[1], yl[1]);
                    ept_swap32(xh[0], yh[0]); ept_swap32(xh[1], yh[1]); ept_swap32(xl[0], yl[0]); ept_swap32(xl[1], yl[1]);
                    // x = columns 0-3, y = columns 4-7 of group 2 gp + h (relative to the tile: 16 gp + 8 h)
                    unsigned* dst = reinterpret_cast<unsigned*>(C) + (int64_t)row * p.ldc + (ocol - 4 * h) + 16 * gp + 8 * h;
                    *reinterpret_cast<ept_u32x4*>(dst) = ept_u32x4{xh[0], xh[1], yh[0], yh[1]};
                    *reinterpret_cast<ept_u32x4*>(dst + 4) = ept_u32x4{xl[0], xl[1], yl[0], yl[1]};
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<ep_f32x4*>(C + (int64_t)row * p.ldc + ocol + 8 * g) = v[g];
            }
        }
    }
    if (o_hyper) {
        float* mk = p.masks + (int64_t)(col_base / (TN * 32)) * p.hyper_pstride;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row0 = row_base + i * 32, z = row0 / p.hyper_rows, n0 = row0 - z * p.hyper_rows;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                float a = dot[i][cc], b = dot[i][cc];
                ept_swap32f(a, b);                               // a = {lo: own, hi: lower half's}, b = {lo: upper half's, hi: own}
                if (cc < p.hyper_c && h == 0) mk[((int64_t)z * p.hyper_c + cc) * p.hyper_rows + n0 + r32] = a + b;
            }
        }
    }
}

// Dispatch on the launch's option set (wave-uniform): the combinations the model's GEMMs use run a specialised instance.
template <int TM, int TN, typename ArgsT>
__device__ __forceinline__ void gemm_store_tile_t(const ArgsT& p, ep_f32x16 (&acc)[TM][TN], int row_base, int col_base, int lane, float* __restrict__ C,
                                                  const float* __restrict__ R) {
    const bool interior = gemm_epilogue_interior<TM, TN>(p, row_base, col_base, C, R);      // wave-uniform
    const bool swiglu = (TN % 2 == 0) && p.act == 3;
    const int opt = (swiglu ? EP_SWIGLU : 0) | ((R && !swiglu) ? EP_RES : 0) | ((p.rowbias && !swiglu) ? EP_ROWBIAS : 0) | (!swiglu && p.act == 1 ? EP_GELU : 0) | (!swiglu && p.act == 2 ? EP_RELU : 0) |
                    ((swiglu && p.stats) ? EP_STATS : 0) | (p.pack_out ? EP_PACK : 0) | (p.ln_c ? EP_LNC : 0) | ((p.pack_out && !p.ln_c && p.out_bound) ? EP_BND : 0) |
                    ((p.hyper && !swiglu) ? EP_HYPER : 0) | (p.no_store ? EP_NOSTORE : 0);
    using std::integral_constant;
#define EPT_CASE(Fv) if (opt == (Fv)) return gemm_store_tile_t_impl<TM, TN, (Fv)>(p, acc, row_base, col_base, lane, C, R, interior)
    if (interior) {
        EPT_CASE(0);                                            // bias only
        EPT_CASE(EP_RES);                                       // attention projection
        EPT_CASE(EP_PACK);                                      // qkv, packed for the attention kernel
        EPT_CASE(EP_LNC | EP_RES);                              // fc2 with the folded LayerNorm
        EPT_CASE(EP_GELU);
        EPT_CASE(EP_ROWBIAS);                                   // PatchEncoder conv2.0 (per-row half; the pooled half arrives as the group's bias row)
        if constexpr (TN == 2) { EPT_CASE(EP_GELU | EP_HYPER | EP_NOSTORE); }      // last Linear of the upscaling MLP + the hyper products
        if constexpr (TN % 2 == 0) {
            EPT_CASE(EP_SWIGLU | EP_STATS | EP_PACK | EP_BND);  // fc1 of the fused EVA02 MLP
            EPT_CASE(EP_SWIGLU | EP_STATS | EP_PACK);
            EPT_CASE(EP_SWIGLU);
        }
    }
#undef EPT_CASE
    gemm_store_tile_t_impl<TM, TN, -1>(p, acc, row_base, col_base, lane, C, R, interior);
}

// Full-row variant for workgroup tiles that span ALL N columns (N = WN * TN * 32: the 128x512 ping-pong tile of PatchEncoder's conv2.0): the
// epilogue applies `Linear -> LayerNorm -> activation` (common.py:493-496: conv2.0 -> conv2.1 LayerNorm -> GELU) to whole rows and writes them
// g8-packed for the next GEMM, so the [rows, N] fp32 activation (537 MB at the benchmark's size) and the separate LayerNorm pass over it never
// exist.  A row lives in one lane pair per wave and in the WN waves of its row band: two-pass statistics (mean, then centred squares) with
// in-lane sums, one half-wave exchange and a WN-way exchange through `red` (LDS, 2 * WN * BM floats; the K loop's ring is dead by now), each in a
// fixed order.  Options: bias, row bias per group of rows (rowgroup % 32 == 0: one bias row per 32-row tile), GELU / ReLU / none, packed output
// with the a-priori scale f16_row_scale(out_k2) (out_k2 >= max |gamma| sqrt(N) + max |beta| bounds every LayerNorm output, hence its GELU).
// Interior tiles only (M % BM == 0; the host checks).  Every wave of the workgroup must call it (two __syncthreads).
template <int TM, int TN, int WN, typename ArgsT>
__device__ __forceinline__ void gemm_store_tile_t_rowln(const ArgsT& p, ep_f32x16 (&acc)[TM][TN], float* __restrict__ red, int BM, int row_base, int tile_row,
                                                        int wn, int lane, float* __restrict__ C) {
#pragma clang fp contract(off)
    const int r32 = lane & 31, h = lane >> 5;
    const int col_base = wn * TN * 32;
    float rsq[TM], sum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { rsq[i] = inv_pow2(p.scaleA[row_base + i * 32 + r32]); sum[i] = 0.f; }
    // ---- pass A: finished pre-LayerNorm values in place, row sums
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        ept_fence();
        const int pcol = col_base + j * 32 + 4 * h;
        ep_f32x4 m[4], b[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const ep_f32x4 sw = ep_load4(p.scaleW + pcol + 8 * g);
            b[g] = p.bias ? ep_load4(p.bias + pcol + 8 * g) : ep_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) m[g][e] = p.alpha * inv_pow2(sw[e]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i > 0) ept_fence();
            ep_f32x4 rb[4];
            if (p.rowbias) {
                const float* rbp = p.rowbias + (int64_t)((row_base + i * 32) / p.rowgroup) * p.ldrb + pcol;
#pragma unroll
                for (int g = 0; g < 4; ++g) rb[g] = ep_load4(rbp + 8 * g);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = __builtin_fmaf(acc[i][j][4 * g + e] * rsq[i], m[g][e], b[g][e]);
                    if (p.rowbias) a += rb[g][e];
                    acc[i][j][4 * g + e] = a;
                    sum[i] += a;
                }
        }
    }
    // row totals: lane pair, then the WN waves of the row band (fixed order)
    float* red2 = red + WN * BM;
    float mean[TM], rstd[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float a = sum[i], b = sum[i];
        ept_swap32f(a, b);
        sum[i] = h == 0 ? a + b : b + a;                         // lower half's partial + upper half's, in both halves
        if (h == 0) red[wn * BM + tile_row + i * 32 + r32] = sum[i];
    }
    __syncthreads();
    const float inv_n = 1.0f / (float)(WN * TN * 32);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) t += red[w * BM + tile_row + i * 32 + r32];
        mean[i] = t * inv_n;
    }
    // ---- pass B: centred squares
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = acc[i][j][r] - mean[i]; q = __builtin_fmaf(d, d, q); }
        float a = q, b = q;
        ept_swap32f(a, b);
        q = h == 0 ? a + b : b + a;
        if (h == 0) red2[wn * BM + tile_row + i * 32 + r32] = q;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) t += red2[w * BM + tile_row + i * 32 + r32];
        rstd[i] = 1.0f / sqrtf(t * inv_n + p.row_ln_eps);
    }
    // ---- pass C: normalise, activate, pack, store
    const float so = f16_row_scale(p.out_k2);
    if (p.pack_out && h == 0 && wn == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) p.out_scale[row_base + i * 32 + r32] = so;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        ept_fence();
        const int pcol = col_base + j * 32 + 4 * h;
        ep_f32x4 gm[4], bt[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) { gm[g] = ep_load4(p.row_ln_g + pcol + 8 * g); bt[g] = ep_load4(p.row_ln_b + pcol + 8 * g); }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (i > 0) ept_fence();
            const int row = row_base + i * 32 + r32;
            ep_f32x4 v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[g][e] = ep_act(__builtin_fmaf((acc[i][j][4 * g + e] - mean[i]) * rstd[i], gm[g][e], bt[g][e]), p.act);
            if (p.pack_out) {
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    unsigned xh[2], xl[2], yh[2], yl[2];
                    psam_split2_f16(v[2 * gp][0], v[2 * gp][1], so, xh[0], xl[0]);
                    psam_split2_f16(v[2 * gp][2], v[2 * gp][3], so, xh[1], xl[1]);
                    psam_split2_f16(v[2 * gp + 1][0], v[2 * gp + 1][1], so, yh[0], yl[0]);
                    psam_split2_f16(v[2 * gp + 1][2], v[2 * gp + 1][3], so, yh[1], yl[1]);
                    ept_swap32(xh[0], yh[0]); ept_swap32(xh[1], yh[1]); ept_swap32(xl[0], yl[0]); ept_swap32(xl[1], yl[1]);
                    unsigned* dst = reinterpret_cast<unsigned*>(C) + (int64_t)row * p.ldc + col_base + j * 32 + 16 * gp + 8 * h;
                    *reinterpret_cast<ept_u32x4*>(dst) = ept_u32x4{xh[0], xh[1], yh[0], yh[1]};
                    *reinterpret_cast<ept_u32x4*>(dst + 4) = ept_u32x4{xl[0], xl[1], yl[0], yl[1]};
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<ep_f32x4*>(C + (int64_t)row * p.ldc + pcol + 8 * g) = v[g];
            }
        }
    }
}
