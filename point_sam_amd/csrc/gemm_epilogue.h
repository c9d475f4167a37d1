// Shared GEMM epilogue: accumulator tiles -> C through a wave-private LDS transpose, so that global memory sees full
// 128..256-byte row segments (float4 per lane) instead of the MFMA layout's 4-byte column-strided scatter.
//
// Why: measured with per-workgroup cycle counters on the bf16x6 kernel (128x128 tile, K = 1024), the direct epilogue --
// 64 global_store_dword per lane, each wave instruction touching 2 rows x 128 B, plus equally scattered residual loads --
// took 37 k cycles per tile against 128 k for the whole K loop.  Here a wave stages 32 rows x (TN*32) columns at a time in
// LDS (row stride +4 floats: conflict-free for the b32 writes and the b128 reads) as a pure transposition, then reads rows
// back and does ALL the arithmetic there (un-scale, bias, per-group row bias, activation or SwiGLU gate, residual):
//   * interior tiles (whole wave tile inside M x N, every operand float4-addressable): 16 (or 8) lanes per row, float4
//     everywhere, the loads of two passes (residual, row bias, row scale) issued together before their first use;
//   * edge tiles: a compact rolled loop, one column per lane, every element bounds-checked.
// C/D layout of v_mfma_f32_32x32x*: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
#pragma once
#include <type_traits>
#include "common.h"

typedef float ep_f32x16 __attribute__((ext_vector_type(16)));
typedef float ep_f32x4 __attribute__((ext_vector_type(4)));

template <int TN>
constexpr int gemm_epilogue_lds_floats_per_wave() { return 32 * (TN * 32 + 4); }

// 1/s for a power-of-two scale s in [2^-126, 2^126] (exact): exponent field 254 - E
__device__ __forceinline__ float inv_pow2(float s) { return __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(unsigned, s)); }
__device__ __forceinline__ ep_f32x4 ep_load4(const float* q) { return *reinterpret_cast<const ep_f32x4*>(q); }
__device__ __forceinline__ float ep_act(float v, int act) { return act == 1 ? gelu_erf(v) : (act == 2 ? fmaxf(v, 0.f) : v); }
// Every DS operation in flight completes here.  Added in round 3 as the presumed cure of 1e-2 wrong fc2 outputs in multi-stream runs (a DS
// ordering hazard was suspected); round 4 bisected the failure at ISA level to something else: the pass loop's packed multiply
// `v_pk_mul_f32 .. op_sel:[0,1]`, which returns wrong lanes 48-63 while another stream's GEMM workgroups are launched on the CU -- the code change
// merely made hipcc pick another instruction form (profiles/r04/r04_hazard.txt; point_sam_amd/isa_lint.py now rejects a library that contains one).
// The drains stay: they cost nothing measurable and keep the loop's DS traffic simple.
__device__ __forceinline__ void ep_lgkm_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Sum over aligned groups of W = 4, 8 or 16 consecutive lanes, every lane ending with the group's sum, in the order of the xor butterfly
// (1, 2, 4, 8) it replaces -- on the DPP path (quad_perm, quad_perm, row_half_mirror, row_mirror: each partner already holds its sub-group's
// sum, so mirror and xor meet the same value) instead of ds_bpermute.  Whole wave active.
template <int W>
__device__ __forceinline__ float ep_group_sum(float v) {
    if constexpr (W == 4 || W == 8 || W == 16) {
        v += dpp_f32<DPP_XOR1>(v); v += dpp_f32<DPP_XOR2>(v);
        if constexpr (W >= 8) v += dpp_f32<DPP_HALF_MIRROR>(v);
        if constexpr (W >= 16) v += dpp_f32<DPP_MIRROR>(v);
    } else {        // widths no gated epilogue is instantiated for at run time (odd tile counts, the 256-column row tiles)
#pragma unroll
        for (int o = 1; o < W; o <<= 1) v += __shfl_xor(v, o, 64);
    }
    return v;
}
__device__ __forceinline__ void ep_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Per-row scalars of a wave tile (TM*32 rows), held lane -> row (row_base + 64 t + lane): the inverse A-row scale and, with a folded
// LayerNorm, the row's mean and rstd.  Loaded ONCE per wave tile with coalesced loads and fetched per pass with a lane shuffle; together
// with the lane's column constants (bias, inverse weight-row scales, folded-LayerNorm c) they form the epilogue's PREFETCH, which a
// kernel issues BEFORE its K loop: the epilogue then starts with every operand it needs except the accumulators (and the residual rows)
// already in registers.  (Measured on the 256x256 ping-pong tile, profiles/r03/r03_epi_pmc.txt: 2 500 instructions per wave took 52 k cycles,
// 33 k of them parked at s_waitcnt -- dependent 4-byte global loads inside every pass and one LDS round trip at a time.)
template <int TM>
struct EpRows {
    static constexpr int NR = (TM * 32 + 63) / 64;
    float rs[NR], mean[NR], rstd[NR];
};
template <int TM, bool SCALED, bool EXT, typename ArgsT>
__device__ __forceinline__ EpRows<TM> gemm_epilogue_rows(const ArgsT& p, int row_base, int lane) {
    EpRows<TM> o;
#pragma unroll
    for (int t = 0; t < EpRows<TM>::NR; ++t) {
        int row = row_base + t * 64 + lane;
        row = row < p.M ? row : p.M - 1;
        o.rs[t] = 1.f; o.mean[t] = 0.f; o.rstd[t] = 1.f;
        if (row >= 0) {
            if constexpr (SCALED) o.rs[t] = p.scaleA[row];      // RAW scale (a power of two): inverted where it is used, so that nothing waits for this load early
            if constexpr (EXT) {
                if (p.ln_c) { o.mean[t] = p.ln_mean[row]; o.rstd[t] = p.ln_rstd[row]; }
                else if (p.out_bound) o.mean[t] = p.out_bound[row];      // the packed output's per-row bound travels in the (unused) mean slot
            }
        }
    }
    return o;
}

// lane -> (row inside a pass, float4 column) mapping of the interior path: non-gated TN*8 float4 per row, gated TN*4 float4 of output per row
template <int TN>
struct EpLane {
    bool swiglu, lane_on;
    float alpha;
    int c4n, rpp, np, rl0, c4, scol, pcol, ocol;
    template <typename ArgsT>
    __device__ __forceinline__ EpLane(const ArgsT& p, int col_base, int lane) {
        swiglu = (TN % 2 == 0) && p.act == 3;
        alpha = swiglu ? 1.f : p.alpha;
        c4n = swiglu ? TN * 4 : TN * 8; rpp = 64 / c4n; np = 32 / rpp;          // rows per pass (1, 2, 4 or 8), np = 4 .. 32 passes per stripe
        rl0 = lane / c4n; c4 = lane % c4n;
        lane_on = rl0 < rpp;                                                    // TN = 3: 48 of the 64 lanes carry a float4
        scol = swiglu ? ((c4 * 4) >> 5) * 64 + ((c4 * 4) & 31) : c4 * 4;       // staged column of the lane's (first) float4
        pcol = col_base + scol;                                                // its column in N (packed, for the gate)
        ocol = swiglu ? (col_base >> 1) + c4 * 4 : pcol;                       // output column
    }
};
template <int TM, int TN, typename ArgsT>
__device__ __forceinline__ bool gemm_epilogue_interior(const ArgsT& p, int row_base, int col_base, const float* C, const float* R) {
    const bool vec_ok = ((p.ldc & 3) == 0) && (((uintptr_t)C & 15) == 0) && (!R || (((p.ldr & 3) == 0) && (((uintptr_t)R & 15) == 0))) &&
                        (!p.rowbias || (((p.ldrb & 3) == 0) && (((uintptr_t)p.rowbias & 15) == 0))) && (!p.bias || (((uintptr_t)p.bias & 15) == 0));
    return vec_ok && row_base + TM * 32 <= p.M && col_base + TN * 32 <= p.N;   // wave-uniform
}
// option bits of a specialised pass loop (gemm_store_tile)
enum : int { EP_SWIGLU = 1, EP_STATS = 2, EP_PACK = 4, EP_LNC = 8, EP_RES = 16, EP_ROWBIAS = 32, EP_GELU = 64, EP_RELU = 128, EP_GMAX = 256, EP_HYPER = 512,
              EP_NOSTORE = 1024, EP_BND = 2048 };

template <int TM>
struct EpPre {
    EpRows<TM> rows;
    ep_f32x4 b0, b1, m0, m1, lnc;
};
template <int TM, int TN, bool SCALED, bool EXT, typename ArgsT>
__device__ __forceinline__ EpPre<TM> gemm_epilogue_prefetch(const ArgsT& p, int row_base, int col_base, int lane, const float* C, const float* R) {
    EpPre<TM> o;
    const EpLane<TN> L(p, col_base, lane);
    o.b0 = ep_f32x4{0.f, 0.f, 0.f, 0.f}; o.b1 = o.b0; o.lnc = o.b0;
    o.m0 = ep_f32x4{1.f, 1.f, 1.f, 1.f}; o.m1 = o.m0;      // RAW weight-row scales (1 when not SCALED): alpha / scale is formed in gemm_store_tile
    o.rows = gemm_epilogue_rows<TM, SCALED, EXT>(p, row_base, lane);
    if (gemm_epilogue_interior<TM, TN>(p, row_base, col_base, C, R) && L.lane_on) {
        if constexpr (EXT) { if (p.ln_c) o.lnc = ep_load4(p.ln_c + L.pcol); }
        if (p.bias) { o.b0 = ep_load4(p.bias + L.pcol); if (L.swiglu) o.b1 = ep_load4(p.bias + L.pcol + 32); }
        if constexpr (SCALED) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { o.m0[e] = p.scaleW[L.pcol + e]; if (L.swiglu) o.m1[e] = p.scaleW[L.pcol + 32 + e]; }
        }
    }
    return o;
}

// ArgsT needs: M, N, act, alpha, bias, rowbias, ldrb, rowgroup, ldc, ldr.   C / R already offset for the batch.
// act == 3 (SwiGLU gate fused into fc1): accumulator tile j = 2q holds g and tile 2q+1 holds x of the same 32 hidden units
// (the packed weight alternates 32-row blocks of fc1_g and fc1_x); the output has N/2 columns, no alpha/rowbias/residual.
// SCALED (gemm_f16x3.hip): ArgsT also has scaleA[M], scaleW[N]; the accumulator is multiplied by 1/(scaleA[row] scaleW[col]).
// EXT (gemm_f16x3p.hip; ArgsT then also has the psam_gemm_fuse_t fields, interior tiles only -- the host guarantees it):
//  * pack_out: C receives the g8-packed form ([hi x8 | lo x8] fp16 per 8 columns, gemm_f16x3p.hip) of the output row scaled by
//    out_scale[row] = f16_row_scale(B^2), B = out_k1 / scaleA[row] + out_k2 -- an a-priori BOUND on the row's magnitude (Cauchy-Schwarz
//    on the A row's maximum and the weight row norms) instead of its maximum, which no single tile knows; the bound costs a few of
//    the 18 binades of full-precision range and cannot overflow.  The output can then feed the next GEMM with no pass in between.
//  * stats (with the SwiGLU gate): per row and per wave-wide column segment (TN*16 gated outputs) the segment's mean and centred sum
//    of squares over the columns < stat_cols -- the partials of a LayerNorm over the gated row (merged by psam_ln_stats_finalize).
//  * gmax_out / gmax_k: the max over every group of gmax_k (32 or 64) consecutive rows of the finished output, per column -> gmax_out
//    [M / gmax_k, N] (PatchEncoder's max-pool over the group members, common.py:491,497); no_store: C itself is not written (when the
//    pooled value is all the caller needs, the [M, N] activation never reaches HBM).
//  * full-row wave tiles only (TN == 8: a wave owns whole 256-column rows -- the decoder's upscaling MLP, mask_decoder.py:53-59,171-176):
//    row_ln_g / row_ln_b / row_ln_eps: LayerNorm over the output row BEFORE the activation (`Linear -> LayerNorm -> GELU` in one
//    epilogue); hyper / masks: the C (<= 4) hyper-network dot products of every finished row, masks[z, c, n] = sum_e hyper[z, c, e]
//    out[z * hyper_rows + n, e], so the [rows, 256] activation itself need not be stored (no_store).  Both are ROW passes over the
//    staged stripe -- lane -> (row lane & 31, column half lane >> 5), 128 sequential columns per lane and ONE cross-lane exchange per
//    row -- not per-row wave reductions: with one wave per SIMD (the 133 KiB tile) 32 dependent shuffle chains per stripe measured
//    ~0.25 ms per GEMM, several times the K loop.
//  * ln_mean / ln_rstd / ln_c: the LayerNorm of the A rows folded into this GEMM -- C = rstd[row] * (A W'^T - mean[row] * c[col]) + bias
//    with W' = W * gamma (columns), c = W' 1, bias = W beta + b.
template <int TM, int TN, bool SCALED = false, bool EXT = false, typename ArgsT>
__device__ __forceinline__ void gemm_store_tile(const ArgsT& p, ep_f32x16 (&acc)[TM][TN], float* __restrict__ lw, int row_base, int col_base,
                                                int lane, float* __restrict__ C, const float* __restrict__ R, const EpPre<TM>* pre = nullptr) {
    const int r32 = lane & 31, h = lane >> 5;
    constexpr int LD = TN * 32 + 4;
    const bool interior = gemm_epilogue_interior<TM, TN>(p, row_base, col_base, C, R);   // wave-uniform
    const EpLane<TN> L(p, col_base, lane);
    const bool swiglu = L.swiglu, lane_on = L.lane_on;
    const float alpha = L.alpha;
    const int rpp = L.rpp, np = L.np, rl0 = L.rl0, c4 = L.c4, scol = L.scol, pcol = L.pcol, ocol = L.ocol;
    [[maybe_unused]] const int c4n = L.c4n;
    const EpPre<TM> own = pre ? *pre : gemm_epilogue_prefetch<TM, TN, SCALED, EXT>(p, row_base, col_base, lane, C, R);
    const ep_f32x4 b0 = own.b0, b1 = own.b1, lnc = own.lnc;
    ep_f32x4 m0, m1;
#pragma unroll
    for (int e = 0; e < 4; ++e) { m0[e] = alpha * inv_pow2(own.m0[e]); m1[e] = alpha * inv_pow2(own.m1[e]); }
    const EpRows<TM>& rows = own.rows;
    constexpr bool FULLROW = EXT && TN == 8;
    constexpr bool ALL_ON = TN != 3;       // 64 % (float4 per row) == 0: every lane carries a float4 of every pass
    constexpr bool HYPER = EXT && (TN == 8 || TN == 2);      // hyper products: full rows (one plane) or 64-column wave tiles (N / 64 partial planes)
    [[maybe_unused]] ep_f32x4 rg = {1.f, 1.f, 1.f, 1.f}, rbt = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FULLROW) { if (interior && p.row_ln_g) { rg = ep_load4(p.row_ln_g + pcol); rbt = ep_load4(p.row_ln_b + pcol); } }
    [[maybe_unused]] ep_f32x4 gm = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};      // running group maximum of this lane's 4 columns (EXT)
    constexpr int NPMAX = TN * 8 > 32 ? 32 : (TN * 8 > 16 ? 16 : 8);

    // one 32-row stripe of the wave tile; instantiated per stripe index (a rolled loop would index the accumulator array at run time and
    // send all of it through scratch memory: seen with TM = 4)
    auto stripe = [&](auto i_c) {
        constexpr int i = decltype(i_c)::value;
        // residual rows of the whole stripe and (when one row-bias row covers the stripe: rowgroup % 32 == 0) the row bias: in flight while the
        // stripe is staged and its passes run
        // (the small wave tiles run at three or four waves per SIMD and must stay inside 128 / 168 registers: they prefetch four passes' worth and
        // fetch the rest, chunk by chunk, in the tail -- their co-resident waves cover that latency)
        constexpr int RP = TM * TN <= 2 ? 4 : (NPMAX < 8 ? NPMAX : 8);
        ep_f32x4 res[RP];
        ep_f32x4 rb_s = {0.f, 0.f, 0.f, 0.f};
        const bool rb_uniform = p.rowbias && (p.rowgroup & 31) == 0;
        auto load_res = [&](int c0) {
#pragma unroll
            for (int q = 0; q < RP; ++q)
                if (c0 + q < np) res[q] = ep_load4(R + (int64_t)(row_base + i * 32 + (c0 + q) * rpp + rl0) * p.ldr + ocol);
        };
        if (interior && lane_on) {
            if (R) load_res(0);
            if (rb_uniform) rb_s = ep_load4(p.rowbias + (int64_t)((row_base + i * 32) / p.rowgroup) * p.ldrb + pcol);
        }
        // ---- stage this 32-row stripe (pure transposition)
#ifdef PSAM_GEMM_ABLATE
        if constexpr (EXT) { if (p.epi_abl & 2) goto staged; }
#endif
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) lw[((r & 3) + 8 * (r >> 2) + 4 * h) * LD + j * 32 + r32] = acc[i][j][r];
#ifdef PSAM_GEMM_ABLATE
    staged:
        if constexpr (EXT) { if (p.epi_abl & 2) { float sum = 0.f; for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) sum += acc[i][j][r]; if (sum == 123.456f) C[0] = sum; } }
#endif
        ep_wave_sync();
        [[maybe_unused]] float row_mean = 0.f, row_rstd = 1.f;      // of row (lane & 31) of this stripe (FULLROW, row_ln)
        [[maybe_unused]] bool finalized = false;
        if constexpr (FULLROW) {
            if (interior && p.row_ln_g && !swiglu) {
                // finish the stripe in place (un-scale, bias) and take each row's LayerNorm statistics, two-pass
                float* rowp = lw + r32 * LD + h * (TN * 16);
                const int colh = col_base + h * (TN * 16);
                float rsr = 1.f;
                if constexpr (SCALED) rsr = inv_pow2(p.scaleA[row_base + i * 32 + r32]);
                float sm = 0.f;
#pragma unroll 4
                for (int c = 0; c < TN * 4; ++c) {
                    ep_f32x4 x = ep_load4(rowp + 4 * c);
                    ep_f32x4 mm = {alpha, alpha, alpha, alpha}, bb = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (SCALED) {
                        const ep_f32x4 sw = ep_load4(p.scaleW + colh + 4 * c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) mm[e] *= inv_pow2(sw[e]);
                    }
                    if (p.bias) bb = ep_load4(p.bias + colh + 4 * c);
                    x = x * rsr * mm + bb;
                    *reinterpret_cast<ep_f32x4*>(rowp + 4 * c) = x;
                    sm += (x[0] + x[1]) + (x[2] + x[3]);
                }
                sm += __shfl_xor(sm, 32, 64);
                const float inv_n = 1.0f / (float)(TN * 32);
                row_mean = sm * inv_n;
                float qq = 0.f;
#pragma unroll 4
                for (int c = 0; c < TN * 4; ++c) {
                    const ep_f32x4 x = ep_load4(rowp + 4 * c);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = x[e] - row_mean; qq += d * d; }
                }
                qq += __shfl_xor(qq, 32, 64);
                row_rstd = 1.0f / sqrtf(qq * inv_n + p.row_ln_eps);
                finalized = true;
                ep_wave_sync();
            }
        }
        if (interior) {
            // ---- the passes of the stripe, as a ROLLED loop (one copy of the code per stripe, resident in the instruction cache after its first
            // pass) that is SPECIALISED on what the launch asks of the epilogue.  History (256x256 ping-pong tile, qkv shape, 32 passes per wave;
            // profiles/r03/r03_epi_*.txt): fully unrolled with every option a run-time branch, 31 us of an 95 us kernel (each pass a separate cold copy
            // of the code); rolled, 20 us -- of which 12 us were the ~110 instructions and ~15 scalar branches of ONE generic pass body and 5 us
            // the stores.  The option set of a launch is fixed, so the loop is instantiated for the combinations the model's GEMMs use (F >= 0:
            // every option test folds at compile time) and once generically (F < 0).
            auto passes = [&](auto f_c) {
                constexpr int F = decltype(f_c)::value;
                const bool o_swiglu = F < 0 ? swiglu : bool(F & EP_SWIGLU);
                const bool o_res = F < 0 ? (R != nullptr) : bool(F & EP_RES);
                const bool o_rowbias = F < 0 ? (p.rowbias != nullptr) : bool(F & EP_ROWBIAS);
                const bool o_rb_uniform = F < 0 ? rb_uniform : true;
                const int o_act = F < 0 ? p.act : ((F & EP_GELU) ? 1 : ((F & EP_RELU) ? 2 : 0));
                bool o_stats = false, o_pack = false, o_lnc = false, o_gmax = false, o_hyper = false, o_nostore = false;
                if constexpr (EXT) {
                    o_stats = F < 0 ? (p.stats != nullptr) : bool(F & EP_STATS);
                    o_pack = F < 0 ? (p.pack_out != 0) : bool(F & EP_PACK);
                    o_lnc = F < 0 ? (p.ln_c != nullptr) : bool(F & EP_LNC);
                    o_gmax = F < 0 ? (p.gmax_out != nullptr) : bool(F & EP_GMAX);
                    o_hyper = HYPER && (F < 0 ? (p.hyper != nullptr) : bool(F & EP_HYPER));
                    o_nostore = F < 0 ? (p.no_store != 0) : bool(F & EP_NOSTORE);
                }
                bool o_bnd = false;      // packed output scaled by a per-row bound the caller supplies (psam_gemm_fuse_t.out_bound)
                if constexpr (EXT) o_bnd = F < 0 ? (o_pack && !o_lnc && p.out_bound != nullptr) : bool(F & EP_BND);
                // what happens to a finished value (after the residual): group maximum, write-back for the hyper row pass, packed or plain store
                auto finish = [&](ep_f32x4 v, int rl, int row, float rsq, float obnd) {
                    if (!o_swiglu) {
                        if constexpr (EXT) {
                            if (o_gmax) gm = ep_f32x4{fmaxf(gm[0], v[0]), fmaxf(gm[1], v[1]), fmaxf(gm[2], v[2]), fmaxf(gm[3], v[3])};
                        }
                        if constexpr (HYPER) {
                            if (o_hyper) *reinterpret_cast<ep_f32x4*>(lw + rl * LD + scol) = v;      // finished value back in place for the row pass below
                        }
                    }
                    bool stored = false;
                    if constexpr (EXT) {
                        if (o_pack) {      // g8-packed output with the bound-derived row scale (see the header of this function)
                            const float bnd = p.out_k1 * rsq + p.out_k2;          // rsq = 1 / scaleA[row]
                            const float so = f16_row_scale(o_bnd ? obnd : (o_swiglu ? bnd * bnd : bnd));
                            if (c4 == 0 && col_base == 0) p.out_scale[row] = so;
                            unsigned h0, l0, h1, l1;
                            psam_split2_f16(v[0], v[1], so, h0, l0);
                            psam_split2_f16(v[2], v[3], so, h1, l1);
                            const bool odd = lane & 1;
                            // lane ^ 1 exchange on the VALU's DPP path (quad_perm), not through the LDS crossbar (ds_bpermute): every lane is active
                            // here (packing exists for the two-tile-wide wave tiles only)
                            const unsigned r0 = (unsigned)dpp_i32<DPP_XOR1>((int)(odd ? h0 : l0)), r1 = (unsigned)dpp_i32<DPP_XOR1>((int)(odd ? h1 : l1));
                            typedef unsigned ep_u32x4 __attribute__((ext_vector_type(4)));
                            *reinterpret_cast<ep_u32x4*>(C + (int64_t)row * p.ldc + ocol) = odd ? ep_u32x4{r0, r1, l0, l1} : ep_u32x4{h0, h1, r0, r1};
                            stored = true;
                        }
                    }
                    if (o_nostore) stored = true;
                    if (!stored) *reinterpret_cast<ep_f32x4*>(C + (int64_t)row * p.ldc + ocol) = v;
                };
                // operands of a pass that come through LDS (staged float4s, row scalars by lane shuffle: every lane takes part)
                struct PassIn { ep_f32x4 v, x; float rsq, lmean, lrstd; };
                auto fetch = [&](int q) {
                    PassIn o;
                    const int rl = q * rpp + rl0;
                    const int srcl = ((i & 1) * 32 + rl) & 63;
                    o.rsq = inv_pow2(__shfl(rows.rs[i >> 1], srcl, 64));
                    o.lmean = 0.f; o.lrstd = 1.f;
                    if (o_lnc) { o.lmean = __shfl(rows.mean[i >> 1], srcl, 64); o.lrstd = __shfl(rows.rstd[i >> 1], srcl, 64); }
                    else if (o_bnd) o.lmean = __shfl(rows.mean[i >> 1], srcl, 64);
                    o.v = ep_f32x4{0.f, 0.f, 0.f, 0.f}; o.x = o.v;
                    if (ALL_ON || lane_on) {
                        o.v = ep_load4(lw + rl * LD + scol);
                        if (o_swiglu) o.x = ep_load4(lw + rl * LD + scol + 32);
                    }
                    return o;
                };
                int np_run = np;
#ifdef PSAM_GEMM_ABLATE
                if constexpr (EXT) { if (p.epi_abl & 1) np_run = 0; }
#endif
#pragma unroll 1
                for (int q = 0; q < np_run; ++q) {
                    // (the round-3 version fetched pass q+1 here, one pass ahead; that form of the loop compiled to `v_pk_mul_f32 .. op_sel:[0,1]` for
                    // lnc * mean, the instruction behind the 1e-2 multi-stream errors -- see ep_lgkm_drain() and profiles/r04/r04_hazard.txt;
                    // tests/test_gpu_kernels.py::test_fused_mlp_bitwise_stable_beside_other_streams and the ISA lint hold the line.)
                    const PassIn cur = fetch(q);
                    const int rl = q * rpp + rl0;
                    const int row = row_base + i * 32 + rl;
                    const float rsq = cur.rsq;
                    if (ALL_ON || lane_on) {
                        ep_f32x4 v = cur.v;
                        if (!finalized) {
                            if constexpr (SCALED) v *= rsq;
                            if (o_lnc) { v = (v * m0 - lnc * cur.lmean) * cur.lrstd + b0; } else v = v * m0 + b0;
                        }
                        if (o_swiglu) {
                            ep_f32x4 x = cur.x;
                            if constexpr (SCALED) x *= rsq;
                            x = x * m1 + b1;
                            v = ep_f32x4{silu(v[0]) * x[0], silu(v[1]) * x[1], silu(v[2]) * x[2], silu(v[3]) * x[3]};
                            ep_lgkm_drain();
                            if constexpr (EXT) {
                                if (o_stats) {     // LayerNorm partials of this row over the wave's TN*16 gated columns (valid ones: < stat_cols)
                                    const int seg0 = col_base >> 1, nv = p.stat_cols - seg0 < TN * 16 ? (p.stat_cols - seg0 > 0 ? p.stat_cols - seg0 : 0) : TN * 16;
                                    float sm = 0.f;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) sm += (ocol + e < p.stat_cols) ? v[e] : 0.f;
                                    sm = ep_group_sum<TN * 4>(sm);
                                    const float mean = nv > 0 ? sm / (float)nv : 0.f;
                                    float m2 = 0.f;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) { const float d = (ocol + e < p.stat_cols) ? v[e] - mean : 0.f; m2 += d * d; }
                                    m2 = ep_group_sum<TN * 4>(m2);
                                    if (c4 == 0) {
                                        float* st = p.stats + ((int64_t)row * p.stat_segs + seg0 / (TN * 16)) * 2;
                                        st[0] = mean; st[1] = m2;
                                    }
                                }
                            }
                        } else {
                            if (o_rowbias) v += o_rb_uniform ? rb_s : ep_load4(p.rowbias + (int64_t)(row / p.rowgroup) * p.ldrb + pcol);
                            if constexpr (FULLROW) {
                                if (finalized) {      // LayerNorm over the 256 columns of this row (rpp == 1: row rl is wave-uniform)
                                    const float mean = __shfl(row_mean, rl, 64), rr = __shfl(row_rstd, rl, 64);
                                    v = (v - mean) * rr * rg + rbt;
                                }
                            }
                            if (o_act == 1) v = ep_f32x4{gelu_erf(v[0]), gelu_erf(v[1]), gelu_erf(v[2]), gelu_erf(v[3])};
                            else if (o_act == 2) v = ep_f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                            ep_lgkm_drain();
                        }
#ifdef PSAM_GEMM_ABLATE
                        if constexpr (EXT) { if (p.epi_abl & 4) { if (v[0] + v[1] + v[2] + v[3] == 123.456f) C[0] = v[0]; continue; } }
#endif
                        if (o_res) *reinterpret_cast<ep_f32x4*>(lw + rl * LD + scol) = v;      // parked: the residual rows (in flight since before the staging) are added below
                        else finish(v, rl, row, rsq, cur.lmean);
                    }
                }
                if (o_res) {     // residual add + store, unrolled over the prefetched rows: every parked float4 is read back first (each lane its own), ONE
                                 // drain, then the adds and stores -- no DS operation is consumed on a counted wait (see ep_lgkm_drain)
#pragma unroll
                    for (int c0 = 0; c0 < NPMAX; c0 += RP) {
                        if (c0 > 0 && c0 < np && lane_on) load_res(c0);
                        ep_f32x4 pv[RP];
                        float prs[RP], pbd[RP];
#pragma unroll
                        for (int q = 0; q < RP; ++q) {
                            prs[q] = 1.f; pbd[q] = 0.f;
                            if (c0 + q < np) {
                                const int rl = (c0 + q) * rpp + rl0;
                                if (o_pack) prs[q] = inv_pow2(__shfl(rows.rs[i >> 1], ((i & 1) * 32 + rl) & 63, 64));
                                if (o_bnd) pbd[q] = __shfl(rows.mean[i >> 1], ((i & 1) * 32 + rl) & 63, 64);
                                if (ALL_ON || lane_on) pv[q] = ep_load4(lw + rl * LD + scol);
                            }
                        }
                        ep_lgkm_drain();
#pragma unroll
                        for (int q = 0; q < RP; ++q) {
                            if (c0 + q < np) {
                                const int rl = (c0 + q) * rpp + rl0;
                                if (ALL_ON || lane_on) finish(pv[q] + res[q], rl, row_base + i * 32 + rl, prs[q], pbd[q]);
                            }
                        }
                    }
                }
            };
            // option set of this launch -> its specialised instance (the FULLROW row-LayerNorm path and per-row row biases stay generic)
            int opt = -1;
            if (!finalized && (!p.rowbias || rb_uniform)) {
                opt = (swiglu ? EP_SWIGLU : 0) | (R ? EP_RES : 0) | (p.rowbias ? EP_ROWBIAS : 0) | (!swiglu && p.act == 1 ? EP_GELU : 0) | (!swiglu && p.act == 2 ? EP_RELU : 0);
                if constexpr (EXT) {
                    opt |= (p.stats ? EP_STATS : 0) | (p.pack_out ? EP_PACK : 0) | (p.ln_c ? EP_LNC : 0) | (p.gmax_out ? EP_GMAX : 0) | ((HYPER && p.hyper) ? EP_HYPER : 0) |
                           (p.no_store ? EP_NOSTORE : 0) | ((p.pack_out && !p.ln_c && p.out_bound) ? EP_BND : 0);
                }
            }
            using std::integral_constant;
            if (opt == 0) passes(integral_constant<int, 0>{});                                                  // qkv, patch_proj, ...: bias only
            else if (opt == EP_RES) passes(integral_constant<int, EP_RES>{});                                   // attention projection
            else if (EXT && opt == (EP_SWIGLU | EP_STATS | EP_PACK | EP_BND)) passes(integral_constant<int, EP_SWIGLU | EP_STATS | EP_PACK | EP_BND>{});   // fc1 of the fused EVA02 MLP
            else if (EXT && opt == (EP_SWIGLU | EP_STATS | EP_PACK)) passes(integral_constant<int, EP_SWIGLU | EP_STATS | EP_PACK>{});   // ... with the (k1 / scale + k2)^2 bound
            else if (EXT && opt == EP_PACK) passes(integral_constant<int, EP_PACK>{});                          // qkv written packed for the attention kernel
            else if (EXT && opt == (EP_LNC | EP_RES)) passes(integral_constant<int, EP_LNC | EP_RES>{});         // fc2 with the folded LayerNorm
            else if (EXT && opt == (EP_PACK | EP_GMAX)) passes(integral_constant<int, EP_PACK | EP_GMAX>{});     // PatchEncoder conv1.3
            else if (opt == EP_ROWBIAS) passes(integral_constant<int, EP_ROWBIAS>{});                           // PatchEncoder conv2.0 (x half)
            else if (EXT && opt == (EP_GMAX | EP_NOSTORE)) passes(integral_constant<int, EP_GMAX | EP_NOSTORE>{});   // PatchEncoder conv2.3
            else if (EXT && HYPER && opt == (EP_GELU | EP_HYPER | EP_NOSTORE)) passes(integral_constant<int, EP_GELU | EP_HYPER | EP_NOSTORE>{});   // upscaling MLP
            else if (opt == EP_GELU) passes(integral_constant<int, EP_GELU>{});
            else if (opt == EP_SWIGLU) passes(integral_constant<int, EP_SWIGLU>{});
            else passes(integral_constant<int, -1>{});
            if constexpr (HYPER) {
                if (p.hyper && !swiglu) {      // hyper-network dot products of the stripe's finished rows (hyper_rows % 32 == 0: one z per stripe)
                    // over this wave's TN*32 columns: the whole row (TN == 8), or plane col_base / 64 of N / 64 partial sums (psam_sum_planes adds them)
                    ep_wave_sync();
                    const int row0 = row_base + i * 32, z = row0 / p.hyper_rows, n0 = row0 - z * p.hyper_rows;
                    const float* rowp = lw + r32 * LD + h * (TN * 16);
                    const float* hb = p.hyper + (int64_t)z * p.hyper_c * p.N + col_base + h * (TN * 16);
                    float* mk = p.masks + (int64_t)(col_base / (TN * 32)) * p.hyper_pstride;
                    float dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
                    for (int c = 0; c < TN * 4; ++c) {
                        const ep_f32x4 x = ep_load4(rowp + 4 * c);
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc)
                            if (cc < p.hyper_c) {
                                const ep_f32x4 hv = ep_load4(hb + (int64_t)cc * p.N + 4 * c);
                                dot[cc] += (x[0] * hv[0] + x[1] * hv[1]) + (x[2] * hv[2] + x[3] * hv[3]);
                            }
                    }
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        dot[cc] += __shfl_xor(dot[cc], 32, 64);
                        if (cc < p.hyper_c && h == 0) mk[((int64_t)z * p.hyper_c + cc) * p.hyper_rows + n0 + r32] = dot[cc];
                    }
                }
            }
            if constexpr (EXT) {
                if (p.gmax_out && (p.gmax_k == 32 || (i & 1))) {      // a whole group of rows has passed through this lane's columns
#pragma unroll
                    for (int o = c4n; o < 64; o <<= 1)
#pragma unroll
                        for (int e = 0; e < 4; ++e) gm[e] = fmaxf(gm[e], __shfl_xor(gm[e], o, 64));
                    if (rl0 == 0) {
                        const int grp = (row_base + i * 32) / p.gmax_k;      // gmax_k = 64: stripe i - 1 starts the group, same quotient
                        *reinterpret_cast<ep_f32x4*>(p.gmax_out + (int64_t)grp * p.gmax_ld + pcol) = gm;
                    }
                    gm = ep_f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                }
            }
        } else {
            // ---- edge tiles: one column per lane, rolled over the rows, everything bounds-checked
            const int wcols = swiglu ? TN * 16 : TN * 32;
#pragma unroll 1
            for (int c = lane; c < wcols; c += 64) {
                const int sc = swiglu ? (c >> 5) * 64 + (c & 31) : c;      // staged column (of g when gated)
                const int pc = col_base + sc;                              // column in N
                const int oc = swiglu ? (col_base >> 1) + c : pc;          // output column
                if (pc >= p.N) continue;
                float mg = alpha, mx = alpha;
                if constexpr (SCALED) { mg *= inv_pow2(p.scaleW[pc]); if (swiglu && pc + 32 < p.N) mx *= inv_pow2(p.scaleW[pc + 32]); }
                const float bg = p.bias ? p.bias[pc] : 0.f, bx = (swiglu && p.bias && pc + 32 < p.N) ? p.bias[pc + 32] : 0.f;
#pragma unroll 1
                for (int rl = 0; rl < 32; ++rl) {
                    const int row = row_base + i * 32 + rl;
                    if (row >= p.M) break;
                    float rm = 1.f;
                    if constexpr (SCALED) rm = inv_pow2(p.scaleA[row]);
                    float v = lw[rl * LD + sc] * rm * mg + bg;
                    if (swiglu) {
                        v = silu(v) * (lw[rl * LD + sc + 32] * rm * mx + bx);
                    } else {
                        if (p.rowbias) v += p.rowbias[(int64_t)(row / p.rowgroup) * p.ldrb + pc];
                        v = ep_act(v, p.act);
                        if (R) v += R[(int64_t)row * p.ldr + pc];
                    }
                    C[(int64_t)row * p.ldc + oc] = v;
                }
            }
        }
        ep_wave_sync();   // the stripe is consumed before the next one overwrites it
    };
    stripe(std::integral_constant<int, 0>{});
    if constexpr (TM > 1) stripe(std::integral_constant<int, 1>{});
    if constexpr (TM > 2) stripe(std::integral_constant<int, 2>{});
    if constexpr (TM > 3) stripe(std::integral_constant<int, 3>{});
    static_assert(TM <= 4, "stripes instantiated for TM <= 4");
}

// Wide wave tiles (one wave owns TN > 4 column tiles of 32 -- the right-sized 256x224 / 256x192 ping-pong configurations, whose waves span the
// whole tile width): the epilogue runs per CHUNK of two column tiles, the shape every fused extra (packed output, row statistics, SwiGLU pairs,
// folded LayerNorm) is written for: 8.5 KiB of LDS per wave instead of 32 x (TN * 32 + 4) floats, and the accumulators of a chunk are dead
// once it is stored.  The chunks are instantiated one after the other (compile-time accumulator indices; a rolled loop selecting the chunk's
// registers at run time kept all TN tiles live beside the pass loop's own registers and spilled ~150 VGPRs); inside a chunk the pass loop stays
// rolled and specialised, so what runs per chunk is one short instruction stream.  A chunk whose second tile lies outside N (N % 64 == 32, or
// the odd last tile of an odd TN) runs the one-tile instance; chunks entirely outside N are skipped.
template <int TM, int TN, bool SCALED = false, bool EXT = false, typename ArgsT>
__device__ __forceinline__ void gemm_store_tile_chunked(const ArgsT& p, ep_f32x16 (&acc)[TM][TN], float* __restrict__ lw, int row_base, int col_base,
                                                        int lane, float* __restrict__ C, const float* __restrict__ R) {
    auto chunk = [&](auto c_c) {
        constexpr int c = decltype(c_c)::value;
        if constexpr (2 * c < TN) {
            const int cb = col_base + c * 64;
            if (cb >= p.N) return;                              // wave-uniform
            const bool single = (2 * c + 1 >= TN) || (cb + 32 >= p.N);
            if (single) {
                ep_f32x16 sub[TM][1];
#pragma unroll
                for (int i = 0; i < TM; ++i) sub[i][0] = acc[i][2 * c];
                gemm_store_tile<TM, 1, SCALED, EXT>(p, sub, lw, row_base, cb, lane, C, R, nullptr);
            } else if constexpr (2 * c + 1 < TN) {
                ep_f32x16 sub[TM][2];
#pragma unroll
                for (int i = 0; i < TM; ++i) { sub[i][0] = acc[i][2 * c]; sub[i][1] = acc[i][2 * c + 1]; }
                gemm_store_tile<TM, 2, SCALED, EXT>(p, sub, lw, row_base, cb, lane, C, R, nullptr);
            }
        }
    };
    using std::integral_constant;
    chunk(integral_constant<int, 0>{}); chunk(integral_constant<int, 1>{}); chunk(integral_constant<int, 2>{}); chunk(integral_constant<int, 3>{});
    static_assert(TN <= 8, "chunks instantiated for TN <= 8");
}
