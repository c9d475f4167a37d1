// Packed-operand fp32-grade GEMM, LOCK-STEP schedule on a ring of k16 UNITS ("f16x3q", round 4): same contract, operands and arithmetic as
// gemm_f16x3p.hip / gemm_f16x3pp.hip (C = act(alpha A W^T + bias) + residual on g8-packed row-scaled hi|lo fp16 operands, hi*lo + lo*hi + hi*hi
// on v_mfma_f32_32x32x16_f16 in the same order => the same bits), register-only epilogue (gemm_epilogue_t.h) only.
//
// Why another structure.  The 128x128 tile of the shipped configuration runs two workgroups per CU -- one workgroup's prologue, epilogue and
// barrier waits are covered by the other's MFMAs -- but every MFMA of it pulls 341 B through the CU's texture path (L2 -> LDS by LDS-DMA),
// and that path, not the matrix pipe, bounds it: LDS-DMA + fragment reads alone take 184 k cycles of the qkv GEMM against 104 k for its MFMAs
// (profiles/r02/r02_gemm_clock_counters.txt); the texture path delivers at most 64 B / clk / CU and in practice a third of that.  The 256-row
// ping-pong tiles halve the bytes per MFMA but hold a whole CU each (160 KiB, 8 waves): no second workgroup, everything outside the K loop
// exposed.  This kernel takes the middle: ONE k16 step per ring unit (64-byte rows), so a 128x256 tile needs 24 KiB per unit and a ring of
// three units 72 KiB -- TWO four-wave workgroups per CU with 256 B per MFMA (-25 %), wave tiles of 64x128 (128 accumulator registers).
//
//   unit u = k16 step u: [BM rows of A | BN rows of W] x 64 B = [hi k0-7 | lo k0-7 | hi k8-15 | lo k8-15]; 1 KiB DMA pieces of 16 rows;
//   chunk c of row r is stored at c ^ ((r >> 2) & 3) (on the SOURCE address: the DMA writes LDS linearly) -> conflict-free ds_read_b128.
//   per step: counted vmcnt (unit t landed, S - 2 later units may be in flight) | s_barrier (unit t visible, unit t - 1 free) | fragment reads of
//   unit t, then its 3 TM TN MFMAs with the DMA issues of unit t + S - 1 spread behind them.
#include <type_traits>
#include "common.h"
#include "gemm_f16x3p_args.h"
#include "gemm_epilogue.h"
#include "gemm_epilogue_t.h"

template <int WM, int WN, int TM, int TN, int S>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_f16x3q_kernel(const F16PArgs p) {
    constexpr int NW = WM * WN, BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int ROWB = 64;
    constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, UNIT = A_BYTES + W_BYTES;
    constexpr int NBLK = UNIT / 1024, A_BLK = A_BYTES / 1024;
    static_assert(NBLK % NW == 0 && S >= 3, "pieces divide among the waves; ring of at least three units");
    constexpr int NL = NBLK / NW;
    constexpr int NMF = 3 * TM * TN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- tile of this workgroup (tile order: gemm_f16x3p.hip)
    const int ntiles = p.tiles_m * p.tiles_n;
    int tile = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, x = tile & 7, y = tile >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int pfull = p.tiles_m * p.panel, pn = tile / pfull, prem = tile - pn * pfull;
    const int pw = p.tiles_n - pn * p.panel < p.panel ? p.tiles_n - pn * p.panel : p.panel;
    const int m0 = (prem / pw) * BM, n0 = (pn * p.panel + prem % pw) * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int r32 = lane & 31, h = lane >> 5;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda * 4), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.ldw * 4), 0, 0x7fffffff, 0x00020000);
    int voff[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int b = wave + i * NW;
        const bool isw = b >= A_BLK;
        const int row = (isw ? b - A_BLK : b) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        const int lim = isw ? p.N - n0 : p.M - m0;
        const int rc = row < lim ? row : lim - 1;
        voff[i] = (int)((int64_t)rc * (isw ? p.ldw : p.lda) * 4) + chunk * 16;
    }
    auto issue_one = [&](int i, int step, int unit) {
        const int b = wave + i * NW;
        unsigned char* dst = smem + unit * UNIT + b * 1024;
        if (b >= A_BLK) P_DMA16(rsW, dst, voff[i], step * ROWB);
        else P_DMA16(rsA, dst, voff[i], step * ROWB);
    };
    int fa_off[2], fw_off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int f = r32 * ROWB + (((2 * h + q) ^ ((r32 >> 2) & 3)) << 4);
        fa_off[q] = wm * TM * 32 * ROWB + f;
        fw_off[q] = A_BYTES + wn * TN * 32 * ROWB + f;
    }
    pf32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nsteps = p.K / 16;                    // K % 32 == 0, K >= 128: nsteps >= 8 > S
    // ---- prologue: S - 1 units in flight
#pragma unroll
    for (int u = 0; u < S - 1; ++u) {
#pragma unroll
        for (int i = 0; i < NL; ++i) issue_one(i, u, u);
    }
    int un = 0;                                     // ring unit of the current step
    auto step = [&](int t, auto issue_c, auto later_c) {
        constexpr bool DO_ISSUE = decltype(issue_c)::value;
        constexpr int LATER = decltype(later_c)::value;            // units issued after unit t that may stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LATER * NL) : "memory");
        asm volatile("s_barrier" ::: "memory");
        const int un_issue = un == 0 ? S - 1 : un - 1;             // unit of step t - 1 = unit of step t + S - 1
        pf16x8 fa[TM][2], fw[TN][2];
        const unsigned char* base = smem + un * UNIT;
        // fragment reads in the order of first use (term 0 = A hi x W lo, term 1 = A lo x W hi, term 2 = A hi x W hi): the first MFMAs wait for the
        // first reads only
        auto rd_a = [&](int i, int q) { fa[i][q] = *reinterpret_cast<const pf16x8*>(base + fa_off[q] + i * 32 * ROWB); };
        auto rd_w = [&](int j, int q) { fw[j][q] = *reinterpret_cast<const pf16x8*>(base + fw_off[q] + j * 32 * ROWB); };
        rd_a(0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) rd_w(j, 1);
#pragma unroll
        for (int i = 1; i < TM; ++i) rd_a(i, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) rd_a(i, 1);
#pragma unroll
        for (int j = 0; j < TN; ++j) rd_w(j, 0);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int PA[3] = {0, 1, 0}, PW[3] = {1, 0, 0};        // term-major: hi*lo, lo*hi, hi*hi (the order of gemm_f16x3p.hip)
        constexpr int GAP = NMF / NL > 0 ? NMF / NL : 1;           // one DMA issue behind every GAP-th MFMA
#pragma unroll
        for (int m = 0; m < NMF; ++m) {
            const int term = m / (TM * TN), ij = m % (TM * TN), i = ij / TN, j = ij % TN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j][PW[term]], fa[i][PA[term]], acc[i][j], 0, 0, 0);      // operands swapped: one output row per lane
            if (DO_ISSUE && m % GAP == GAP - 1 && m / GAP < NL) issue_one(m / GAP, t + S - 1, un_issue);
            __builtin_amdgcn_sched_barrier(0);
        }
        un = un == S - 1 ? 0 : un + 1;
    };
    using std::integral_constant;
    using std::true_type;
    using std::false_type;
    int t = 0;
#pragma unroll 1
    for (; t + (S - 1) < nsteps; ++t) step(t, true_type{}, integral_constant<int, S - 2>{});
    // tail: the last S - 1 steps issue nothing; `left` later units are still in flight at tail position j
    auto tail = [&](auto j_c) {
        constexpr int j = decltype(j_c)::value;
        constexpr int left = S - 2 - j;
        step(t, false_type{}, integral_constant<int, (left > 0 ? left : 0)>{});
        ++t;
    };
    tail(integral_constant<int, 0>{});
    if constexpr (S >= 3) tail(integral_constant<int, 1>{});
    if constexpr (S >= 4) tail(integral_constant<int, 2>{});
    static_assert(S <= 4, "tail unrolled for S <= 4");

    gemm_store_tile_t<TM, TN>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, lane, p.C, p.residual);
}

template <int WM, int WN, int TM, int TN, int S>
static int32_t launch_q(F16PArgs& p, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NW = WM * WN;
    constexpr int lds = S * (BM + BN) * 64;
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (TN % 2 != 0 && p.act == 3) {
        psam_set_error("psam_gemm_f16x3p: this tile configuration cannot apply the SwiGLU epilogue");
        return PSAM_EINVAL;
    }
    p.tiles_m = (int)psam_cdiv(p.M, BM);
    p.tiles_n = (int)psam_cdiv(p.N, BN);
    p.panel = f16x3p_panel(p.tiles_m, p.tiles_n, BM, BN, p.K);
    static unsigned long long attr_done = 0;
    if (!f16x3p_reserve_lds(&gemm_f16x3q_kernel<WM, WN, TM, TN, S>, lds, attr_done)) {
        psam_set_error("psam_gemm_f16x3p: cannot reserve LDS");
        return PSAM_EINVAL;
    }
    hipLaunchKernelGGL((gemm_f16x3q_kernel<WM, WN, TM, TN, S>), dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(64 * NW), lds, stream, p);
    return psam_launch_status("psam_gemm_f16x3p: launch failed");
}

// cfg 80 ..: configurations of this kernel; every one needs the register epilogue (f16x3p_use_register_epilogue) and no split-K
bool f16x3q_supports(int cfg, const F16PArgs& p) {
    if (cfg < 80 || cfg > 84 || !f16x3p_use_register_epilogue(p) || p.hyper || p.ksplit > 1) return false;
    const int wt_m = cfg == 83 ? 32 : 64, wt_n = cfg == 80 ? 128 : (cfg == 83 ? 256 : (cfg == 84 ? 96 : 64));      // wave tile
    if (p.act == 3 && cfg == 84) return false;
    const bool fused = p.pack_out || p.stats || p.ln_c;
    if (fused && (p.M % wt_m != 0 || p.N % wt_n != 0)) return false;      // the fused extras exist for interior wave tiles only
    return true;
}

int32_t launch_f16x3q(int cfg, F16PArgs& p, hipStream_t stream) {
    switch (cfg) {
        case 80: return launch_q<2, 2, 2, 4, 3>(p, stream);      // 128x256, 4 waves of 64x128, 3 units (72 KiB): two workgroups per CU, 256 B per MFMA
        case 81: return launch_q<2, 2, 2, 2, 3>(p, stream);      // 128x128, 4 waves of 64x64, 3 units (48 KiB): three per CU (A/B against cfg 21)
        case 82: return launch_q<2, 2, 2, 2, 4>(p, stream);      // 128x128, 4 units (64 KiB): two per CU
        case 83: return launch_q<4, 1, 1, 8, 3>(p, stream);      // 128x256, 4 waves of 32x256
        case 84: return launch_q<2, 2, 2, 3, 3>(p, stream);      // 128x192, 4 waves of 64x96 (60 KiB; no SwiGLU): qkv 4096x3072 = 512 tiles
        default: break;
    }
    psam_set_error("psam_gemm_f16x3p: unknown unit-ring config");
    return PSAM_EINVAL;
}
