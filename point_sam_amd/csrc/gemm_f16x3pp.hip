// Packed-operand fp32-grade GEMM, PING-PONG schedule ("f16x3pp"): same contract, operands, arithmetic and epilogues as gemm_f16x3p.hip
// (C = act(alpha A W^T + bias + rowbias) + residual on g8-packed row-scaled hi|lo fp16 operands, hi*lo + lo*hi + hi*hi on
// v_mfma_f32_32x32x16_f16, fp32 accumulation in the same order => bit-identical results), a different execution structure.
//
// Why (profiles/r02/r02_gemm_clock_counters.txt): in the lock-step ring kernel every wave of a CU reads fragments at the same time, issues
// its MFMAs at the same time and waits at the slab barrier at the same time -- the matrix pipe idles while the data path runs (busy
// 0.44), and with two 128x128 workgroups per CU the L2 -> LDS path carries 42 B/clk/CU at full MFMA rate against ~32 B/clk/CU measured.
// Here ONE workgroup of 8 waves owns a CU and a tile of up to 256x256 (half the operand bytes per flop), and its waves form two groups
// of four -- one wave of each group per SIMD -- that run the SAME program one barrier interval apart:
//
//      interval      group 0                         group 1
//      I(2p)         LOAD(p): fragment reads,        COMPUTE(p-1): MFMAs only
//                    LDS-DMA issue for a later slab
//      I(2p+1)       COMPUTE(p): MFMAs only          LOAD(p)
//
// so on every SIMD one wave feeds the matrix pipe while the other does all of its LDS / DMA work, and the roles swap at each barrier.
// A phase p covers P k16-steps (P = 1: 24 MFMAs per wave for a 128x64 wave tile; P = 2: a whole 32-k slab); fragments are single
// buffered (loaded in one interval, consumed in the next).
//
// Ring of S units, one k16 step each (64 B per row: DMA pieces of 16 rows x 64 B, chunk swizzle on the SOURCE address).  Every phase is the
// same: LOAD(ph) reads the phase's P units, then issues this wave's pieces of the P steps that take the units phase ph-1 occupied (free:
// group 1 finished reading them one barrier ago) -- an even DMA stream, S - 2P steps of look-ahead beyond the next phase.  A wave waits
// (counted vmcnt) for its own pieces of phase q before the barrier that precedes the interval of phase q's first read: group 0 at the end
// of COMPUTE(q-1), group 1 at the end of LOAD(q-1) -- the same physical barrier.
// (First version of this kernel: 32-k slabs of 128-byte rows in a 2-stage ring, refill issued in one burst -> the burst's issue time
// stretched two of every four intervals and the refill had less than a slab time to land: 98 us on qkv against 79 for the lock-step kernel.)
#include <type_traits>
#include "common.h"
#include "gemm_f16x3p_args.h"
#include "gemm_epilogue.h"
#include "gemm_epilogue_t.h"

// ABL (measurement builds, -DPSAM_GEMM_ABLATE): 1 = no epilogue, 2 = no DMA after the prologue, 4 = no MFMA, 16 = no fragment reads after
// the first phase.
// OCC: waves per SIMD the register allocation must admit (2 = one 8-wave workgroup per CU, up to 256 registers; 4 = TWO workgroups per CU, 128 registers:
// the 64x64 wave tile with single-step phases -- 64 accumulator + 32 fragment registers -- on a ring of three units (72 KiB)).
template <int GWM, int WN, int TM, int TN, int S, int P, int PRIO, int ABL = 0, int TR = 0, int OCC = 2>
__global__ __launch_bounds__(512, OCC) void gemm_f16x3pp_kernel(const F16PArgs p) {
    static_assert(GWM * WN == 4, "four waves per group");
    static_assert((P == 1 || P == 2) && S >= 2 * P && S - 2 * P <= 4, "phase = one or two k16 steps; ring of at least two phases");
    constexpr int BM = 2 * GWM * TM * 32, BN = WN * TN * 32;
    constexpr int ROWB = 64;                        // bytes per row per ring unit: one k16 step = [hi k0-7 | lo k0-7 | hi k8-15 | lo k8-15]
    constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, UNIT = A_BYTES + W_BYTES;
    constexpr int NBLK = UNIT / 1024, A_BLK = A_BYTES / 1024;       // 1 KiB DMA pieces: 16 rows x 64 B
    // Pieces of a unit are dealt to the eight waves round-robin: NL per wave, except that with NBLK % 8 != 0 (the right-sized 256x224 / 256x192
    // tiles) the waves NFULL .. 7 carry one piece less -- their counted vmcnt waits use NL - 1 per step (PP_WAITV: a wave-uniform branch over
    // two immediates), nothing is loaded twice.
    constexpr int NL = (NBLK + 7) / 8;              // DMA instructions per wave per step (waves < NFULL)
    constexpr int NFULL = NBLK - 8 * (NL - 1);      // 8 when the pieces divide evenly
    constexpr bool CHUNKED = TN > 4;                // wide wave tiles: epilogue per chunk of two column tiles (gemm_store_tile_chunked)
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
    const int grp = wave >> 2, wg = wave & 3;       // waves 0-3: group 0, 4-7: group 1 (consecutive waves go to different SIMDs)
    const int wm = grp * GWM + wg / WN, wn = wg % WN;
    const int r32 = lane & 31, h = lane >> 5;
    const bool shortw = NFULL < 8 && wave >= NFULL;      // wave-uniform
#define PP_WAITV(k)                                                                                   \
    do {                                                                                              \
        if (shortw) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((k) * (NL - 1)) : "memory");             \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((k) * NL) : "memory");                          \
    } while (0)

    // ---- DMA source offsets: piece b = wave + 8 i covers rows 16b .. 16b+15 of the unit (A rows, then W rows); lane -> (row 16b + lane/4,
    // slot lane%4), which must receive chunk slot ^ swz(row) of that row's 64 bytes, swz(r) = (r >> 2) & 3: the rows of a ds_read_b128
    // lane group that share a 64-byte quarter of the 256-byte bank row then sit on four different 16-byte slots (conflict-free).
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda * 4), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.ldw * 4), 0, 0x7fffffff, 0x00020000);
    int voff[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int b = wave + i * 8;
        const bool isw = b >= A_BLK;
        const int row = (isw ? b - A_BLK : b) * 16 + (lane >> 2);
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        const int lim = isw ? p.N - n0 : p.M - m0;
        const int rc = row < lim ? row : lim - 1;
        voff[i] = (int)((int64_t)rc * (isw ? p.ldw : p.lda) * 4) + chunk * 16;
    }
    auto issue_one = [&](int i, int step, int unit) {
        if (NFULL < 8 && i == NL - 1 && shortw) return;
        const int b = wave + i * 8;
        unsigned char* dst = smem + unit * UNIT + b * 1024;
        if (b >= A_BLK) P_DMA16(rsW, dst, voff[i], step * ROWB);
        else P_DMA16(rsA, dst, voff[i], step * ROWB);
    };

    // ---- fragment offsets inside a unit: row r32 of a 32-row tile, chunk 2 h + q (q = hi / lo plane)
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

    // the epilogue's operands (per-row inverse A scales / folded-LayerNorm statistics, the lane's column constants): loaded here, consumed
    // after the K loop
    // (not with two-step phases of the 128x64 wave tile: 128 accumulator + 96 fragment registers leave no room for them through the loop)
    constexpr bool PREFETCH_EPI = !(P == 2 && TM * TN >= 8) && !CHUNKED && !TR && OCC <= 2;
    EpPre<TM> epre;
    if constexpr (PREFETCH_EPI) epre = gemm_epilogue_prefetch<TM, TN, true, true>(p, m0 + wm * TM * 32, n0 + wn * TN * 32, lane, p.C, p.residual);
    pf16x8 fa[P][TM][2], fw[P][TN][2];
    const int nsteps = p.K / 16;                    // K % 32 == 0: even
    const int nph = nsteps / P;

    // ---- prologue: the whole ring in flight (steps 0 .. S-1; nsteps >= 8 >= S), phase 0 visible
#pragma unroll
    for (int u = 0; u < S; ++u) {
#pragma unroll
        for (int i = 0; i < NL; ++i) issue_one(i, u, u);
    }
    PP_WAITV(S - P);
    asm volatile("s_barrier" ::: "memory");
    if (grp == 1) asm volatile("s_barrier" ::: "memory");      // the stagger: group 1 runs one interval behind
    if (PRIO == 2 && grp == 1) __builtin_amdgcn_s_setprio(1);   // static priority for the younger half (MI355X_MICROARCH.md, item 4)

    // Wait until this wave's pieces of phase q (q >= 1) have landed.  At that point the wave has issued the prologue and the refills of
    // LOAD(1 .. q-1): steps up to min((q-1) P + S - 1, nsteps - 1); everything issued after the last step of phase q may stay in flight.
    auto wait_phase = [&](int q) {
        int last = (q - 1) * P + S - 1;
        last = last < nsteps - 1 ? last : nsteps - 1;
        const int later = last - (q * P + P - 1);        // 0 .. S - 2P
        if (later >= 4) PP_WAITV(4);
        else if (later == 3) PP_WAITV(3);
        else if (later == 2) PP_WAITV(2);
        else if (later == 1) PP_WAITV(1);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    int un = 0;             // ring unit of the phase's first step
    // One phase.  STEADY (compile-time): 1 <= ph <= (nsteps - S) / P -- every refill step exists and the wait count is the constant
    // (S - 2P) NL, so the loop body carries no bookkeeping between the last MFMA / ds_read and the barrier.
    auto phase = [&](int ph, auto steady_c) {
        constexpr bool STEADY = decltype(steady_c)::value;
        // ================= LOAD(ph): fragments of the phase's P steps, then the refills of the units phase ph-1 occupied
        {
            if (!((ABL & 16) && ph > 0)) {
#pragma unroll
                for (int s = 0; s < P; ++s) {
                    const int u = un + s >= S ? un + s - S : un + s;
                    const unsigned char* base = smem + u * UNIT;
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int q = 0; q < 2; ++q) fa[s][i][q] = *reinterpret_cast<const pf16x8*>(base + fa_off[q] + i * 32 * ROWB);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int q = 0; q < 2; ++q) fw[s][j][q] = *reinterpret_cast<const pf16x8*>(base + fw_off[q] + j * 32 * ROWB);
                }
            }
            if (!(ABL & 2) && (STEADY || ph >= 1)) {
#pragma unroll
                for (int s = 0; s < P; ++s) {
                    const int step = (ph - 1) * P + S + s;           // goes where step (ph-1) P + s was: group 1 finished reading it a barrier ago
                    int u = un - P + s; u = u < 0 ? u + S : u;
                    if (STEADY || step < nsteps) {
#pragma unroll
                        for (int i = 0; i < NL; ++i) issue_one(i, step, u);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (grp == 1) {
                if (STEADY) PP_WAITV(S - 2 * P);
                else if (ph + 1 < nph) wait_phase(ph + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        // ================= COMPUTE(ph)
        {
            if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < P; ++s) {
                constexpr int PA[3] = {0, 1, 0}, PW[3] = {1, 0, 0};      // term-major: hi*lo, lo*hi, hi*hi (the order of gemm_f16x3p.hip)
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if (ABL & 4) asm volatile("" ::"v"(fa[s][i][PA[term]]), "v"(fw[s][j][PW[term]]));
                            else if (TR) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[s][j][PW[term]], fa[s][i][PA[term]], acc[i][j], 0, 0, 0);
                            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s][i][PA[term]], fw[s][j][PW[term]], acc[i][j], 0, 0, 0);
                        }
            }
            if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
            if (grp == 0) {
                if (STEADY) PP_WAITV(S - 2 * P);
                else if (ph + 1 < nph) wait_phase(ph + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        un = un + P >= S ? un + P - S : un + P;
    };
    {
        const int nsteady_end = (nsteps - S) / P;       // last steady phase (>= 1: nsteps >= 8 >= S + P)
        int ph = 0;
        phase(ph++, std::false_type{});
#pragma unroll 1
        for (; ph <= nsteady_end; ++ph) phase(ph, std::true_type{});
#pragma unroll 1
        for (; ph < nph; ++ph) phase(ph, std::false_type{});
    }
    if (PRIO == 2 && grp == 1) __builtin_amdgcn_s_setprio(0);
    if (grp == 0) asm volatile("s_barrier" ::: "memory");      // balances group 1's extra barrier

    // ---- epilogue (gemm_epilogue.h): every wave is done with the ring, no DMA in flight
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ABL & 1) {
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
        if (sum == 123.456f) p.C[0] = sum;
        return;
    }
    if constexpr (TR == 2) {       // full-row tile (tiles_n == 1): Linear -> LayerNorm -> activation -> packed rows (gemm_epilogue_t.h)
        gemm_store_tile_t_rowln<TM, TN, WN>(p, acc, reinterpret_cast<float*>(smem), BM, m0 + wm * TM * 32, wm * TM * 32, wn, lane, p.C);
    } else if constexpr (TR) {
        gemm_store_tile_t<TM, TN>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, lane, p.C, p.residual);
    } else if constexpr (CHUNKED) {
        gemm_store_tile_chunked<TM, TN, true, true>(p, acc, reinterpret_cast<float*>(smem) + wave * gemm_epilogue_lds_floats_per_wave<2>(), m0 + wm * TM * 32,
                                                    n0 + wn * TN * 32, lane, p.C, p.residual);
    } else {
        if (ABL & 32) {     // everything of the epilogue except the global stores of C
            F16PArgs q = p;
            q.no_store = 1;
            gemm_store_tile<TM, TN, true, true>(q, acc, reinterpret_cast<float*>(smem) + wave * gemm_epilogue_lds_floats_per_wave<TN>(), m0 + wm * TM * 32,
                                                n0 + wn * TN * 32, lane, p.C, p.residual, PREFETCH_EPI ? &epre : nullptr);
            return;
        }
        gemm_store_tile<TM, TN, true, true>(p, acc, reinterpret_cast<float*>(smem) + wave * gemm_epilogue_lds_floats_per_wave<TN>(), m0 + wm * TM * 32,
                                            n0 + wn * TN * 32, lane, p.C, p.residual, PREFETCH_EPI ? &epre : nullptr);
    }
#undef PP_WAITV
}

template <int GWM, int WN, int TM, int TN, int S, int P, int PRIO, int ABL = 0, int TR = 0, int OCC = 2>
static int32_t launch_pp(F16PArgs& p, hipStream_t stream) {
    constexpr int BM = 2 * GWM * TM * 32, BN = WN * TN * 32;
    constexpr int ring = S * (BM + BN) * 64, epi = 8 * (TN > 4 ? gemm_epilogue_lds_floats_per_wave<2>() : gemm_epilogue_lds_floats_per_wave<TN>()) * 4;
    constexpr int lds = ring > epi ? ring : epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (TN % 2 != 0 && p.act == 3) {
        psam_set_error("psam_gemm_f16x3p: this tile configuration cannot apply the SwiGLU epilogue");
        return PSAM_EINVAL;
    }
    p.tiles_m = (int)psam_cdiv(p.M, BM);
    p.tiles_n = (int)psam_cdiv(p.N, BN);
    p.panel = f16x3p_panel(p.tiles_m, p.tiles_n, BM, BN, p.K);
    static unsigned long long attr_done = 0;
    if (!f16x3p_reserve_lds(&gemm_f16x3pp_kernel<GWM, WN, TM, TN, S, P, PRIO, ABL, TR, OCC>, lds, attr_done)) {
        psam_set_error("psam_gemm_f16x3p: cannot reserve LDS");
        return PSAM_EINVAL;
    }
    hipLaunchKernelGGL((gemm_f16x3pp_kernel<GWM, WN, TM, TN, S, P, PRIO, ABL, TR, OCC>), dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(512), lds, stream, p);
    return psam_launch_status("psam_gemm_f16x3p: launch failed");
}

// Which fused extras a ping-pong configuration's wave tile supports (same rules as the lock-step kernel: SwiGLU pairs accumulator tiles
// (even TN); row statistics and hyper products are written for two-tile-wide wave tiles; the group maximum needs 64-row wave tiles).
bool f16x3pp_supports(int cfg, int act, bool stats, bool gmax, bool hyper) {
    int tm, tn;
    switch (cfg) {
        case 50: case 51: case 52: case 59: case 60: case 61: tm = 4; tn = 2; break;
        case 53: tm = 2; tn = 4; break;
        case 55: case 56: case 65: case 66: case 67: tm = 2; tn = 2; break;
        case 57: case 58: tm = 1; tn = 2; break;
        case 62: case 63: case 64: tm = 1; tn = 2; break;      // wide wave tiles run the two-tile epilogue chunk by chunk (odd widths: no SwiGLU, checked at launch)
        default: return false;
    }
    if (act == 3 && ((tn & 1) || cfg == 62)) return false;      // SwiGLU pairs column tiles (2q, 2q+1): even widths only (cfg 62 is seven tiles wide)
    if ((stats || hyper) && tn != 2) return false;
    if (gmax && tm < 2) return false;
    return true;
}

// Where the ping-pong kernel replaces the lock-step one.  Measured (profiles/r03/r03_gemm_pp_bench.log, r03_bench_ab.log): alone it wins on fc1
// (256x128 tile: 128 vs 132 us), on the mini-PointNet's conv2.3 (256x256: 416 vs 431 us) and, as 128x128, on proj / fc2 (30 vs 31, 65.5 vs
// 67.5 us); in three-stream layer loops the 256x256 tile is the fastest arrangement measured (268 vs 281 us per layer).  In the pipelined
// benchmark (two dense streams + the tokenizer stream, HIP graphs) none of it shows: 751 clouds/s with and without it, 736 / 722 when every
// encoder GEMM runs on it -- one 8-wave workgroup per CU cannot share a CU with the other batch's kernels, and that co-scheduling is worth
// more than the kernel's own gain.  So the default is OFF.  PSAM_GEMM_PP (read once): 0 (default) = never; 1 = fc1 and conv2.3; 2 = every
// encoder-sized GEMM on the 256x128 tile as well; 3 = on the 256x256 tile where N allows.
int f16x3pp_pick(int M, int N, int K, int act) {
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("PSAM_GEMM_PP"); mode = e ? atoi(e) : 0; }
    if (mode == 0 || K < 128 || (K & 31)) return -1;
    if (M >= 32768) return (N >= 256 && (N & 127) == 0 && K >= 512) ? 51 : -1;      // mini-PointNet conv2.3: 256x256 tiles, many rounds
    if (mode == 4) {
        // right-sized tiles: among the 256-row ping-pong tiles (256x256 / 256x224 / 256x192; SwiGLU pairs need an even number of column tiles)
        // the width whose tile count fills whole rounds of #CU - 8 workgroup slots best -- wide GEMMs only (N >= 1536: with 128-column tiles
        // and one workgroup per CU the narrow ones leave half of the chip idle, the lock-step kernel keeps them)
        if (M < 2048 || (M & 255) || (N & 127) || N < 1536) return -1;
        static int ncu = 0;
        if (!ncu) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256; }
        const int slots = ncu > 16 ? ncu - 8 : ncu;
        int best = -1; double best_cost = 1e300;
        const int cfgs[3] = {51, 62, 63}, widths[3] = {256, 224, 192};
        for (int i = 0; i < 3; ++i) {
            if (act == 3 && (widths[i] / 32) % 2) continue;
            const int64_t tiles = (int64_t)(M / 256) * ((N + widths[i] - 1) / widths[i]);
            const double rounds = (double)((tiles + slots - 1) / slots);
            const double cost = rounds * widths[i] * (1.0 + 24.0 / widths[i]);      // per-round time ~ tile width + a fixed part (prologue, epilogue ramp)
            if (cost < best_cost) { best_cost = cost; best = cfgs[i]; }
        }
        return best;
    }
    if (mode >= 5 && mode <= 8) {
        // round 5: two ping-pong workgroups per CU.  5 = the wide encoder GEMMs (qkv, fc1: N >= 2048) on 256x128 tiles (cfg 65); 6 = the narrow ones (proj,
        // fc2) on the 128x128 two-per-CU tile (cfg 58) as well; 7 = as 5 with priority around the MFMAs (cfg 66); 8 = as 5 on 128x256 tiles (cfg 67)
        if (M < 2048 || (M & 255) || (N & 255)) return -1;
        if (N >= 2048) return mode == 7 ? 66 : (mode == 8 ? 67 : 65);
        return mode == 6 ? 58 : -1;
    }
    if (M >= 2048 && (M & 255) == 0 && (N & 127) == 0) {
        if (act == 3) return 55;                                                     // fc1 (SwiGLU): 256x128 tiles
        if (mode == 3) return (N & 255) == 0 ? 51 : 55;
        if (mode >= 2) return 55;
    }
    return -1;
}

int32_t launch_f16x3pp(int cfg, F16PArgs& p, hipStream_t stream) {
#ifdef PSAM_GEMM_ABLATE
    if (cfg >= 1000) {    // 1000 + epilogue ablation bits (1 = no pass loop, 2 = no staging writes, 4 = passes compute but do not store): cfg 51 otherwise
        p.epi_abl = cfg - 1000;
        return launch_pp<1, 4, 4, 2, 5, 1, 0, 0>(p, stream);
    }
    if (cfg >= 200) {     // 200 + 64 * which + ablation bits; which: 0 = cfg 50, 1 = cfg 55, 2 = cfg 57, 3 = cfg 60
        const int which = (cfg - 200) / 64, abl = (cfg - 200) % 64;
#define PP_ABL(B)                                                                \
    case B:                                                                      \
        if (which == 0) return launch_pp<1, 4, 4, 2, 5, 1, 1, B>(p, stream);     \
        if (which == 1) return launch_pp<2, 2, 2, 2, 6, 2, 1, B>(p, stream);     \
        if (which == 3) return launch_pp<1, 4, 4, 2, 5, 2, 0, B>(p, stream);     \
        return launch_pp<2, 2, 1, 2, 8, 2, 1, B>(p, stream);
        switch (abl) { PP_ABL(0) PP_ABL(1) PP_ABL(2) PP_ABL(3) PP_ABL(4) PP_ABL(5) PP_ABL(16) PP_ABL(19) PP_ABL(23) PP_ABL(32) PP_ABL(34) default: break; }
#undef PP_ABL
    }
#endif
#ifdef PSAM_PP_ONLY      // probe builds (scripts/kernel_resources.py -DPSAM_PP_ONLY=65 -save-temps): one configuration, seconds to compile
    if (cfg == PSAM_PP_ONLY) {
        switch (PSAM_PP_ONLY) {
            case 65: return launch_pp<2, 2, 2, 2, 3, 1, 0, 0, 0, 4>(p, stream);
            case 58: return launch_pp<2, 2, 1, 2, 4, 2, 1>(p, stream);
            default: break;
        }
    }
    return PSAM_EINVAL;
#else
    switch (cfg) {
        case 50: return launch_pp<1, 4, 4, 2, 5, 1, 1>(p, stream);      // 256x256, waves of 128x64, 5 units (160 KiB), priority around the MFMAs
        case 51: return f16x3p_use_register_epilogue(p) ? launch_pp<1, 4, 4, 2, 5, 1, 0, 0, 1>(p, stream) : launch_pp<1, 4, 4, 2, 5, 1, 0>(p, stream);      //   no priority changes
        case 52: return launch_pp<1, 4, 4, 2, 5, 1, 2>(p, stream);      //   static priority for group 1
        case 59: return launch_pp<1, 4, 4, 2, 4, 1, 1>(p, stream);      //   4 units (128 KiB)
        case 60: return launch_pp<1, 4, 4, 2, 5, 2, 0>(p, stream);      //   two-step phases (48 MFMAs between barriers), 5 units
        case 61: return launch_pp<1, 4, 4, 2, 4, 2, 0>(p, stream);      //   two-step phases, 4 units
        case 53: return launch_pp<2, 2, 2, 4, 5, 1, 1>(p, stream);      // 256x256, waves of 64x128
        case 55: return f16x3p_use_register_epilogue(p) ? launch_pp<2, 2, 2, 2, 6, 2, 1, 0, 1>(p, stream)
                                                        : launch_pp<2, 2, 2, 2, 6, 2, 1>(p, stream);      // 256x128, waves of 64x64, 6 units (144 KiB), two-step phases (24 MFMAs)
        case 56: return launch_pp<2, 2, 2, 2, 6, 1, 1>(p, stream);      //   one-step phases (12 MFMAs)
        case 57: return f16x3p_use_register_epilogue(p) ? launch_pp<2, 2, 1, 2, 8, 2, 1, 0, 1>(p, stream)
                                                        : launch_pp<2, 2, 1, 2, 8, 2, 1>(p, stream);      // 128x128, waves of 32x64, 8 units (128 KiB), two-step phases (12 MFMAs)
        case 58: return launch_pp<2, 2, 1, 2, 4, 2, 1>(p, stream);      //   4 units (64 KiB): two workgroups per CU
        // round 5: TWO ping-pong workgroups per CU (72 KiB, 128 registers): four waves per SIMD, two of them in a compute interval at any time
        case 65: return launch_pp<2, 2, 2, 2, 3, 1, 0, 0, 0, 4>(p, stream);      // 256x128, waves of 64x64, 3 units, one-step phases (12 MFMAs)
        case 66: return launch_pp<2, 2, 2, 2, 3, 1, 1, 0, 0, 4>(p, stream);      //   priority around the MFMAs
        case 67: return launch_pp<1, 4, 2, 2, 3, 1, 0, 0, 0, 4>(p, stream);      // 128x256, waves of 64x64
        // right-sized tiles (round 4): eight waves of 32 rows x the whole tile width, so that a launch's tiles fill #CU - 8 workgroup slots in whole rounds
        case 62: return launch_pp<4, 1, 1, 7, 5, 1, 1>(p, stream);      // 256x224 (150 KiB): qkv 4096x3072 = 224 tiles, one round
        case 63: return launch_pp<4, 1, 1, 6, 5, 1, 1>(p, stream);      // 256x192 (140 KiB): fc1 4096x5504 = 464 tiles, two rounds
        case 64: return launch_pp<4, 1, 1, 8, 5, 1, 1>(p, stream);      // 256x256 with the same wave layout (160 KiB)
#ifdef PSAM_BUILD_EXPERIMENTS
        case 70: return launch_pp<1, 4, 2, 4, 3, 1, 1, 0, 2>(p, stream);      // 128x512 full-row tile, waves of 64x128, 3 units (120 KiB): row LayerNorm epilogue (N == 512)
#endif
        default: break;
    }
    psam_set_error("psam_gemm_f16x3p: unknown ping-pong config");
    return PSAM_EINVAL;
#endif
}
