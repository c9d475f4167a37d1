// fp32-GRADE GEMM on the fp16 matrix pipe, both operands PRE-PACKED ("f16x3p"): the production kernel of the large GEMMs.
//
//   C = act(alpha * A @ W^T + bias + rowbias[row / rowgroup]) + residual          (contract and epilogues of gemm.hip, 2-D form)
//
// Arithmetic: every operand row is scaled by a power of two (row maximum in [2^14, 2^15)) and every element split
// into hi + lo fp16; a product is hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with fp32 accumulation, un-scaled in the epilogue.
// What is new is the data path.  Both operands arrive in the "g8" packed form (psam_pack_rows_f16x2_g8, the LayerNorm / attention /
// GEMM-epilogue producers): the container is still one 32-bit word per element, and every group of 8 consecutive k holds
//      [hi k0..k7 : 8 x fp16 = 16 B][lo k0..k7 : 16 B]
// so one 16-byte chunk IS one matrix-instruction operand (8 k of one row).  With nothing left to compute while staging, a K slab
// moves global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging registers, no ds_write, no VALU in the main loop.
//
//   * slab = 32 k = 128 B per row; LDS stage = [BM rows of A | BN rows of W] x 128 B; chunk c of row r is stored at position
//     c ^ ((r >> 1) & 7)  -> every ds_read_b128 lane group touches 16 distinct 16-byte slots of the 256-byte bank row (conflict-free).
//     The DMA writes LDS linearly (wave base + lane * 16), so the swizzle is applied to the per-lane SOURCE address; each wave
//     instruction fetches 8 whole 128-byte lines.
//   * ring of S stages, ONE raw s_barrier per slab, counted vmcnt: at slab t every wave waits for its own pieces of slab t (+LA),
//     the barrier publishes them and retires every read of slab t-1, then the pieces of slab t+S-1 are issued into the stage slab
//     t-1 occupied -- S-1 slab times for the data to land.  LA = 1: the barrier of slab t also guarantees slab t+1, so the first
//     fragments of slab t+1 are read under the last MFMAs of slab t (no fragment latency exposed after the barrier).
//   * fragments double-buffered over the two k16 steps of a slab; one MFMA per issue slot with one ds_read / one DMA issue behind it.
//   * epilogue: gemm_epilogue.h (LDS transposition, float4 rows).
// Tile shapes (waves WM x WN, each TM x TN accumulator tiles of 32x32) and S are template parameters; the host picks per shape.
#include <type_traits>
#include <cstdlib>
#include "common.h"
#include <utility>
#include "gemm_f16x3p_args.h"
#include "gemm_epilogue.h"
#include "gemm_epilogue_t.h"

// ABL (measurement builds only, -DPSAM_GEMM_ABLATE): 1 = no epilogue, 2 = no DMA after the prologue, 4 = no MFMA, 8 = every tile loads the
// operand panels of tile (0, 0) (perfect L2 sharing), 16 = no fragment reads after the first slab.
// PF (where the DMA of a slab is issued): 0 = behind the fragment reads of the slab S-1 earlier, one per MFMA slot; 1 = all of it right
// after that slab's barrier; 2 (S = 2 only) = a second barrier per slab once every wave holds its step-1 fragments frees the stage half
// a slab early: the DMA of slab t+2 is issued in the middle of slab t (1.5 slab times ahead instead of 1).
// TR = 1: MFMA operands swapped (accumulator tiles transposed: one output row per lane) + the register-only epilogue of gemm_epilogue_t.h.
template <int WM, int WN, int TM, int TN, int S, int LA, int ABL = 0, int PF = 0, int TR = 0>
__global__ __launch_bounds__(64 * WM * WN, (TR && WM * WN == 4) ? 2 : 1) void gemm_f16x3p_kernel(const F16PArgs p) {      // TR, four waves: two workgroups per CU must fit (<= 256 registers)
    constexpr int NW = WM * WN, BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int ROWB = 128;                                   // bytes per row per slab (32 k x 4 B)
    constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
    constexpr int NBLK = STAGE / 1024, A_BLK = A_BYTES / 1024;  // 1 KiB DMA pieces (8 rows) per slab
    static_assert(NBLK % NW == 0, "pieces divide among the waves");
    constexpr int NL = NBLK / NW;                               // DMA instructions per wave per slab
    static_assert(S >= 2 && LA >= 0 && LA <= 1 && S - 2 - LA >= 0, "ring depth");
    static_assert(PF != 2 || (S == 2 && LA == 0), "mid-slab release is written for the two-stage ring");
    constexpr int NFR = 2 * (TM + TN);                          // fragment reads per k16 step
    constexpr int NMF = 3 * TM * TN;                            // MFMAs per k16 step
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // ---- tile of this workgroup: consecutive workgroup ids go to different XCDs (each with its own 4 MiB L2), so XCD x takes the x-th
    // CONTIGUOUS eighth of the tile order; the order walks column panels of `panel` tiles row-major (panel == tiles_n: plain row-major,
    // an XCD reads an eighth of A and all of W; panel ~ tiles_n / 8: an XCD keeps its W panel in L2 and streams A once)
    const int ntiles = p.tiles_m * p.tiles_n;
    int tile = blockIdx.x, split = 0;
    if (p.ksplit > 1) { split = tile / ntiles; tile -= split * ntiles; }      // split-K: `ksplit` workgroups per tile, consecutive K ranges
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
    // ABL & 64 (measurement builds): where a wave's cycles go.  Per slab six stamps -- slab top | DMA wait over | barrier passed | step-0 fragments in
    // registers | step-0 MFMAs and step-1 reads issued | mid-slab barrier passed -- accumulated per wave into: vm, b1, f0, m0, b2, m1 (written at the end
    // with the prologue / epilogue times, the absolute start / end stamps and HW_ID, 16 words per wave).
    unsigned tb[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tk0 = 0, tk1 = 0, tq = 0;
    if (ABL & 64) tk0 = gemm_now();

    // ---- DMA source offsets: piece b = wave + i NW covers rows 8b..8b+7 of the stage; lane -> (row b*8 + lane/8, slot lane%8),
    // which must receive chunk slot ^ swz(row) of that row.  Rows past M / N are clamped (their products are never stored).
    const int m0l = (ABL & 8) ? 0 : m0, n0l = (ABL & 8) ? 0 : n0;
    const int nslabs_all = p.K / 32;
    const int slab0 = p.ksplit > 1 ? (int)((int64_t)split * nslabs_all / p.ksplit) : 0;
    const int nslabs = p.ksplit > 1 ? (int)((int64_t)(split + 1) * nslabs_all / p.ksplit) - slab0 : nslabs_all;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0l * p.lda * 4 + slab0 * ROWB), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0l * p.ldw * 4 + slab0 * ROWB), 0, 0x7fffffff, 0x00020000);
    int voff[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int b = wave + i * NW;
        const bool isw = b >= A_BLK;
        const int row = (isw ? b - A_BLK : b) * 8 + (lane >> 3);            // row inside its operand's region
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        const int lim = isw ? p.N - n0l : p.M - m0l;
        const int rc = row < lim ? row : lim - 1;
        voff[i] = (int)((int64_t)rc * (isw ? p.ldw : p.lda) * 4) + chunk * 16;
    }
    auto issue = [&](int slab, int stage) {     // DMA the wave's NL pieces of `slab` into ring stage `stage`
        const int koff = slab * ROWB;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int b = wave + i * NW;
            unsigned char* dst = smem + stage * STAGE + b * 1024;
            if (b >= A_BLK) P_DMA16(rsW, dst, voff[i], koff);
            else P_DMA16(rsA, dst, voff[i], koff);
        }
    };
    auto issue_one = [&](int i, int slab, int stage) {
        if (ABL & 2) return;
        const int b = wave + i * NW;
        unsigned char* dst = smem + stage * STAGE + b * 1024;
        if (b >= A_BLK) P_DMA16(rsW, dst, voff[i], slab * ROWB);
        else P_DMA16(rsA, dst, voff[i], slab * ROWB);
    };

    // ---- fragment offsets inside a stage: row r32 of a 32-row tile, chunk 4 s + 2 h + q (s = k16 step, q = hi / lo plane)
    int fa_off[2][2], fw_off[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = r32 * ROWB + (((4 * s + 2 * h + q) ^ ((r32 >> 1) & 7)) << 4);
            fa_off[s][q] = wm * TM * 32 * ROWB + f;
            fw_off[s][q] = A_BYTES + wn * TN * 32 * ROWB + f;
        }

    pf32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // the epilogue's operands (per-row scalars, column constants): in flight from here on for the wave tiles that have the registers to hold
    // them through the K loop (gemm_epilogue.h)
    constexpr bool PREFETCH_EPI = TM * TN >= 4 && TN <= 4 && !TR;
    EpPre<TM> epre;
    float* const Cout = p.sk_part ? p.C : p.C + (int64_t)split * p.plane;
    if constexpr (PREFETCH_EPI) epre = gemm_epilogue_prefetch<TM, TN, true, true>(p, m0 + wm * TM * 32, n0 + wn * TN * 32, lane, Cout, p.residual);
    pf16x8 f0a[TM][2], f0w[TN][2], f1a[TM][2], f1w[TN][2];
    // fragment read n in [0, NFR) of step s from `stage`: n -> (operand, tile, plane)
    bool first_slab = true;
    auto read_frag = [&](int n, int s, int stage, pf16x8 (&fa)[TM][2], pf16x8 (&fw)[TN][2]) {
        if ((ABL & 16) && !first_slab) return;
        const unsigned char* base = smem + stage * STAGE;
        if (n < 2 * TM) { const int i = n >> 1, q = n & 1; fa[i][q] = *reinterpret_cast<const pf16x8*>(base + fa_off[s][q] + i * 32 * ROWB); }
        else { const int m = n - 2 * TM, j = m >> 1, q = m & 1; fw[j][q] = *reinterpret_cast<const pf16x8*>(base + fw_off[s][q] + j * 32 * ROWB); }
    };
    auto mfma = [&](int m, const pf16x8 (&fa)[TM][2], const pf16x8 (&fw)[TN][2]) {     // m in [0, NMF): term-major (hi*lo, lo*hi, hi*hi)
        constexpr int PA[3] = {0, 1, 0}, PW[3] = {1, 0, 0};
        const int term = m / (TM * TN), ij = m % (TM * TN), i = ij / TN, j = ij % TN;
        if (ABL & 4) asm volatile("" ::"v"(fa[i][PA[term]]), "v"(fw[j][PW[term]]));
        else if (TR) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j][PW[term]], fa[i][PA[term]], acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[i][PA[term]], fw[j][PW[term]], acc[i][j], 0, 0, 0);
    };

    // ---- prologue: S-1 slabs in flight
#pragma unroll
    for (int u = 0; u < S - 1 + (PF == 2 ? 1 : 0); ++u) issue(u, u);
    if (LA) {   // slab 0 visible, its first fragments in registers
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL * (S - 2)) : "memory");
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int n = 0; n < NFR; ++n) read_frag(n, 0, 0, f0a, f0w);
    }
    int st = 0;                    // ring stage of the current slab
    // One slab.  WAITN: DMA instructions of later slabs that may stay in flight at the wait; DO_ISSUE: a slab t+S-1 exists.
    auto body = [&](int t, auto waitn_c, auto issue_c, auto next_c) {
        constexpr int WAITN = decltype(waitn_c)::value;
        constexpr bool DO_ISSUE = decltype(issue_c)::value, HAS_NEXT = decltype(next_c)::value;
        unsigned long long q0 = 0, q1 = 0;
        if (ABL & 64) { q0 = gemm_now(); if (tq) tb[5] += (unsigned)(q0 - tq); }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
        if (ABL & 64) { q1 = gemm_now(); tb[0] += (unsigned)(q1 - q0); }
        asm volatile("s_barrier" ::: "memory");
        if (ABL & 64) { q0 = gemm_now(); tb[1] += (unsigned)(q0 - q1); }
        const int st_issue = st == 0 ? S - 1 : st - 1;          // stage of slab t-1 = stage of slab t+S-1
        const int st_next = st == S - 1 ? 0 : st + 1;
        if (PF == 1 && DO_ISSUE) {
#pragma unroll
            for (int i = 0; i < NL; ++i) issue_one(i, t + S - 1, st_issue);
        }
        if (!LA) {
#pragma unroll
            for (int n = 0; n < NFR; ++n) read_frag(n, 0, st, f0a, f0w);
        }
        if (ABL & 64) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); q1 = gemm_now(); tb[2] += (unsigned)(q1 - q0); }
        __builtin_amdgcn_sched_barrier(0);
        // step 0: NMF MFMAs; behind them the NFR fragment reads of step 1 (all of them: step 1 needs them), then DMA issues
        constexpr int P0 = (NFR + NMF - 1) / NMF;                               // extra operations per MFMA slot in step 0
        constexpr int D0 = PF == 1 ? NL : (NMF * P0 - NFR < NL ? NMF * P0 - NFR : NL);   // DMA issues placed in (or before) step 0
#pragma unroll
        for (int m = 0; m < NMF; ++m) {
            mfma(m, f0a, f0w);
#pragma unroll
            for (int k = m * P0; k < (m + 1) * P0; ++k) {
                if (k < NFR) read_frag(k, 1, st, f1a, f1w);
                else if (PF == 0 && DO_ISSUE && k - NFR < NL) issue_one(k - NFR, t + S - 1, st_issue);
                else if (PF == 2 && DO_ISSUE && k - NFR < NL) issue_one(k - NFR, t + 2, st);
            }
            if (PF == 2 && (m + 1) * P0 >= NFR && m * P0 < NFR) {   // every step-1 fragment read is issued: wait for them, release the stage
                if (ABL & 64) { q0 = gemm_now(); tb[3] += (unsigned)(q0 - q1); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
                if (ABL & 64) { tq = gemm_now(); tb[4] += (unsigned)(tq - q0); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // step 1: NMF MFMAs; behind them the rest of the DMA issues and (LA) the first fragments of the next slab
        constexpr int REST = NL - D0, N1 = REST + (LA ? NFR : 0), P1 = N1 > 0 ? (N1 + NMF - 1) / NMF : 1;
#pragma unroll
        for (int m = 0; m < NMF; ++m) {
            mfma(m, f1a, f1w);
#pragma unroll
            for (int k = m * P1; k < (m + 1) * P1; ++k) {
                if (k < REST) { if (DO_ISSUE) issue_one(D0 + k, PF == 2 ? t + 2 : t + S - 1, PF == 2 ? st : st_issue); }
                else if (LA && HAS_NEXT && k - REST < NFR) read_frag(k - REST, 0, st_next, f0a, f0w);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        st = st_next;
        first_slab = false;
    };
    using std::integral_constant;
    int t = 0;
    if constexpr (PF == 2) {
        // slabs t and t+1 are in flight at the top of slab t; slab t+2 is issued mid-slab
        for (; t + 2 < nslabs; ++t) body(t, integral_constant<int, NL>{}, integral_constant<bool, true>{}, integral_constant<bool, true>{});
        body(t, integral_constant<int, NL>{}, integral_constant<bool, false>{}, integral_constant<bool, true>{});
        ++t;
        body(t, integral_constant<int, 0>{}, integral_constant<bool, false>{}, integral_constant<bool, false>{});
    } else {
        for (; t + (S - 1) < nslabs; ++t) body(t, integral_constant<int, NL*(S - 2 - LA)>{}, integral_constant<bool, true>{}, integral_constant<bool, true>{});
        // tail: the last S-1 slabs issue nothing; at tail position j the slabs still in flight after the needed one(s): S-2-j-LA
        auto tail = [&](auto j_c) {
            constexpr int j = decltype(j_c)::value;
            constexpr int left = S - 2 - j - LA;
            body(t, integral_constant<int, (left > 0 ? left : 0) * NL>{}, integral_constant<bool, false>{}, integral_constant<bool, (j < S - 2)>{});
            ++t;
        };
        if constexpr (S >= 2) tail(integral_constant<int, 0>{});
        if constexpr (S >= 3) tail(integral_constant<int, 1>{});
        if constexpr (S >= 4) tail(integral_constant<int, 2>{});
        if constexpr (S >= 5) tail(integral_constant<int, 3>{});
        static_assert(S <= 5, "tail unrolled for S <= 5");
    }

    // ---- epilogue (gemm_epilogue.h): every wave is done with the ring, no DMA in flight
    if (ABL & 64) { tk1 = gemm_now(); if (tq) tb[5] += (unsigned)(tk1 - tq); }
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
    if constexpr (!TR || NW == 8) if (p.sk_part) {      // (the four-wave register-epilogue kernels have no registers to spare for the fix-up; the eight-wave ones hold 32 accumulator registers)
        // ---- split-K fix-up.  Every split of a tile parks its accumulators (raw MFMA layout: [register quad][wave][lane] float4, 1 KiB per store
        // instruction) with device-coherent stores (sc1: written through the XCD's L2), counts itself in, and all but the last arrival are done.  The
        // last one reads the ksplit partials back (its own too: same bits, and the sum order s = 0 .. ksplit-1 then never depends on who came last),
        // zeroes the counter for the next launch and runs the epilogue -- any epilogue -- on the sum.  No reduction launch, no fp32 planes.
        constexpr int QUADS = TM * TN * 4, TILE_F4 = QUADS * NW * 64;
        constexpr int SC1 = 16;
        const int tile_id = blockIdx.x - split * ntiles;      // the un-permuted id: any bijection serves
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.sk_part + (int64_t)tile_id * p.ksplit * TILE_F4 * 4), 0, 0x7fffffff, 0x00020000);
        const int lane_off = (wave * 64 + lane) * 16;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    // (element by element into floats first: __builtin_bit_cast applied to a vector ELEMENT expression read element 0 every time)
                    const float a0 = acc[i][j][4 * r4], a1 = acc[i][j][4 * r4 + 1], a2 = acc[i][j][4 * r4 + 2], a3 = acc[i][j][4 * r4 + 3];
                    const ep_f32x4 vf = {a0, a1, a2, a3};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu32x4, vf), rs, (split * QUADS + (i * TN + j) * 4 + r4) * (NW * 64 * 16) + lane_off, 0, SC1);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's partials are acknowledged by the memory side
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);
        if (tid == 0) *flag = (int)__hip_atomic_fetch_add(p.sk_count + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int arrived = *flag;
        __syncthreads();      // the epilogue below stages through this LDS
        if (arrived != p.ksplit - 1) return;
        if (tid == 0) __hip_atomic_store(p.sk_count + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int s = 0; s < p.ksplit; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const ep_f32x4 v = __builtin_bit_cast(ep_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (s * QUADS + (i * TN + j) * 4 + r4) * (NW * 64 * 16) + lane_off, 0, SC1));
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * r4 + e] += v[e];
                    }
        }
    }
    if constexpr (TR) gemm_store_tile_t<TM, TN>(p, acc, m0 + wm * TM * 32, n0 + wn * TN * 32, lane, Cout, p.residual);
    else gemm_store_tile<TM, TN, true, true>(p, acc, reinterpret_cast<float*>(smem) + wave * gemm_epilogue_lds_floats_per_wave<TN>(), m0 + wm * TM * 32,
                                             n0 + wn * TN * 32, lane, Cout, p.residual, PREFETCH_EPI ? &epre : nullptr);
    if ((ABL & 64) && p.dbg) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the wave's stores are acknowledged: the end of its epilogue
        const unsigned long long te = gemm_now();
        if (lane == 0) {
            unsigned* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 16;
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i] = tb[i];
            d[6] = (unsigned)(tk1 - tk0) - (tb[0] + tb[1] + tb[2] + tb[3] + tb[4] + tb[5]);      // prologue: kernel entry -> top of the first slab
            d[7] = (unsigned)(te - tk1);                                                          // epilogue incl. the final barrier and store acknowledgement
            d[8] = (unsigned)tk0; d[9] = (unsigned)(tk0 >> 32); d[10] = (unsigned)te; d[11] = (unsigned)(te >> 32);
            d[12] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
            d[13] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
            d[14] = (unsigned)nslabs; d[15] = (unsigned)blockIdx.x;
        }
    }
}

// ---------------------------------------------------------------------------------------------- row scales
// scale[r] = 2^(14 - e), e = floor(log2(max_k |X[r,k]|)) clamped below at -112; 1 for an all-zero / non-finite row (common.h).
__global__ __launch_bounds__(256) void row_scale_f16_kernel(const float* __restrict__ X, int64_t ldx, int rows, int cols, float* __restrict__ scale) {
    typedef float rs_f32x4 __attribute__((ext_vector_type(4)));
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* x = X + (int64_t)row * ldx;
    float m = 0.f;
    if (((ldx & 3) == 0) && (((uintptr_t)X & 15) == 0)) {
        const int c4 = cols >> 2;
        for (int c = lane; c < c4; c += 64) {
            const rs_f32x4 v = *reinterpret_cast<const rs_f32x4*>(x + c * 4);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
        for (int c = (c4 << 2) + lane; c < cols; c += 64) m = fmaxf(m, fabsf(x[c]));
    } else {
        for (int c = lane; c < cols; c += 64) m = fmaxf(m, fabsf(x[c]));
    }
    m = wave_max(m);
    if (lane == 0) scale[row] = f16_row_scale(m);
}

PSAM_API int32_t psam_row_scale_f16(const float* X, int64_t ldx, int32_t rows, int32_t cols, float* scale, hipStream_t stream) {
    PSAM_REQUIRE(X && scale, PSAM_EINVAL, "psam_row_scale_f16: null pointer");
    PSAM_REQUIRE(rows > 0 && cols > 0 && ldx >= cols, PSAM_EINVAL, "psam_row_scale_f16: bad shape");
    hipLaunchKernelGGL(row_scale_f16_kernel, dim3((unsigned)psam_cdiv(rows, 4)), dim3(256), 0, stream, X, ldx, rows, cols, scale);
    return psam_launch_status("psam_row_scale_f16: launch failed");
}

// ---------------------------------------------------------------------------------------------- g8 packing of an fp32 matrix
// P[r, 8g .. 8g+7] (32-bit containers) = [hi(8 x fp16) | lo(8 x fp16)] of scale[r] * X[r, 8g .. 8g+7]; columns K .. Kp-1 (Kp = K rounded
// up to 32, the GEMM's slab) are written as zeros.  In place (P == X, ldp == ldx) is allowed when K % 8 == 0.
__global__ __launch_bounds__(256) void pack_rows_g8_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ scale, int rows,
                                                           int K, int g8n, unsigned* __restrict__ P, int64_t ldp) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)rows * g8n) return;
    const int r = (int)(t / g8n), g = (int)(t % g8n);
    const float* x = X + (int64_t)r * ldx + g * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = g * 8 + e < K ? x[e] : 0.f;
    const float s = scale[r];
    unsigned hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) psam_split2_f16(v[2 * e], v[2 * e + 1], s, hi[e], lo[e]);
    unsigned* o = P + (int64_t)r * ldp + g * 8;
    *reinterpret_cast<pu32x4*>(o) = pu32x4{hi[0], hi[1], hi[2], hi[3]};
    *reinterpret_cast<pu32x4*>(o + 4) = pu32x4{lo[0], lo[1], lo[2], lo[3]};
}

PSAM_API int32_t psam_pack_rows_f16x2_g8(const float* X, int64_t ldx, const float* scale, int32_t rows, int32_t K, void* P, int64_t ldp,
                                         hipStream_t stream) {
    PSAM_REQUIRE(X && scale && P, PSAM_EINVAL, "psam_pack_rows_f16x2_g8: null pointer");
    const int Kp = (K + 31) / 32 * 32;
    PSAM_REQUIRE(rows > 0 && K > 0 && ldx >= K && ldp >= Kp, PSAM_EINVAL, "psam_pack_rows_f16x2_g8: bad shape (ldp >= K rounded up to 32)");
    PSAM_REQUIRE((ldp & 7) == 0 && ((uintptr_t)P & 31) == 0, PSAM_EALIGN, "psam_pack_rows_f16x2_g8: packed rows must be 32-byte aligned");
    PSAM_REQUIRE((const void*)X != (const void*)P || (ldx == ldp && (K & 7) == 0), PSAM_EINVAL, "psam_pack_rows_f16x2_g8: in place needs ldp == ldx, K % 8 == 0");
    const int64_t total = (int64_t)rows * (Kp / 8);
    hipLaunchKernelGGL(pack_rows_g8_kernel, dim3((unsigned)psam_cdiv(total, 256)), dim3(256), 0, stream, X, ldx, scale, rows, K, Kp / 8, (unsigned*)P, ldp);
    return psam_launch_status("psam_pack_rows_f16x2_g8: launch failed");
}

// Row scale + g8 packing in one pass (an fp32 activation that no LayerNorm produced, e.g. the attention output): one wave per row, the
// row stays in registers between the maximum and the split.  K <= NV4 * 256, K % 4 == 0, rows 16-byte aligned.
template <int NV4>
__global__ __launch_bounds__(256) void scale_pack_rows_g8_kernel(const float* __restrict__ X, int64_t ldx, int rows, int K, unsigned* __restrict__ P,
                                                                 int64_t ldp, float* __restrict__ scale, const float* __restrict__ add = nullptr,
                                                                 int64_t ldadd = 0, int rows_per_set = 1, int rep = 1) {
    typedef float sp_f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int row = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6);
    if (row >= rows) return;
    const sp_f32x4* xr = reinterpret_cast<const sp_f32x4*>(X + (int64_t)row * ldx);
    // optional broadcast addend (psam_scale_pack_rows_g8_add): set z = row / rows_per_set reads the rows of set z / rep; the sum is add + x, one
    // fp32 addition per element, as psam_add_bcast writes it
    const sp_f32x4* ar = add ? reinterpret_cast<const sp_f32x4*>(add + ((int64_t)(row / (rep * rows_per_set)) * rows_per_set + row % rows_per_set) * ldadd) : nullptr;
    const int c4n = K >> 2;
    sp_f32x4 v[NV4];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = i * 64 + lane;
        v[i] = xr[c < c4n ? c : c4n - 1];                       // unconditional (clamped) loads
        if (ar) v[i] = ar[c < c4n ? c : c4n - 1] + v[i];
        if (c >= c4n) v[i] = sp_f32x4{0.f, 0.f, 0.f, 0.f};
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[i][0]), fabsf(v[i][1]))), fmaxf(fabsf(v[i][2]), fabsf(v[i][3])));
    }
    amax = wave_max(amax);
    const float sc = f16_row_scale(amax);
    if (lane == 0) scale[row] = sc;
    const int klim = (K + 31) & ~31;
    const bool odd = lane & 1;
    unsigned* prow = P + (int64_t)row * ldp;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = i * 64 + lane;
        unsigned h0, l0, h1, l1;
        psam_split2_f16(v[i][0], v[i][1], sc, h0, l0);
        psam_split2_f16(v[i][2], v[i][3], sc, h1, l1);
        const unsigned r0 = __shfl_xor(odd ? h0 : l0, 1, 64), r1 = __shfl_xor(odd ? h1 : l1, 1, 64);
        if (c * 4 < klim) *reinterpret_cast<pu32x4*>(prow + c * 4) = odd ? pu32x4{r0, r1, l0, l1} : pu32x4{h0, h1, r0, r1};
    }
}

// scale[r] = f16 row scale of X[r, :K]; P[r, :] = g8-packed scale[r] * X[r, :K], zero-padded to K rounded up to 32 (ldp >= that).
PSAM_API int32_t psam_scale_pack_rows_g8(const float* X, int64_t ldx, int32_t rows, int32_t K, void* P, int64_t ldp, float* scale, hipStream_t stream) {
    PSAM_REQUIRE(X && P && scale, PSAM_EINVAL, "psam_scale_pack_rows_g8: null pointer");
    const int Kp = (K + 31) / 32 * 32;
    PSAM_REQUIRE(rows > 0 && K > 0 && K <= 6144 && (K & 3) == 0 && ldx >= K && ldp >= Kp, PSAM_EINVAL,
                 "psam_scale_pack_rows_g8: bad shape (K % 4 == 0, K <= 6144, ldp >= K rounded up to 32)");
    PSAM_REQUIRE((ldx & 3) == 0 && ((uintptr_t)X & 15) == 0 && (ldp & 7) == 0 && ((uintptr_t)P & 31) == 0, PSAM_EALIGN,
                 "psam_scale_pack_rows_g8: rows of X 16-byte, of P 32-byte aligned");
    const dim3 grid((unsigned)psam_cdiv(rows, 4)), block(256);
#define SP_LAUNCH(R) hipLaunchKernelGGL(scale_pack_rows_g8_kernel<R>, grid, block, 0, stream, X, ldx, rows, K, (unsigned*)P, ldp, scale, (const float*)nullptr, (int64_t)0, 1, 1)
    if (Kp <= 256) SP_LAUNCH(1);
    else if (Kp <= 512) SP_LAUNCH(2);
    else if (Kp <= 1024) SP_LAUNCH(4);
    else if (Kp <= 2048) SP_LAUNCH(8);
    else if (Kp <= 4096) SP_LAUNCH(16);
    else SP_LAUNCH(24);
#undef SP_LAUNCH
    return psam_launch_status("psam_scale_pack_rows_g8: launch failed");
}

PSAM_API int32_t psam_scale_pack_rows_g8_add(const float* X, int64_t ldx, const float* add, int64_t ldadd, int32_t rows_per_set, int32_t rep, int32_t rows, int32_t K,
                                             void* P, int64_t ldp, float* scale, hipStream_t stream) {
    PSAM_REQUIRE(X && add && P && scale, PSAM_EINVAL, "psam_scale_pack_rows_g8_add: null pointer");
    const int Kp = (K + 31) / 32 * 32;
    PSAM_REQUIRE(rows > 0 && K > 0 && K <= 6144 && (K & 3) == 0 && ldx >= K && ldadd >= K && ldp >= Kp && rows_per_set > 0 && rep > 0, PSAM_EINVAL,
                 "psam_scale_pack_rows_g8_add: bad shape (K % 4 == 0, K <= 6144, ldp >= K rounded up to 32)");
    PSAM_REQUIRE(((ldx | ldadd) & 3) == 0 && (((uintptr_t)X | (uintptr_t)add) & 15) == 0 && (ldp & 7) == 0 && ((uintptr_t)P & 31) == 0, PSAM_EALIGN,
                 "psam_scale_pack_rows_g8_add: rows of X / add 16-byte, of P 32-byte aligned");
    const dim3 grid((unsigned)psam_cdiv(rows, 4)), block(256);
#define SPA_LAUNCH(R) hipLaunchKernelGGL(scale_pack_rows_g8_kernel<R>, grid, block, 0, stream, X, ldx, rows, K, (unsigned*)P, ldp, scale, add, ldadd, rows_per_set, rep)
    if (Kp <= 256) SPA_LAUNCH(1);
    else if (Kp <= 512) SPA_LAUNCH(2);
    else if (Kp <= 1024) SPA_LAUNCH(4);
    else if (Kp <= 2048) SPA_LAUNCH(8);
    else if (Kp <= 4096) SPA_LAUNCH(16);
    else SPA_LAUNCH(24);
#undef SPA_LAUNCH
    return psam_launch_status("psam_scale_pack_rows_g8_add: launch failed");
}

// Both packed forms of the decoder's patch rows in one pass: P_sum / scale_sum = the rows of X + add (k = keys + key_pe, the A operand of the k / q
// projections), P_x / scale_x = the rows of X alone (the A operand of the v projection) -- transformer.py:160-170.  K <= 256: a float4 per lane.
__global__ __launch_bounds__(256) void scale_pack_rows_g8_dual_kernel(const float* __restrict__ X, int64_t ldx, int rows, int K, const float* __restrict__ add,
                                                                      int64_t ldadd, int rows_per_set, int rep, unsigned* __restrict__ Ps, float* __restrict__ ss,
                                                                      unsigned* __restrict__ Px, float* __restrict__ sx, int64_t ldp) {
    typedef float sp_f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int row = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6);
    if (row >= rows) return;
    const int c4n = K >> 2, c = lane < c4n ? lane : c4n - 1;
    sp_f32x4 x = reinterpret_cast<const sp_f32x4*>(X + (int64_t)row * ldx)[c];
    sp_f32x4 v = reinterpret_cast<const sp_f32x4*>(add + ((int64_t)(row / (rep * rows_per_set)) * rows_per_set + row % rows_per_set) * ldadd)[c] + x;
    if (lane >= c4n) { x = sp_f32x4{0.f, 0.f, 0.f, 0.f}; v = x; }
    const float scs = f16_row_scale(wave_max(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])))));
    const float scx = f16_row_scale(wave_max(fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])))));
    if (lane == 0) { ss[row] = scs; sx[row] = scx; }
    const int klim = (K + 31) & ~31;
    const bool odd = lane & 1;
    auto put = [&](const sp_f32x4& t, float sc, unsigned* prow) {
        unsigned h0, l0, h1, l1;
        psam_split2_f16(t[0], t[1], sc, h0, l0);
        psam_split2_f16(t[2], t[3], sc, h1, l1);
        const unsigned r0 = __shfl_xor(odd ? h0 : l0, 1, 64), r1 = __shfl_xor(odd ? h1 : l1, 1, 64);
        if (lane * 4 < klim) *reinterpret_cast<pu32x4*>(prow + lane * 4) = odd ? pu32x4{r0, r1, l0, l1} : pu32x4{h0, h1, r0, r1};
    };
    put(v, scs, Ps + (int64_t)row * ldp);
    put(x, scx, Px + (int64_t)row * ldp);
}

PSAM_API int32_t psam_scale_pack_rows_g8_add_dual(const float* X, int64_t ldx, const float* add, int64_t ldadd, int32_t rows_per_set, int32_t rep, int32_t rows, int32_t K,
                                                  void* P_sum, float* scale_sum, void* P_x, float* scale_x, int64_t ldp, hipStream_t stream) {
    PSAM_REQUIRE(X && add && P_sum && scale_sum && P_x && scale_x, PSAM_EINVAL, "psam_scale_pack_rows_g8_add_dual: null pointer");
    const int Kp = (K + 31) / 32 * 32;
    PSAM_REQUIRE(rows > 0 && K > 0 && K <= 256 && (K & 3) == 0 && ldx >= K && ldadd >= K && ldp >= Kp && rows_per_set > 0 && rep > 0, PSAM_EINVAL,
                 "psam_scale_pack_rows_g8_add_dual: bad shape (K % 4 == 0, K <= 256, ldp >= K rounded up to 32)");
    PSAM_REQUIRE(((ldx | ldadd) & 3) == 0 && (((uintptr_t)X | (uintptr_t)add) & 15) == 0 && (ldp & 7) == 0 && (((uintptr_t)P_sum | (uintptr_t)P_x) & 31) == 0, PSAM_EALIGN,
                 "psam_scale_pack_rows_g8_add_dual: rows of X / add 16-byte, of the packed outputs 32-byte aligned");
    hipLaunchKernelGGL(scale_pack_rows_g8_dual_kernel, dim3((unsigned)psam_cdiv(rows, 4)), dim3(256), 0, stream, X, ldx, rows, K, add, ldadd, rows_per_set, rep,
                       (unsigned*)P_sum, scale_sum, (unsigned*)P_x, scale_x, ldp);
    return psam_launch_status("psam_scale_pack_rows_g8_add_dual: launch failed");
}

// ---------------------------------------------------------------------------------------------- host
static int g_f16x3p_cfg = -1;
PSAM_API void psam_gemm_f16x3p_force_config(int32_t cfg) { g_f16x3p_cfg = cfg; }
#ifdef PSAM_BUILD_EXPERIMENTS
// Persistent form of the 128x128 register-epilogue configuration (gemm_f16x3c.hip: whole tiles from a queue, one continuous slab stream per workgroup)
// for batch-sized launches: -1 = default (environment PSAM_GEMM_CONTINUOUS, else OFF: measured neutral), 0 = never, 1 = wherever it applies.
static int g_f16x3p_continuous = -1;
PSAM_API void psam_gemm_f16x3p_force_continuous(int32_t mode) { g_f16x3p_continuous = mode; }
static bool f16x3p_continuous_enabled() {
    if (g_f16x3p_continuous >= 0) return g_f16x3p_continuous != 0;
    static int on = -1;
    if (on < 0) { const char* e = getenv("PSAM_GEMM_CONTINUOUS"); on = e ? (atoi(e) != 0) : 0; }
    return on != 0;
}
#endif
#ifdef PSAM_GEMM_ABLATE
static unsigned* g_f16x3p_dbg = nullptr;      // measurement builds: 16 words per wave of the timing instances (ABL & 64)
extern "C" __attribute__((visibility("default"))) void psam_gemm_f16x3p_set_timing_buffer(void* buf) { g_f16x3p_dbg = (unsigned*)buf; }
#endif
// Epilogue of the packed-operand GEMMs: 1 = the register-only epilogue on transposed accumulator tiles (gemm_epilogue_t.h) wherever the launch's
// options allow it, 0 = always the LDS-transposition epilogue (gemm_epilogue.h), -1 = the default (environment PSAM_GEMM_TR, else 1).
static int g_f16x3p_tr = -1;
PSAM_API void psam_gemm_f16x3p_force_epilogue(int32_t mode) { g_f16x3p_tr = mode; }
static int f16x3p_epilogue_mode() {
    int mode = g_f16x3p_tr;
    if (mode < 0) {
        static int env = -2;
        if (env == -2) { const char* e = getenv("PSAM_GEMM_TR"); env = e ? atoi(e) : 1; }
        mode = env;
    }
    return mode;
}
// Whether psam_gemm_f16x3p_ex takes psam_gemm_fuse_t.row_ln_* (Linear -> LayerNorm -> activation in one GEMM) for N output columns: 256 always
// (full-row wave tiles, LDS epilogue), 512 with the register epilogue (128x512 workgroup tiles).
// N == 512 is OFF by default (environment PSAM_GEMM_ROWLN512=1 turns it on): measured SLOWER than the two launches it replaces at the benchmark's
// size -- PatchEncoder 1.27 ms against 1.05 ms (profiles/r04/r04_rowln512.txt): one 8-wave workgroup per CU runs its K loop (8 steps), the three row
// passes (128 erf-GELUs per lane) and 256 KiB of stores back to back with nothing to overlap them, where GEMM + LayerNorm kernel overlap across
// many resident workgroups.  Kept reachable, parity-tested (tests/test_gpu_kernels.py::test_gemm_row_ln_512).
PSAM_API int32_t psam_gemm_f16x3p_fused_row_ln(int32_t N) {
    if (N == 256) return 1;
#ifndef PSAM_BUILD_EXPERIMENTS
    return 0;      // the 128x512 full-row tile is an experiments-build configuration (measured slower, see above)
#endif
    static int on = -1;
    if (on < 0) { const char* e = getenv("PSAM_GEMM_ROWLN512"); on = e ? atoi(e) : 0; }
    return N == 512 && on > 0 && f16x3p_epilogue_mode() > 0 ? 1 : 0;
}
bool f16x3p_use_register_epilogue(const F16PArgs& p) {
    int mode = g_f16x3p_tr;
    if (mode < 0) {
        static int env = -2;
        if (env == -2) { const char* e = getenv("PSAM_GEMM_TR"); env = e ? atoi(e) : 1; }
        mode = env;
    }
    if (mode <= 0) return false;
    if (p.gmax_out || p.row_ln_g || (p.no_store && !p.hyper)) return false;      // options only gemm_epilogue.h implements
    // per-group row bias: implemented, bitwise equal, but only when forced -- its one user (PatchEncoder conv2.0, K = 128, 512 MB of fp32 output)
    // is bound by the stores, and one-row-per-lane 16-byte stores lose against the LDS epilogue's full rows (247 vs 235 us,
    // profiles/r04/r04_gemm_experiments.txt)
    if (p.rowbias && (g_f16x3p_tr <= 0 || (p.ldrb & 3) != 0 || (((uintptr_t)p.rowbias) & 15) != 0 || p.act == 3)) return false;
    if (p.hyper && (p.hyper_rows % 32 != 0 || (((uintptr_t)p.hyper) & 15) != 0 || (p.N & 3) != 0)) return false;
    if ((((uintptr_t)p.scaleW | (uintptr_t)p.bias | (uintptr_t)p.ln_c) & 15) != 0) return false;      // float4 loads of the column constants
    return true;
}

// Tile configuration for a shape.  Measured per-CU rates of the configurations are within ~15 % of each other once a CU is busy
// (profiles/r02/r02_gemm_p_sweep_*.log; the kernel is power-limited, profiles/r02/r02_gemm_power_limit.txt); what differs is how many rounds
// of workgroups a launch needs, how full the last one is -- and how a launch behaves when a few CUs are NOT available: the tokenizer
// of the next batch (FPS: one 1024-thread workgroup per cloud, a whole CU each, ~2 ms per step) runs beside the dense stage, and a
// launch of exactly #CU one-per-CU workgroups then needs a second round for the last few tiles (twice the time).  So the rounds are
// counted on #CU - 8 and configurations with two workgroups per CU (64-70 KiB of LDS) are candidates for the shapes that would
// otherwise sit exactly at one round.
//   cost = rounds x tile area x (K + 300) x penalty x share
// (300 ~ the epilogue of a tile in k-steps; penalty: operand bytes per flop of the smaller tiles, LDS-DMA path ~32 B/clk/CU; share: two
// resident tiles on a CU each progress at ~60 % of a lone tile's speed).
static int f16x3p_pick(int M, int N, int K, int act, bool two_wide_only) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    }
    static int reserve = -1;
    static double share2 = 0.0, pen9 = 0.0;
    if (reserve < 0) {      // tuning hooks (environment, read once): CUs assumed busy elsewhere; slowdown of two co-resident tiles
        const char* e = getenv("PSAM_GEMM_RESERVE_CUS");
        reserve = e ? atoi(e) : 8;
        const char* f = getenv("PSAM_GEMM_SHARE");
        share2 = f ? atof(f) : 1.6;
        const char* g = getenv("PSAM_GEMM_PEN9");
        pen9 = g ? atof(g) : 1.15;
    }
    const int ncu_eff = ncu > 2 * reserve ? ncu - reserve : ncu;
    struct Cand { int cfg, bm, bn, per_cu; bool swiglu, two_wide; double pen; };
    // 41 = the eight-wave 128x128 tile on a FIVE-stage ring (exactly 160 KiB), 42 = 128x96 on five stages (140 KiB; single-cloud launches of one round): the same
    // bits as cfg 9, and ALONE on the chip faster where the K loop is bound by the slabs in flight (giant qkv at one cloud 28.1 -> 25.5 -> 23.8 us, ViT-L fc2 at
    // M = 2048 42.7 -> 41.0 us; profiles/r06/r06_small_m.txt) -- but in the two-stream pipelines they LOSE 4 % (cfg #5 146.3 -> 140.4 sessions/s, cfg #3 102.8 ->
    // 98.6 clouds/s, profiles/r06/r06_small_m_ring.txt): a workgroup that holds all of a CU's LDS keeps the other stream's kernels off that CU.  OFF by default
    // (PSAM_GEMM_SMALL_M_RING=1 / force_config 41, 42 switch them in: a single stream of work, e.g. an interactive predictor, gains).
    static const Cand cands[] = {{14, 256, 256, 1, true, false, 1.0}, {23, 256, 192, 1, false, false, 1.0}, {4, 256, 128, 1, true, true, 1.05},
                                 {41, 128, 128, 1, true, true, 1.15}, {21, 128, 128, 2, true, true, 1.15},  {28, 128, 128, 2, true, true, 1.2},
                                 {42, 128, 96, 1, false, false, 1.15}};
    static int small_ring = -1;
    if (small_ring < 0) { const char* e = getenv("PSAM_GEMM_SMALL_M_RING"); small_ring = e ? atoi(e) : 0; }
    int best = 41;
    double best_cost = 1e300;
    for (const Cand& c : cands) {
        if (act == 3 && !c.swiglu) continue;
        if (two_wide_only && !c.two_wide) continue;
        const int64_t tiles = psam_cdiv(M, c.bm) * psam_cdiv(N, c.bn);
        const double rounds = (double)psam_cdiv(tiles, (int64_t)ncu_eff * c.per_cu);
        if (c.cfg == 42 && (!small_ring || M > 1024 || rounds > 1.0)) continue;
        const double share = (c.per_cu == 2 && tiles * 2 > (int64_t)ncu_eff * 3) ? share2 : 1.0;
        const double cost = rounds * c.bm * c.bn * (K + 300.0) * (c.cfg == 41 ? pen9 : c.pen) * share;
        if (cost < best_cost) { best_cost = cost; best = c.cfg; }
    }
    if (best == 41 && !small_ring) best = 9;
    // The eight-wave 128x128 tile comes as cfg 9 (four stages, 128 KiB: one workgroup per CU) and cfg 28 (two stages with the mid-slab release, 70 KiB: two
    // per CU) -- the same wave tiles, the same bits.  Alone cfg 9 is 5-10 % faster on every single-cloud shape; in the two-stream pipelines the answer depends on
    // the rows (profiles/r06/r06_sub9.txt): at M = 512 (giant, one cloud) cfg 9 wins (145.9 vs 138.4 sessions/s), at M = 2048 (ViT-L, N = 131072) cfg 28 wins
    // (106.1 vs 103.2 clouds/s: with several rounds of tiles the LDS it leaves free lets the other stream's workgroups onto the CU).  PSAM_GEMM_SUB9=<cfg>
    // overrides (A/B; 9 = always cfg 9).
    static int sub9 = -1;
    if (sub9 < 0) { const char* e = getenv("PSAM_GEMM_SUB9"); sub9 = e ? atoi(e) : 0; }
    if (best == 9) best = sub9 > 0 ? sub9 : (M >= 2048 ? 28 : 9);
    return best;
}

template <int WM, int WN, int TM, int TN, int S, int LA, int ABL = 0, int PF = 0, int TR = 0>
static int32_t launch_f16x3p(F16PArgs& p, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NW = WM * WN;
    constexpr int ring = S * (BM + BN) * 128, epi = TR ? 0 : NW * gemm_epilogue_lds_floats_per_wave<TN>() * 4;
    constexpr int lds = ring > epi ? ring : epi;
    static_assert(lds <= 160 * 1024, "LDS budget");
    if (TN % 2 != 0 && p.act == 3) {      // the SwiGLU gate pairs accumulator tiles (2q, 2q+1)
        psam_set_error("psam_gemm_f16x3p: this tile configuration cannot apply the SwiGLU epilogue");
        return PSAM_EINVAL;
    }
    p.tiles_m = (int)psam_cdiv(p.M, BM);
    p.tiles_n = (int)psam_cdiv(p.N, BN);
    p.panel = f16x3p_panel(p.tiles_m, p.tiles_n, BM, BN, p.K);
    static unsigned long long attr_done = 0;   // per device (bit = device id)
    if (!f16x3p_reserve_lds(&gemm_f16x3p_kernel<WM, WN, TM, TN, S, LA, ABL, PF, TR>, lds, attr_done)) {
        psam_set_error("psam_gemm_f16x3p: cannot reserve LDS");
        return PSAM_EINVAL;
    }
    hipLaunchKernelGGL((gemm_f16x3p_kernel<WM, WN, TM, TN, S, LA, ABL, PF, TR>), dim3((unsigned)(p.tiles_m * p.tiles_n * p.ksplit)), dim3(64 * NW), lds, stream, p);
    return psam_launch_status("psam_gemm_f16x3p: launch failed");
}

// Optional fused extras of psam_gemm_f16x3p_ex: psam_gemm_fuse_t, include/pointsam_hip.h (part of this translation unit through common.h).

// partial planes the hyper products of an N-column GEMM are delivered in: 1 with the row-LayerNorm (full-row) epilogue, N / 64 otherwise
PSAM_API int32_t psam_gemm_f16x3p_hyper_planes(int32_t N, int32_t with_row_ln) { return with_row_ln ? 1 : N / 64; }

// segments (of 32 gated columns) per row of the stats buffer of a SwiGLU GEMM with N packed weight rows
PSAM_API int32_t psam_gemm_f16x3p_stat_segs(int32_t N) { return (N / 2 + 31) / 32; }

static void f16x3p_cfg_tile(int cfg, int& bm, int& bn, int& per_cu) {
    switch (cfg) {
        case 4: bm = 256; bn = 128; per_cu = 1; break;
        case 14: bm = 256; bn = 256; per_cu = 1; break;
        case 12: case 23: bm = 256; bn = 192; per_cu = 1; break;
        case 9: case 29: case 41: bm = 128; bn = 128; per_cu = 1; break;
        case 42: bm = 128; bn = 96; per_cu = 1; break;
        default: bm = 128; bn = 128; per_cu = 2; break;      // 0, 21, 28
    }
}

// Arrival counters of the split-K fix-up: the first SK_MAX_TILES ints of the CALLER's counter block (psam_gemm_fuse_t.counters, PSAM_COUNTER_BYTES,
// include/pointsam_hip.h): zero before the first launch, left zero by every launch (the last workgroup of a tile resets its word).  The library keeps no
// state and allocates nothing; without a block the launch writes partial planes and a reduction pass adds them.  PSAM_GEMM_SPLITK_FIXUP=0 switches
// the fix-up off.
constexpr int64_t SK_MAX_TILES = PSAM_CNT_GEMM_N;
static int g_f16x3p_sk_fixup = -1;      // -1: PSAM_GEMM_SPLITK_FIXUP (default on); 0 / 1 forced (psam_gemm_f16x3p_force_splitk_fixup)
static bool f16x3p_splitk_fixup_enabled() {
    if (g_f16x3p_sk_fixup >= 0) return g_f16x3p_sk_fixup != 0;
    static int on = -1;
    if (on < 0) { const char* e = getenv("PSAM_GEMM_SPLITK_FIXUP"); on = e ? (atoi(e) != 0) : 1; }
    return on != 0;
}
PSAM_API void psam_gemm_f16x3p_force_splitk_fixup(int32_t mode) { g_f16x3p_sk_fixup = mode; }

// Split-K factor for a shape (1: none).  A launch whose tiles cover less than half of the CUs (M = 512 rows of one cloud: 44 tiles of
// 128x128 for the N = 1408 GEMMs of the giant encoder) leaves the rest of the chip idle for a K loop of up to 192 slabs; `ks` workgroups
// per tile share the slabs (>= 8 each) and psam_gemm_f16x3p_ex adds the partial planes in a fixed order (deterministic).
// PSAM_GEMM_SPLITK: 0 = never, n > 1 = always n (tuning).
PSAM_API int32_t psam_gemm_f16x3p_splitk(int32_t M, int32_t N, int32_t K, int32_t act) {
    static int forced = -1;
    if (forced < 0) { const char* e = getenv("PSAM_GEMM_SPLITK"); forced = e ? atoi(e) : 1; }
    if (act == 3 || K < 1024 || (K & 31) || forced == 0) return 1;
    if (forced == 1 && M > 2048) return 1;      // a batch of clouds: its other GEMM stream fills the idle CUs, the extra reduction pass only costs (r03 profile)
    const int nslabs = K / 32;
    if (forced > 1) return forced <= nslabs / 4 ? forced : (nslabs / 4 > 1 ? nslabs / 4 : 1);
    int cfg = g_f16x3p_cfg;
    if (cfg < 0 || cfg >= 50) cfg = f16x3p_pick(M, N, K, act, true);
    int bm, bn, per_cu;
    f16x3p_cfg_tile(cfg, bm, bn, per_cu);
    int ncu = 256, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    const int64_t tiles = psam_cdiv(M, bm) * psam_cdiv(N, bn), slots = (int64_t)(ncu - 8) * per_cu;
    // Measured per shape, alone on the chip (profiles/r06/r06_small_m.txt): splitting pays for 44 tiles on 248 slots (giant proj 24.7 -> 18.6 us, fc2 82.6 -> 38.7)
    // and LOSES for 132 and 192 tiles (giant qkv 27.9 -> 41.2 us with two splits, fc1 30.1 -> 44.1 with three: 8.6 - 12.6 MB of partial planes per split
    // through the fabric and a serial fix-up for a K loop that was only 44 slabs long).  Until round 5 the limit was 0.55 of the slots.
    static int frac_pct = -1;
    if (frac_pct < 0) { const char* e = getenv("PSAM_GEMM_SPLITK_MAX_FILL_PCT"); frac_pct = e ? atoi(e) : 30; }
    if (tiles * 100 > slots * frac_pct) return 1;
    int ks = (int)((slots + tiles / 2) / tiles);
    if (ks > 4) ks = 4;
    if (ks > nslabs / 8) ks = nslabs / 8;
    return ks > 1 ? ks : 1;
}

// C = act(alpha * sum_s ws[s] + bias) + residual over `ks` planes, s ascending (the sum order is fixed: results do not depend on timing)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int64_t plane, int ks, int M, int N, const float* __restrict__ bias,
                                                            const float* __restrict__ residual, int64_t ldr, float alpha, int act, float* __restrict__ C,
                                                            int64_t ldc) {
    typedef float sk_f32x4 __attribute__((ext_vector_type(4)));
    const int n4 = N >> 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)M * n4) return;
    const int r = (int)(t / n4), c = (int)(t % n4) * 4;
    const float* w = ws + (int64_t)r * N + c;
    sk_f32x4 a = *reinterpret_cast<const sk_f32x4*>(w);
    for (int s = 1; s < ks; ++s) a += *reinterpret_cast<const sk_f32x4*>(w + s * plane);
    a *= alpha;
    if (bias) a += *reinterpret_cast<const sk_f32x4*>(bias + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = ep_act(a[e], act);
    if (residual) a += *reinterpret_cast<const sk_f32x4*>(residual + (int64_t)r * ldr + c);
    *reinterpret_cast<sk_f32x4*>(C + (int64_t)r * ldc + c) = a;
}

// A [M, K] and W [N, K]: g8-packed, row-scaled (scaleA[M], scaleW[N] powers of two); K % 32 == 0 (pad with zeros), K >= 128.
PSAM_API int32_t psam_gemm_f16x3p_ex(const void* A, int64_t lda, const float* scaleA, const void* W, int64_t ldw, const float* scaleW, float* C,
                                     int64_t ldc, const float* bias, const float* residual, int64_t ldr, const float* rowbias, int64_t ldrb,
                                     int32_t rowgroup, int32_t M, int32_t N, int32_t K, float alpha, int32_t act, const psam_gemm_fuse_t* fuse,
                                     hipStream_t stream) {
    PSAM_REQUIRE(A && W && C && scaleA && scaleW, PSAM_EINVAL, "psam_gemm_f16x3p: null pointer");
    PSAM_REQUIRE(M > 0 && N > 0 && K >= 128 && (K & 31) == 0, PSAM_EINVAL, "psam_gemm_f16x3p: bad shape (K % 32 == 0, K >= 128)");
    PSAM_REQUIRE(act >= 0 && act <= 3, PSAM_EINVAL, "psam_gemm_f16x3p: bad activation code");
    PSAM_REQUIRE(!rowbias || rowgroup > 0, PSAM_EINVAL, "psam_gemm_f16x3p: rowbias needs rowgroup > 0");
    PSAM_REQUIRE((lda & 7) == 0 && (ldw & 7) == 0 && ((uintptr_t)A & 31) == 0 && ((uintptr_t)W & 31) == 0, PSAM_EALIGN,
                 "psam_gemm_f16x3p: packed rows must be 32-byte aligned (lda, ldw multiples of 8)");
    PSAM_REQUIRE((int64_t)256 * lda * 4 + K * 4 < ((int64_t)1 << 31) && (int64_t)256 * ldw * 4 + K * 4 < ((int64_t)1 << 31), PSAM_EINVAL,
                 "psam_gemm_f16x3p: leading dimension too large for 32-bit tile offsets");
    PSAM_REQUIRE(act != 3 || ((N & 63) == 0 && !residual && !rowbias), PSAM_EINVAL, "psam_gemm_f16x3p: SwiGLU epilogue needs N % 64 == 0, no residual/rowbias");
    F16PArgs p;
    p.A = (const unsigned char*)A; p.W = (const unsigned char*)W; p.C = C; p.bias = bias; p.residual = residual; p.rowbias = rowbias;
    p.scaleA = scaleA; p.scaleW = scaleW; p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.ldrb = ldrb;
    p.M = M; p.N = N; p.K = K; p.rowgroup = rowgroup > 0 ? rowgroup : 1; p.act = act; p.alpha = alpha;
    p.out_scale = nullptr; p.out_k1 = p.out_k2 = 0.f; p.pack_out = 0; p.out_bound = nullptr; p.stats = nullptr; p.stat_cols = 0; p.stat_segs = 0;
    p.ln_mean = p.ln_rstd = p.ln_c = nullptr;
    p.gmax_out = nullptr; p.gmax_ld = 0; p.gmax_k = 0; p.no_store = 0;
    p.row_ln_g = p.row_ln_b = nullptr; p.row_ln_eps = 0.f; p.hyper = nullptr; p.masks = nullptr; p.hyper_c = 0; p.hyper_rows = 1; p.hyper_pstride = 0; p.epi_abl = 0;
    p.ksplit = 1; p.plane = 0; p.sk_part = nullptr; p.sk_count = nullptr; p.dbg = nullptr;
#ifdef PSAM_GEMM_ABLATE
    p.dbg = g_f16x3p_dbg;
#endif
    int cfg = g_f16x3p_cfg;
    if (cfg < 0) cfg = f16x3p_pick(M, N, K, act, false);
    if (fuse && fuse->splitk > 1) {
        const int ks = fuse->splitk;
        PSAM_REQUIRE(!fuse->pack_out && !fuse->stats && !fuse->ln_c && !fuse->gmax_out && !fuse->row_ln_g && !fuse->hyper, PSAM_EINVAL,
                     "psam_gemm_f16x3p_ex: split-K does not combine with the fused epilogue extras");
        PSAM_REQUIRE(act != 3 && !rowbias && ks <= K / 128, PSAM_EINVAL, "psam_gemm_f16x3p_ex: split-K needs act != SwiGLU, no rowbias, >= 4 slabs per split");
        PSAM_REQUIRE(fuse->splitk_ws && fuse->splitk_plane >= (int64_t)M * N && (N & 3) == 0 && (fuse->splitk_plane & 3) == 0, PSAM_EINVAL,
                     "psam_gemm_f16x3p_ex: split-K needs a workspace of splitk planes of >= M * N floats, N % 4 == 0");
        PSAM_REQUIRE((((uintptr_t)fuse->splitk_ws | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)residual) & 15) == 0 && (ldc & 3) == 0 && (ldr & 3) == 0,
                     PSAM_EALIGN, "psam_gemm_f16x3p_ex: split-K needs 16-byte aligned rows");
        if (cfg >= 50 || cfg == 30 || cfg == 31) cfg = f16x3p_pick(M, N, K, act, true);
        // in-kernel fix-up (the last workgroup of a tile sums the partials and runs the epilogue) where the workspace holds the tiles' raw accumulators
        // and this stream has its arrival counters; otherwise partial planes + the reduction launch
        int bm = 0, bn = 0, per_cu = 0;
        f16x3p_cfg_tile(cfg, bm, bn, per_cu);
        const int64_t sk_tiles = psam_cdiv(M, bm) * psam_cdiv(N, bn);
        int* counters = (f16x3p_splitk_fixup_enabled() && sk_tiles <= SK_MAX_TILES && sk_tiles * bm * bn <= fuse->splitk_plane) ? (fuse->counters ? fuse->counters + PSAM_CNT_GEMM : nullptr) : nullptr;
        p.ksplit = ks; p.plane = fuse->splitk_plane;
        if (counters) { p.sk_part = fuse->splitk_ws; p.sk_count = counters; }
        else { p.C = fuse->splitk_ws; p.ldc = N; p.bias = nullptr; p.residual = nullptr; p.act = 0; p.alpha = 1.f; }
        int32_t rc = PSAM_EINVAL;
        switch (cfg) {
            case 0: rc = launch_f16x3p<2, 2, 2, 2, 2, 0>(p, stream); break;
            case 4: rc = launch_f16x3p<4, 2, 2, 2, 3, 0>(p, stream); break;
            case 9: rc = launch_f16x3p<4, 2, 1, 2, 4, 1>(p, stream); break;
            case 12: rc = launch_f16x3p<4, 2, 2, 3, 2, 0>(p, stream); break;
            case 14: rc = launch_f16x3p<4, 2, 2, 4, 2, 0>(p, stream); break;
            case 21: rc = launch_f16x3p<2, 2, 2, 2, 2, 0, 0, 2>(p, stream); break;
            case 23: rc = launch_f16x3p<4, 2, 2, 3, 2, 0, 0, 2>(p, stream); break;
            case 28: rc = launch_f16x3p<4, 2, 1, 2, 2, 0, 0, 2>(p, stream); break;
            case 29: rc = launch_f16x3p<4, 2, 1, 2, 4, 1, 0, 0, 1>(p, stream); break;
            case 41: rc = launch_f16x3p<4, 2, 1, 2, 5, 1>(p, stream); break;
            case 42: rc = launch_f16x3p<4, 1, 1, 3, 5, 1>(p, stream); break;
            default: psam_set_error("psam_gemm_f16x3p_ex: split-K has no such tile configuration"); return PSAM_EINVAL;
        }
        if (rc != PSAM_OK || counters) return rc;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)psam_cdiv((int64_t)M * (N / 4), 256)), dim3(256), 0, stream, (const float*)fuse->splitk_ws,
                           fuse->splitk_plane, ks, M, N, bias, residual, ldr, alpha, act, C, ldc);
        return psam_launch_status("psam_gemm_f16x3p_ex: split-K reduction launch failed");
    }
    if (fuse && fuse->hyper && !fuse->row_ln_g) {
        // hyper products from the 64-column wave tiles of the 128x128 / 256x128 configurations: N / 64 partial planes, added by psam_sum_planes
        PSAM_REQUIRE((M & 255) == 0 && (N & 127) == 0 && act != 3, PSAM_EINVAL, "psam_gemm_f16x3p_ex: hyper products need M % 256 == 0, N % 128 == 0, no SwiGLU");
        PSAM_REQUIRE(!fuse->stats && !fuse->ln_c && !fuse->gmax_out && !fuse->pack_out, PSAM_EINVAL, "psam_gemm_f16x3p_ex: hyper products do not combine with other extras");
        PSAM_REQUIRE(fuse->masks && fuse->hyper_c > 0 && fuse->hyper_c <= 4 && fuse->hyper_rows > 0 && fuse->hyper_rows % 32 == 0 &&
                     (N == 64 || fuse->hyper_pstride >= (int64_t)(M / fuse->hyper_rows) * fuse->hyper_c * fuse->hyper_rows), PSAM_EINVAL,
                     "psam_gemm_f16x3p_ex: hyper products need masks, 1 <= hyper_c <= 4, hyper_rows % 32 == 0 and a plane stride covering [Z, C, rows]");
        PSAM_REQUIRE(((uintptr_t)fuse->hyper & 15) == 0, PSAM_EALIGN, "psam_gemm_f16x3p_ex: 16-byte alignment");
        p.hyper = fuse->hyper; p.masks = fuse->masks; p.hyper_c = fuse->hyper_c; p.hyper_rows = fuse->hyper_rows; p.hyper_pstride = fuse->hyper_pstride;
        p.no_store = fuse->no_store;
        // The register epilogue sums a row's products in another order than the LDS epilogue (same accuracy, other rounding): ONE configuration for every
        // M, so that a cloud's logits do not depend on how many clouds share the launch (tests/test_gpu_e2e.py::test_properties_full_size).
        if (g_f16x3p_cfg < 0 && f16x3p_use_register_epilogue(p)) cfg = 21;
        else if (cfg != 4 && cfg != 9 && cfg != 41 && cfg != 21 && cfg != 28) cfg = f16x3p_pick(M, N, K, act, true);
#ifdef PSAM_BUILD_EXPERIMENTS
    } else if (fuse && fuse->row_ln_g && N == 512) {
        // full-row tile 128x512 on the ping-pong kernel with the register epilogue: Linear (+ row bias per group) -> LayerNorm -> activation -> packed rows
        PSAM_REQUIRE(f16x3p_epilogue_mode() > 0, PSAM_EINVAL, "psam_gemm_f16x3p_ex: row LayerNorm over N == 512 needs the register epilogue (psam_gemm_f16x3p_force_epilogue)");
        PSAM_REQUIRE((M & 127) == 0 && act != 3 && fuse->row_ln_b && !fuse->hyper && !fuse->stats && !fuse->ln_c && !fuse->gmax_out && !fuse->no_store && !residual, PSAM_EINVAL,
                     "psam_gemm_f16x3p_ex: row LayerNorm over N == 512 needs M % 128 == 0 and combines only with bias / rowbias / activation / packed output");
        PSAM_REQUIRE(!rowbias || ((p.rowgroup & 31) == 0 && (ldrb & 3) == 0 && ((uintptr_t)rowbias & 15) == 0), PSAM_EINVAL,
                     "psam_gemm_f16x3p_ex: row LayerNorm over N == 512: rowbias needs rowgroup % 32 == 0 and 16-byte aligned rows");
        PSAM_REQUIRE(!fuse->pack_out || (fuse->out_scale && fuse->out_k1 == 0.f && fuse->out_k2 > 0.f && (ldc & 7) == 0 && ((uintptr_t)C & 31) == 0), PSAM_EINVAL,
                     "psam_gemm_f16x3p_ex: packed output after a row LayerNorm takes the a-priori bound in out_k2 (out_k1 == 0) and 32-byte aligned rows");
        PSAM_REQUIRE((((uintptr_t)fuse->row_ln_g | (uintptr_t)fuse->row_ln_b | (uintptr_t)scaleW | (uintptr_t)bias | (uintptr_t)C) & 15) == 0 && (ldc & 3) == 0, PSAM_EALIGN,
                     "psam_gemm_f16x3p_ex: 16-byte alignment");
        p.row_ln_g = fuse->row_ln_g; p.row_ln_b = fuse->row_ln_b; p.row_ln_eps = fuse->row_ln_eps;
        p.pack_out = fuse->pack_out; p.out_scale = fuse->out_scale; p.out_k1 = 0.f; p.out_k2 = fuse->out_k2;
        return launch_f16x3pp(70, p, stream);
#endif
    } else if (fuse && (fuse->row_ln_g || fuse->hyper)) {
        // full-row epilogues: a wave owns whole rows of N == 256 columns (128x256 tiles, four waves of 32 rows)
        PSAM_REQUIRE(N == 256 && (M & 127) == 0 && act != 3, PSAM_EINVAL, "psam_gemm_f16x3p_ex: row LayerNorm / hyper products need N == 256, M % 128 == 0");
        PSAM_REQUIRE(!fuse->stats && !fuse->ln_c && !fuse->gmax_out, PSAM_EINVAL, "psam_gemm_f16x3p_ex: row epilogues do not combine with stats / folded LN / group max");
        PSAM_REQUIRE(!fuse->row_ln_g || fuse->row_ln_b, PSAM_EINVAL, "psam_gemm_f16x3p_ex: row LayerNorm needs gamma and beta");
        PSAM_REQUIRE(!fuse->hyper || (fuse->masks && fuse->hyper_c > 0 && fuse->hyper_c <= 4 && fuse->hyper_rows > 0), PSAM_EINVAL,
                     "psam_gemm_f16x3p_ex: hyper products need masks, 1 <= hyper_c <= 4, hyper_rows > 0");
        PSAM_REQUIRE(!fuse->pack_out || (fuse->out_scale && (ldc & 7) == 0 && ((uintptr_t)C & 31) == 0), PSAM_EINVAL,
                     "psam_gemm_f16x3p_ex: packed output needs out_scale and 32-byte aligned output rows");
        PSAM_REQUIRE(!fuse->hyper || fuse->hyper_rows % 32 == 0, PSAM_EINVAL, "psam_gemm_f16x3p_ex: hyper_rows must be a multiple of 32");
        PSAM_REQUIRE(!fuse->no_store || fuse->hyper, PSAM_EINVAL, "psam_gemm_f16x3p_ex: no_store needs an epilogue product that keeps the result");
        PSAM_REQUIRE((((uintptr_t)fuse->row_ln_g | (uintptr_t)fuse->row_ln_b | (uintptr_t)fuse->hyper) & 15) == 0, PSAM_EALIGN, "psam_gemm_f16x3p_ex: 16-byte alignment");
        p.row_ln_g = fuse->row_ln_g; p.row_ln_b = fuse->row_ln_b; p.row_ln_eps = fuse->row_ln_eps;
        p.hyper = fuse->hyper; p.masks = fuse->masks; p.hyper_c = fuse->hyper_c; p.hyper_rows = fuse->hyper_rows;
        p.pack_out = fuse->pack_out; p.out_scale = fuse->out_scale; p.out_k1 = fuse->out_k1; p.out_k2 = fuse->out_k2; p.no_store = fuse->no_store;
        return launch_f16x3p<4, 1, 1, 8, 2, 0, 0, 2>(p, stream);
    }
    if (fuse && !fuse->hyper && (fuse->pack_out || fuse->stats || fuse->ln_c || fuse->gmax_out)) {
        // The fused epilogue paths exist for interior tiles of the two-tile-wide wave tiles only: whole 256-row / 128-column tiles.
        PSAM_REQUIRE((M & 255) == 0 && (N & 127) == 0, PSAM_EINVAL, "psam_gemm_f16x3p_ex: fused extras need M % 256 == 0 and N % 128 == 0");
        PSAM_REQUIRE(!fuse->pack_out || (fuse->out_scale && (ldc & 7) == 0 && ((uintptr_t)C & 31) == 0), PSAM_EINVAL,
                     "psam_gemm_f16x3p_ex: packed output needs out_scale and 32-byte aligned output rows");
        PSAM_REQUIRE(!fuse->stats || (act == 3 && fuse->stat_cols > 0 && fuse->stat_cols <= N / 2), PSAM_EINVAL,
                     "psam_gemm_f16x3p_ex: row statistics come with the SwiGLU epilogue (0 < stat_cols <= N / 2)");
        PSAM_REQUIRE(!fuse->ln_c || (fuse->ln_mean && fuse->ln_rstd && act != 3), PSAM_EINVAL, "psam_gemm_f16x3p_ex: folded LayerNorm needs mean, rstd, c (no SwiGLU)");
        PSAM_REQUIRE(((uintptr_t)fuse->ln_c & 15) == 0, PSAM_EALIGN, "psam_gemm_f16x3p_ex: ln_c must be 16-byte aligned");
        p.out_scale = fuse->out_scale; p.out_k1 = fuse->out_k1; p.out_k2 = fuse->out_k2; p.pack_out = fuse->pack_out;
        p.out_bound = fuse->pack_out ? fuse->out_bound : nullptr;
        PSAM_REQUIRE(!p.out_bound || !fuse->ln_c, PSAM_EINVAL, "psam_gemm_f16x3p_ex: out_bound does not combine with the folded LayerNorm");
        p.stats = fuse->stats; p.stat_cols = fuse->stat_cols; p.stat_segs = psam_gemm_f16x3p_stat_segs(N);
        p.ln_mean = fuse->ln_mean; p.ln_rstd = fuse->ln_rstd; p.ln_c = fuse->ln_c;
        PSAM_REQUIRE(!fuse->gmax_out || ((fuse->gmax_k == 32 || fuse->gmax_k == 64) && act != 3 && (fuse->gmax_ld & 3) == 0 && fuse->gmax_ld >= N &&
                                         ((uintptr_t)fuse->gmax_out & 15) == 0), PSAM_EINVAL,
                     "psam_gemm_f16x3p_ex: group maximum needs groups of 32 or 64 rows, no SwiGLU, a 16-byte aligned [M / k, >= N] output");
        PSAM_REQUIRE(!fuse->no_store || fuse->gmax_out, PSAM_EINVAL, "psam_gemm_f16x3p_ex: no_store only together with the group maximum");
        p.gmax_out = fuse->gmax_out; p.gmax_ld = fuse->gmax_ld; p.gmax_k = fuse->gmax_k; p.no_store = fuse->no_store;
        // group maximum: wave tiles of 64 rows (two stripes): 256x128 (cfg 4), 256x256 (cfg 14, N % 256 == 0), 128x128 of four waves (cfg 21);
        // row statistics / everything else: wave tiles two 32-column tiles wide (cfg 4, 9, 21, 28)
        if (fuse->gmax_out) { if (cfg != 4 && cfg != 14 && cfg != 21) cfg = (N % 256 == 0 && !fuse->stats) ? 14 : 4; }
        else if (cfg != 4 && cfg != 9 && cfg != 41 && cfg != 21 && cfg != 28) cfg = f16x3p_pick(M, N, K, act, true);
    }
    {   // ping-pong kernel (gemm_f16x3pp.hip) where it measured faster (f16x3pp_pick), or where a forced configuration names it
        const bool w_stats = p.stats != nullptr, w_gmax = p.gmax_out != nullptr, w_hyper = p.hyper != nullptr;
        const bool fused_any = w_stats || w_gmax || w_hyper || p.pack_out || p.ln_c;
        const bool shape_ok = !fused_any || ((M & 255) == 0 && (N & 127) == 0);
        const int forced = g_f16x3p_cfg;
        if (forced >= 50 && forced < 100) { if (shape_ok && f16x3pp_supports(forced, act, w_stats, w_gmax, w_hyper)) cfg = forced; }
        else if (forced < 0) {
            const int pp = f16x3pp_pick(M, N, K, act);
            if (pp >= 0 && shape_ok && f16x3pp_supports(pp, act, w_stats, w_gmax, w_hyper)) cfg = pp;
        }
    }
#ifdef PSAM_BUILD_EXPERIMENTS
    {   // unit-ring kernel (gemm_f16x3q.hip): a forced configuration 80 .., or the environment's choice for the large encoder GEMMs
        int q = (g_f16x3p_cfg >= 80 && g_f16x3p_cfg < 90) ? g_f16x3p_cfg : 0;
        if (!q && g_f16x3p_cfg < 0) {
            static int envq = -1;
            if (envq < 0) { const char* e = getenv("PSAM_GEMM_Q"); envq = e ? atoi(e) : 0; }
            if (envq >= 80 && M >= 2048 && N >= 1024 && K >= 512) q = (envq == 84 && (act == 3 || N % 192 != 0)) ? 80 : envq;
        }
        if (q && f16x3q_supports(q, p)) return launch_f16x3q(q, p, stream);
    }
#endif
    if (cfg >= 80 && cfg < 90) cfg = f16x3p_pick(M, N, K, act, true);      // (unit-ring configurations: experiments builds only)
#ifdef PSAM_BUILD_EXPERIMENTS
    // persistent forms of cfg 21, both measured and not adopted (profiles/r05/r05_continuous_sweep.txt, r05_streamk_sweep.txt): 94 = whole tiles from per-XCD
    // queues, one continuous slab stream per workgroup (gemm_f16x3c.hip: the same bits as cfg 21, the same time; 95: one workgroup per CU; PSAM_GEMM_CONTINUOUS=1
    // switches it in for the batch-sized launches), 90 .. 93 = even shares of the K slabs (stream-K, gemm_f16x3s.hip: slower)
    if (cfg == 94 || cfg == 95 || (cfg == 21 && g_f16x3p_cfg < 0 && M >= 2048 && f16x3p_continuous_enabled())) {
        int32_t rc = PSAM_OK;
        if (f16x3p_use_register_epilogue(p) && launch_f16x3c(p, stream, rc, cfg == 95 ? 1 : 2)) return rc;
        if (cfg >= 90) cfg = 21;
    }
    if (cfg >= 90 && cfg <= 93) {
        int32_t rc = PSAM_OK;
        if (f16x3p_use_register_epilogue(p) && launch_f16x3s(p, stream, rc, cfg - 90)) return rc;
        cfg = 21;
    }
#endif
    if (cfg >= 90 && cfg < 100) cfg = 21;
#ifdef PSAM_GEMM_ABLATE
    if (cfg >= 3000 && cfg < 3100) {   // 3000 + ablation bits: the PRODUCTION instance (128x128, four waves, mid-slab release, register epilogue); 64 = timing
        switch (cfg - 3000) {
            case 0: return launch_f16x3p<2, 2, 2, 2, 2, 0, 0, 2, 1>(p, stream);
            case 64: return launch_f16x3p<2, 2, 2, 2, 2, 0, 64, 2, 1>(p, stream);
            case 1: return launch_f16x3p<2, 2, 2, 2, 2, 0, 1, 2, 1>(p, stream);
            case 2: return launch_f16x3p<2, 2, 2, 2, 2, 0, 2, 2, 1>(p, stream);
            case 4: return launch_f16x3p<2, 2, 2, 2, 2, 0, 4, 2, 1>(p, stream);
            case 5: return launch_f16x3p<2, 2, 2, 2, 2, 0, 5, 2, 1>(p, stream);
            case 19: return launch_f16x3p<2, 2, 2, 2, 2, 0, 19, 2, 1>(p, stream);
            default: break;
        }
    }
#endif
    if (cfg >= 50 && (cfg < 100 || cfg >= 200)) return launch_f16x3pp(cfg, p, stream);
#ifdef PSAM_GEMM_ABLATE
    if (cfg >= 100) {   // 100 + 32 * which + ablation bits; which: 0 = 128x128 4 waves S2, 1 = 256x128 8 waves S3, 2 = 256x192 S2, 3 = 256x256 S2
        const int which = (cfg - 100) / 32, abl = (cfg - 100) % 32;
#define ABL_CASE(B)                                                                      \
    case B:                                                                              \
        if (which == 0) return launch_f16x3p<2, 2, 2, 2, 2, 0, B>(p, stream);            \
        if (which == 1) return launch_f16x3p<4, 2, 2, 2, 3, 0, B>(p, stream);            \
        if (which == 2) return launch_f16x3p<4, 2, 2, 3, 2, 0, B>(p, stream);            \
        return launch_f16x3p<4, 2, 2, 4, 2, 0, B>(p, stream);
        switch (abl) { ABL_CASE(0) ABL_CASE(1) ABL_CASE(2) ABL_CASE(3) ABL_CASE(4) ABL_CASE(5) ABL_CASE(16) ABL_CASE(19) ABL_CASE(23) default: break; }
#undef ABL_CASE
    }
#endif
    switch (cfg) {   // the configurations that won somewhere in the sweeps (profiles/r02/r02_gemm_p_sweep_*.log); numbering kept from the sweeps
        case 0: return launch_f16x3p<2, 2, 2, 2, 2, 0>(p, stream);            // 128x128, 4 waves of 64x64, 2 stages (64 KiB): 2 workgroups per CU
        case 4: return launch_f16x3p<4, 2, 2, 2, 3, 0>(p, stream);            // 256x128, 8 waves, 3 stages (144 KiB)
        case 9: return launch_f16x3p<4, 2, 1, 2, 4, 1>(p, stream);            // 128x128, 8 waves of 32x64, 4 stages + look-ahead fragments (128 KiB)
        case 12: return launch_f16x3p<4, 2, 2, 3, 2, 0>(p, stream);           // 256x192, 8 waves of 64x96, 2 stages (112 KiB); no SwiGLU epilogue
        case 14: return launch_f16x3p<4, 2, 2, 4, 2, 0>(p, stream);           // 256x256, 8 waves of 64x128, 2 stages (128 KiB)
        case 21: return f16x3p_use_register_epilogue(p) ? launch_f16x3p<2, 2, 2, 2, 2, 0, 0, 2, 1>(p, stream)
                                                        : launch_f16x3p<2, 2, 2, 2, 2, 0, 0, 2>(p, stream);     // 128x128, 4 waves, mid-slab stage release
        case 23: return launch_f16x3p<4, 2, 2, 3, 2, 0, 0, 2>(p, stream);     // 256x192, mid-slab stage release
        case 28: return launch_f16x3p<4, 2, 1, 2, 2, 0, 0, 2>(p, stream);     // 128x128, 8 waves of 32x64, 2 stages, mid-slab release (70 KiB): 2 per CU
        case 29: return f16x3p_use_register_epilogue(p) ? launch_f16x3p<4, 2, 1, 2, 4, 1, 0, 0, 1>(p, stream)      // cfg 9 with the register epilogue (round 6)
                                                        : launch_f16x3p<4, 2, 1, 2, 4, 1>(p, stream);
        // deeper rings for single-cloud shapes, whose K loop is bound by the LDS-DMA round trip / slabs in flight (round 6, profiles/r06/r06_small_m.txt)
        case 41: return launch_f16x3p<4, 2, 1, 2, 5, 1>(p, stream);           // cfg 9 with FIVE stages (exactly 160 KiB): four slabs in flight instead of three
        case 42: return launch_f16x3p<4, 1, 1, 3, 5, 1>(p, stream);           // 128x96, 4 waves of 32x96, five stages (140 KiB); no SwiGLU epilogue
        // 30 / 31: three workgroups per CU.  Alone they win on the short launches (proj 38.4 -> 32.3 us, up.3 233 -> 205 us), in the pipelined
        // bench (two batches' kernels co-scheduled) they lose 1.5 % (profiles/r02/r02_gemm_tri_tile.txt): reachable through force_config only
        case 30: return launch_f16x3p<2, 2, 2, 1, 2, 0, 0, 2>(p, stream);     // 128x64, 4 waves of 64x32, 48 KiB (no SwiGLU / fused extras)
        case 31: return launch_f16x3p<2, 2, 1, 2, 2, 0, 0, 2>(p, stream);     // 64x128, 4 waves of 32x64, 48 KiB: 3 workgroups per CU
        case 40: return launch_f16x3p<4, 1, 1, 8, 2, 0, 0, 2>(p, stream);     // 128x256, 4 waves of 32x256 (whole rows per wave: row epilogues), 133 KiB
        default: break;
    }
    psam_set_error("psam_gemm_f16x3p: unknown config");
    return PSAM_EINVAL;
}

PSAM_API int32_t psam_gemm_f16x3p(const void* A, int64_t lda, const float* scaleA, const void* W, int64_t ldw, const float* scaleW, float* C,
                                  int64_t ldc, const float* bias, const float* residual, int64_t ldr, const float* rowbias, int64_t ldrb,
                                  int32_t rowgroup, int32_t M, int32_t N, int32_t K, float alpha, int32_t act, hipStream_t stream) {
    return psam_gemm_f16x3p_ex(A, lda, scaleA, W, ldw, scaleW, C, ldc, bias, residual, ldr, rowbias, ldrb, rowgroup, M, N, K, alpha, act, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------- LayerNorm statistics from the partials
// stats [rows, segs, 2] = (mean, centred sum of squares) of consecutive 32-column segments of a row (the last used one may hold fewer
// columns); merged in a FIXED order (Chan et al.: exact in real arithmetic, no cancellation; bit-reproducible) -> mean[rows],
// rstd[rows] = 1 / sqrt(M2 / cols + eps).
// One wave per row: lane l merges segments l, l + 64, ... in order, then the 64 lane partials merge in a fixed butterfly.
__device__ __forceinline__ void chan_merge(float& n, float& mu, float& m2, float nb, float mb, float qb) {
    const float nn = n + nb;
    if (nn > 0.f) {
        const float d = mb - mu, f = nb / nn;
        mu += d * f;
        m2 += qb + d * d * (n * f);
        n = nn;
    }
}
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ stats, int rows, int segs, int cols, float eps,
                                                                float* __restrict__ mean, float* __restrict__ rstd) {
    typedef float st_f32x2 __attribute__((ext_vector_type(2)));
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const st_f32x2* st = reinterpret_cast<const st_f32x2*>(stats + (int64_t)row * segs * 2);
    float n = 0.f, mu = 0.f, m2 = 0.f;
    for (int t = lane; t * 32 < cols; t += 64) {
        const st_f32x2 v = st[t];
        chan_merge(n, mu, m2, (float)(cols - t * 32 < 32 ? cols - t * 32 : 32), v[0], v[1]);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float nb = __shfl_xor(n, o, 64), mb = __shfl_xor(mu, o, 64), qb = __shfl_xor(m2, o, 64);
        // both partners must compute the SAME merged triple: order the pair by lane (lower lane first)
        const bool lo = (lane & o) == 0;
        float n0 = lo ? n : nb, mu0 = lo ? mu : mb, m20 = lo ? m2 : qb;
        chan_merge(n0, mu0, m20, lo ? nb : n, lo ? mb : mu, lo ? qb : m2);
        n = n0; mu = mu0; m2 = m20;
    }
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = 1.0f / sqrtf(m2 / (float)cols + eps);
    }
}

PSAM_API int32_t psam_ln_stats_finalize(const float* stats, int32_t rows, int32_t segs, int32_t cols, float eps, float* mean, float* rstd,
                                        hipStream_t stream) {
    PSAM_REQUIRE(stats && mean && rstd, PSAM_EINVAL, "psam_ln_stats_finalize: null pointer");
    PSAM_REQUIRE(rows > 0 && cols > 0 && segs * 32 >= cols, PSAM_EINVAL, "psam_ln_stats_finalize: bad shape");
    hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((unsigned)psam_cdiv(rows, 4)), dim3(256), 0, stream, stats, rows, segs, cols, eps, mean, rstd);
    if (psam_ablate_repeat() & 4) hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((unsigned)psam_cdiv(rows, 4)), dim3(256), 0, stream, stats, rows, segs, cols, eps, mean, rstd);
    return psam_launch_status("psam_ln_stats_finalize: launch failed");
}
