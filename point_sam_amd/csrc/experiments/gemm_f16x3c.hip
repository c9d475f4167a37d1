// Packed-operand fp32-grade GEMM, CONTINUOUS schedule ("f16x3c", round 5): the production tile of gemm_f16x3p.hip -- 128x128, four waves of 64x64, two
// 32-k stages with the mid-slab release, LDS-DMA of g8-packed operands, hi*lo + lo*hi + hi*hi on v_mfma_f32_32x32x16_f16, register epilogue
// (gemm_epilogue_t.h) -- run by workgroups that stay on their CU, draw WHOLE tiles from a queue and keep ONE uninterrupted stream of K slabs going
// across tile boundaries.
//
// Why (profiles/r05/r05_gemm_timing.txt, r05_streamk_timing.txt: s_memtime stamps per wave).  With two workgroups resident on a CU the K loop of the
// production kernel already runs at ~95 % of the matrix pipe (a slab in ~1620 clk per wave, two waves per SIMD); one resident workgroup reaches ~0.65.
// What a launch loses is what surrounds the loop, per tile: 3.3 k clk from the workgroup's start to its first slab (arguments, descriptors, the first
// DMA's round trip), the dispatch of a new workgroup into the freed slot, and 8.8 k clk of epilogue (dependent constant loads, the stores) in which
// the workgroup issues no MFMA -- 12 k of a tile's 71 k, covered only as far as the other resident workgroup happens to be inside its loop.
// Even shares of the K slabs (gemm_f16x3s.hip, stream-K) made it worse: parking / counting / combining parts costs more than the imbalance it removes.
// Here:
//   * a workgroup never restarts: the DMA of the NEXT tile's first two slabs is issued during the last two slabs of the current tile, into the ring
//     stages those slabs free -- the K loop never drains, there is no prologue after the first tile;
//   * tiles come from per-XCD queues (XCD x draws the x-th eighth of the panel order, as gemm_f16x3p.hip's tile permutation deals them; an empty queue
//     falls through to the next XCD's): a workgroup that finds its slot late (the tokenizer of the next batch holds 8 CUs for a part of every step)
//     simply draws less -- no tile is split, a tile's bits are those of the one-workgroup-per-tile kernel;
//   * the next tile id is drawn (one returning atomic by one lane) six slabs before it is needed and handed to the workgroup through LDS;
//   * the epilogue of tile i runs between the last slab of tile i and the first slab of tile i + 1, whose operands are already in LDS: the first
//     fragments are read without any wait -- the epilogue's own loads were issued after that DMA and have returned (the VM counter is in order);
//     nothing waits for the epilogue's stores.
// Same products, same summation order per tile as cfg 21: the same bits.
#include <type_traits>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>
#include "common.h"
#include "gemm_f16x3p_args.h"
#include "gemm_epilogue.h"
#include "gemm_epilogue_t.h"

namespace {
constexpr int CK_NW = 4, CK_BM = 128, CK_BN = 128, CK_ROWB = 128, CK_A_BYTES = CK_BM * CK_ROWB, CK_STAGE = 2 * CK_A_BYTES;
constexpr int CK_NL = CK_STAGE / 1024 / CK_NW;      // 8 DMA pieces (1 KiB: 8 rows x 128 B) per wave per slab
constexpr int CK_A_BLK = CK_A_BYTES / 1024;         // pieces 0 .. 15: A rows, 16 .. 31: W rows
constexpr int CK_LDS = 2 * CK_STAGE + 64;           // ring of two stages + the hand-over word of the tile queue
constexpr int CK_AHEAD = 6;                         // the next tile is drawn this many slabs before the current one ends (it is needed two slabs before)
}  // namespace

struct F16CArgs {
    F16PArgs g;
    int* queue;        // [0..7] per-XCD heads, [8] workgroups finished; all zero between launches (the last workgroup to finish resets them)
    int nslabs;        // K / 32 (>= 16)
    int tiles;
};

template <int F, bool TIMING = false>
__global__ __launch_bounds__(256, 2) void gemm_f16x3c_kernel(const F16CArgs a) {
    const F16PArgs& p = a.g;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;
    const int xcd = blockIdx.x & 7;      // (where workgroup ids are dealt round-robin to the XCDs this is the workgroup's XCD: a matter of L2 locality only)
    int* ctl = reinterpret_cast<int*>(smem + 2 * CK_STAGE);      // (re-read across the slabs' barriers: asm statements with a memory clobber)
    unsigned tb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tk0 = 0, tq = 0;
    if (TIMING) { tk0 = gemm_now(); tq = tk0; }
#define CK_STAMP(slot) do { if (TIMING) { const unsigned long long _t = gemm_now(); tb[slot] += (unsigned)(_t - tq); tq = _t; } } while (0)

    // ---- the queue: XCD x owns tiles [x T / 8, (x + 1) T / 8) of the panel order; draw from the own queue first, then from the others in turn
    // A draw = ONE returning atomic on the own XCD's head by one lane.  A returned atomic costs a memory round trip (and the heads are contended: the
    // workgroups of an XCD run in step), and hipcc waits for a returned atomic where it is issued: the instruction is written in assembly, which the
    // compiler does not count, and its result is read only after a point where the VM counter proves it complete -- at the kernel's start an explicit
    // drain, inside the K loop the counted wait of the slab after next (the atomic is then older than the eight DMA pieces that may stay in flight).
    // No stealing from other XCDs' queues: a dry queue is the launch's tail, and every probe of another head is another contended round trip.
    const int q_lo = (int)((int64_t)xcd * a.tiles / 8), q_n = (int)((int64_t)(xcd + 1) * a.tiles / 8) - q_lo;
    int* const q_head = a.queue + xcd;
    auto draw_issue = [&](int& ret) {      // one lane; `ret` must not be read before the matching wait
        asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(ret) : "v"(q_head), "v"(1) : "memory");
    };
    auto resolve = [&](int id) -> int { return id < q_n ? q_lo + id : -1; };
    auto tile_origin = [&](int tile, int& m0, int& n0) {      // the panel order of gemm_f16x3p.hip
        const int pfull = p.tiles_m * p.panel, pn = tile / pfull, prem = tile - pn * pfull;
        const int pw = p.tiles_n - pn * p.panel < p.panel ? p.tiles_n - pn * p.panel : p.panel;
        m0 = (prem / pw) * CK_BM;
        n0 = (pn * p.panel + prem % pw) * CK_BN;
    };

    // fragment offsets inside a stage: row r32 of a 32-row tile, chunk 4 s + 2 h + pl (s = k16 step, pl = hi / lo plane), swizzled as the DMA stores them
    int fa_off[2][2], fw_off[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const int f = r32 * CK_ROWB + (((4 * s + 2 * h + pl) ^ ((r32 >> 1) & 7)) << 4);
            fa_off[s][pl] = wm * 64 * CK_ROWB + f;
            fw_off[s][pl] = CK_A_BYTES + wn * 64 * CK_ROWB + f;
        }
    // per-lane DMA source offsets relative to a tile's first row / column (whole tiles only: the host checks M, N % 128 == 0)
    int voff[CK_NL];
#pragma unroll
    for (int i = 0; i < CK_NL; ++i) {
        const int b = wave + i * CK_NW;
        const bool isw = b >= CK_A_BLK;
        const int row = (isw ? b - CK_A_BLK : b) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        voff[i] = (int)((int64_t)row * (isw ? p.ldw : p.lda) * 4) + chunk * 16;
    }
    // descriptors of the tile whose slabs are being ISSUED (the current tile, or from its last two slabs on the next one)
    __amdgpu_buffer_rsrc_t rsA, rsW;
    auto dma_setup = [&](int m0, int n0) {
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)m0 * p.lda * 4), 0, 0x7fffffff, 0x00020000);
        rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)n0 * p.ldw * 4), 0, 0x7fffffff, 0x00020000);
    };
    auto issue_one = [&](int i, int slab, int stage) {
        const int b = wave + i * CK_NW;
        unsigned char* dst = smem + stage * CK_STAGE + b * 1024;
        if (b >= CK_A_BLK) P_DMA16(rsW, dst, voff[i], slab * CK_ROWB);
        else P_DMA16(rsA, dst, voff[i], slab * CK_ROWB);
    };

    pf32x16 acc[2][2];
    pf16x8 f0a[2][2], f0w[2][2], f1a[2][2], f1w[2][2];
    auto read_frag = [&](int n, int s, int stage, pf16x8 (&fa)[2][2], pf16x8 (&fw)[2][2]) {
        const unsigned char* base = smem + stage * CK_STAGE;
        if (n < 4) { const int i = n >> 1, pl = n & 1; fa[i][pl] = *reinterpret_cast<const pf16x8*>(base + fa_off[s][pl] + i * 32 * CK_ROWB); }
        else { const int m = n - 4, j = m >> 1, pl = m & 1; fw[j][pl] = *reinterpret_cast<const pf16x8*>(base + fw_off[s][pl] + j * 32 * CK_ROWB); }
    };
    auto mfma = [&](int m, const pf16x8 (&fa)[2][2], const pf16x8 (&fw)[2][2]) {      // term-major (hi*lo, lo*hi, hi*hi), as gemm_f16x3p.hip
        constexpr int PA[3] = {0, 1, 0}, PW[3] = {1, 0, 0};
        const int term = m / 4, ij = m % 4, i = ij / 2, j = ij % 2;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j][PW[term]], fa[i][PA[term]], acc[i][j], 0, 0, 0);
    };
    // One slab read from ring stage `st`; the DMA it issues (after its mid-slab barrier has freed the stage) is slab `islab` of the tile the descriptors
    // point at.  WAIT: 1 = counted wait for this slab's own pieces (the 8 newest operations of the VM queue may stay in flight), 0 = everything,
    // -1 = none (the slab landed before an epilogue whose loads have returned).
    int st = 0;      // ring stage of the slab being read: alternates from slab to slab, ACROSS tile boundaries too (an odd slab count flips a tile's parity)
    auto body = [&](int islab, auto wait_c, auto issue_c) {
        constexpr int WAIT = decltype(wait_c)::value;
        constexpr bool DO_ISSUE = decltype(issue_c)::value;
        if (WAIT == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CK_NL) : "memory");
        else if (WAIT == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int n = 0; n < 8; ++n) read_frag(n, 0, st, f0a, f0w);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            mfma(m, f0a, f0w);
            if (m < 8) read_frag(m, 1, st, f1a, f1w);
            else if (DO_ISSUE) issue_one(m - 8, islab, st);
            if (m == 7) {      // every step-1 fragment read is issued: wait for them, release the stage
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            mfma(m, f1a, f1w);
            if (m < 4 && DO_ISSUE) issue_one(4 + m, islab, st);
            __builtin_amdgcn_sched_barrier(0);
        }
        st ^= 1;
    };
    using std::integral_constant;
    using I1 = integral_constant<int, 1>; using I0 = integral_constant<int, 0>; using IM = integral_constant<int, -1>;
    using BT = integral_constant<bool, true>; using BF = integral_constant<bool, false>;

    // ---- first tile
    {
        int first = 0;
        if (tid == 0) {
            draw_issue(first);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(first)::"memory");
            ctl[0] = resolve(first);
        }
    }
    __syncthreads();
    int cur = __builtin_amdgcn_readfirstlane(ctl[0]);
    __syncthreads();
    const int n = a.nslabs;
    int m0 = 0, n0 = 0;
    bool after_epilogue = false;
    if (cur >= 0) {
        tile_origin(cur, m0, n0);
        dma_setup(m0, n0);
#pragma unroll
        for (int i = 0; i < CK_NL; ++i) issue_one(i, 0, 0);
#pragma unroll
        for (int i = 0; i < CK_NL; ++i) issue_one(i, 1, 1);
    }
    if (TIMING) { tq = gemm_now(); tb[3] = (unsigned)(tq - tk0); }
    while (cur >= 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // slabs 0 and 1: in LDS already when they follow an epilogue (no wait), otherwise the counted wait; from slab 2 on the counted wait
        if (after_epilogue) { body(2, IM{}, BT{}); body(3, IM{}, BT{}); }
        else { body(2, I1{}, BT{}); body(3, I1{}, BT{}); }
        int nxt = -1, nm0 = 0, nn0 = 0;
        int t = 2;
        for (; t < n - CK_AHEAD; ++t) body(t + 2, I1{}, BT{});
        // slab n - 6: one lane draws the next tile (a returning atomic: one more operation in wave 0's VM queue -- its counted waits then cover a DMA piece
        // more than they need, never less); after slab n - 5 its result is resolved and handed over through LDS; the barriers of slab n - 4 publish it
        int drawn = 0;
        if (tid == 0) draw_issue(drawn);
        body(t + 2, I1{}, BT{}); ++t;
        body(t + 2, I1{}, BT{}); ++t;      // its counted wait leaves only this slab's own eight DMA pieces in flight: the atomic has returned
        if (tid == 0) { asm volatile("" : "+v"(drawn)); ctl[0] = resolve(drawn); }
        body(t + 2, I1{}, BT{}); ++t;      // (its first barrier orders the LDS store before the reads below)
        nxt = __builtin_amdgcn_readfirstlane(ctl[0]);
        for (; t < n - 2; ++t) body(t + 2, I1{}, BT{});
        // slabs n - 2, n - 1: their DMA is the next tile's slab 0 / 1 (nothing when the queues are empty)
        if (nxt >= 0) {
            tile_origin(nxt, nm0, nn0);
            dma_setup(nm0, nn0);
            body(0, I1{}, BT{});
            body(1, I1{}, BT{});
        } else {
            body(0, I1{}, BF{});
            body(0, I0{}, BF{});
        }
        CK_STAMP(0);
        // ---- epilogue: accumulators -> C in registers; the ring is not touched
        gemm_store_tile_t_impl<2, 2, F, F16PArgs, false>(p, acc, m0 + wm * 64, n0 + wn * 64, lane, p.C, p.residual, true);
        if (TIMING) { tb[2] += 1; }
        CK_STAMP(1);
        cur = nxt; m0 = nm0; n0 = nn0;
        after_epilogue = true;
    }
    // ---- the last workgroup to finish leaves the queue zeroed for the next launch
    if (tid == 0) {
        const int done = __hip_atomic_fetch_add(a.queue + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
#pragma unroll
            for (int k = 0; k < 9; ++k) __hip_atomic_store(a.queue + k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (TIMING && p.dbg) {
        const unsigned long long te = gemm_now();
        if (lane == 0) {
            unsigned* d = p.dbg + ((size_t)blockIdx.x * CK_NW + wave) * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) d[i] = tb[i];
            d[10] = (unsigned)tk0; d[11] = (unsigned)(tk0 >> 32); d[12] = (unsigned)te; d[13] = (unsigned)(te >> 32);
            d[14] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
            d[15] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        }
    }
#undef CK_STAMP
}

// ---------------------------------------------------------------------------------------------- host
namespace {
// per (device, stream): the queue words; allocated on first use (not inside a graph capture)
static int* ck_queue(hipStream_t stream) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, int*> table;
    int dev = 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipGetDevice(&dev) != hipSuccess || hipStreamIsCapturing(stream, &cs) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto it = table.find({dev, stream});
    if (it != table.end()) return it->second;
    if (cs != hipStreamCaptureStatusNone) return nullptr;
    int* q = nullptr;
    if (hipMalloc(&q, 64 * sizeof(int)) != hipSuccess) return nullptr;
    if (hipMemsetAsync(q, 0, 64 * sizeof(int), stream) != hipSuccess) { (void)hipFree(q); return nullptr; }
    table[{dev, stream}] = q;
    return q;
}

template <int F, bool TIMING = false>
static int32_t launch_c(const F16CArgs& a, int grid, hipStream_t stream) {
#ifdef PSAM_GEMM_ABLATE
    if (!TIMING && a.g.dbg) return launch_c<F, true>(a, grid, stream);      // a timing buffer is set: the stamped instance
#endif
    static unsigned long long attr_done = 0;
    if (!f16x3p_reserve_lds(&gemm_f16x3c_kernel<F, TIMING>, CK_LDS, attr_done)) {
        psam_set_error("psam_gemm_f16x3p: cannot reserve LDS");
        return PSAM_EINVAL;
    }
    hipLaunchKernelGGL((gemm_f16x3c_kernel<F, TIMING>), dim3((unsigned)grid), dim3(256), CK_LDS, stream, a);
    return psam_launch_status("psam_gemm_f16x3p: launch failed");
}
}  // namespace

// Option set of a launch as the EP_* bits of the register epilogue, or -1 when the continuous kernel does not serve it.
static int f16x3c_option_set(const F16PArgs& p) {
    if (p.gmax_out || p.row_ln_g || p.hyper || p.no_store || p.rowbias || p.ksplit > 1) return -1;
    const bool swiglu = p.act == 3;
    const int opt = (swiglu ? EP_SWIGLU : 0) | ((p.residual && !swiglu) ? EP_RES : 0) | (!swiglu && p.act == 1 ? EP_GELU : 0) | (!swiglu && p.act == 2 ? EP_RELU : 0) |
                    ((swiglu && p.stats) ? EP_STATS : 0) | (p.pack_out ? EP_PACK : 0) | (p.ln_c ? EP_LNC : 0) | ((p.pack_out && !p.ln_c && p.out_bound) ? EP_BND : 0);
    switch (opt) {
        case 0: case EP_RES: case EP_PACK: case EP_LNC | EP_RES: case EP_GELU: case EP_SWIGLU | EP_STATS | EP_PACK | EP_BND: case EP_SWIGLU | EP_STATS | EP_PACK: case EP_SWIGLU: return opt;
        default: return -1;
    }
}

// gemm_f16x3p.hip calls this for the launches its 128x128 register-epilogue configuration would take: true = launched here (rc holds the status).
// wgs_per_cu: 2 (default) or 1 (measurement).
bool launch_f16x3c(F16PArgs& p, hipStream_t stream, int32_t& rc, int wgs_per_cu) {
    const int opt = f16x3c_option_set(p);
    if (opt < 0 || (p.M & 127) || (p.N & 127) || (p.K & 31) || p.K < 512) return false;
    // interior-tile conditions of the register epilogue (alignment of C / residual / bias rows), checked once for the launch
    const bool vec_ok = ((p.ldc & 3) == 0) && (((uintptr_t)p.C & 15) == 0) && (!p.residual || (((p.ldr & 3) == 0) && (((uintptr_t)p.residual & 15) == 0))) &&
                        (!p.bias || (((uintptr_t)p.bias & 15) == 0)) && ((((uintptr_t)p.scaleW | (uintptr_t)p.ln_c) & 15) == 0);
    if (!vec_ok) return false;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    }
    F16CArgs a;
    p.tiles_m = p.M / CK_BM; p.tiles_n = p.N / CK_BN;
    p.panel = f16x3p_panel(p.tiles_m, p.tiles_n, CK_BM, CK_BN, p.K);
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
    if (tiles >= ((int64_t)1 << 28)) return false;
    a.nslabs = p.K / 32;
    a.tiles = (int)tiles;
    int grid = (wgs_per_cu == 1 ? 1 : 2) * ncu;
    if (grid > tiles) grid = (int)tiles;
    int* q = ck_queue(stream);
    if (!q) return false;
    a.g = p; a.queue = q;
    switch (opt) {
        case 0: rc = launch_c<0>(a, grid, stream); break;
        case EP_RES: rc = launch_c<EP_RES>(a, grid, stream); break;
        case EP_PACK: rc = launch_c<EP_PACK>(a, grid, stream); break;
        case EP_LNC | EP_RES: rc = launch_c<EP_LNC | EP_RES>(a, grid, stream); break;
        case EP_GELU: rc = launch_c<EP_GELU>(a, grid, stream); break;
        case EP_SWIGLU | EP_STATS | EP_PACK | EP_BND: rc = launch_c<EP_SWIGLU | EP_STATS | EP_PACK | EP_BND>(a, grid, stream); break;
        case EP_SWIGLU | EP_STATS | EP_PACK: rc = launch_c<EP_SWIGLU | EP_STATS | EP_PACK>(a, grid, stream); break;
        case EP_SWIGLU: rc = launch_c<EP_SWIGLU>(a, grid, stream); break;
        default: return false;
    }
    return true;
}

// After a failed launch: re-zero the stream's queue words (psam_gemm_f16x3p_reset_splitk_state calls this too).
void f16x3c_reset_state(hipStream_t stream) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
    int* q = ck_queue(stream);
    if (q) (void)hipMemsetAsync(q, 0, 64 * sizeof(int), stream);
}
