// Packed-operand fp32-grade GEMM, PERSISTENT / STREAM-K schedule ("f16x3s", round 5): the production tile of gemm_f16x3p.hip -- 128x128, four waves
// of 64x64, two 32-k stages with the mid-slab release, LDS-DMA of g8-packed operands, hi*lo + lo*hi + hi*hi on v_mfma_f32_32x32x16_f16, register
// epilogue (gemm_epilogue_t.h) -- run by workgroups that STAY on their CU and share the launch's K slabs evenly.
//
// Why (profiles/r05/r05_gemm_timing.txt: s_memtime stamps per wave in the production kernel).  With two workgroups resident on a CU -- two waves per
// SIMD -- the K loop is matrix-pipe bound: a wave needs ~1530 clk per slab for 768 clk of MFMA issue, two of them fill the pipe.  What the launch
// loses is everything around the loop:
//   * a workgroup's prologue (3.3 k clk until its first slab has landed) and epilogue (8.8 k clk: dependent constant loads, stores, the drain) issue no
//     MFMA, and workgroups that were dispatched together reach them together;
//   * tiles do not divide among the CUs' 512 slots: qkv (4096 x 3072) is 768 tiles = 1.5 per slot, so after the first round a CU runs ONE workgroup
//     -- a lone wave per SIMD reaches 0.42 of the pipe (1843 clk per slab: fragment latency, two barriers and eight DMA issues sit in its own issue
//     stream) -- for a third of the launch (45 k of 142 k clk per CU); proj / fc2 (256 tiles) never have a second workgroup at all;
//   * every new workgroup costs a dispatch and a cold prologue.
// Here the grid is 2 x #CU persistent workgroups (8 | grid; workgroup w runs on XCD w % 8).  The launch's work is the sequence of (tile, slab) units
// in tile order; XCD x takes the x-th eighth of it (the tile order is the panel order of gemm_f16x3p.hip: an XCD keeps its W panel in L2), workgroup j
// of the XCD the j-th share of that eighth, rounded so that no piece of a tile is shorter than 4 slabs.  A workgroup therefore runs: possibly the END
// of a tile someone else started, whole tiles, possibly the START of a tile someone else finishes -- all CUs finish together.  A tile covered by
// several workgroups is combined by the split-K fix-up of gemm_f16x3p.hip, generalised: every part parks its raw accumulators (device-coherent sc1
// stores, 64 KiB, slot 2 w + {0: the part starts the workgroup's range, 1: it starts at a tile boundary}), counts itself in on the tile's counter, and
// the LAST arrival adds the parts in K order (its own re-read: the sum never depends on who came last), resets the counter and runs the epilogue.
// Between two pieces the workgroup issues the first two slabs of the NEXT piece before it runs the epilogue (or parks the accumulators) of the
// finished one: the ring is free -- the register epilogue does not touch LDS -- so the next prologue's DMA latency and the epilogue's load / store
// latencies overlap.
// Results: a whole tile's bits are those of gemm_f16x3p.hip's cfg 21; a split tile's differ in the last bits (partial sums over K ranges), and WHICH
// tiles are split depends on the launch's shape (deterministic for a shape, run to run and graph replay to graph replay).
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
constexpr int SK_NW = 4, SK_BM = 128, SK_BN = 128, SK_ROWB = 128, SK_A_BYTES = SK_BM * SK_ROWB, SK_STAGE = 2 * SK_A_BYTES;
constexpr int SK_NL = SK_STAGE / 1024 / SK_NW;      // 8 DMA pieces (1 KiB: 8 rows x 128 B) per wave per slab
constexpr int SK_A_BLK = SK_A_BYTES / 1024;         // pieces 0 .. 15: A rows, 16 .. 31: W rows
constexpr int SK_MINS = 4;                          // no piece of a tile shorter than this many slabs
constexpr int SK_LDS = 2 * SK_STAGE + 64;           // ring of two stages + the fix-up's flag word
constexpr int SK_PART_FLOATS = 128 * 128;           // one parked accumulator tile
}  // namespace

struct F16SArgs {
    F16PArgs g;
    float* part;       // [2 * grid] parked accumulator tiles
    int* count;        // [tiles] arrival counters, zero between launches
    int nslabs;        // K / 32
    int units;         // tiles * nslabs
    int per_xcd;       // grid / 8
    unsigned inv_per_xcd;     // ceil(2^32 / per_xcd), inv_nslabs = ceil(2^32 / nslabs): v / d == umulhi(v, inv_d) for v * d < 2^32 (the host checks the ranges)
    unsigned inv_nslabs;
    int whole;         // 1: no tile is split -- workgroup q runs the tiles q, q + grid, ... (persistence and the pipelined prologue only: A/B against the shares)
};

// first unit of workgroup q (linear order: XCD-major) -- q in [0, grid]; boundaries closer than SK_MINS slabs to a tile boundary snap onto it
__device__ __forceinline__ int sk_bound(const F16SArgs& a, int q) {      // multiply-high by precomputed inverses: no division
    const int x = (int)__umulhi((unsigned)q, a.inv_per_xcd), j = q - x * a.per_xcd;
    const int lo = (int)(((int64_t)x * a.units) >> 3), hi = (int)(((int64_t)(x + 1) * a.units) >> 3);
    int u = lo + (int)__umulhi((unsigned)j * (unsigned)(hi - lo), a.inv_per_xcd);
    const int r = u - (int)__umulhi((unsigned)u, a.inv_nslabs) * a.nslabs;
    if (r < SK_MINS) u -= r;
    else if (a.nslabs - r < SK_MINS) u += a.nslabs - r;
    return u;
}

// TIMING (measurement builds, -DPSAM_GEMM_ABLATE): s_memtime stamps per wave -> a.g.dbg, 16 words per wave: cycles in [0] K loops, [1] issuing the next
// piece's first slabs, [2] parking a part, [3] counting in, [4] combining the parts, [5] epilogues, [6] kernel entry -> first loop; [7] pieces,
// [8] slabs, [9] tiles finished, [10..13] absolute start / end stamps, [14] HW_ID, [15] XCC_ID.
template <int F, bool TIMING = false>
__global__ __launch_bounds__(256, 2) void gemm_f16x3s_kernel(const F16SArgs a) {
    const F16PArgs& p = a.g;
    unsigned tb[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tk0 = 0, tq = 0;
    if (TIMING) tk0 = gemm_now();
#define SK_STAMP(slot) do { if (TIMING) { const unsigned long long _t = gemm_now(); tb[slot] += (unsigned)(_t - tq); tq = _t; } } while (0)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;
    const int q = (blockIdx.x & 7) * a.per_xcd + (blockIdx.x >> 3);      // this workgroup in the XCD-major order
    const int grid = 8 * a.per_xcd;
    const int u_begin = a.whole ? q * a.nslabs : sk_bound(a, q);
    const int u_end = a.whole ? (q * a.nslabs < a.units ? a.units : u_begin) : sk_bound(a, q + 1);      // (whole: the end of the last tile; pieces end at tile boundaries)

    // fragment offsets inside a stage: row r32 of a 32-row tile, chunk 4 s + 2 h + pl (s = k16 step, pl = hi / lo plane), swizzled as the DMA stores them
    int fa_off[2][2], fw_off[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const int f = r32 * SK_ROWB + (((4 * s + 2 * h + pl) ^ ((r32 >> 1) & 7)) << 4);
            fa_off[s][pl] = wm * 64 * SK_ROWB + f;
            fw_off[s][pl] = SK_A_BYTES + wn * 64 * SK_ROWB + f;
        }

    // ---- a piece = (tile, slabs [s0, s1)); its DMA state: buffer descriptors at the tile's first row / column and slab s0, per-lane source offsets
    struct Piece { int tile, s0, s1, m0, n0; };
    auto piece_at = [&](int u) {
        Piece pc;
        pc.tile = (int)__umulhi((unsigned)u, a.inv_nslabs);
        pc.s0 = u - pc.tile * a.nslabs;
        const int left = u_end - u;
        pc.s1 = (!a.whole && pc.s0 + left < a.nslabs) ? pc.s0 + left : a.nslabs;
        const int pfull = p.tiles_m * p.panel, pn = pc.tile / pfull, prem = pc.tile - pn * pfull;
        const int pw = p.tiles_n - pn * p.panel < p.panel ? p.tiles_n - pn * p.panel : p.panel;
        pc.m0 = (prem / pw) * SK_BM;
        pc.n0 = (pn * p.panel + prem % pw) * SK_BN;
        return pc;
    };
    // per-lane source offsets relative to the tile's first row / column (the same for every tile: whole tiles only, the host checks M, N % 128 == 0);
    // the buffer descriptors carry the tile and the piece's first slab
    int voff[SK_NL];
#pragma unroll
    for (int i = 0; i < SK_NL; ++i) {
        const int b = wave + i * SK_NW;
        const bool isw = b >= SK_A_BLK;
        const int row = (isw ? b - SK_A_BLK : b) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);
        voff[i] = (int)((int64_t)row * (isw ? p.ldw : p.lda) * 4) + chunk * 16;
    }
    __amdgpu_buffer_rsrc_t rsA, rsW;
    auto dma_setup = [&](const Piece& pc) {
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)pc.m0 * p.lda * 4 + (int64_t)pc.s0 * SK_ROWB), 0, 0x7fffffff, 0x00020000);
        rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (int64_t)pc.n0 * p.ldw * 4 + (int64_t)pc.s0 * SK_ROWB), 0, 0x7fffffff, 0x00020000);
    };
    auto issue_one = [&](int i, int slab, int stage) {      // slab counted from the piece's first
        const int b = wave + i * SK_NW;
        unsigned char* dst = smem + stage * SK_STAGE + b * 1024;
        if (b >= SK_A_BLK) P_DMA16(rsW, dst, voff[i], slab * SK_ROWB);
        else P_DMA16(rsA, dst, voff[i], slab * SK_ROWB);
    };
    auto issue_two = [&]() {      // the first two slabs of a piece (every piece has >= SK_MINS slabs)
#pragma unroll
        for (int i = 0; i < SK_NL; ++i) issue_one(i, 0, 0);
#pragma unroll
        for (int i = 0; i < SK_NL; ++i) issue_one(i, 1, 1);
    };

    pf32x16 acc[2][2];
    pf16x8 f0a[2][2], f0w[2][2], f1a[2][2], f1w[2][2];
    auto read_frag = [&](int n, int s, int stage, pf16x8 (&fa)[2][2], pf16x8 (&fw)[2][2]) {      // n in [0, 8): A tile 0 hi, lo, A tile 1 hi, lo, W ...
        const unsigned char* base = smem + stage * SK_STAGE;
        if (n < 4) { const int i = n >> 1, pl = n & 1; fa[i][pl] = *reinterpret_cast<const pf16x8*>(base + fa_off[s][pl] + i * 32 * SK_ROWB); }
        else { const int m = n - 4, j = m >> 1, pl = m & 1; fw[j][pl] = *reinterpret_cast<const pf16x8*>(base + fw_off[s][pl] + j * 32 * SK_ROWB); }
    };
    auto mfma = [&](int m, const pf16x8 (&fa)[2][2], const pf16x8 (&fw)[2][2]) {      // m in [0, 12): term-major (hi*lo, lo*hi, hi*hi), as gemm_f16x3p.hip
        constexpr int PA[3] = {0, 1, 0}, PW[3] = {1, 0, 0};
        const int term = m / 4, ij = m % 4, i = ij / 2, j = ij % 2;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fw[j][PW[term]], fa[i][PA[term]], acc[i][j], 0, 0, 0);      // operands swapped: one output row per lane
    };
    int st = 0;
    // One slab (gemm_f16x3p.hip, S = 2, PF = 2).  FIRST: the piece's first slab -- its DMA was issued before the previous piece's epilogue, whose loads and
    // stores share the VM counter and may return out of order with the DMA: a full drain instead of the counted wait.
    auto body = [&](int t, auto waitn_c, auto issue_c) {
        constexpr int WAITN = decltype(waitn_c)::value;
        constexpr bool DO_ISSUE = decltype(issue_c)::value;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int n = 0; n < 8; ++n) read_frag(n, 0, st, f0a, f0w);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            mfma(m, f0a, f0w);
            if (m < 8) read_frag(m, 1, st, f1a, f1w);
            else if (DO_ISSUE) issue_one(m - 8, t + 2, st);
            if (m == 7) {      // every step-1 fragment read is issued: wait for them, release the stage
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("s_barrier" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            mfma(m, f1a, f1w);
            if (m < 4 && DO_ISSUE) issue_one(4 + m, t + 2, st);
            __builtin_amdgcn_sched_barrier(0);
        }
        st ^= 1;
    };
    using std::integral_constant;

    int* flag = reinterpret_cast<int*>(smem + 2 * SK_STAGE);
    if (u_begin >= u_end) return;
    Piece cur = piece_at(u_begin);
    dma_setup(cur);
    issue_two();
    if (TIMING) { tq = gemm_now(); tb[6] = (unsigned)(tq - tk0); }
    for (;;) {
        const int n = cur.s1 - cur.s0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        st = 0;
        // ---- K loop over the piece's slabs
        body(0, integral_constant<int, 0>{}, integral_constant<bool, true>{});      // n >= 4: slab 2 exists
        int t = 1;
        for (; t + 2 < n; ++t) body(t, integral_constant<int, SK_NL>{}, integral_constant<bool, true>{});
        body(t, integral_constant<int, SK_NL>{}, integral_constant<bool, false>{});
        body(t + 1, integral_constant<int, 0>{}, integral_constant<bool, false>{});
        // ---- every wave is done with the ring: the next piece's first two slabs go out before this piece's epilogue
        const int u_next = a.whole ? (cur.tile + grid) * a.nslabs : cur.tile * a.nslabs + cur.s1;
        const bool more = u_next < u_end;
        if (TIMING) { tb[7] += 1; tb[8] += (unsigned)(cur.s1 - cur.s0); }
        SK_STAMP(0);
        const Piece done = cur;
        // (no barrier: after the last slab's mid-slab barrier no wave reads the ring any more -- its step-1 fragments are in registers)
        if (more) {
            cur = piece_at(u_next);
            dma_setup(cur);
            issue_two();
        }
        SK_STAMP(1);
        // ---- whole tile: the epilogue; part of a tile: park, count in, and the last arrival combines
        bool finish = true;
        if (done.s0 != 0 || done.s1 != a.nslabs) {
            const int T0 = done.tile * a.nslabs, T1 = T0 + a.nslabs;
            int first = q, last = q;
            while (sk_bound(a, first) > T0) --first;
            while (sk_bound(a, last + 1) < T1) ++last;
            const int nparts = last - first + 1;
            constexpr int SC1 = 16;
            const int my_slot = 2 * q + (T0 + done.s0 == u_begin ? 0 : 1);
            const int lane_off = (wave * 64 + lane) * 16;
            {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.part + (int64_t)my_slot * SK_PART_FLOATS), 0, SK_PART_FLOATS * 4, 0x00020000);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const float a0 = acc[i][j][4 * r4], a1 = acc[i][j][4 * r4 + 1], a2 = acc[i][j][4 * r4 + 2], a3 = acc[i][j][4 * r4 + 3];
                            const ep_f32x4 vf = {a0, a1, a2, a3};
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu32x4, vf), rs, ((i * 2 + j) * 4 + r4) * (SK_NW * 64 * 16) + lane_off, 0, SC1);
                        }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's part is acknowledged by the memory side
            SK_STAMP(2);
            __syncthreads();
            if (tid == 0) *flag = (int)__hip_atomic_fetch_add(a.count + done.tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int arrived = *flag;
            __syncthreads();      // (the flag word is rewritten by the next partial piece)
            finish = arrived == nparts - 1;
            SK_STAMP(3);
            if (finish) {
                if (tid == 0) __hip_atomic_store(a.count + done.tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                for (int k = 0; k < nparts; ++k) {      // K order: part k belongs to workgroup first + k
                    const int qq = first + k;
                    const int b0 = sk_bound(a, qq);
                    const int slot = 2 * qq + (b0 >= T0 ? 0 : 1);      // its piece of this tile starts its own range (b0 >= T0) or at the tile boundary
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.part + (int64_t)slot * SK_PART_FLOATS), 0, SK_PART_FLOATS * 4, 0x00020000);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int r4 = 0; r4 < 4; ++r4) {
                                const ep_f32x4 v = __builtin_bit_cast(ep_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((i * 2 + j) * 4 + r4) * (SK_NW * 64 * 16) + lane_off, 0, SC1));
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[i][j][4 * r4 + e] += v[e];
                            }
                }
                SK_STAMP(4);
            }
        }
        if (finish) {
            gemm_store_tile_t_impl<2, 2, F, F16PArgs, false>(p, acc, done.m0 + wm * 64, done.n0 + wn * 64, lane, p.C, p.residual, true);
            if (TIMING) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tb[9] += 1; }
            SK_STAMP(5);
        }
        if (!more) break;
    }
    if (TIMING && p.dbg) {
        const unsigned long long te = gemm_now();
        if (lane == 0) {
            unsigned* d = p.dbg + ((size_t)blockIdx.x * SK_NW + wave) * 16;
#pragma unroll
            for (int i = 0; i < 10; ++i) d[i] = tb[i];
            d[10] = (unsigned)tk0; d[11] = (unsigned)(tk0 >> 32); d[12] = (unsigned)te; d[13] = (unsigned)(te >> 32);
            d[14] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
            d[15] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        }
    }
#undef SK_STAMP
}

// ---------------------------------------------------------------------------------------------- host
namespace {
struct SkBlock { float* part; int* count; int grid; int tiles; };
constexpr int SK_MAX_TILES_S = 16384;

// per (device, stream): the parked-tile workspace (2 x grid x 64 KiB) and the arrival counters; allocated on first use (not inside a graph capture)
static SkBlock sk_block(hipStream_t stream, int grid) {
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, SkBlock> table;
    int dev = 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipGetDevice(&dev) != hipSuccess || hipStreamIsCapturing(stream, &cs) != hipSuccess) return {nullptr, nullptr, 0, 0};
    std::lock_guard<std::mutex> lock(mu);
    auto it = table.find({dev, stream});
    if (it != table.end() && it->second.grid >= grid) return it->second;
    if (cs != hipStreamCaptureStatusNone) return {nullptr, nullptr, 0, 0};
    if (it != table.end()) return {nullptr, nullptr, 0, 0};      // (a block sized for a smaller grid: the caller keeps the ordinary kernel)
    void* part = nullptr; void* count = nullptr;
    if (hipMalloc(&part, (size_t)2 * grid * SK_PART_FLOATS * 4) != hipSuccess) return {nullptr, nullptr, 0, 0};
    if (hipMalloc(&count, SK_MAX_TILES_S * sizeof(int)) != hipSuccess || hipMemsetAsync(count, 0, SK_MAX_TILES_S * sizeof(int), stream) != hipSuccess) {
        (void)hipFree(part); if (count) (void)hipFree(count);
        return {nullptr, nullptr, 0, 0};
    }
    const SkBlock b = {static_cast<float*>(part), static_cast<int*>(count), grid, SK_MAX_TILES_S};
    table[{dev, stream}] = b;
    return b;
}

template <int F, bool TIMING = false>
static int32_t launch_s(const F16SArgs& a, int grid, hipStream_t stream) {
#ifdef PSAM_GEMM_ABLATE
    if (!TIMING && a.g.dbg) return launch_s<F, true>(a, grid, stream);      // a timing buffer is set (psam_gemm_f16x3p_set_timing_buffer): the stamped instance
#endif
    static unsigned long long attr_done = 0;
    if (!f16x3p_reserve_lds(&gemm_f16x3s_kernel<F, TIMING>, SK_LDS, attr_done)) {
        psam_set_error("psam_gemm_f16x3p: cannot reserve LDS");
        return PSAM_EINVAL;
    }
    hipLaunchKernelGGL((gemm_f16x3s_kernel<F, TIMING>), dim3((unsigned)grid), dim3(256), SK_LDS, stream, a);
    return psam_launch_status("psam_gemm_f16x3p: launch failed");
}
}  // namespace

// Option set of a launch as the EP_* bits of the register epilogue, or -1 when the stream-K kernel does not serve it.
static int f16x3s_option_set(const F16PArgs& p) {
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
// mode: 0 = even shares of the K slabs (stream-K) on 2 x (#CU - reserve) workgroups, 1 = whole tiles dealt round-robin to the persistent workgroups,
// 2 = even shares on one workgroup per CU, 3 = even shares on 2 x #CU workgroups -- 1 .. 3 are measurement variants (psam_gemm_f16x3p_force_config 91 .. 93).
// reserve (PSAM_GEMM_RESERVE_CUS, default 8): the tokenizer of the next batch (FPS: one 1024-thread workgroup per cloud, a whole CU each for ~1.3 ms of a
// 10 ms step) runs beside the dense stage; a persistent grid that counts on every CU would leave 16 of its workgroups waiting for a second round then.
bool launch_f16x3s(F16PArgs& p, hipStream_t stream, int32_t& rc, int mode) {
    const int opt = f16x3s_option_set(p);
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
    F16SArgs a;
    p.tiles_m = p.M / SK_BM; p.tiles_n = p.N / SK_BN;
    p.panel = f16x3p_panel(p.tiles_m, p.tiles_n, SK_BM, SK_BN, p.K);
    const int64_t tiles = (int64_t)p.tiles_m * p.tiles_n;
    a.nslabs = p.K / 32;
    if (tiles > SK_MAX_TILES_S || tiles * a.nslabs >= ((int64_t)1 << 30)) return false;
    a.units = (int)(tiles * a.nslabs);
    static int reserve = -1;
    if (reserve < 0) { const char* e = getenv("PSAM_GEMM_RESERVE_CUS"); reserve = e ? atoi(e) : 8; }
    const int cus = (mode == 0 && ncu > 2 * reserve) ? ncu - reserve : ncu;
    int grid = mode == 2 ? cus : 2 * cus;
    if ((int64_t)grid * 16 > a.units) grid = a.units / 16;      // at least 16 slabs per workgroup
    if (mode == 1 && grid > tiles) grid = (int)tiles;
    grid &= ~7;
    // ranges of the multiply-high divisions in sk_bound / piece_at: u * nslabs, (j * span) * per_xcd < 2^32
    if (grid < 8 || (int64_t)a.units * a.nslabs >= ((int64_t)1 << 32) || (int64_t)(grid / 8) * (grid / 8) * (a.units / 8 + 1) >= ((int64_t)1 << 32)) return false;
    a.inv_nslabs = (unsigned)((((uint64_t)1 << 32) + a.nslabs - 1) / a.nslabs);
    a.inv_per_xcd = (unsigned)((((uint64_t)1 << 32) + grid / 8 - 1) / (grid / 8));
    a.whole = mode == 1 ? 1 : 0;
    const SkBlock blk = sk_block(stream, 2 * ncu);
    if (!blk.part) return false;
    a.g = p; a.part = blk.part; a.count = blk.count; a.per_xcd = grid / 8;
    switch (opt) {
        case 0: rc = launch_s<0>(a, grid, stream); break;
        case EP_RES: rc = launch_s<EP_RES>(a, grid, stream); break;
        case EP_PACK: rc = launch_s<EP_PACK>(a, grid, stream); break;
        case EP_LNC | EP_RES: rc = launch_s<EP_LNC | EP_RES>(a, grid, stream); break;
        case EP_GELU: rc = launch_s<EP_GELU>(a, grid, stream); break;
        case EP_SWIGLU | EP_STATS | EP_PACK | EP_BND: rc = launch_s<EP_SWIGLU | EP_STATS | EP_PACK | EP_BND>(a, grid, stream); break;
        case EP_SWIGLU | EP_STATS | EP_PACK: rc = launch_s<EP_SWIGLU | EP_STATS | EP_PACK>(a, grid, stream); break;
        case EP_SWIGLU: rc = launch_s<EP_SWIGLU>(a, grid, stream); break;
        default: return false;
    }
    return true;
}

// After a failed launch: re-zero the stream's arrival counters (see psam_gemm_f16x3p_reset_splitk_state, which calls this too).
void f16x3s_reset_state(hipStream_t stream) {
    int ncu = 256, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
    const SkBlock blk = sk_block(stream, 2 * ncu);
    if (blk.count) (void)hipMemsetAsync(blk.count, 0, SK_MAX_TILES_S * sizeof(int), stream);
}
