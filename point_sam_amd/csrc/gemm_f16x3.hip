// fp32-GRADE GEMM on the fp16 matrix pipe ("f16x3"): same contract and epilogues as gemm.hip (2-D form),
//
//   C = act(alpha * A @ W^T + bias + rowbias[row / rowgroup]) + residual,      A, W, C fp32 in HBM.
//
// Every operand row is first scaled by a power of two s (exact) that puts its largest magnitude in [2^14, 2^15) --
// psam_row_scale_f16 computes s per row; for a static weight once at load time -- then every element is split into two
// fp16 numbers, s*x = hi + lo + e with hi = RNE_f16(s*x), lo = RNE_f16(s*x - hi) (the subtraction is exact, one FMA):
// 11 + 11 significand bits plus the sign of lo, |e| <= 2^-22 |s*x| for elements within 2^-18 of the row maximum and
// <= 2^-25 absolute (2^-39 of the row maximum) below that, where lo goes subnormal.  A product a*w is computed as
// hi*hi + hi*lo + lo*hi (each fp16 x fp16 product is exact in fp32; v_mfma_f32_32x32x16_f16 accumulates in fp32); the
// dropped lo*lo term is <= 2^-22 |a*w|.  The accumulator is multiplied by 1/(sA[row] sW[col]) (exact) in the epilogue.
// Net: per-product relative error <= ~3*2^-22 with random sign, against 2^-24 for an exact fp32 product -- in a K-term dot
// product both are buried under the fp32 accumulation round-off, and the measured error against an fp64 reference equals
// that of the f32-MFMA kernel (tests/test_gpu_kernels.py, scripts/gemm_split_bench.py) at 3/16 of its matrix-pipe time
// and half that of the bf16x6 scheme (gemm_split.hip, which needs no row scales and is exact to 2^-24 for any range).
//
// Kernel structure = gemm_split.hip with two planes instead of three: tile 128x128 (2x2 waves of 64x64) or 128x64 (4 waves
// of 32x64), slabs of 32 k, split in registers while a slab moves global -> LDS, LDS rows of 64 bytes with the 16-byte
// chunk XOR-swizzled by (row>>2)&3, epilogue through the LDS transpose of gemm_epilogue.h.
#include "common.h"
#include "gemm_epilogue.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct F16x3Args {
    const float* A; const float* W; float* C;
    const float* bias; const float* residual; const float* rowbias;
    const float* scaleA; const float* scaleW;       // per-row powers of two (psam_row_scale_f16)
    int64_t lda, ldw, ldc, ldr, ldrb;
    int M, N, K, rowgroup, act;
    float alpha;
    int tiles_m, tiles_n;
    // split-K over two workgroups for the tiles >= split_from (pipelined kernel): partner partials through `ws`, see the kernel
    int split_from; unsigned epoch; float* ws; unsigned* flags;
};

constexpr int HG_BK = 32;
constexpr int HG_ROWB = HG_BK * 2;                       // bytes per row of one fp16 plane

// x (2 floats, already scaled) -> packed fp16 pairs hi, lo
__device__ __forceinline__ void split2(const f32x2 x, const float s, unsigned& hi, unsigned& lo) {
    const f32x2 xs = x * s;
    const f16x2 h = __builtin_convertvector(xs, f16x2);
    const f32x2 r = xs - __builtin_convertvector(h, f32x2);
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, l);
}

// ---------------------------------------------------------------------------------------------- row scales
// scale[r] = 2^(14 - e), e = floor(log2(max_k |X[r,k]|)) clamped below at -112; 1 for an all-zero / non-finite row.
__global__ __launch_bounds__(256) void row_scale_f16_kernel(const float* __restrict__ X, int64_t ldx, int rows, int cols, float* __restrict__ scale) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* x = X + (int64_t)row * ldx;
    float m = 0.f;
    if (((ldx & 3) == 0) && (((uintptr_t)X & 15) == 0)) {
        const int c4 = cols >> 2;
        for (int c = lane; c < c4; c += 64) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + c * 4);
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

// ---------------------------------------------------------------------------------------------- packed operands
// "f16x2-packed" rows: the container is still one 32-bit word per element ([R, K] with leading dimension ld), but every group of
// four consecutive k holds [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3] (fp16) of the row-scaled values -- exactly the two 8-byte LDS
// slots the GEMM stages, so a packed operand moves global -> VGPR -> LDS with no arithmetic.  Static weights are packed once;
// in place (P == X) is allowed.
__global__ __launch_bounds__(256) void pack_rows_f16x2_kernel(const float* __restrict__ X, int64_t ldx, const float* __restrict__ scale, int rows,
                                                              int k4, unsigned* __restrict__ P, int64_t ldp) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)rows * k4) return;
    const int r = (int)(t / k4), g = (int)(t % k4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(X + (int64_t)r * ldx + g * 4);
    const float s = scale[r];
    unsigned h0, l0, h1, l1;
    split2(f32x2{v[0], v[1]}, s, h0, l0);
    split2(f32x2{v[2], v[3]}, s, h1, l1);
    *reinterpret_cast<u32x4*>(P + (int64_t)r * ldp + g * 4) = u32x4{h0, h1, l0, l1};
}

PSAM_API int32_t psam_pack_rows_f16x2(const float* X, int64_t ldx, const float* scale, int32_t rows, int32_t K, void* P, int64_t ldp,
                                      hipStream_t stream) {
    PSAM_REQUIRE(X && scale && P, PSAM_EINVAL, "psam_pack_rows_f16x2: null pointer");
    PSAM_REQUIRE(rows > 0 && K > 0 && (K & 3) == 0 && ldx >= K && ldp >= K, PSAM_EINVAL, "psam_pack_rows_f16x2: bad shape (K % 4 == 0)");
    PSAM_REQUIRE(((ldx | ldp) & 3) == 0 && (((uintptr_t)X | (uintptr_t)P) & 15) == 0, PSAM_EALIGN, "psam_pack_rows_f16x2: 16-byte alignment");
    const int64_t total = (int64_t)rows * (K / 4);
    hipLaunchKernelGGL(pack_rows_f16x2_kernel, dim3((unsigned)psam_cdiv(total, 256)), dim3(256), 0, stream, X, ldx, scale, rows, K / 4, (unsigned*)P, ldp);
    return psam_launch_status("psam_pack_rows_f16x2: launch failed");
}

// ---------------------------------------------------------------------------------------------- GEMM
// WM x WN waves (4 in total), each TM x TN accumulator tiles of 32x32.  FDB: double-buffer the fragment registers
// across the two k16 steps of a slab.
template <int WM, int WN, int TM, int TN, bool FDB>
__global__ __launch_bounds__(256) void gemm_f16x3_kernel(const F16x3Args p) {
    static_assert(WM * WN == 4, "256-thread workgroup");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_F4 = BM * 8 / 256, W_F4 = BN * 8 / 256;   // float4 per thread per slab
    constexpr int PLANE_A = BM * HG_ROWB, PLANE_W = BN * HG_ROWB;
    constexpr int OPER_BYTES = 2 * PLANE_A + 2 * PLANE_W, EPI_BYTES = 4 * gemm_epilogue_lds_floats_per_wave<TN>() * 4;
    // LDS: [A planes hi,lo][W planes hi,lo]; the epilogue staging reuses it
    __shared__ __attribute__((aligned(16))) unsigned char smem[OPER_BYTES > EPI_BYTES ? OPER_BYTES : EPI_BYTES];
    unsigned char* sA = smem;
    unsigned char* sW = smem + 2 * PLANE_A;

    const int ntiles = p.tiles_m * p.tiles_n;
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const float* A = p.A;
    const float* W = p.W;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r32 = lane & 31, h = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // global -> register staging: thread (lr, lc4) owns 4 consecutive k of rows lr + 32 i of both operands
    const int lr = tid >> 3, lc4 = tid & 7;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((((int64_t)p.M - 1) * p.lda + p.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)((((int64_t)p.N - 1) * p.ldw + p.K) * 4), 0x00020000);
    const int voA = (int)(((int64_t)(m0 + lr) * p.lda + lc4 * 4) * 4), voW = (int)(((int64_t)(n0 + lr) * p.ldw + lc4 * 4) * 4);
    const int stepA = (int)(32 * p.lda * 4), stepW = (int)(32 * p.ldw * 4);
    constexpr int OOB = 0x7ffffff0;
    float sca[A_F4], scw[W_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) sca[i] = m0 + i * 32 + lr < p.M ? p.scaleA[m0 + i * 32 + lr] : 1.f;
#pragma unroll
    for (int i = 0; i < W_F4; ++i) scw[i] = n0 + i * 32 + lr < p.N ? p.scaleW[n0 + i * 32 + lr] : 1.f;
    f32x4 ra0[A_F4], rw0[W_F4];
    auto load_slab = [&](int k0) {
        const bool kok = k0 + lc4 * 4 < p.K;
#pragma unroll
        for (int i = 0; i < A_F4; ++i)
            ra0[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (kok && m0 + i * 32 + lr < p.M) ? voA + i * stepA : OOB, k0 * 4, 0));
#pragma unroll
        for (int i = 0; i < W_F4; ++i)
            rw0[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (kok && n0 + i * 32 + lr < p.N) ? voW + i * stepW : OOB, k0 * 4, 0));
    };
    // LDS byte offset of this thread's 8-byte slot inside a plane: row lr (+32 i), 16-byte chunk lc4>>1 swizzled, half lc4&1
    const int st_off = lr * HG_ROWB + ((((lc4 >> 1) ^ ((lr >> 2) & 3)) << 4) | ((lc4 & 1) << 3));
    // split registers: [row stripe][plane hi/lo] -> 2 packed words (4 fp16)
    u32x2 spa[A_F4][2], spw[W_F4][2];
    auto split_regs = [&]() {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            unsigned h0, l0, h1, l1;
            split2(f32x2{ra0[i][0], ra0[i][1]}, sca[i], h0, l0);
            split2(f32x2{ra0[i][2], ra0[i][3]}, sca[i], h1, l1);
            spa[i][0] = u32x2{h0, h1}; spa[i][1] = u32x2{l0, l1};
        }
#pragma unroll
        for (int i = 0; i < W_F4; ++i) {
            unsigned h0, l0, h1, l1;
            split2(f32x2{rw0[i][0], rw0[i][1]}, scw[i], h0, l0);
            split2(f32x2{rw0[i][2], rw0[i][3]}, scw[i], h1, l1);
            spw[i][0] = u32x2{h0, h1}; spw[i][1] = u32x2{l0, l1};
        }
    };
    auto store_split = [&]() {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) *reinterpret_cast<u32x2*>(sA + q * PLANE_A + i * 32 * HG_ROWB + st_off) = spa[i][q];
#pragma unroll
            for (int i = 0; i < W_F4; ++i) *reinterpret_cast<u32x2*>(sW + q * PLANE_W + i * 32 * HG_ROWB + st_off) = spw[i][q];
        }
    };
    // fragment addresses: row r32 of 32-row tile, chunk (2s + h) swizzled
    int frag_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) frag_off[s] = r32 * HG_ROWB + (((2 * s + h) ^ ((r32 >> 2) & 3)) << 4);
    const unsigned char* a_base = sA + wm * TM * 32 * HG_ROWB;
    const unsigned char* w_base = sW + wn * TN * 32 * HG_ROWB;
    auto load_frags = [&](int s, f16x8 (&af)[TM][2], f16x8 (&wf)[TN][2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i][q] = *reinterpret_cast<const f16x8*>(a_base + q * PLANE_A + i * 32 * HG_ROWB + frag_off[s]);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j][q] = *reinterpret_cast<const f16x8*>(w_base + q * PLANE_W + j * 32 * HG_ROWB + frag_off[s]);
        }
    };
    // the two small partial products first (hi*lo, lo*hi), then hi*hi; term-major so that the accumulator tiles rotate
#define HG_TERM(AF, WF, PA, PW)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)               \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AF[i][PA], WF[j][PW], acc[i][j], 0, 0, 0);
#define HG_STEP(AF, WF) HG_TERM(AF, WF, 0, 1) HG_TERM(AF, WF, 1, 0) HG_TERM(AF, WF, 0, 0)

    const int nslabs = (p.K + HG_BK - 1) / HG_BK;
    // Pipeline (as gemm_split.hip).  At the top of slab t: sp = split slab t (registers); (ra0, rw0) = fp32 slab t+1 (landed
    // or landing).  Slab t: [barrier: LDS free] sp -> LDS [barrier: LDS ready]; fragments of k16-step 0; then 2 x 3*TM*TN
    // MFMAs, in whose shadow issue: the fragment reads of step 1, the split of slab t+1 into sp and the refill of the fp32
    // registers with slab t+2.  Loads past K return zeros (bounds-checked descriptor) and are unconditional: counted vmcnt.
    f16x8 af0[TM][2], wf0[TN][2], af1[TM][2], wf1[TN][2];
    load_slab(0);
    split_regs();
    load_slab(HG_BK);
    constexpr int NM = 3 * TM * TN, NFR = 2 * (TM + TN);
    constexpr int NVALU = (A_F4 + W_F4) * 12 / (2 * NM) + 2;     // split VALU per MFMA slot (6 per float2, 2 float2 per float4)
    for (int t = 0; t < nslabs; ++t) {
        __syncthreads();
        store_split();
        __syncthreads();
        load_frags(0, af0, wf0);
        if (FDB) {
            load_frags(1, af1, wf1);
            HG_STEP(af0, wf0)
            split_regs();                 // slab t+1
            HG_STEP(af1, wf1)
        } else {
            HG_STEP(af0, wf0)
            load_frags(1, af0, wf0);
            split_regs();                 // slab t+1
            HG_STEP(af0, wf0)
        }
        load_slab((t + 2) * HG_BK);
        // requested issue order (one scheduling region: the loop body after the second barrier)
        __builtin_amdgcn_sched_group_barrier(0x100, NFR, 0);      // fragments of step 0
        if (FDB) {
            static_assert(2 * NM >= NFR, "fragment reads hide behind the MFMAs");
#pragma unroll
            for (int i = 0; i < NFR; ++i) {                        // first MFMAs, fragment reads of step 1 behind them
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, NVALU, 0);
            }
#pragma unroll
            for (int i = 0; i < 2 * NM - NFR; ++i) {              // remaining MFMAs with the split arithmetic behind them
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, NVALU, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, NVALU, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, NFR, 0);
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, NVALU, 0);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x2, 64, 0);
        __builtin_amdgcn_sched_group_barrier(0x20, A_F4 + W_F4, 0);
    }
#undef HG_STEP
#undef HG_TERM

    // ---- epilogue: un-scale, then LDS transpose -> row-contiguous float4 stores (gemm_epilogue.h)
    __syncthreads();   // every wave is done reading operand fragments
    gemm_store_tile<TM, TN, true>(p, acc, reinterpret_cast<float*>(smem) + wave * gemm_epilogue_lds_floats_per_wave<TN>(), m0 + wm * TM * 32,
                                  n0 + wn * TN * 32, lane, p.C, p.residual);
}

// ---------------------------------------------------------------------------------------------- pipelined 128x128 kernel
// Same arithmetic, software-pipelined across k slabs (measured on the kernel above: a lone workgroup spends 2100 cycles per
// slab for 768 cycles of MFMA -- barrier, LDS write, barrier and fragment latency are all exposed between the MFMA bursts).
// LDS is double buffered (2 x 32 KiB), ONE barrier per slab, and every data movement of slab t+1 / t+3 is issued in the
// shadow of the 24 MFMAs of slab t:
//   top of slab t:  LDS[t&1] = split slab t;  F0 = fragments (t, k16 step 0);  R[(t+1)&1] = fp32 slab t+1;  R[t&1] <- slab t+2 in flight
//   region A:  F1 <- LDS[t&1] (step 1)      | 12 MFMAs of step 0 (F0) + 4 of step 1 (F1)
//              split R[(t+1)&1] -> LDS[(t+1)&1];  R[(t+1)&1] <- global slab t+3 (two slab periods to land)
//   barrier    (LDS[(t+1)&1] complete; every wave has its F1, so LDS[t&1] may be overwritten next slab)
//   region B:  F0 <- LDS[(t+1)&1] (step 0 of slab t+1)     | remaining 8 MFMAs of step 1
// APK / WPK: operand already f16x2-packed (psam_pack_rows_f16x2): staged without arithmetic.  NSETS: operand register sets =
// prefetch distance in slabs (2 in production: ~215-230 registers, two workgroups per CU; 3 and 4 exist for measurement).
template <bool APK, bool WPK, int NSETS>
__global__ __launch_bounds__(256) void gemm_f16x3_pipe_kernel(const F16x3Args p) {
    constexpr int TM = 2, TN = 2, WN = 2;
    constexpr int BM = 128, BN = 128;
    constexpr int NF4 = 4;                                     // float4 per thread per slab per operand
    constexpr int PLANE = 128 * HG_ROWB;                       // 8 KiB
    constexpr int STAGE = 4 * PLANE;                           // A hi, A lo, W hi, W lo
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    // Workgroup -> (tile, K half).  Tiles below split_from are computed whole (half = -1).  A tile >= split_from is computed by TWO
    // workgroups, each over half of the k slabs: with fewer than 2 tiles per CU that doubles the resident workgroups (latency
    // hiding) and with a last round at most half full it halves that round.  The partner (half 1) parks its accumulators in `ws`
    // and raises the tile's flag to this launch's epoch; the owner (half 0) adds them to its own in a FIXED order (own + partner:
    // bit-reproducible, unlike atomics on C) and runs the epilogue.  Partners are 8 workgroup ids apart: the same XCD under the
    // round-robin dispatch, so the partial normally never leaves that XCD's L2; the fences are agent-scope regardless.
    const int ntiles = p.tiles_m * p.tiles_n;
    int tile = blockIdx.x, half = -1;
    if (tile >= p.split_from) {
        const int sidx = tile - p.split_from;                  // 16 q + 8 half + x  ->  split tile 8 q + x
        half = (sidx >> 3) & 1;
        tile = p.split_from + ((sidx >> 4) << 3) + (sidx & 7);
    }
    const int split_idx = tile - p.split_from;
    if ((ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r32 = lane & 31, h = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int lr = tid >> 3, lc4 = tid & 7;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((((int64_t)p.M - 1) * p.lda + p.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)((((int64_t)p.N - 1) * p.ldw + p.K) * 4), 0x00020000);
    const int voA = (int)(((int64_t)(m0 + lr) * p.lda + lc4 * 4) * 4), voW = (int)(((int64_t)(n0 + lr) * p.ldw + lc4 * 4) * 4);
    const int stepA = (int)(32 * p.lda * 4), stepW = (int)(32 * p.ldw * 4);
    constexpr int OOB = 0x7ffffff0;
    float sca[NF4], scw[NF4];
    int offA[NF4], offW[NF4];                                  // row-bound check folded into the offset once
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
        const bool oka = m0 + i * 32 + lr < p.M, okw = n0 + i * 32 + lr < p.N;
        sca[i] = oka ? p.scaleA[m0 + i * 32 + lr] : 1.f;
        scw[i] = okw ? p.scaleW[n0 + i * 32 + lr] : 1.f;
        offA[i] = oka ? voA + i * stepA : OOB;
        offW[i] = okw ? voW + i * stepW : OOB;
    }
    f32x4 ra[NSETS][NF4], rw[NSETS][NF4];
    auto load_slab = [&](int k0, f32x4 (&a)[NF4], f32x4 (&w)[NF4]) {
        const bool kok = k0 + lc4 * 4 < p.K;
#pragma unroll
        for (int i = 0; i < NF4; ++i) a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, kok ? offA[i] : OOB, k0 * 4, 0));
#pragma unroll
        for (int i = 0; i < NF4; ++i) w[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, kok ? offW[i] : OOB, k0 * 4, 0));
    };
    const int st_off = lr * HG_ROWB + ((((lc4 >> 1) ^ ((lr >> 2) & 3)) << 4) | ((lc4 & 1) << 3));
    auto stage4 = [&](unsigned char* st, const f32x4 v, const float sc, const bool packed) {   // one float4 -> its hi and lo 8-byte slots
        if (packed) {
            const u32x4 raw = __builtin_bit_cast(u32x4, v);
            *reinterpret_cast<u32x2*>(st) = u32x2{raw[0], raw[1]};
            *reinterpret_cast<u32x2*>(st + PLANE) = u32x2{raw[2], raw[3]};
        } else {
            unsigned h0, l0, h1, l1;
            split2(f32x2{v[0], v[1]}, sc, h0, l0);
            split2(f32x2{v[2], v[3]}, sc, h1, l1);
            *reinterpret_cast<u32x2*>(st) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(st + PLANE) = u32x2{l0, l1};
        }
    };
    auto split_store = [&](int buf, const f32x4 (&a)[NF4], const f32x4 (&w)[NF4]) {
        unsigned char* st = smem + buf * STAGE + st_off;
#pragma unroll
        for (int i = 0; i < NF4; ++i) stage4(st + i * 32 * HG_ROWB, a[i], sca[i], APK);
#pragma unroll
        for (int i = 0; i < NF4; ++i) stage4(st + 2 * PLANE + i * 32 * HG_ROWB, w[i], scw[i], WPK);
    };
    int frag_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) frag_off[s] = r32 * HG_ROWB + (((2 * s + h) ^ ((r32 >> 2) & 3)) << 4);
    const int a_off = wm * TM * 32 * HG_ROWB, w_off = 2 * PLANE + wn * TN * 32 * HG_ROWB;
    auto load_frags = [&](int buf, int s, f16x8 (&af)[TM][2], f16x8 (&wf)[TN][2]) {
        const unsigned char* b = smem + buf * STAGE + frag_off[s];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i][q] = *reinterpret_cast<const f16x8*>(b + a_off + q * PLANE + i * 32 * HG_ROWB);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j][q] = *reinterpret_cast<const f16x8*>(b + w_off + q * PLANE + j * 32 * HG_ROWB);
        }
    };
#define HP_TERM(AF, WF, PA, PW)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)               \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(AF[i][PA], WF[j][PW], acc[i][j], 0, 0, 0);

    f16x8 af0[TM][2], wf0[TN][2], af1[TM][2], wf1[TN][2];
    const int nslabs_all = (p.K + HG_BK - 1) / HG_BK;
    const int kbeg = half == 1 ? (nslabs_all + 1) / 2 : 0;                       // first slab of this workgroup's range
    const int nslabs = half < 0 ? nslabs_all : (half == 0 ? (nslabs_all + 1) / 2 : nslabs_all - kbeg);
    // (the pipeline stages up to three slabs past the end of its range: never multiplied, and inside the descriptor bounds or zero)
    // Hand-interleaved slab body.  The compiler's own order (even with sched_group_barrier requests) clumped the MFMAs
    // (12 back to back, then ~150 VALU/LDS instructions with the matrix pipe idle); here every slot is ONE MFMA followed by one
    // chunk of the split work (half a float4: scale, cvt_pk, exact residual, cvt_pk -- ~6 VALU ~ the 32 cycles the MFMA
    // occupies the pipe) and sched_barrier(0) pins the order.  16 chunks (8 float4 x 2 halves) ride on the first 16 MFMAs.
    auto mfma = [&](int m, const f16x8 (&af)[TM][2], const f16x8 (&wf)[TN][2]) {   // m in [0, 12): term-major
        constexpr int PA[3] = {0, 1, 0}, PW[3] = {1, 0, 0};                        // hi*lo, lo*hi, hi*hi
        const int term = m >> 2, i = (m >> 1) & 1, j = m & 1;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][PA[term]], wf[j][PW[term]], acc[i][j], 0, 0, 0);
    };
    unsigned ph0, pl0;   // first half of the float4 being split
    // chunk c in [0, 16): half of float4 q = c>>1 (0..3 = A stripes, 4..7 = W stripes) of slab t+1; the odd chunk also stores the
    // float4's two 8-byte slots and immediately re-issues the register's global load for slab t+3 (knext): every load gets
    // ~2 slab periods to land instead of ~1.3 when all eight are issued after the last chunk.
    auto split_chunk = [&](int c, int buf, int knext, f32x4 (&a)[NF4], f32x4 (&w)[NF4]) {
        const int q = c >> 1;
        const bool isw = q >= NF4;
        const int i = isw ? q - NF4 : q;
        const f32x4 v = isw ? w[i] : a[i];
        const float sc = isw ? scw[i] : sca[i];
        const bool packed = isw ? WPK : APK;
        if ((c & 1) == 0) {
            if (!packed) split2(f32x2{v[0], v[1]}, sc, ph0, pl0);
        } else {
            unsigned char* st = smem + buf * STAGE + st_off + (isw ? 2 * PLANE : 0) + i * 32 * HG_ROWB;
            if (packed) {
                const u32x4 raw = __builtin_bit_cast(u32x4, v);
                *reinterpret_cast<u32x2*>(st) = u32x2{raw[0], raw[1]};
                *reinterpret_cast<u32x2*>(st + PLANE) = u32x2{raw[2], raw[3]};
            } else {
                unsigned h1, l1;
                split2(f32x2{v[2], v[3]}, sc, h1, l1);
                *reinterpret_cast<u32x2*>(st) = u32x2{ph0, h1};
                *reinterpret_cast<u32x2*>(st + PLANE) = u32x2{pl0, l1};
            }
            const bool kok = knext + lc4 * 4 < p.K;
            if (isw) w[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, kok ? offW[i] : OOB, knext * 4, 0));
            else a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, kok ? offA[i] : OOB, knext * 4, 0));
        }
    };
    auto slab_body = [&](int t, int buf, f32x4 (&a_next)[NF4], f32x4 (&w_next)[NF4]) {
        // ---- region A: F1 <- LDS[buf]; 12 MFMAs of step 0 + 4 of step 1, the split of slab t+1 into LDS[buf^1] behind them
        load_frags(buf, 1, af1, wf1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 12; ++m) {
            mfma(m, af0, wf0);
            split_chunk(m, buf ^ 1, (kbeg + t + 1 + NSETS) * HG_BK, a_next, w_next);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            mfma(m, af1, wf1);
            split_chunk(12 + m, buf ^ 1, (kbeg + t + 1 + NSETS) * HG_BK, a_next, w_next);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // ---- region B: F0 <- LDS[buf^1] (slab t+1, step 0; after the last slab the all-zero slab, unused); last 8 MFMAs
        load_frags(buf ^ 1, 0, af0, wf0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 4; m < 12; ++m) mfma(m, af1, wf1);
        __builtin_amdgcn_sched_barrier(0);
    };
    // slab s lives in register set s % NSETS: slabs 0..NSETS-1 first, then set 0 (staged as slab 0) is refilled with slab NSETS
#pragma unroll
    for (int u = 0; u < NSETS; ++u) load_slab((kbeg + u) * HG_BK, ra[u], rw[u]);
    split_store(0, ra[0], rw[0]);
    load_slab((kbeg + NSETS) * HG_BK, ra[0], rw[0]);
    __syncthreads();
    load_frags(0, 0, af0, wf0);
    constexpr int UNR = NSETS % 2 == 0 ? NSETS : 2 * NSETS;   // the LDS stage alternates per slab, the register set cycles per NSETS
    int t = 0;
    for (; t + UNR <= nslabs; t += UNR) {       // straight-line groups: every vmcnt wait stays counted
#pragma unroll
        for (int u = 0; u < UNR; ++u) slab_body(t + u, u & 1, ra[(u + 1) % NSETS], rw[(u + 1) % NSETS]);
    }
#pragma unroll
    for (int u = 0; u < UNR - 1; ++u)
        if (t + u < nslabs) slab_body(t + u, u & 1, ra[(u + 1) % NSETS], rw[(u + 1) % NSETS]);
#undef HP_TERM

    if (half >= 0) {   // split-K hand-over: [split tile][wave][tile i][tile j][reg][lane] floats, 256-byte rows per wave instruction
        // Every access to the partial and the flag is an AGENT-scope relaxed atomic (sc1: written through / read at the device
        // coherence point), ordered by plain waitcnt + barrier -- coherent across XCDs without the bulk L2 write-back / invalidate
        // of an agent-scope release/acquire FENCE (measured: with fences the split launches were 2x slower than unsplit ones).
        float* part = p.ws + ((size_t)split_idx * 4 + wave) * (TM * TN * 16 * 64) + lane;
        if (half == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __hip_atomic_store(part + ((i * TN + j) * 16 + r) * 64, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // = wait for this wave's stores to be acknowledged
            __syncthreads();
            if (tid == 0) __hip_atomic_store(p.flags + split_idx, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (tid == 0)
            while (__hip_atomic_load(p.flags + split_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.epoch) __builtin_amdgcn_s_sleep(8);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[i][j][r] += __hip_atomic_load(part + ((i * TN + j) * 16 + r) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();   // every wave is done reading operand fragments
    gemm_store_tile<TM, TN, true>(p, acc, reinterpret_cast<float*>(smem) + wave * gemm_epilogue_lds_floats_per_wave<TN>(), m0 + wm * TM * 32,
                                  n0 + wn * TN * 32, lane, p.C, p.residual);
}

static int g_f16x3_cfg = -1;  // tuning hook: 0 = 128x128 (2x2 waves of 64x64), 1 = 128x64 (4x1 waves of 32x64), 2 = 128x128 without fragment double
// buffering (fewer registers), 3 = 128x128 software-pipelined (double-buffered LDS), -1 = auto
PSAM_API void psam_gemm_f16x3_force_config(int32_t cfg) { g_f16x3_cfg = cfg; }
static int g_f16x3_deep = -1;  // tuning hook: 2 / 3 / 4 force that many operand register sets in the pipelined kernel, -1 = auto
PSAM_API void psam_gemm_f16x3_force_deep(int32_t sets) { g_f16x3_deep = sets; }

// C = act(alpha * A @ W^T + bias + rowbias[row/rowgroup]) + residual; scaleA[M], scaleW[N] from psam_row_scale_f16.
// a_packed / w_packed: that operand is the f16x2-packed form of the row-scaled matrix (psam_pack_rows_f16x2 with the same scales).
static int f16x3_slots() {   // resident 64-KiB-LDS workgroups on the device: 2 per CU
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        slots = 2 * cus;
    }
    return slots;
}
constexpr size_t F16X3_FLAG_BYTES = 16384, F16X3_PART_BYTES = 4 * 4 * 16 * 64 * sizeof(float);   // per split tile: 4 waves x 4 acc tiles

// Workspace for the split-K hand-over of psam_gemm_f16x3_ws: zero it ONCE (flags), then pass it with a launch-unique, non-zero,
// never-repeating `epoch` to every call that uses it; one workspace per stream that runs such GEMMs concurrently.
PSAM_API size_t psam_gemm_f16x3_workspace_bytes(void) { return F16X3_FLAG_BYTES + (size_t)(f16x3_slots() / 2) * F16X3_PART_BYTES; }

PSAM_API int32_t psam_gemm_f16x3_ws(const void* A, int64_t lda, const float* scaleA, int32_t a_packed, const void* W, int64_t ldw,
                                    const float* scaleW, int32_t w_packed, float* C, int64_t ldc, const float* bias, const float* residual,
                                    int64_t ldr, const float* rowbias, int64_t ldrb, int32_t rowgroup, int32_t M, int32_t N, int32_t K,
                                    float alpha, int32_t act, void* ws, size_t ws_bytes, uint32_t epoch, hipStream_t stream) {
    PSAM_REQUIRE(A && W && C && scaleA && scaleW, PSAM_EINVAL, "psam_gemm_f16x3: null pointer");
    PSAM_REQUIRE(M > 0 && N > 0 && K > 0, PSAM_EINVAL, "psam_gemm_f16x3: bad shape");
    PSAM_REQUIRE(act >= 0 && act <= 3, PSAM_EINVAL, "psam_gemm_f16x3: bad activation code");
    PSAM_REQUIRE(!rowbias || rowgroup > 0, PSAM_EINVAL, "psam_gemm_f16x3: rowbias needs rowgroup > 0");
    PSAM_REQUIRE((K & 3) == 0 && (lda & 3) == 0 && (ldw & 3) == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, PSAM_EALIGN,
                 "psam_gemm_f16x3: K, lda, ldw must be multiples of 4 and A, W 16-byte aligned");
    PSAM_REQUIRE(((int64_t)M - 1) * lda + K < ((int64_t)1 << 29) - 8 && ((int64_t)N - 1) * ldw + K < ((int64_t)1 << 29) - 8, PSAM_EINVAL,
                 "psam_gemm_f16x3: one operand matrix must span < 2 GiB (32-bit buffer offsets)");
    PSAM_REQUIRE(act != 3 || ((N & 63) == 0 && !residual && !rowbias), PSAM_EINVAL,
                 "psam_gemm_f16x3: SwiGLU epilogue needs N % 64 == 0, no residual/rowbias");
    F16x3Args p;
    p.A = (const float*)A; p.W = (const float*)W; p.C = C; p.bias = bias; p.residual = residual; p.rowbias = rowbias; p.scaleA = scaleA; p.scaleW = scaleW;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.ldrb = ldrb;
    p.M = M; p.N = N; p.K = K; p.rowgroup = rowgroup > 0 ? rowgroup : 1; p.act = act; p.alpha = alpha;
    int cfg = g_f16x3_cfg;
    // measured (scripts/gemm_split_bench.py): the software-pipelined 128x128 kernel wins on every shape of the path
    if (a_packed || w_packed) cfg = 3;   // packed operands exist only in the pipelined kernel
    if (cfg < 0) cfg = N > 64 ? 3 : 1;
    const int bn = cfg == 1 ? 64 : 128;
    p.tiles_m = (int)psam_cdiv(M, 128);
    p.tiles_n = (int)psam_cdiv(N, bn);
    const int ntiles = p.tiles_m * p.tiles_n;
    // split-K policy (pipelined kernel, workspace given): all tiles when there are fewer than two per slot-pair (N = 1024 GEMMs at
    // M = 4096: 256 tiles on 512 slots), else the tiles of a last round that is at most half full (qkv: 768 = 512 + 256)
    int nsplit = 0;
    if (cfg == 3 && ws && K >= 16 * HG_BK) {
        const int slots = f16x3_slots(), rem = ntiles % slots;
        if (2 * ntiles <= slots) nsplit = ntiles;
        else if (rem > 0 && 2 * rem <= slots) nsplit = rem;
        nsplit &= ~7;
        PSAM_REQUIRE(nsplit == 0 || (epoch != 0 && ws_bytes >= F16X3_FLAG_BYTES + (size_t)nsplit * F16X3_PART_BYTES && (size_t)nsplit * 4 <= F16X3_FLAG_BYTES),
                     PSAM_EINVAL, "psam_gemm_f16x3_ws: workspace too small (psam_gemm_f16x3_workspace_bytes) or epoch == 0");
    }
    p.split_from = ntiles - nsplit; p.epoch = epoch; p.flags = (unsigned*)ws; p.ws = (float*)((char*)ws + F16X3_FLAG_BYTES);
    const dim3 grid((unsigned)(cfg == 3 ? ntiles + nsplit : ntiles));
    if (cfg == 0) hipLaunchKernelGGL((gemm_f16x3_kernel<2, 2, 2, 2, true>), grid, dim3(256), 0, stream, p);
    else if (cfg == 2) hipLaunchKernelGGL((gemm_f16x3_kernel<2, 2, 2, 2, false>), grid, dim3(256), 0, stream, p);
    else if (cfg == 3) {
        // operand register sets (prefetch distance): 2.  Deeper (3: 246 registers on paper but the allocator spills the second
        // wave per SIMD; 4: one wave per SIMD) was measured on one-tile-per-CU launches (proj, fc2) and on full launches: no gain
        // on the former (35.2 / 33.8 / 34.2 us: not load-latency bound), 20-30 % slower on the latter (occupancy).  Hook kept.
        const int deep = g_f16x3_deep >= 0 ? g_f16x3_deep : 2;
#define PIPE_LAUNCH(AP, WP)                                                                                          \
    do {                                                                                                             \
        if (deep >= 4) hipLaunchKernelGGL((gemm_f16x3_pipe_kernel<AP, WP, 4>), grid, dim3(256), 0, stream, p);      \
        else if (deep == 3) hipLaunchKernelGGL((gemm_f16x3_pipe_kernel<AP, WP, 3>), grid, dim3(256), 0, stream, p); \
        else hipLaunchKernelGGL((gemm_f16x3_pipe_kernel<AP, WP, 2>), grid, dim3(256), 0, stream, p);                \
    } while (0)
        if (a_packed && w_packed) PIPE_LAUNCH(true, true);
        else if (a_packed) PIPE_LAUNCH(true, false);
        else if (w_packed) PIPE_LAUNCH(false, true);
        else PIPE_LAUNCH(false, false);
#undef PIPE_LAUNCH
    } else hipLaunchKernelGGL((gemm_f16x3_kernel<4, 1, 1, 2, false>), grid, dim3(256), 0, stream, p);
    return psam_launch_status("psam_gemm_f16x3: launch failed");
}

PSAM_API int32_t psam_gemm_f16x3_ex(const void* A, int64_t lda, const float* scaleA, int32_t a_packed, const void* W, int64_t ldw,
                                    const float* scaleW, int32_t w_packed, float* C, int64_t ldc, const float* bias, const float* residual,
                                    int64_t ldr, const float* rowbias, int64_t ldrb, int32_t rowgroup, int32_t M, int32_t N, int32_t K,
                                    float alpha, int32_t act, hipStream_t stream) {
    return psam_gemm_f16x3_ws(A, lda, scaleA, a_packed, W, ldw, scaleW, w_packed, C, ldc, bias, residual, ldr, rowbias, ldrb, rowgroup, M, N, K, alpha,
                              act, nullptr, 0, 0, stream);
}

PSAM_API int32_t psam_gemm_f16x3(const float* A, int64_t lda, const float* scaleA, const float* W, int64_t ldw, const float* scaleW, float* C,
                                 int64_t ldc, const float* bias, const float* residual, int64_t ldr, const float* rowbias, int64_t ldrb,
                                 int32_t rowgroup, int32_t M, int32_t N, int32_t K, float alpha, int32_t act, hipStream_t stream) {
    return psam_gemm_f16x3_ex(A, lda, scaleA, 0, W, ldw, scaleW, 0, C, ldc, bias, residual, ldr, rowbias, ldrb, rowgroup, M, N, K, alpha, act, stream);
}
