// fp32-ACCURATE GEMM on the bf16 matrix pipe ("bf16x6"): same contract and epilogues as gemm.hip,
//
//   C[z] = act(alpha * A[z] @ W[z]^T + bias + rowbias[row / rowgroup]) + residual[z],      A, W, C fp32 in HBM.
//
// Every fp32 operand x is split EXACTLY into three bf16 numbers, x = hi + mid + lo (hi = RNE_bf16(x), mid = RNE_bf16(x-hi),
// lo = x-hi-mid: 3 x 8 significand bits = the 24 of fp32; the two subtractions are exact in fp32).  A product a*b is the sum
// of 9 partial products; the six with weight >= 2^-17 are computed (hh, hm, mh, mm, hl, lh), the dropped three are
// <= 2^-24 relative -- below fp32 round-off.  Each bf16 x bf16 product is exact in fp32 and v_mfma_f32_32x32x16_bf16
// accumulates in fp32, so the result has fp32-GEMM accuracy (tests: error vs an fp64 reference is the same as the
// f32-MFMA kernel's) at 6/16 of the matrix-pipe time of v_mfma_f32_32x32x2_f32.
//
// The split happens in registers while a K slab moves global -> LDS (v_cvt_pk_bf16_f32 + shifts + exact subtracts), so
// operands stay plain fp32 in HBM and no other kernel changes.  Tile 128x128 (2x2 waves of 64x64) or 128x64 (4 waves of
// 32x64), slabs of 32 k; LDS holds the three bf16 planes of both operands (48 / 36 KiB, single buffered); rows are 64 bytes with the
// 16-byte chunk index XOR-swizzled by (row>>2)&3: conflict-free ds_read_b128 fragments and ds_write_b64 staging.
#include "common.h"
#include "gemm_epilogue.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct SplitGemmArgs {
    const float* A; const float* W; float* C;
    const float* bias; const float* residual; const float* rowbias;
    int64_t lda, ldw, ldc, ldr, ldrb;
    int64_t sA1, sA2, sW1, sW2, sC1, sC2, sR1, sR2;
    int M, N, K, batch2, rowgroup, act;
    float alpha;
    int tiles_m, tiles_n;
};

constexpr int SG_BK = 32;
constexpr int SG_ROWB = SG_BK * 2;                       // bytes per row of one bf16 plane

// x (2 floats) -> packed bf16 pairs hi, mid, lo with x == hi + mid + lo exactly
__device__ __forceinline__ void split3(const f32x2 x, unsigned& hi, unsigned& mid, unsigned& lo) {
    const bf16x2 h = __builtin_convertvector(x, bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    mid = __builtin_bit_cast(unsigned, m);
    lo = __builtin_bit_cast(unsigned, l);
}

// WM x WN waves (4 in total), each TM x TN accumulator tiles of 32x32.  FDB: double-buffer the fragment registers
// across the two k16 steps of a slab (costs 12*(TM+TN) VGPRs).
template <int WM, int WN, int TM, int TN, bool FDB>
__global__ __launch_bounds__(256) void gemm_bf16x6_kernel(const SplitGemmArgs p) {
    static_assert(WM * WN == 4, "256-thread workgroup");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_F4 = BM * 8 / 256, W_F4 = BN * 8 / 256;   // float4 per thread per slab
    constexpr int PLANE_A = BM * SG_ROWB, PLANE_W = BN * SG_ROWB;
    // LDS: [A planes hi,mid,lo][W planes hi,mid,lo]
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * PLANE_A + 3 * PLANE_W];
    unsigned char* sA = smem;
    unsigned char* sW = smem + 3 * PLANE_A;

    const int ntiles = p.tiles_m * p.tiles_n;
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int z = blockIdx.z, z1 = z / p.batch2, z2 = z % p.batch2;
    const float* A = p.A + z1 * p.sA1 + z2 * p.sA2;
    const float* W = p.W + z1 * p.sW1 + z2 * p.sW2;

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
    const int st_off = lr * SG_ROWB + ((((lc4 >> 1) ^ ((lr >> 2) & 3)) << 4) | ((lc4 & 1) << 3));
    // split registers: [row stripe][plane hi/mid/lo] -> 2 packed words (4 bf16)
    u32x2 spa[A_F4][3], spw[W_F4][3];
    auto split_regs = [&]() {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            unsigned h0, m0_, l0, h1, m1, l1;
            split3(f32x2{ra0[i][0], ra0[i][1]}, h0, m0_, l0);
            split3(f32x2{ra0[i][2], ra0[i][3]}, h1, m1, l1);
            spa[i][0] = u32x2{h0, h1}; spa[i][1] = u32x2{m0_, m1}; spa[i][2] = u32x2{l0, l1};
        }
#pragma unroll
        for (int i = 0; i < W_F4; ++i) {
            unsigned h0, m0_, l0, h1, m1, l1;
            split3(f32x2{rw0[i][0], rw0[i][1]}, h0, m0_, l0);
            split3(f32x2{rw0[i][2], rw0[i][3]}, h1, m1, l1);
            spw[i][0] = u32x2{h0, h1}; spw[i][1] = u32x2{m0_, m1}; spw[i][2] = u32x2{l0, l1};
        }
    };
    auto store_split = [&]() {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) *reinterpret_cast<u32x2*>(sA + q * PLANE_A + i * 32 * SG_ROWB + st_off) = spa[i][q];
#pragma unroll
            for (int i = 0; i < W_F4; ++i) *reinterpret_cast<u32x2*>(sW + q * PLANE_W + i * 32 * SG_ROWB + st_off) = spw[i][q];
        }
    };
    // fragment addresses: row r32 of 32-row tile, chunk (2s + h) swizzled
    int frag_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) frag_off[s] = r32 * SG_ROWB + (((2 * s + h) ^ ((r32 >> 2) & 3)) << 4);
    const unsigned char* a_base = sA + wm * TM * 32 * SG_ROWB;
    const unsigned char* w_base = sW + wn * TN * 32 * SG_ROWB;
    auto load_frags = [&](int s, bf16x8 (&af)[TM][3], bf16x8 (&wf)[TN][3]) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i][q] = *reinterpret_cast<const bf16x8*>(a_base + q * PLANE_A + i * 32 * SG_ROWB + frag_off[s]);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j][q] = *reinterpret_cast<const bf16x8*>(w_base + q * PLANE_W + j * 32 * SG_ROWB + frag_off[s]);
        }
    };
    // smallest partial products first (hl, lh, mm), then hm, mh, then hh; term-major so that the accumulator tiles rotate
#define SG_TERM(AF, WF, PA, PW)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)               \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[i][PA], WF[j][PW], acc[i][j], 0, 0, 0);
#define SG_STEP(AF, WF) SG_TERM(AF, WF, 0, 2) SG_TERM(AF, WF, 2, 0) SG_TERM(AF, WF, 1, 1) SG_TERM(AF, WF, 0, 1) SG_TERM(AF, WF, 1, 0) SG_TERM(AF, WF, 0, 0)

    const int nslabs = (p.K + SG_BK - 1) / SG_BK;
    // Pipeline.  At the top of slab t: sp = split slab t (registers); (ra0, rw0) = fp32 slab t+1 (landed or landing).
    // Slab t: [barrier: LDS free] sp -> LDS [barrier: LDS ready]; fragments of k16-step 0; then 2 x 6*TM*TN MFMAs, in whose
    // shadow issue: the fragment reads of step 1, the split of slab t+1 into sp and the refill of the fp32 registers with
    // slab t+2.  Loads past K return zeros (bounds-checked descriptor) and are unconditional: counted vmcnt waits.
    bf16x8 af0[TM][3], wf0[TN][3], af1[TM][3], wf1[TN][3];
    load_slab(0);
    split_regs();
    load_slab(SG_BK);
    constexpr int NM = 6 * TM * TN, NFR = 3 * (TM + TN);
    for (int t = 0; t < nslabs; ++t) {
        __syncthreads();
        store_split();
        __syncthreads();
        load_frags(0, af0, wf0);
        if (FDB) {
            load_frags(1, af1, wf1);
            SG_STEP(af0, wf0)
            split_regs();                 // slab t+1
            SG_STEP(af1, wf1)
        } else {
            SG_STEP(af0, wf0)
            load_frags(1, af0, wf0);
            split_regs();                 // slab t+1
            SG_STEP(af0, wf0)
        }
        load_slab((t + 2) * SG_BK);
        // requested issue order (one scheduling region: the loop body after the second barrier)
        __builtin_amdgcn_sched_group_barrier(0x100, NFR, 0);      // fragments of step 0
        if (FDB) {
#pragma unroll
            for (int i = 0; i < NFR; ++i) {                        // step-0 MFMAs, fragment reads of step 1 behind them
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
            }
#pragma unroll
            for (int i = 0; i < 2 * NM - NFR; ++i) {              // remaining MFMAs with the split arithmetic behind them
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 6, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, NFR, 0);
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x2, 6, 0);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x2, 64, 0);
        __builtin_amdgcn_sched_group_barrier(0x20, A_F4 + W_F4, 0);
    }
#undef SG_STEP
#undef SG_TERM

    // ---- epilogue: LDS transpose -> row-contiguous float4 stores (gemm_epilogue.h)
    float* C = p.C + z1 * p.sC1 + z2 * p.sC2;
    const float* R = p.residual ? p.residual + z1 * p.sR1 + z2 * p.sR2 : nullptr;
    static_assert(4 * gemm_epilogue_lds_floats_per_wave<TN>() * 4 <= 3 * PLANE_A + 3 * PLANE_W, "epilogue staging fits the operand LDS");
    __syncthreads();   // every wave is done reading operand fragments
    gemm_store_tile<TM, TN>(p, acc, reinterpret_cast<float*>(smem) + wave * gemm_epilogue_lds_floats_per_wave<TN>(), m0 + wm * TM * 32,
                            n0 + wn * TN * 32, lane, C, R);
}

static int g_split_cfg = -1;  // tuning hook: 0 = 128x128 (2x2 waves of 64x64), 1 = 128x64 (4x1 waves of 32x64), -1 = auto
PSAM_API void psam_gemm_bf16x6_force_config(int32_t cfg) { g_split_cfg = cfg; }

// Same argument list as psam_gemm_f32 (include/pointsam_hip.h).
PSAM_API int32_t psam_gemm_bf16x6(const float* A, int64_t lda, int64_t sA1, int64_t sA2, const float* W, int64_t ldw, int64_t sW1,
                                  int64_t sW2, float* C, int64_t ldc, int64_t sC1, int64_t sC2, const float* bias, const float* residual,
                                  int64_t ldr, int64_t sR1, int64_t sR2, const float* rowbias, int64_t ldrb, int32_t rowgroup, int32_t M,
                                  int32_t N, int32_t K, int32_t batch1, int32_t batch2, float alpha, int32_t act, hipStream_t stream) {
    PSAM_REQUIRE(A && W && C, PSAM_EINVAL, "psam_gemm_bf16x6: null pointer");
    PSAM_REQUIRE(M > 0 && N > 0 && K > 0 && batch1 > 0 && batch2 > 0, PSAM_EINVAL, "psam_gemm_bf16x6: bad shape");
    PSAM_REQUIRE(act >= 0 && act <= 3, PSAM_EINVAL, "psam_gemm_bf16x6: bad activation code");
    PSAM_REQUIRE((int64_t)batch1 * batch2 <= 65535, PSAM_EINVAL, "psam_gemm_bf16x6: batch > 65535");
    PSAM_REQUIRE(!rowbias || rowgroup > 0, PSAM_EINVAL, "psam_gemm_bf16x6: rowbias needs rowgroup > 0");
    PSAM_REQUIRE((K & 3) == 0 && (lda & 3) == 0 && (ldw & 3) == 0 && (sA1 & 3) == 0 && (sA2 & 3) == 0 && (sW1 & 3) == 0 &&
                     (sW2 & 3) == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
                 PSAM_EALIGN, "psam_gemm_bf16x6: K, lda, ldw, batch strides must be multiples of 4 and A, W 16-byte aligned");
    PSAM_REQUIRE(((int64_t)M - 1) * lda + K < ((int64_t)1 << 29) - 8 && ((int64_t)N - 1) * ldw + K < ((int64_t)1 << 29) - 8, PSAM_EINVAL,
                 "psam_gemm_bf16x6: one operand matrix must span < 2 GiB (32-bit buffer offsets); split the batch");
    PSAM_REQUIRE(act != 3 || ((N & 63) == 0 && !residual && !rowbias), PSAM_EINVAL,
                 "psam_gemm_bf16x6: SwiGLU epilogue needs N % 64 == 0, no residual/rowbias");
    SplitGemmArgs p;
    p.A = A; p.W = W; p.C = C; p.bias = bias; p.residual = residual; p.rowbias = rowbias;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.ldrb = ldrb;
    p.sA1 = sA1; p.sA2 = sA2; p.sW1 = sW1; p.sW2 = sW2; p.sC1 = sC1; p.sC2 = sC2; p.sR1 = sR1; p.sR2 = sR2;
    p.M = M; p.N = N; p.K = K; p.batch2 = batch2; p.rowgroup = rowgroup > 0 ? rowgroup : 1; p.act = act; p.alpha = alpha;
    int cfg = g_split_cfg;
    // measured (scripts/gemm_split_bench.py): both shapes saturate near 158 TFLOP/s fp32-equivalent once >= 2 workgroups
    // share a CU; with fewer than 512 128x128 tiles (N = 1024 GEMMs at M = 4096) the 128x64 shape keeps 2+ per CU.
    if (cfg < 0) cfg = (psam_cdiv(M, 128) * psam_cdiv(N, 128) * (int64_t)batch1 * batch2 >= 512 && K > 256) ? 0 : 1;
    const int bn = cfg == 0 ? 128 : 64;
    p.tiles_m = (int)psam_cdiv(M, 128);
    p.tiles_n = (int)psam_cdiv(N, bn);
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n), 1, (unsigned)((int64_t)batch1 * batch2));
    if (cfg == 0) hipLaunchKernelGGL((gemm_bf16x6_kernel<2, 2, 2, 2, true>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gemm_bf16x6_kernel<4, 1, 1, 2, false>), grid, dim3(256), 0, stream, p);
    return psam_launch_status("psam_gemm_bf16x6: launch failed");
}
