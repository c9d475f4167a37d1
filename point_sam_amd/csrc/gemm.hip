// fp32-exact GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32): the dense contractions of the path --
// nn.Linear everywhere (pc_sam/model/common.py:486-497, pc_encoder.py:99-116,127-143, timm Eva blocks,
// transformer.py:199-202,248-249, mask_decoder.py:53-59,201-203).
//
//   C[z] = act(alpha * A[z] @ W[z]^T + bias + rowbias[row / rowgroup]) + residual[z]
//
// A is [M,K] row-major (lda), W is [N,K] row-major (nn.Linear layout, ldw), both K-contiguous ("NT").  f32-input
// MFMA is bit-for-bit an fp32 fma chain, so results stay within fp32 round-off of the CPU oracle -- this is the
// precision the north_star tolerance (1e-3 on logits through 24-40 blocks) needs; the peak it is priced against is
// the 157.3 TFLOP/s f32 matrix rate, not the bf16 one.
//
// Tiling: 256 threads = 2x2 waves, each wave TM x TN MFMA tiles of 32x32; K is consumed in 32-wide slabs staged
// through LDS (unpadded 128-byte rows with an XOR chunk swizzle: conflict-free ds_read_b128), double-buffered with
// the next slab prefetched into registers while the current one feeds the matrix pipe.  One ds_read_b128 per operand
// row feeds 4 MFMAs: lanes 0-31 take k = 8s..8s+3, lanes 32-63 take k = 8s+4..8s+7 (A and W use the same
// permutation of k, so the sum over k is unchanged).
#include "common.h"
#include "gemm_epilogue.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const float* A; const float* W; float* C;
    const float* bias;      // [N] or null
    const float* residual;  // same indexing as C (ldr, batch strides sR*) or null; may alias C
    const float* rowbias;   // [ceil(M/rowgroup), ldrb] or null (per-group broadcast row, e.g. the max-pooled half of conv2.0)
    int64_t lda, ldw, ldc, ldr, ldrb;
    int64_t sA1, sA2, sW1, sW2, sC1, sC2, sR1, sR2;  // batch strides (elements): z1 = z / batch2, z2 = z % batch2
    int M, N, K, batch2, rowgroup, act;               // act: 0 none, 1 GELU(erf), 2 ReLU, 3 SwiGLU gate (paired tiles)
    float alpha;
    int tiles_m, tiles_n;
};

constexpr int GEMM_BK = 32;   // K slab: one 128-byte row segment per operand row

// LDS image of a slab: rows of 32 floats (8 x 16-byte chunks), NO padding; chunk c of row r lives at chunk slot
// c ^ ((r >> 1) & 7).  Conflict-free for ds_read_b128 (each 16-lane service group sees 16 distinct rows mod 16 ->
// 16 distinct 4-bank slots) and for ds_write_b128 (8 lanes cover one whole row).  Unpadded rows keep a 128x64
// double-buffered tile at 48 KiB, i.e. 3 workgroups (12 waves) per CU.
__device__ __forceinline__ int lds_chunk_off(int row, int chunk) { return row * GEMM_BK + ((chunk ^ ((row >> 1) & 7)) << 2); }

template <int WM, int WN, int TM, int TN>  // WM x WN waves (WM*WN == 4), each TM x TN MFMA tiles of 32x32
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs p) {
    static_assert(WM * WN == 4, "256-thread workgroup");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int A_F4 = BM * (GEMM_BK / 4) / 256;  // float4 loads per thread per slab
    constexpr int W_F4 = BN * (GEMM_BK / 4) / 256;
    __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * GEMM_BK];   // [A buf 0,1][W buf 0,1]; reused by the epilogue
    float (*sA)[BM * GEMM_BK] = reinterpret_cast<float (*)[BM * GEMM_BK]>(smem);
    float (*sW)[BN * GEMM_BK] = reinterpret_cast<float (*)[BN * GEMM_BK]>(smem + 2 * BM * GEMM_BK);

    // XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs, so give each XCD a contiguous
    // range of tiles (they share A row panels / W column panels in that XCD's private L2).
    const int ntiles = p.tiles_m * p.tiles_n;
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);
    const int tm_i = tile / p.tiles_n, tn_i = tile % p.tiles_n;
    const int m0 = tm_i * BM, n0 = tn_i * BN;
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

    // Two register sets for global->LDS staging: the loads of slab t+2 are issued while slab t is computed, so a
    // load has ~1.5 slab times (2-4 us) to return -- operands mostly come from the Infinity Cache / HBM (A and W exceed
    // the 4 MiB XCD L2) and one slab time is not enough under load (measured: +40 % MFMA idle with distance 1).
    f32x4 ra0[A_F4], rw0[W_F4], ra1[A_F4], rw1[W_F4];
    const int lr = tid >> 3, lc4 = tid & 7;             // this thread's (row, 16-byte chunk) inside a 32-row stripe
    const int st_off = lds_chunk_off(lr, lc4);           // (row >> 1) & 7 is the same for row and row + 32*i

    // Operands are read through buffer descriptors: rows >= M / >= N and the K tail fall outside num_records and
    // return 0 from the hardware bounds check -- no exec-mask branches around the loads, so the compiler can keep the
    // two register sets in flight with COUNTED vmcnt waits instead of draining to vmcnt(0).
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((((int64_t)p.M - 1) * p.lda + p.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)((((int64_t)p.N - 1) * p.ldw + p.K) * 4), 0x00020000);
    const int voA = (int)(((int64_t)(m0 + lr) * p.lda + lc4 * 4) * 4), voW = (int)(((int64_t)(n0 + lr) * p.ldw + lc4 * 4) * 4);
    const int stepA = (int)(32 * p.lda * 4), stepW = (int)(32 * p.ldw * 4);
    constexpr int OOB = 0x7ffffff0;  // >= num_records (operands are < 2 GiB, checked on the host)
    auto load_slab = [&](int k0, f32x4 (&ra)[A_F4], f32x4 (&rw)[W_F4]) {
        const bool kok = k0 + lc4 * 4 < p.K;
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const bool ok = kok && (m0 + i * 32 + lr < p.M);
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? voA + i * stepA : OOB, k0 * 4, 0));
        }
#pragma unroll
        for (int i = 0; i < W_F4; ++i) {
            const bool ok = kok && (n0 + i * 32 + lr < p.N);
            rw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, ok ? voW + i * stepW : OOB, k0 * 4, 0));
        }
    };
    auto store_slab = [&](int buf, const f32x4 (&ra)[A_F4], const f32x4 (&rw)[W_F4]) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) *reinterpret_cast<f32x4*>(&sA[buf][i * 32 * GEMM_BK + st_off]) = ra[i];
#pragma unroll
        for (int i = 0; i < W_F4; ++i) *reinterpret_cast<f32x4*>(&sW[buf][i * 32 * GEMM_BK + st_off]) = rw[i];
    };
    // fragment of k-chunk s (k = 8s .. 8s+7): lanes 0-31 take 8s..8s+3, lanes 32-63 take 8s+4..8s+7
    int frag_off[GEMM_BK / 8];
#pragma unroll
    for (int s = 0; s < GEMM_BK / 8; ++s) frag_off[s] = lds_chunk_off(r32, 2 * s + h);
    const int a_row0 = wm * TM * 32 * GEMM_BK, w_row0 = wn * TN * 32 * GEMM_BK;
    auto load_frags = [&](int buf, int s, f32x4 (&af)[TM], f32x4 (&wf)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(&sA[buf][a_row0 + i * 32 * GEMM_BK + frag_off[s]]);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const f32x4*>(&sW[buf][w_row0 + j * 32 * GEMM_BK + frag_off[s]]);
    };

    // Software pipeline (one barrier per slab, no LDS latency exposed to the matrix pipe):
    //   slab t+2: global -> register set (t&1) at the top of slab t;
    //   slab t+1: register set ((t+1)&1) -> LDS[buf^1] before the 3rd MFMA chunk of slab t;
    //   fragments of chunk s+1 are read from LDS while chunk s feeds the MFMAs, also across the slab barrier.
    f32x4 af[2][TM], wf[2][TN];
    const int nslabs = (p.K + GEMM_BK - 1) / GEMM_BK;
    auto slab_body = [&](int t, int buf, f32x4 (&ra_far)[A_F4], f32x4 (&rw_far)[W_F4], f32x4 (&ra_near)[A_F4], f32x4 (&rw_near)[W_F4]) {
#pragma unroll
        for (int s = 0; s < GEMM_BK / 8; ++s) {
            if (s < GEMM_BK / 8 - 1) {
                load_frags(buf, s + 1, af[(s + 1) & 1], wf[(s + 1) & 1]);
            } else {
                __syncthreads();  // all waves: reads of LDS[buf] done, writes of LDS[buf^1] visible
                load_frags(buf ^ 1, 0, af[0], wf[0]);  // (after the last slab this reads the all-zero slab; unused)
            }
            if (s == GEMM_BK / 8 - 2) {  // unconditional, also for the last slab: keeps every vmcnt wait counted
                store_slab(buf ^ 1, ra_near, rw_near);                              // slab t+1 (loaded during slab t-1)
                // registers free again: slab t+3.  Issued unconditionally (past the end of K it is an all-out-of-bounds
                // load returning zeros) so that the number of younger loads in flight is a compile-time constant.
                load_slab((t + 3) * GEMM_BK, ra_near, rw_near);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s & 1][i][q], wf[s & 1][j][q], acc[i][j], 0, 0, 0);
        }
        (void)t; (void)ra_far; (void)rw_far;
    };
    load_slab(0, ra0, rw0);
    load_slab(GEMM_BK, ra1, rw1);
    store_slab(0, ra0, rw0);
    load_slab(2 * GEMM_BK, ra0, rw0);
    __syncthreads();
    load_frags(0, 0, af[0], wf[0]);
    // invariant at the top of slab t: LDS[t&1] holds slab t; set ((t+1)&1) holds (or is receiving) slab t+1;
    // set (t&1) holds (or is receiving) slab t+2.
    int t = 0;
    for (; t + 1 < nslabs; t += 2) {  // both bodies unconditional inside the loop (counted vmcnt needs straight-line code)
        slab_body(t, 0, ra0, rw0, ra1, rw1);
        slab_body(t + 1, 1, ra1, rw1, ra0, rw0);
    }
    if (t < nslabs) slab_body(t, 0, ra0, rw0, ra1, rw1);

    // epilogue: LDS transpose -> row-contiguous float4 stores (gemm_epilogue.h).  act == 3 is the SwiGLU gate fused into fc1
    // (timm SwiGLU: act(fc1_g x) * fc1_x x): the packed weight alternates 32-row blocks of fc1_g and fc1_x, so accumulator
    // tile j=2q holds g and tile j=2q+1 holds x for the SAME 32 hidden units in the same lane/register; output has N/2 columns.
    float* C = p.C + z1 * p.sC1 + z2 * p.sC2;
    const float* R = p.residual ? p.residual + z1 * p.sR1 + z2 * p.sR2 : nullptr;
    static_assert(4 * gemm_epilogue_lds_floats_per_wave<TN>() <= 2 * (BM + BN) * GEMM_BK, "epilogue staging fits the operand LDS");
    __syncthreads();   // every wave is done reading operand fragments
    gemm_store_tile<TM, TN>(p, acc, smem + wave * gemm_epilogue_lds_floats_per_wave<TN>(), m0 + wm * TM * 32, n0 + wn * TN * 32, lane, C, R);
}

static int g_force_cfg = -1;  // test/bench hook: 0=128x128, 1=128x64, 2=64x64, -1=auto
PSAM_API void psam_gemm_force_config(int32_t cfg) { g_force_cfg = cfg; }

// Fully general entry point (strided-batched, two-level batch index z = z1*batch2 + z2).
PSAM_API int32_t psam_gemm_f32(const float* A, int64_t lda, int64_t sA1, int64_t sA2, const float* W, int64_t ldw, int64_t sW1,
                               int64_t sW2, float* C, int64_t ldc, int64_t sC1, int64_t sC2, const float* bias, const float* residual,
                               int64_t ldr, int64_t sR1, int64_t sR2, const float* rowbias, int64_t ldrb, int32_t rowgroup, int32_t M,
                               int32_t N, int32_t K, int32_t batch1, int32_t batch2, float alpha, int32_t act, hipStream_t stream) {
    PSAM_REQUIRE(A && W && C, PSAM_EINVAL, "psam_gemm_f32: null pointer");
    PSAM_REQUIRE(M > 0 && N > 0 && K > 0 && batch1 > 0 && batch2 > 0, PSAM_EINVAL, "psam_gemm_f32: bad shape");
    PSAM_REQUIRE(act >= 0 && act <= 3, PSAM_EINVAL, "psam_gemm_f32: bad activation code");
    PSAM_REQUIRE((int64_t)batch1 * batch2 <= 65535, PSAM_EINVAL, "psam_gemm_f32: batch > 65535");
    PSAM_REQUIRE(!rowbias || rowgroup > 0, PSAM_EINVAL, "psam_gemm_f32: rowbias needs rowgroup > 0");
    // float4 staging: K-contiguous operands whose rows start on 16-byte boundaries
    PSAM_REQUIRE((K & 3) == 0 && (lda & 3) == 0 && (ldw & 3) == 0 && (sA1 & 3) == 0 && (sA2 & 3) == 0 && (sW1 & 3) == 0 &&
                     (sW2 & 3) == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0,
                 PSAM_EALIGN, "psam_gemm_f32: K, lda, ldw, batch strides must be multiples of 4 and A, W 16-byte aligned");
    PSAM_REQUIRE(((int64_t)M - 1) * lda + K < ((int64_t)1 << 29) - 8 && ((int64_t)N - 1) * ldw + K < ((int64_t)1 << 29) - 8, PSAM_EINVAL,
                 "psam_gemm_f32: one operand matrix must span < 2 GiB (32-bit buffer offsets); split the batch");
    GemmArgs p;
    p.A = A; p.W = W; p.C = C; p.bias = bias; p.residual = residual; p.rowbias = rowbias;
    p.lda = lda; p.ldw = ldw; p.ldc = ldc; p.ldr = ldr; p.ldrb = ldrb;
    p.sA1 = sA1; p.sA2 = sA2; p.sW1 = sW1; p.sW2 = sW2; p.sC1 = sC1; p.sC2 = sC2; p.sR1 = sR1; p.sR2 = sR2;
    p.M = M; p.N = N; p.K = K; p.batch2 = batch2; p.rowgroup = rowgroup > 0 ? rowgroup : 1; p.act = act; p.alpha = alpha;
    const int64_t batch = (int64_t)batch1 * batch2;
    int cfg = g_force_cfg;
    if (cfg < 0) {
        // Measured on MI355X (scripts/gemm_bench.py): with f32 MFMA the matrix pipe is slow enough that the extra
        // operand traffic of small tiles is free, while more resident waves (3 WG/CU at 128x64, 5 at 64x64) and finer
        // tile quantisation over 256 CUs are not: 128x64 for the ViT GEMMs, 64x64 for short-K / small problems.
        const int64_t tmid = psam_cdiv(M, 128) * psam_cdiv(N, 64) * batch;
        cfg = (K <= 256 || M <= 64 || tmid < 512) ? 2 : 1;
    }
    if (act == 3) {
        PSAM_REQUIRE((N & 63) == 0 && !residual && !rowbias, PSAM_EINVAL, "psam_gemm_f32: SwiGLU epilogue needs N % 64 == 0, no residual/rowbias");
        if (cfg == 2) cfg = 1;  // needs paired accumulator tiles (TN even)
    }
    const int bm = cfg == 2 ? 64 : 128, bn = cfg == 0 ? 128 : 64;
    p.tiles_m = (int)psam_cdiv(M, bm);
    p.tiles_n = (int)psam_cdiv(N, bn);
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n), 1, (unsigned)batch);
    if (cfg == 0) hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 2, 2>), grid, dim3(256), 0, stream, p);       // 128x128
    else if (cfg == 1) hipLaunchKernelGGL((gemm_nt_kernel<4, 1, 1, 2>), grid, dim3(256), 0, stream, p);  // 128x64, wave = 32x64
    else hipLaunchKernelGGL((gemm_nt_kernel<2, 2, 1, 1>), grid, dim3(256), 0, stream, p);                // 64x64
    return psam_launch_status("psam_gemm_f32: launch failed");
}

// nn.Linear convenience: y[M,N] = act(x[M,K] @ W[N,K]^T + bias) (+ residual[M,N]); all row-major contiguous except
// explicit leading dimensions.
PSAM_API int32_t psam_linear(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* residual,
                             int64_t ldr, float* y, int64_t ldy, int32_t M, int32_t N, int32_t K, int32_t act, hipStream_t stream) {
    return psam_gemm_f32(x, ldx, 0, 0, W, ldw, 0, 0, y, ldy, 0, 0, bias, residual, ldr, 0, 0, nullptr, 0, 0, M, N, K, 1, 1, 1.0f, act,
                         stream);
}

// ------------------------------------------------------------------------------------------------
// Linear layer on a handful of rows (M <= 64: the decoder's 7 output tokens per prompt -- q/k/v/out projections and the token MLP of
// the two-way transformer, transformer.py:109-236): y = act(x W^T + b) + residual with exact fp32 products (v_mfma_f32_16x16x4_f32).
// The general kernel above needs ~10 us for such a launch (one or two workgroups walking K in eight dependent slabs); here one
// workgroup owns 16 output columns and all <= 64 rows (wave w: rows 16w .. 16w+15), so N / 16 workgroups run, and every lane fetches
// float4s of its x row and its W row for 128 k at a time, double-buffered -- the whole K = 256 of a projection is two load rounds.
// One float4 feeds four MFMAs: element e of lane group g is k-slot 16s + 4g + e for BOTH operands (a sum over k does not care about
// the order of k).
// ------------------------------------------------------------------------------------------------
typedef float sk_f32x4 __attribute__((ext_vector_type(4)));
constexpr int SK_CH = 8;      // float4 per operand per lane and round = 128 k
__global__ __launch_bounds__(256) void linear_skinny_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ W, int64_t ldw,
                                                            const float* __restrict__ bias, const float* __restrict__ res, int64_t ldr,
                                                            float* __restrict__ y, int64_t ldy, int M, int N, int K, int act) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int row = wave * 16 + r, col = blockIdx.x * 16 + r;
    const float* xp = x + (int64_t)(row < M ? row : 0) * ldx + 4 * g;
    const float* wp = W + (int64_t)(col < N ? col : 0) * ldw + 4 * g;
    const bool cok = col < N;
    sk_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    sk_f32x4 xa[SK_CH], wb[SK_CH], xn[SK_CH], wn[SK_CH];
    // Loads are UNCONDITIONAL (rows / columns past the end read a clamped row and are never stored; a 16-k slot past K -- K % 16 == 0: inside or outside
    // as a whole -- reads slot 0 and gets a zero weight): a load under a branch costs a full wait at the join, and the sixteen loads of a round became
    // sixteen round trips (round 5: linear_skinny 5.6 -> see profiles/r05/r05_click_kernels_*.txt).
    auto load = [&](sk_f32x4 (&xr)[SK_CH], sk_f32x4 (&wr)[SK_CH], int kc) {
#pragma unroll
        for (int s = 0; s < SK_CH; ++s) {
            const int k = kc + 16 * s, kk = k < K ? k : 0;
            xr[s] = *reinterpret_cast<const sk_f32x4*>(xp + kk);
            wr[s] = *reinterpret_cast<const sk_f32x4*>(wp + kk);
            if (k >= K) wr[s] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    load(xa, wb, 0);
    for (int kc = 0; kc < K; kc += 16 * SK_CH) {
        const bool more = kc + 16 * SK_CH < K;
        if (more) load(xn, wn, kc + 16 * SK_CH);
        __builtin_amdgcn_sched_barrier(0);      // the loads of a round are all in flight before the MFMAs start waiting
#pragma unroll
        for (int s = 0; s < SK_CH; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s][e], wb[s][e], acc, 0, 0, 0);
        if (more) {
#pragma unroll
            for (int s = 0; s < SK_CH; ++s) { xa[s] = xn[s]; wb[s] = wn[s]; }
        }
    }
    // D layout: lane holds rows 4g .. 4g+3 of the wave's 16, column r
    if (cok) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int orow = wave * 16 + 4 * g + v;
            if (orow < M) {
                float val = acc[v] + bv;
                if (act == 1) val = gelu_erf(val);
                else if (act == 2) val = fmaxf(val, 0.f);
                if (res) val += res[(int64_t)orow * ldr + col];
                y[(int64_t)orow * ldy + col] = val;
            }
        }
    }
}

// Up to three skinny Linears that share their input rows in ONE launch (grid.y = job): the q / k / v projections of a decoder attention
// (transformer.py:214-236), each job with its own weight, bias, output and -- optionally -- a second input added to x on load (q = k = queries +
// query_pe, v = queries: transformer.py:153-170), so that neither the positional add nor three separate launches exist.  Same arithmetic per
// output element as linear_skinny_kernel on the pre-added input (x + xadd is one fp32 add, as psam_add_bcast makes it): the same bits.
__global__ __launch_bounds__(256) void linear_skinny_multi_kernel(const psam_skinny_jobs_t jobs, int64_t ldx, int64_t ldxa, int64_t ldw, int M, int K) {
    const psam_skinny_job_t jb = jobs.job[blockIdx.y];
    const int N = jb.N;
    if ((int)blockIdx.x * 16 >= N) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int row = wave * 16 + r, col = blockIdx.x * 16 + r;
    const float* xp = jb.x + (int64_t)(row < M ? row : 0) * ldx + 4 * g;
    const float* xa = jb.xadd ? jb.xadd + (int64_t)(row < M ? row : 0) * ldxa + 4 * g : nullptr;
    const float* wp = jb.W + (int64_t)(col < N ? col : 0) * ldw + 4 * g;
    const bool cok = col < N;
    sk_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    sk_f32x4 xr[SK_CH], wr[SK_CH];
    for (int kc = 0; kc < K; kc += 16 * SK_CH) {
#pragma unroll
        for (int s = 0; s < SK_CH; ++s) {      // unconditional loads, see linear_skinny_kernel
            const int k = kc + 16 * s, kk = k < K ? k : 0;
            xr[s] = *reinterpret_cast<const sk_f32x4*>(xp + kk);
            wr[s] = *reinterpret_cast<const sk_f32x4*>(wp + kk);
            if (k >= K) wr[s] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (xa) {      // one uniform branch around the addend's loads, all issued before the first addition waits
            sk_f32x4 ad[SK_CH];
#pragma unroll
            for (int s = 0; s < SK_CH; ++s) { const int k = kc + 16 * s; ad[s] = *reinterpret_cast<const sk_f32x4*>(xa + (k < K ? k : 0)); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < SK_CH; ++s) xr[s] += ad[s];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < SK_CH; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[s][e], wr[s][e], acc, 0, 0, 0);
    }
    if (cok) {
        const float bv = jb.bias ? jb.bias[col] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int orow = wave * 16 + 4 * g + v;
            if (orow < M) {
                float val = acc[v] + bv;
                if (jb.act == 1) val = gelu_erf(val);
                else if (jb.act == 2) val = fmaxf(val, 0.f);
                jb.y[(int64_t)orow * jb.ldy + col] = val;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Several Linears over the same few hundred to few thousand input rows in ONE launch, exact fp32 products: the patch side of a decoder layer projects
// its 512 rows per cloud three times (k of tokens -> patches and q of patches -> tokens from keys + key_pe, v from keys: transformer.py:160-175;
// N = 128, K = 256).  As packed-operand GEMMs that was a packing pass and two or three launches of 8-10 us each (tiles of 128 rows: four to eight
// workgroups, all latency); here a workgroup owns 32 rows x 32 columns of one job (wave: one 16 x 16 tile, v_mfma_f32_16x16x4_f32 as in
// linear_skinny_kernel), grid = (rows / 32, columns / 32, jobs) -- 192 workgroups for the three projections of one cloud --, the positional addend
// (broadcast over `rep` row sets) is added while the rows are loaded, and K runs in double-buffered rounds of 128.
// ------------------------------------------------------------------------------------------------
template <bool FULL>      // FULL: K % 128 == 0 -- every load of a round is inside K, none sits under a condition (a load under a branch makes the compiler wait for it at the join)
__global__ __launch_bounds__(256) void linear_rows_multi_kernel(const psam_skinny_jobs_t jobs, int64_t ldx, int64_t ldxa, int rows_per_set, int rep, int64_t ldw,
                                                                int M, int K) {
    const psam_skinny_job_t jb = jobs.job[blockIdx.z];
    const int N = jb.N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * 32 + (wave & 1) * 16, col0 = blockIdx.y * 32 + (wave >> 1) * 16;
    if (row0 >= M || col0 >= N) return;      // wave-uniform
    const int row = row0 + r < M ? row0 + r : M - 1, colc = col0 + r < N ? col0 + r : N - 1;
    const float* xp = jb.x + (int64_t)row * ldx + 4 * g;
    const float* xa = jb.xadd ? jb.xadd + ((int64_t)(row / (rep * rows_per_set)) * rows_per_set + row % rows_per_set) * ldxa + 4 * g : nullptr;
    const float* wp = jb.W + (int64_t)colc * ldw + 4 * g;
    sk_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    sk_f32x4 xr[SK_CH], wr[SK_CH], xn[SK_CH], wn[SK_CH];
    auto load = [&](sk_f32x4 (&xv)[SK_CH], sk_f32x4 (&wv)[SK_CH], int kc) {
#pragma unroll
        for (int s = 0; s < SK_CH; ++s) {
            const int k = kc + 16 * s;
            const int kk = FULL || k < K ? k : 0;      // K % 16 == 0: a 16-k slot is inside or outside as a whole; outside: any address, the weight is zeroed
            xv[s] = *reinterpret_cast<const sk_f32x4*>(xp + kk);
            wv[s] = *reinterpret_cast<const sk_f32x4*>(wp + kk);
            if (!FULL && k >= K) wv[s] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (xa) {      // one uniform branch around the eight addend loads, all issued before the first addition waits
            sk_f32x4 ad[SK_CH];
#pragma unroll
            for (int s = 0; s < SK_CH; ++s) { const int k = kc + 16 * s; ad[s] = *reinterpret_cast<const sk_f32x4*>(xa + (FULL || k < K ? k : 0)); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < SK_CH; ++s) xv[s] += ad[s];
        }
    };
    load(xr, wr, 0);
    for (int kc = 0; kc < K; kc += 16 * SK_CH) {
        const bool more = kc + 16 * SK_CH < K;
        if (more) load(xn, wn, kc + 16 * SK_CH);
        __builtin_amdgcn_sched_barrier(0);      // the next round's loads are all in flight before this round's MFMAs start
#pragma unroll
        for (int s = 0; s < SK_CH; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[s][e], wr[s][e], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
#pragma unroll
            for (int s = 0; s < SK_CH; ++s) { xr[s] = xn[s]; wr[s] = wn[s]; }
        }
    }
    // D layout: lane holds rows 4g .. 4g+3 of the wave's 16, column r
    const int col = col0 + r;
    if (col < N) {
        const float bv = jb.bias ? jb.bias[col] : 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int orow = row0 + 4 * g + v;
            if (orow < M) {
                float val = acc[v] + bv;
                if (jb.act == 1) val = gelu_erf(val);
                else if (jb.act == 2) val = fmaxf(val, 0.f);
                jb.y[(int64_t)orow * jb.ldy + col] = val;
            }
        }
    }
}

// jobs->n Linears y_i [M, N_i] = act_i((x_i + xadd_i[broadcast]) W_i^T + bias_i) over the same M rows in one launch (any M; K % 16 == 0), exact fp32
// products.  xadd_i (optional) has rows_per_set-row sets, set (row / (rep * rows_per_set)) is added to row `row`: the decoder's key_pe, shared by the
// `rep` prompt sets of a cloud.
PSAM_API int32_t psam_linear_rows_multi(const psam_skinny_jobs_t* jobs, int64_t ldx, int64_t ldxadd, int32_t rows_per_set, int32_t rep, int64_t ldw, int64_t M, int32_t K,
                                        hipStream_t stream) {
    PSAM_REQUIRE(jobs && jobs->n >= 1 && jobs->n <= PSAM_SKINNY_MAX_JOBS, PSAM_EINVAL, "psam_linear_rows_multi: one to PSAM_SKINNY_MAX_JOBS jobs");
    PSAM_REQUIRE(M > 0 && M < ((int64_t)1 << 31) - 32 && K > 0 && (K & 15) == 0 && ((ldx | ldw | ldxadd) & 3) == 0 && rows_per_set > 0 && rep > 0, PSAM_EINVAL,
                 "psam_linear_rows_multi: need M > 0, K % 16 == 0, rows 16-byte aligned");
    int nmax = 0;
    for (int i = 0; i < jobs->n; ++i) {
        const psam_skinny_job_t& j = jobs->job[i];
        PSAM_REQUIRE(j.x && j.W && j.y && j.N > 0 && j.act >= 0 && j.act <= 2, PSAM_EINVAL, "psam_linear_rows_multi: bad job");
        PSAM_REQUIRE((((uintptr_t)j.x | (uintptr_t)j.W | (uintptr_t)j.xadd) & 15) == 0, PSAM_EALIGN, "psam_linear_rows_multi: rows must be 16-byte aligned");
        nmax = j.N > nmax ? j.N : nmax;
    }
    const dim3 grid((unsigned)psam_cdiv(M, 32), (unsigned)psam_cdiv(nmax, 32), (unsigned)jobs->n);
    if ((K & 127) == 0) hipLaunchKernelGGL(linear_rows_multi_kernel<true>, grid, dim3(256), 0, stream, *jobs, ldx, ldxadd, rows_per_set, rep, ldw, (int)M, K);
    else hipLaunchKernelGGL(linear_rows_multi_kernel<false>, grid, dim3(256), 0, stream, *jobs, ldx, ldxadd, rows_per_set, rep, ldw, (int)M, K);
    return psam_launch_status("psam_linear_rows_multi: launch failed");
}

PSAM_API int32_t psam_linear_skinny_multi(const psam_skinny_jobs_t* jobs, int64_t ldx, int64_t ldxadd, int64_t ldw, int32_t M, int32_t K, hipStream_t stream) {
    PSAM_REQUIRE(jobs && jobs->n >= 1 && jobs->n <= PSAM_SKINNY_MAX_JOBS, PSAM_EINVAL, "psam_linear_skinny_multi: one to PSAM_SKINNY_MAX_JOBS jobs");
    PSAM_REQUIRE(M > 0 && M <= 64 && K > 0 && (K & 15) == 0 && ((ldx | ldw | ldxadd) & 3) == 0, PSAM_EINVAL, "psam_linear_skinny_multi: need 0 < M <= 64, K % 16 == 0, rows 16-byte aligned");
    int nmax = 0;
    for (int i = 0; i < jobs->n; ++i) {
        const psam_skinny_job_t& j = jobs->job[i];
        PSAM_REQUIRE(j.x && j.W && j.y && j.N > 0 && j.act >= 0 && j.act <= 2, PSAM_EINVAL, "psam_linear_skinny_multi: bad job");
        PSAM_REQUIRE((((uintptr_t)j.x | (uintptr_t)j.W | (uintptr_t)j.xadd) & 15) == 0, PSAM_EALIGN, "psam_linear_skinny_multi: rows must be 16-byte aligned");
        nmax = j.N > nmax ? j.N : nmax;
    }
    hipLaunchKernelGGL(linear_skinny_multi_kernel, dim3((unsigned)psam_cdiv(nmax, 16), (unsigned)jobs->n), dim3(256), 0, stream, *jobs, ldx, ldxadd, ldw, M, K);
    return psam_launch_status("psam_linear_skinny_multi: launch failed");
}

// ------------------------------------------------------------------------------------------------
// Skinny Linear + residual + LayerNorm in ONE launch: the token side of the two-way decoder ends every attention and its MLP with
// `queries = norm(queries + out_proj(a))` / `norm3(queries + lin2(relu(lin1 queries)))` (transformer.py:153-176) on <= 64 rows of 256 columns -- two
// launches of ~5 us each whose second one only waits for the first, and for lin2 (K = 2048) a Linear whose 16 workgroups walk K in 16 dependent
// load rounds (23 us).  Here grid = (16 column blocks, ksplit K ranges): every workgroup does the arithmetic of linear_skinny_kernel on its K range
// (one or two load rounds), parks its 16 columns in `tmp` with device-coherent stores (sc1: written through the XCD's L2), counts itself in, and the
// LAST workgroup to arrive adds the ksplit partials in fixed order, the bias and the residual and normalises the rows (a wave per row, float4 per
// lane: the arithmetic of layernorm_v4_kernel<1>) -- the in-kernel fix-up of the split-K GEMM (gemm_f16x3p.hip) applied to a row operation.  The
// counter is left at zero for the next launch.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_skinny_ln_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ W, int64_t ldw,
                                                               const float* __restrict__ bias, const float* __restrict__ res, int64_t ldr,
                                                               const float* __restrict__ lnw, const float* __restrict__ lnb, float eps, float* __restrict__ tmp,
                                                               float* __restrict__ y, int64_t ldy, int M, int K, int kper, int* __restrict__ counter) {
    constexpr int N = 256, SC1 = 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int row = wave * 16 + r, col = blockIdx.x * 16 + r;
    const int k0 = blockIdx.y * kper, k1 = k0 + kper < K ? k0 + kper : K;
    const float* xp = x + (int64_t)(row < M ? row : 0) * ldx + 4 * g;
    const float* wp = W + (int64_t)col * ldw + 4 * g;
    sk_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    sk_f32x4 xr[SK_CH], wr[SK_CH];
    for (int kc = k0; kc < k1; kc += 16 * SK_CH) {
#pragma unroll
        for (int s = 0; s < SK_CH; ++s) {      // unconditional loads, see linear_skinny_kernel
            const int k = kc + 16 * s, kk = k < k1 ? k : k0;
            xr[s] = *reinterpret_cast<const sk_f32x4*>(xp + kk);
            wr[s] = *reinterpret_cast<const sk_f32x4*>(wp + kk);
            if (k >= k1) wr[s] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < SK_CH; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[s][e], wr[s][e], acc, 0, 0, 0);
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)tmp, 0, 0x7fffffff, 0x00020000);
    const int plane = M * N;      // floats per K range
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int orow = wave * 16 + 4 * g + v;
        const float av = acc[v];      // (into a float first: __builtin_bit_cast applied to a vector ELEMENT expression read element 0 every time)
        if (orow < M) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, av), rs, ((int)blockIdx.y * plane + orow * N + col) * 4, 0, SC1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's columns are acknowledged by the memory side
    __shared__ int flag;
    __syncthreads();
    if (threadIdx.x == 0) flag = (int)__hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (flag != (int)(gridDim.x * gridDim.y) - 1) return;
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    typedef unsigned sk_u32x4 __attribute__((ext_vector_type(4)));
    const float inv = 1.0f / (float)N;
    const sk_f32x4 w4 = *reinterpret_cast<const sk_f32x4*>(lnw + lane * 4), b4 = *reinterpret_cast<const sk_f32x4*>(lnb + lane * 4);
    const sk_f32x4 bv = bias ? *reinterpret_cast<const sk_f32x4*>(bias + lane * 4) : sk_f32x4{0.f, 0.f, 0.f, 0.f};
    const int ksplit = (int)gridDim.y;
    for (int orow = wave; orow < M; orow += 4) {
        sk_f32x4 v = __builtin_bit_cast(sk_f32x4, (sk_u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, (orow * N + lane * 4) * 4, 0, SC1));
        for (int ks = 1; ks < ksplit; ++ks)
            v += __builtin_bit_cast(sk_f32x4, (sk_u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, (ks * plane + orow * N + lane * 4) * 4, 0, SC1));
        v += bv;
        if (res) v += *reinterpret_cast<const sk_f32x4*>(res + (int64_t)orow * ldr + lane * 4);
        const float s = (v[0] + v[1]) + (v[2] + v[3]);
        const float mean = wave_sum(s) * inv;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv + eps);
        const sk_f32x4 o = (v - mean) * rstd * w4 + b4;
        *reinterpret_cast<sk_f32x4*>(y + (int64_t)orow * ldy + lane * 4) = o;
    }
}

// K ranges of psam_linear_skinny_ln: one per 256 k (two load rounds), at most 8
static int skinny_ln_ksplit(int K) { const int s = (K + 255) / 256; return s < 1 ? 1 : (s > 8 ? 8 : s); }
PSAM_API size_t psam_linear_skinny_ln_tmp_floats(int32_t M, int32_t K) { return M > 0 && K > 0 ? (size_t)skinny_ln_ksplit(K) * (size_t)M * 256 : 0; }

// y [M, 256] = LayerNorm_256(x [M, K] W [256, K]^T + bias + residual) * ln_w + ln_b for M <= 64 rows (residual optional, may alias y), one launch.
// tmp: psam_linear_skinny_ln_tmp_floats(M, K) floats of scratch.  counters: the caller's arrival-counter block (PSAM_COUNTER_BYTES,
// include/pointsam_hip.h; word PSAM_CNT_ROW is used).
PSAM_API int32_t psam_linear_skinny_ln(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* residual, int64_t ldr,
                                       const float* ln_w, const float* ln_b, float eps, float* tmp, float* y, int64_t ldy, int32_t M, int32_t N, int32_t K,
                                       int32_t* counters, hipStream_t stream) {
    PSAM_REQUIRE(x && W && y && ln_w && ln_b && tmp && counters, PSAM_EINVAL, "psam_linear_skinny_ln: null pointer (the arrival-counter block is required)");
    PSAM_REQUIRE(M > 0 && M <= 64 && N == 256 && K > 0 && (K & 15) == 0, PSAM_EINVAL, "psam_linear_skinny_ln: need 0 < M <= 64, N == 256, K % 16 == 0");
    PSAM_REQUIRE(((ldx | ldw | ldy | (residual ? ldr : 0)) & 3) == 0 &&
                     (((uintptr_t)x | (uintptr_t)W | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)ln_w | (uintptr_t)ln_b | (uintptr_t)tmp | (uintptr_t)bias) & 15) == 0,
                 PSAM_EALIGN, "psam_linear_skinny_ln: rows must be 16-byte aligned");
    int* counter = counters + PSAM_CNT_ROW;
    const int ksplit = skinny_ln_ksplit(K);
    const int kper = ((K + ksplit - 1) / ksplit + 15) & ~15;
    hipLaunchKernelGGL(linear_skinny_ln_kernel, dim3(N / 16, ksplit), dim3(256), 0, stream, x, ldx, W, ldw, bias, residual, ldr, ln_w, ln_b, eps, tmp, y, ldy, M, K, kper,
                       counter);
    return psam_launch_status("psam_linear_skinny_ln: launch failed");
}

// ------------------------------------------------------------------------------------------------
// Linear (K small) + residual + LayerNorm over MANY rows of 256 columns in one launch: `keys = norm4(keys + out_proj(attn))` of the decoder's patch
// side (transformer.py:170-175: 512 rows per cloud, K = 128) was pack + packed GEMM + LayerNorm = three launches of 5 + 8 + 5 us.  Here a workgroup
// owns 16 whole rows: wave w of eight computes columns 32 w .. 32 w + 31 (two 16 x 16 tiles, v_mfma_f32_16x16x4_f32: exact fp32 products as in
// linear_skinny_kernel), the tile is staged in LDS and every wave normalises two of the rows (float4 per lane, the arithmetic of
// layernorm_v4_kernel<1>).  The weight (K * 1 KiB) is read once per 16 rows, from L2.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void linear_ln256_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ W, int64_t ldw,
                                                           const float* __restrict__ bias, const float* __restrict__ res, int64_t ldr,
                                                           const float* __restrict__ lnw, const float* __restrict__ lnb, float eps, float* __restrict__ y,
                                                           int64_t ldy, int M, int K) {
    constexpr int N = 256, LD = N + 4;
    __shared__ __attribute__((aligned(16))) float s_t[16 * LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;      // 8 waves: wave w owns columns 32 w .. 32 w + 31
    const int row0 = blockIdx.x * 16, row = row0 + r;
    const float* xp = x + (int64_t)(row < M ? row : M - 1) * ldx + 4 * g;
    const float* wp = W + (int64_t)(wave * 32 + r) * ldw + 4 * g;
    sk_f32x4 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) acc[t] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kc = 0; kc < K; kc += 16 * SK_CH) {
        sk_f32x4 xr[SK_CH], wr[2][SK_CH];
#pragma unroll
        for (int s = 0; s < SK_CH; ++s) {
            const int k = kc + 16 * s;
            const int kk = k < K ? k : 0;      // unconditional loads (K % 16 == 0: a 16-k slot is inside or outside as a whole; outside: the weight is zeroed)
            xr[s] = *reinterpret_cast<const sk_f32x4*>(xp + kk);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                wr[t][s] = *reinterpret_cast<const sk_f32x4*>(wp + (int64_t)t * 16 * ldw + kk);
                if (k >= K) wr[t][s] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __builtin_amdgcn_sched_barrier(0);      // all 24 loads of the round are issued before the first MFMA waits (left alone, the scheduler interleaves three loads, a full wait, eight MFMAs: eight round trips)
#pragma unroll
        for (int s = 0; s < SK_CH; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[s][e], wr[t][s][e], acc[t], 0, 0, 0);
    }
    // D layout: lane holds rows 4g .. 4g+3, column r of each 16-column tile
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) s_t[(4 * g + v) * LD + wave * 32 + t * 16 + r] = acc[t][v];
    __syncthreads();
    const float inv = 1.0f / (float)N;
    const sk_f32x4 w4 = *reinterpret_cast<const sk_f32x4*>(lnw + lane * 4), b4 = *reinterpret_cast<const sk_f32x4*>(lnb + lane * 4);
    const sk_f32x4 bv = bias ? *reinterpret_cast<const sk_f32x4*>(bias + lane * 4) : sk_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = wave * 2 + i, orow = row0 + lr;
        if (orow >= M) break;      // wave-uniform
        sk_f32x4 v = *reinterpret_cast<const sk_f32x4*>(s_t + lr * LD + lane * 4);
        v += bv;
        if (res) v += *reinterpret_cast<const sk_f32x4*>(res + (int64_t)orow * ldr + lane * 4);
        const float sm = (v[0] + v[1]) + (v[2] + v[3]);
        const float mean = wave_sum(sm) * inv;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv + eps);
        const sk_f32x4 o = (v - mean) * rstd * w4 + b4;
        *reinterpret_cast<sk_f32x4*>(y + (int64_t)orow * ldy + lane * 4) = o;
    }
}

// y [M, 256] = LayerNorm_256(x [M, K] W [256, K]^T + bias + residual) * ln_w + ln_b for any M (residual optional, may alias y), K % 16 == 0, K <= 512.
PSAM_API int32_t psam_linear_ln256(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* residual, int64_t ldr,
                                   const float* ln_w, const float* ln_b, float eps, float* y, int64_t ldy, int64_t M, int32_t N, int32_t K, hipStream_t stream) {
    PSAM_REQUIRE(x && W && y && ln_w && ln_b, PSAM_EINVAL, "psam_linear_ln256: null pointer");
    PSAM_REQUIRE(M > 0 && M < ((int64_t)1 << 31) - 16 && N == 256 && K > 0 && K <= 512 && (K & 15) == 0, PSAM_EINVAL, "psam_linear_ln256: need N == 256, K % 16 == 0, K <= 512");
    PSAM_REQUIRE(((ldx | ldw | ldy | (residual ? ldr : 0)) & 3) == 0 &&
                     (((uintptr_t)x | (uintptr_t)W | (uintptr_t)y | (uintptr_t)residual | (uintptr_t)ln_w | (uintptr_t)ln_b | (uintptr_t)bias) & 15) == 0,
                 PSAM_EALIGN, "psam_linear_ln256: rows must be 16-byte aligned");
    hipLaunchKernelGGL(linear_ln256_kernel, dim3((unsigned)psam_cdiv(M, 16)), dim3(512), 0, stream, x, ldx, W, ldw, bias, residual, ldr, ln_w, ln_b, eps, y, ldy, (int)M, K);
    return psam_launch_status("psam_linear_ln256: launch failed");
}

// y [M, N] = act(x [M, K] W [N, K]^T + bias) + residual for M <= 64, K % 16 == 0, 16-byte aligned rows (ld % 4 == 0).
PSAM_API int32_t psam_linear_skinny(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* residual, int64_t ldr,
                                    float* y, int64_t ldy, int32_t M, int32_t N, int32_t K, int32_t act, hipStream_t stream) {
    PSAM_REQUIRE(x && W && y, PSAM_EINVAL, "psam_linear_skinny: null pointer");
    PSAM_REQUIRE(M > 0 && M <= 64 && N > 0 && K > 0 && (K & 15) == 0 && act >= 0 && act <= 2, PSAM_EINVAL,
                 "psam_linear_skinny: need 0 < M <= 64, K % 16 == 0, act in {none, gelu, relu}");
    PSAM_REQUIRE(((ldx | ldw) & 3) == 0 && (((uintptr_t)x | (uintptr_t)W) & 15) == 0, PSAM_EALIGN, "psam_linear_skinny: rows must be 16-byte aligned");
    hipLaunchKernelGGL(linear_skinny_kernel, dim3((unsigned)psam_cdiv(N, 16)), dim3(256), 0, stream, x, ldx, W, ldw, bias, residual, ldr, y, ldy, M, N, K, act);
    return psam_launch_status("psam_linear_skinny: launch failed");
}
