// bf16x6 GEMM with PRE-PACKED WEIGHTS: y = act(x @ W^T + bias ...) where W is a static nn.Linear weight.
//
// Same arithmetic as gemm_split.hip (exact 3-way bf16 split of both operands, 6 partial products, fp32 accumulation), but
// the weight is split ONCE at model load (psam_pack_weight_bf16x3) and stored in MFMA fragment order: for every 32-row
// tile nt, k16-step ks and plane q (hi/mid/lo) one 1-KiB block holds, lane by lane, the 8 bf16 a lane feeds to
// v_mfma_f32_32x32x16_bf16 (lane = 32*h + r: row nt*32+r, k = ks*16 + 8h .. +7).  A wave then loads its W fragments with
// fully coalesced 16-byte-per-lane loads straight into VGPRs: the weight never touches LDS and costs no split arithmetic.
// Measured motivation (SQ counters on the symmetric kernel): the LDS array was 61-93 % busy, mostly with the slow
// ds_write path (~85 B/clk) of 48 KiB per slab; here only the activation operand goes through LDS (24 KiB/slab), which
// also leaves room to double-buffer it: ONE barrier per 32-k slab, and the fragments of the next slab are read behind it
// while the second half of the current slab's MFMAs runs.
//
// Activations stay plain fp32 in HBM and are split on the fly as before.  Tile 128x128, 2x2 waves of 64x64.
#include "common.h"
#include "gemm_epilogue.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct PackWArgs {
    const float* A; const unsigned char* Wpk; float* C;
    const float* bias; const float* residual; const float* rowbias;
    int64_t lda, ldc, ldr, ldrb;
    int M, N, K, KS, NT, rowgroup, act;   // KS = k16 steps of the packed weight (K rounded up to 32, / 16); NT = ceil(N/32)
    float alpha;
    int tiles_m, tiles_n;
};

constexpr int PW_ROWB = 64;  // bytes per LDS row of one bf16 plane (32 k)

__device__ __forceinline__ void pw_split3(const f32x2 x, unsigned& hi, unsigned& mid, unsigned& lo) {
    const bf16x2 h = __builtin_convertvector(x, bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    mid = __builtin_bit_cast(unsigned, m);
    lo = __builtin_bit_cast(unsigned, l);
}

// ---------------------------------------------------------------------------------------------- weight packing
__global__ void pack_weight_kernel(const float* __restrict__ W, int64_t ldw, int N, int K, int KS, int64_t nblocks, unsigned char* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t blk = t >> 6;
    if (blk >= nblocks) return;
    const int lane = (int)(t & 63), r = lane & 31, h = lane >> 5;
    const int nt = (int)(blk / KS), ks = (int)(blk % KS);
    const int n = nt * 32 + r, k0 = ks * 16 + 8 * h;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (n < N && k0 + e < K) ? W[(int64_t)n * ldw + k0 + e] : 0.f;
    u32x4 hi, mid, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned a, b, c;
        pw_split3(f32x2{v[2 * e], v[2 * e + 1]}, a, b, c);
        hi[e] = a; mid[e] = b; lo[e] = c;
    }
    unsigned char* o = out + blk * 3 * 1024 + lane * 16;
    *reinterpret_cast<u32x4*>(o) = hi;
    *reinterpret_cast<u32x4*>(o + 1024) = mid;
    *reinterpret_cast<u32x4*>(o + 2048) = lo;
}

PSAM_API size_t psam_packed_weight_bytes(int32_t N, int32_t K) {
    if (N <= 0 || K <= 0) return 0;
    return (size_t)psam_cdiv(N, 32) * (size_t)(psam_cdiv(K, 32) * 2) * 3 * 1024;
}

// W [N,K] fp32 (nn.Linear layout, leading dimension ldw) -> fragment-ordered bf16x3 planes (psam_packed_weight_bytes bytes)
PSAM_API int32_t psam_pack_weight_bf16x3(const float* W, int64_t ldw, int32_t N, int32_t K, void* out, hipStream_t stream) {
    PSAM_REQUIRE(W && out && N > 0 && K > 0 && ldw >= K, PSAM_EINVAL, "psam_pack_weight_bf16x3: bad argument");
    PSAM_REQUIRE(((uintptr_t)out & 15) == 0, PSAM_EALIGN, "psam_pack_weight_bf16x3: output must be 16-byte aligned");
    const int KS = (int)psam_cdiv(K, 32) * 2;
    const int64_t nblocks = psam_cdiv(N, 32) * (int64_t)KS;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)psam_cdiv(nblocks * 64, 256)), dim3(256), 0, stream, W, ldw, N, K, KS, nblocks,
                       (unsigned char*)out);
    return psam_launch_status("psam_pack_weight_bf16x3: launch failed");
}

// ---------------------------------------------------------------------------------------------- GEMM
template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_bf16x6_pw_kernel(const PackWArgs p) {
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
    constexpr int A_F4 = BM * 8 / 256;
    constexpr int PLANE_A = BM * PW_ROWB;
    __shared__ __attribute__((aligned(16))) unsigned char sA[2][3 * PLANE_A];   // double-buffered activation planes

    const int ntiles = p.tiles_m * p.tiles_n;
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- activation operand: global fp32 -> registers -> split -> LDS planes
    const int lr = tid >> 3, lc4 = tid & 7;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)((((int64_t)p.M - 1) * p.lda + p.K) * 4), 0x00020000);
    const int voA = (int)(((int64_t)(m0 + lr) * p.lda + lc4 * 4) * 4);
    const int stepA = (int)(32 * p.lda * 4);
    constexpr int OOB = 0x7ffffff0;
    f32x4 ra[A_F4];
    auto load_a = [&](int k0) {
        const bool kok = k0 + lc4 * 4 < p.K;
#pragma unroll
        for (int i = 0; i < A_F4; ++i)
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (kok && m0 + i * 32 + lr < p.M) ? voA + i * stepA : OOB, k0 * 4, 0));
    };
    const int st_off = lr * PW_ROWB + ((((lc4 >> 1) ^ ((lr >> 2) & 3)) << 4) | ((lc4 & 1) << 3));
    u32x2 spa[A_F4][3];
    auto split_a = [&]() {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            unsigned h0, m0_, l0, h1, m1, l1;
            pw_split3(f32x2{ra[i][0], ra[i][1]}, h0, m0_, l0);
            pw_split3(f32x2{ra[i][2], ra[i][3]}, h1, m1, l1);
            spa[i][0] = u32x2{h0, h1}; spa[i][1] = u32x2{m0_, m1}; spa[i][2] = u32x2{l0, l1};
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < A_F4; ++i) *reinterpret_cast<u32x2*>(&sA[buf][q * PLANE_A + i * 32 * PW_ROWB + st_off]) = spa[i][q];
    };
    int frag_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) frag_off[s] = (wm * TM * 32 + r32) * PW_ROWB + (((2 * s + h) ^ ((r32 >> 2) & 3)) << 4);
    auto load_af = [&](int buf, int s, bf16x8 (&af)[TM][3]) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i][q] = *reinterpret_cast<const bf16x8*>(&sA[buf][q * PLANE_A + i * 32 * PW_ROWB + frag_off[s]]);
    };

    // ---- weight operand: packed fragment blocks, global -> VGPR
    const int64_t wbytes = (int64_t)p.NT * p.KS * 3 * 1024;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wpk, 0, (int)wbytes, 0x00020000);
    const int nt0 = n0 / 32 + wn * TN;
    auto load_wf = [&](int ks, bf16x8 (&wf)[TN][3]) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const bool ok = (nt0 + j < p.NT) && (ks < p.KS);
            const int base = ok ? ((nt0 + j) * p.KS + ks) * 3072 + lane * 16 : OOB;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                wf[j][q] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, base, ok ? q * 1024 : 0, 0));
        }
    };

#define PW_TERM(AF, WF, PA, PW_)                                                                                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j)               \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[i][PA], WF[j][PW_], acc[i][j], 0, 0, 0);
#define PW_STEP(AF, WF) PW_TERM(AF, WF, 0, 2) PW_TERM(AF, WF, 2, 0) PW_TERM(AF, WF, 1, 1) PW_TERM(AF, WF, 0, 1) PW_TERM(AF, WF, 1, 0) PW_TERM(AF, WF, 0, 0)

    const int nslabs = (p.K + 31) / 32;
    bf16x8 af0[TM][3], af1[TM][3], wfa[TN][3], wfb[TN][3];
    // prologue: slab 0 into LDS[0]; slab 1 split in registers; slab 2 in flight; W fragments of steps 0 and 1 in flight
    load_a(0);
    load_wf(0, wfa);
    load_wf(1, wfb);
    split_a();
    load_a(32);
    store_a(0);
    split_a();       // slab 1 (waits for its load)
    load_a(64);
    __syncthreads();
    load_af(0, 0, af0);
    for (int t = 0; t < nslabs; ++t) {
        const int cur = t & 1, nxt = cur ^ 1;
        store_a(nxt);                       // slab t+1 planes (buffer nxt was last read during slab t-1, before its barrier)
        load_af(cur, 1, af1);               // fragments of step 1 of this slab
        PW_STEP(af0, wfa)                   // step 0
        load_wf(2 * t + 2, wfa);            // W fragments of the next slab's step 0 (zeros past the end)
        __syncthreads();                    // writes of nxt visible; every wave holds its step-1 fragments of cur
        load_af(nxt, 0, af0);               // next slab's step-0 fragments, hidden behind step 1
        PW_STEP(af1, wfb)                   // step 1
        load_wf(2 * t + 3, wfb);
        split_a();                          // slab t+2 -> split registers
        load_a((t + 3) * 32);               // slab t+3 in flight
    }
#undef PW_STEP
#undef PW_TERM

    // ---- epilogue: LDS transpose -> row-contiguous float4 stores (gemm_epilogue.h)
    static_assert(4 * gemm_epilogue_lds_floats_per_wave<TN>() * 4 <= 2 * 3 * PLANE_A, "epilogue staging fits the activation LDS");
    __syncthreads();   // every wave is done reading activation fragments
    gemm_store_tile<TM, TN>(p, acc, reinterpret_cast<float*>(&sA[0][0]) + wave * gemm_epilogue_lds_floats_per_wave<TN>(), m0 + wm * TM * 32,
                            n0 + wn * TN * 32, lane, p.C, p.residual);
}

static int g_pw_cfg = -1;  // tuning hook: 0 = 128x128, 1 = 128x64, -1 = auto
PSAM_API void psam_gemm_bf16x6_pw_force_config(int32_t cfg) { g_pw_cfg = cfg; }

// C = act(alpha * A @ W^T + bias + rowbias[row/rowgroup]) + residual with W given as psam_pack_weight_bf16x3(W[N,K]).
PSAM_API int32_t psam_gemm_bf16x6_pw(const float* A, int64_t lda, const void* Wpk, float* C, int64_t ldc, const float* bias,
                                     const float* residual, int64_t ldr, const float* rowbias, int64_t ldrb, int32_t rowgroup, int32_t M,
                                     int32_t N, int32_t K, float alpha, int32_t act, hipStream_t stream) {
    PSAM_REQUIRE(A && Wpk && C, PSAM_EINVAL, "psam_gemm_bf16x6_pw: null pointer");
    PSAM_REQUIRE(M > 0 && N > 0 && K > 0, PSAM_EINVAL, "psam_gemm_bf16x6_pw: bad shape");
    PSAM_REQUIRE(act >= 0 && act <= 3, PSAM_EINVAL, "psam_gemm_bf16x6_pw: bad activation code");
    PSAM_REQUIRE(!rowbias || rowgroup > 0, PSAM_EINVAL, "psam_gemm_bf16x6_pw: rowbias needs rowgroup > 0");
    PSAM_REQUIRE((K & 3) == 0 && (lda & 3) == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)Wpk & 15) == 0, PSAM_EALIGN,
                 "psam_gemm_bf16x6_pw: K, lda multiples of 4; A and the packed weight 16-byte aligned");
    PSAM_REQUIRE(((int64_t)M - 1) * lda + K < ((int64_t)1 << 29) - 8 && psam_packed_weight_bytes(N, K) < ((size_t)1 << 31) - 16, PSAM_EINVAL,
                 "psam_gemm_bf16x6_pw: operands must span < 2 GiB (32-bit buffer offsets)");
    PSAM_REQUIRE(act != 3 || ((N & 63) == 0 && !residual && !rowbias), PSAM_EINVAL,
                 "psam_gemm_bf16x6_pw: SwiGLU epilogue needs N % 64 == 0, no residual/rowbias");
    PackWArgs p;
    p.A = A; p.Wpk = (const unsigned char*)Wpk; p.C = C; p.bias = bias; p.residual = residual; p.rowbias = rowbias;
    p.lda = lda; p.ldc = ldc; p.ldr = ldr; p.ldrb = ldrb;
    p.M = M; p.N = N; p.K = K; p.KS = (int)psam_cdiv(K, 32) * 2; p.NT = (int)psam_cdiv(N, 32);
    p.rowgroup = rowgroup > 0 ? rowgroup : 1; p.act = act; p.alpha = alpha;
    int cfg = g_pw_cfg;
    if (cfg < 0) cfg = (psam_cdiv(M, 128) * psam_cdiv(N, 128) >= 512) ? 0 : 1;
    if (act == 3) cfg = 0;  // the gate pairs two accumulator tiles of one wave (TN = 2)
    const int bn = cfg == 0 ? 128 : 64;
    p.tiles_m = (int)psam_cdiv(M, 128);
    p.tiles_n = (int)psam_cdiv(N, bn);
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n));
    if (cfg == 0) hipLaunchKernelGGL((gemm_bf16x6_pw_kernel<2, 2>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gemm_bf16x6_pw_kernel<2, 1>), grid, dim3(256), 0, stream, p);
    return psam_launch_status("psam_gemm_bf16x6_pw: launch failed");
}
