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
// through LDS (row stride 36 floats: conflict-free ds_read_b128 for the 16-lane groups), double-buffered with the
// next slab prefetched into registers while the current one feeds the matrix pipe.  One ds_read_b128 per operand
// row feeds 4 MFMAs: lanes 0-31 take k = 8s..8s+3, lanes 32-63 take k = 8s+4..8s+7 (A and W use the same
// permutation of k, so the sum over k is unchanged).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
    const float* A; const float* W; float* C;
    const float* bias;      // [N] or null
    const float* residual;  // same indexing as C (ldr, batch strides sR*) or null; may alias C
    const float* rowbias;   // [ceil(M/rowgroup), ldrb] or null (per-group broadcast row, e.g. the max-pooled half of conv2.0)
    int64_t lda, ldw, ldc, ldr, ldrb;
    int64_t sA1, sA2, sW1, sW2, sC1, sC2, sR1, sR2;  // batch strides (elements): z1 = z / batch2, z2 = z % batch2
    int M, N, K, batch2, rowgroup, act;               // act: 0 none, 1 GELU(erf), 2 ReLU
    float alpha;
    int tiles_m, tiles_n;
};

constexpr int GEMM_BK = 32;
constexpr int GEMM_LD = GEMM_BK + 4;

template <int TM, int TN, int ABL>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs p) {
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
    constexpr int A_F4 = BM * (GEMM_BK / 4) / 256;  // float4 loads per thread per slab
    constexpr int W_F4 = BN * (GEMM_BK / 4) / 256;
    __shared__ __attribute__((aligned(16))) float sA[2][BM * GEMM_LD];
    __shared__ __attribute__((aligned(16))) float sW[2][BN * GEMM_LD];

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
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[A_F4], rw[W_F4];
    const int lr = tid >> 3, lc = (tid & 7) * 4;  // this thread's (row, k) inside a 32-row stripe of a slab

    auto load_slab = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int row = m0 + i * 32 + lr, k = k0 + lc;
            ra[i] = (row < p.M && k < p.K) ? *reinterpret_cast<const f32x4*>(A + (int64_t)row * p.lda + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < W_F4; ++i) {
            const int row = n0 + i * 32 + lr, k = k0 + lc;
            rw[i] = (row < p.N && k < p.K) ? *reinterpret_cast<const f32x4*>(W + (int64_t)row * p.ldw + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_slab = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) *reinterpret_cast<f32x4*>(&sA[buf][(i * 32 + lr) * GEMM_LD + lc]) = ra[i];
#pragma unroll
        for (int i = 0; i < W_F4; ++i) *reinterpret_cast<f32x4*>(&sW[buf][(i * 32 + lr) * GEMM_LD + lc]) = rw[i];
    };

    const int nslabs = (p.K + GEMM_BK - 1) / GEMM_BK;
    load_slab(0);
    store_slab(0);
    __syncthreads();
    for (int t = 0; t < nslabs; ++t) {
        const int buf = (ABL == 2 || ABL == 3) ? 0 : (t & 1);
        if (ABL != 1 && ABL != 3) { if (t + 1 < nslabs) load_slab((t + 1) * GEMM_BK); }
        const float* a_base = &sA[buf][(wm * TM * 32 + r32) * GEMM_LD + h * 4];
        const float* w_base = &sW[buf][(wn * TN * 32 + r32) * GEMM_LD + h * 4];
#pragma unroll
        for (int s = 0; s < GEMM_BK / 8; ++s) {
            f32x4 af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * GEMM_LD + s * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *reinterpret_cast<const f32x4*>(w_base + j * 32 * GEMM_LD + s * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][q], wf[j][q], acc[i][j], 0, 0, 0);
        }
        if (ABL == 0 || ABL == 1) { if (t + 1 < nslabs) store_slab(buf ^ 1); __syncthreads(); }
        if (ABL == 2) { asm volatile("" :: "v"(ra[0][0]), "v"(rw[0][0])); }
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* C = p.C + z1 * p.sC1 + z2 * p.sC2;
    const float* R = p.residual ? p.residual + z1 * p.sR1 + z2 * p.sR2 : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + r32;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= p.M) continue;
                float v = acc[i][j][r] * p.alpha + bv;
                if (p.rowbias) v += p.rowbias[(int64_t)(row / p.rowgroup) * p.ldrb + col];
                if (p.act == 1) v = gelu_erf(v);
                else if (p.act == 2) v = fmaxf(v, 0.f);
                if (R) v += R[(int64_t)row * p.ldr + col];
                C[(int64_t)row * p.ldc + col] = v;
            }
        }
}


extern "C" __attribute__((visibility("default"))) int gemm_ablate(const float* A, const float* W, float* C, const float* bias, int M, int N, int K, int abl, hipStream_t stream) {
    GemmArgs p; p.A=A; p.W=W; p.C=C; p.bias=bias; p.residual=nullptr; p.rowbias=nullptr; p.lda=K; p.ldw=K; p.ldc=N; p.ldr=0; p.ldrb=0;
    p.sA1=p.sA2=p.sW1=p.sW2=p.sC1=p.sC2=p.sR1=p.sR2=0; p.M=M; p.N=N; p.K=K; p.batch2=1; p.rowgroup=1; p.act=0; p.alpha=1.f;
    p.tiles_m=(M+127)/128; p.tiles_n=(N+127)/128;
    dim3 grid(p.tiles_m*p.tiles_n,1,1);
    switch(abl){
      case 0: hipLaunchKernelGGL((gemm_nt_kernel<2,2,0>), grid, dim3(256), 0, stream, p); break;
      case 1: hipLaunchKernelGGL((gemm_nt_kernel<2,2,1>), grid, dim3(256), 0, stream, p); break;
      case 2: hipLaunchKernelGGL((gemm_nt_kernel<2,2,2>), grid, dim3(256), 0, stream, p); break;
      case 3: hipLaunchKernelGGL((gemm_nt_kernel<2,2,3>), grid, dim3(256), 0, stream, p); break;
    }
    return (int)hipGetLastError();
}
void psam_set_error(const char*) {}
