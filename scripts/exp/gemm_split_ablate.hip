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
// operands stay plain fp32 in HBM and no other kernel changes.  Tile 128x128, 2x2 waves of 64x64, slabs of 32 k; LDS
// holds the three bf16 planes of both operands (48 KiB, single buffered, 2 workgroups per CU); rows are 64 bytes with the
// 16-byte chunk index XOR-swizzled by (row>>2)&3: conflict-free ds_read_b128 fragments and ds_write_b64 staging.
#include "common.h"

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
constexpr int SG_BM = 128, SG_BN = 128;
constexpr int SG_ROWB = SG_BK * 2;                       // bytes per row of one bf16 plane
constexpr int SG_PLANE_A = SG_BM * SG_ROWB;              // bytes per plane
constexpr int SG_PLANE_W = SG_BN * SG_ROWB;

// x (2 floats) -> packed bf16 pairs hi, mid, lo with x == hi + mid + lo exactly
template <int ABL> __device__ __forceinline__ void split3(const f32x2 x, unsigned& hi, unsigned& mid, unsigned& lo) {
    if (ABL == 1 || ABL >= 6) { hi = (__builtin_bit_cast(unsigned, x[0]) >> 16) | (__builtin_bit_cast(unsigned, x[1]) & 0xffff0000u); mid = 0; lo = 0; return; }
    const bf16x2 h = __builtin_convertvector(x, bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    mid = __builtin_bit_cast(unsigned, m);
    lo = __builtin_bit_cast(unsigned, l);
}

template <int ABL>
__global__ __launch_bounds__(256) void gemm_bf16x6_kernel(const SplitGemmArgs p) {
    // LDS: [A planes hi,mid,lo][W planes hi,mid,lo]
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * SG_PLANE_A + 3 * SG_PLANE_W];
    unsigned char* sA = smem;
    unsigned char* sW = smem + 3 * SG_PLANE_A;

    const int ntiles = p.tiles_m * p.tiles_n;
    int tile = blockIdx.x;
    if ((ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);
    const int m0 = (tile / p.tiles_n) * SG_BM, n0 = (tile % p.tiles_n) * SG_BN;
    const int z = blockIdx.z, z1 = z / p.batch2, z2 = z % p.batch2;
    const float* A = p.A + z1 * p.sA1 + z2 * p.sA2;
    const float* W = p.W + z1 * p.sW1 + z2 * p.sW2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // global -> register staging: thread (lr, lc4) owns 4 consecutive k of rows lr, lr+32, lr+64, lr+96 of both operands
    const int lr = tid >> 3, lc4 = tid & 7;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((((int64_t)p.M - 1) * p.lda + p.K) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, (int)((((int64_t)p.N - 1) * p.ldw + p.K) * 4), 0x00020000);
    const int voA = (int)(((int64_t)(m0 + lr) * p.lda + lc4 * 4) * 4), voW = (int)(((int64_t)(n0 + lr) * p.ldw + lc4 * 4) * 4);
    const int stepA = (int)(32 * p.lda * 4), stepW = (int)(32 * p.ldw * 4);
    constexpr int OOB = 0x7ffffff0;
    f32x4 ra0[4], rw0[4];
    auto load_slab = [&](int k0, f32x4 (&ra)[4], f32x4 (&rw)[4]) {
        const bool kok = k0 + lc4 * 4 < p.K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (kok && m0 + i * 32 + lr < p.M) ? voA + i * stepA : OOB, k0 * 4, 0));
            rw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, (kok && n0 + i * 32 + lr < p.N) ? voW + i * stepW : OOB, k0 * 4, 0));
        }
    };
    // LDS byte offset of this thread's 8-byte slot inside a plane: row lr (+32 i), 16-byte chunk lc4>>1 swizzled, half lc4&1
    const int st_off = lr * SG_ROWB + ((((lc4 >> 1) ^ ((lr >> 2) & 3)) << 4) | ((lc4 & 1) << 3));
    // split registers: [operand A/W][row stripe i][plane hi/mid/lo] -> 2 packed words (4 bf16)
    u32x2 sp[2][4][3];
    auto split_regs = [&](const f32x4 (&ra)[4], const f32x4 (&rw)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            unsigned h0, m0_, l0, h1, m1, l1;
            split3<ABL>(f32x2{ra[i][0], ra[i][1]}, h0, m0_, l0);
            split3<ABL>(f32x2{ra[i][2], ra[i][3]}, h1, m1, l1);
            sp[0][i][0] = u32x2{h0, h1}; sp[0][i][1] = u32x2{m0_, m1}; sp[0][i][2] = u32x2{l0, l1};
            split3<ABL>(f32x2{rw[i][0], rw[i][1]}, h0, m0_, l0);
            split3<ABL>(f32x2{rw[i][2], rw[i][3]}, h1, m1, l1);
            sp[1][i][0] = u32x2{h0, h1}; sp[1][i][1] = u32x2{m0_, m1}; sp[1][i][2] = u32x2{l0, l1};
        }
    };
    auto store_split = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                *reinterpret_cast<u32x2*>(sA + q * SG_PLANE_A + i * 32 * SG_ROWB + st_off) = sp[0][i][q];
                *reinterpret_cast<u32x2*>(sW + q * SG_PLANE_W + i * 32 * SG_ROWB + st_off) = sp[1][i][q];
            }
    };
    // fragment addresses: row r32 of 32-row tile, chunk (2s + h) swizzled
    int frag_off[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) frag_off[s] = r32 * SG_ROWB + (((2 * s + h) ^ ((r32 >> 2) & 3)) << 4);
    const unsigned char* a_base = sA + wm * 64 * SG_ROWB;
    const unsigned char* w_base = sW + wn * 64 * SG_ROWB;

    auto compute_slab = [&]() {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 af[2][3], wf[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    af[i][q] = *reinterpret_cast<const bf16x8*>(a_base + q * SG_PLANE_A + i * 32 * SG_ROWB + frag_off[s]);
                    wf[i][q] = *reinterpret_cast<const bf16x8*>(w_base + q * SG_PLANE_W + i * 32 * SG_ROWB + frag_off[s]);
                }
            // smallest partial products first (hl, lh, mm), then hm, mh, then hh; term-major so that the four accumulator
            // tiles rotate (an accumulator is reused only after three other MFMAs)
#define SG_TERM(PA, PW)                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA], wf[j][PW], acc[i][j], 0, 0, 0);
            if (ABL != 2) { SG_TERM(0, 2) SG_TERM(2, 0) SG_TERM(1, 1) SG_TERM(0, 1) SG_TERM(1, 0) } SG_TERM(0, 0)
#undef SG_TERM
        }
    };

    const int nslabs = (p.K + SG_BK - 1) / SG_BK;
    // Pipeline.  At the top of slab t: sp = split slab t (registers); (ra0, rw0) = fp32 slab t+1 (landed or landing).
    // Slab t: [barrier: LDS free] sp -> LDS [barrier: LDS ready] then the MFMAs of slab t with, in their shadow (VALU and
    // loads issue while the matrix pipe works), the split of slab t+1 into sp and the refill of the fp32 registers with
    // slab t+2 (a full slab period, ~2 us with two workgroups per CU, to land).  One fp32 set only: with two the kernel
    // needs > 256 registers and drops to one wave per SIMD.  Loads past K return zeros (bounds-checked descriptor).
    load_slab(0, ra0, rw0);
    split_regs(ra0, rw0);
    load_slab(SG_BK, ra0, rw0);
    for (int t = 0; t < nslabs; ++t) {
        if (ABL != 4 && ABL < 5) { __syncthreads(); store_split(); __syncthreads(); }
        if (ABL == 5 || ABL == 6) asm volatile("" ::: "memory");
        compute_slab();
        split_regs(ra0, rw0);                 // slab t+1
        if (ABL != 3 && ABL < 5) load_slab((t + 2) * SG_BK, ra0, rw0);
        // Ask the scheduler for: fragment reads + MFMAs of the first k16 step, then the second step's MFMAs with the
        // split arithmetic of the next slab in their shadow (as late as possible: its operands were loaded one slab
        // ago), then the global loads of slab t+2.
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
        for (int i = 0; i < 24; ++i) __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, 7, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x2, 64, 0);
        __builtin_amdgcn_sched_group_barrier(0x20, 8, 0);
    }

    // ---- epilogue (C/D layout of 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5))
    float* C = p.C + z1 * p.sC1 + z2 * p.sC2;
    const float* R = p.residual ? p.residual + z1 * p.sR1 + z2 * p.sR2 : nullptr;
    if (p.act == 3) {  // SwiGLU gate: accumulator tile j=0 holds g, j=1 holds x of the same 32 hidden units (see gemm.hip)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row0 = m0 + (wm * 2 + i) * 32;
            const int colg = n0 + wn * 64 + r32;
            if (colg >= p.N) continue;
            const float bg = p.bias ? p.bias[colg] : 0.f, bx = p.bias ? p.bias[colg + 32] : 0.f;
            const int ocol = (n0 + wn * 64) / 2 + r32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= p.M) continue;
                C[(int64_t)row * p.ldc + ocol] = silu(acc[i][0][r] + bg) * (acc[i][1][r] + bx);
            }
        }
        return;
    }
    const bool group_uniform = p.rowbias && (p.rowgroup & 31) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row0 = m0 + (wm * 2 + i) * 32;
        const float* rb_tile = group_uniform ? p.rowbias + (int64_t)(row0 / p.rowgroup) * p.ldrb : nullptr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + (wn * 2 + j) * 32 + r32;
            if (col >= p.N) continue;
            float bv = p.bias ? p.bias[col] : 0.f;
            if (rb_tile) bv += rb_tile[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row >= p.M) continue;
                float v = acc[i][j][r] * p.alpha + bv;
                if (p.rowbias && !group_uniform) v += p.rowbias[(int64_t)(row / p.rowgroup) * p.ldrb + col];
                if (p.act == 1) v = gelu_erf(v);
                else if (p.act == 2) v = fmaxf(v, 0.f);
                if (R) v += R[(int64_t)row * p.ldr + col];
                C[(int64_t)row * p.ldc + col] = v;
            }
        }
    }
}


extern "C" __attribute__((visibility("default"))) int gemm_split_ablate(const float* A, const float* W, float* C, const float* bias, int M, int N, int K, int abl, hipStream_t stream) {
    SplitGemmArgs p; p.A=A; p.W=W; p.C=C; p.bias=bias; p.residual=nullptr; p.rowbias=nullptr; p.lda=K; p.ldw=K; p.ldc=N; p.ldr=0; p.ldrb=0;
    p.sA1=p.sA2=p.sW1=p.sW2=p.sC1=p.sC2=p.sR1=p.sR2=0; p.M=M; p.N=N; p.K=K; p.batch2=1; p.rowgroup=1; p.act=0; p.alpha=1.f;
    p.tiles_m=(M+127)/128; p.tiles_n=(N+127)/128;
    dim3 grid(p.tiles_m*p.tiles_n,1,1);
    switch(abl){
      case 0: hipLaunchKernelGGL((gemm_bf16x6_kernel<0>), grid, dim3(256), 0, stream, p); break;
      case 1: hipLaunchKernelGGL((gemm_bf16x6_kernel<1>), grid, dim3(256), 0, stream, p); break;
      case 2: hipLaunchKernelGGL((gemm_bf16x6_kernel<2>), grid, dim3(256), 0, stream, p); break;
      case 3: hipLaunchKernelGGL((gemm_bf16x6_kernel<3>), grid, dim3(256), 0, stream, p); break;
      case 4: hipLaunchKernelGGL((gemm_bf16x6_kernel<4>), grid, dim3(256), 0, stream, p); break;
      case 5: hipLaunchKernelGGL((gemm_bf16x6_kernel<5>), grid, dim3(256), 0, stream, p); break;
      case 6: hipLaunchKernelGGL((gemm_bf16x6_kernel<6>), grid, dim3(256), 0, stream, p); break;
      case 7: hipLaunchKernelGGL((gemm_bf16x6_kernel<7>), grid, dim3(256), 0, stream, p); break;
    }
    return (int)hipGetLastError();
}
void psam_set_error(const char*) {}
