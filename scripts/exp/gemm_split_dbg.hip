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

__global__ __launch_bounds__(256) void gemm_bf16x6_kernel(const SplitGemmArgs p, unsigned long long* dbg) {
    unsigned long long tstamp[6]; unsigned long long rt0 = 0, rt1 = 0; int nst = 0; const unsigned long long t_entry = __builtin_readcyclecounter(); unsigned long long t_loop0 = 0, t_loop1 = 0;
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
            split3(f32x2{ra[i][0], ra[i][1]}, h0, m0_, l0);
            split3(f32x2{ra[i][2], ra[i][3]}, h1, m1, l1);
            sp[0][i][0] = u32x2{h0, h1}; sp[0][i][1] = u32x2{m0_, m1}; sp[0][i][2] = u32x2{l0, l1};
            split3(f32x2{rw[i][0], rw[i][1]}, h0, m0_, l0);
            split3(f32x2{rw[i][2], rw[i][3]}, h1, m1, l1);
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

    auto load_frags = [&](int s, bf16x8 (&af)[2][3], bf16x8 (&wf)[2][3]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                af[i][q] = *reinterpret_cast<const bf16x8*>(a_base + q * SG_PLANE_A + i * 32 * SG_ROWB + frag_off[s]);
                wf[i][q] = *reinterpret_cast<const bf16x8*>(w_base + q * SG_PLANE_W + i * 32 * SG_ROWB + frag_off[s]);
            }
    };
    // smallest partial products first (hl, lh, mm), then hm, mh, then hh; term-major so that the four accumulator tiles
    // rotate (an accumulator is reused only after three other MFMAs)
#define SG_TERM(AF, WF, PA, PW)                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                 \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[i][PA], WF[j][PW], acc[i][j], 0, 0, 0);
#define SG_STEP(AF, WF) SG_TERM(AF, WF, 0, 2) SG_TERM(AF, WF, 2, 0) SG_TERM(AF, WF, 1, 1) SG_TERM(AF, WF, 0, 1) SG_TERM(AF, WF, 1, 0) SG_TERM(AF, WF, 0, 0)

    const int nslabs = (p.K + SG_BK - 1) / SG_BK;
    // Pipeline.  At the top of slab t: sp = split slab t (registers); (ra0, rw0) = fp32 slab t+1 (landed or landing).
    // Slab t: [barrier: LDS free] sp -> LDS [barrier: LDS ready]; fragments of k16-step 0; then 2 x 24 MFMAs, in whose
    // shadow issue: the fragment reads of step 1, the split of slab t+1 into sp (~250 VALU: up to 7 fit behind one
    // 32-cycle MFMA) and the refill of the fp32 registers with slab t+2.  Measured with SQ counters on the unpipelined
    // version: 30 % of a wave's time went to issuing that VALU/LDS work serially and 18 % to waits, matrix pipe 47 % busy.
    // Two workgroups share a CU (one wave of each per SIMD).  They start together and do identical work, so without help
    // they run in lock-step: both in their MFMA phase (each at half rate), then both staging (matrix pipe idle) -- measured
    // 47 % pipe utilisation.  A static priority for every other hardware wave slot breaks the symmetry: the favoured wave
    // runs its MFMA phase at full rate while the other stages, and the two settle half a period apart.
    if (__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1) __builtin_amdgcn_s_setprio(1);  // HW_REG_HW_ID[3:0] = wave slot in the SIMD
    bf16x8 af0[2][3], wf0[2][3], af1[2][3], wf1[2][3];
    load_slab(0, ra0, rw0);
    split_regs(ra0, rw0);
    load_slab(SG_BK, ra0, rw0);
    t_loop0 = __builtin_readcyclecounter();
    for (int t = 0; t < nslabs; ++t) {
        if (t == 8) { tstamp[0] = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
        __syncthreads();
        if (t == 8) tstamp[1] = __builtin_readcyclecounter();
        store_split();
        __syncthreads();
        if (t == 8) tstamp[2] = __builtin_readcyclecounter();
        if (t == 9) tstamp[4] = __builtin_readcyclecounter();
        if (t == 24) { tstamp[5] = __builtin_readcyclecounter(); rt1 = __builtin_amdgcn_s_memrealtime(); }
        load_frags(0, af0, wf0);
        load_frags(1, af1, wf1);
        SG_STEP(af0, wf0)
        split_regs(ra0, rw0);                 // slab t+1
        SG_STEP(af1, wf1)
        load_slab((t + 2) * SG_BK, ra0, rw0);
        // requested issue order (one scheduling region: the loop body after the second barrier)
        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);      // fragments of step 0
#pragma unroll
        for (int i = 0; i < 12; ++i) {                            // step-0 MFMAs, fragment reads of step 1 behind them
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, 4, 0);
        }
#pragma unroll
        for (int i = 0; i < 36; ++i) {                            // remaining MFMAs with the split arithmetic behind them
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, 6, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x2, 64, 0);
        __builtin_amdgcn_sched_group_barrier(0x20, 8, 0);
    }
#undef SG_STEP
#undef SG_TERM

    t_loop1 = __builtin_readcyclecounter();
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
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* d = dbg + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
        d[0] = tstamp[0]; d[1] = tstamp[1]; d[2] = tstamp[2]; d[3] = tstamp[4]; d[4] = tstamp[5];
        d[5] = t_loop0 - t_entry; d[6] = t_loop1 - t_loop0; d[7] = __builtin_readcyclecounter() - t_loop1;
    }
}

extern "C" __attribute__((visibility("default"))) int gemm_split_dbg(const float* A, const float* W, float* C, const float* bias, int M, int N, int K, unsigned long long* dbg, hipStream_t stream) {
    SplitGemmArgs p; p.A=A; p.W=W; p.C=C; p.bias=bias; p.residual=nullptr; p.rowbias=nullptr; p.lda=K; p.ldw=K; p.ldc=N; p.ldr=0; p.ldrb=0;
    p.sA1=p.sA2=p.sW1=p.sW2=p.sC1=p.sC2=p.sR1=p.sR2=0; p.M=M; p.N=N; p.K=K; p.batch2=1; p.rowgroup=1; p.act=0; p.alpha=1.f;
    p.tiles_m=(M+127)/128; p.tiles_n=(N+127)/128;
    hipLaunchKernelGGL(gemm_bf16x6_kernel, dim3(p.tiles_m*p.tiles_n), dim3(256), 0, stream, p, dbg);
    return (int)hipGetLastError();
}
void psam_set_error(const char*) {}
