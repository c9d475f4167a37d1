// Flash-style fp32 attention on the gfx950 matrix cores for the ViT patch encoder (timm EvaAttention ->
// F.scaled_dot_product_attention, no mask, no RoPE: the reference calls block(x) without `rope`,
// pc_sam/model/pc_encoder.py:138-139).  out = softmax(q k^T * scale) v, per (cloud, head).
//
// Layout trick (both products keep the probabilities in registers, no LDS round trip, no cross-lane traffic
// besides one lane^32 exchange per 32 keys):
//   S^T = K Q^T   -> v_mfma_f32_32x32x2_f32 with A = K tile (rows = keys), B = Q^T (cols = query rows):
//                    lane (q = lane&31, h = lane>>5) ends up with S[q][key] for 16 keys of the 32-key sub-tile,
//                    i.e. every lane owns ONE query row -> running max / sum / rescale are lane-local.
//   O^T = V^T P^T -> A = V^T (rows = channels), B = P^T: the C/D registers of the first product are, register
//                    for register, the B operand of the second (key(r,h) = (r&3)+8*(r>>2)+4*h on both sides).
// K/V tiles of 64 keys are staged in LDS (K rows padded by 4 floats: conflict-free ds_read_b128), double
// buffered with a register prefetch; Q stays in registers, pre-scaled by scale*log2(e) so softmax uses v_exp_f32.
#include "common.h"
#include <math.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int FA_BQ = 128;   // query rows per workgroup (4 waves x 32)
constexpr int FA_BKV = 64;   // keys per LDS tile

struct FlashArgs {
    const float* q; const float* k; const float* v; float* o;
    int64_t ldq, ldk, ldv, ldo, sq, sk, sv, so;
    int H, Lq, Lk, B;
    float scale_log2e;
    // packed output (flash_attn_f16x3_kernel only): o receives the g8-packed rows (the A operand of the output projection,
    // gemm_f16x3p.hip), scaled by o_scale[row] = the power of two that puts the BOUND k1 / min_rows(a_scale) + k2 of |V| over the cloud
    // into [2^14, 2^15) -- attention outputs are convex combinations of V rows, so the bound holds for them; a_scale = the row scales
    // of the qkv GEMM's A operand (LayerNorm output), k1 = 2^15 sqrt(D) max_n ||W_v[n]||, k2 = max |b_v|.
    const float* a_scale; float* o_scale; float k1, k2;
#ifdef PSAM_ATTN_ABLATE
    int abl;      // scripts/exp/attn_abl.*: 1 no per-tile convert+store, 2 no tile loads, 4 no exp/split, 8 no S MFMAs, 16 no PV MFMAs
#endif
};
#ifdef PSAM_ATTN_ABLATE
#define FA_ABL(bit) (p.abl & (bit))
static int g_attn_abl = 0;
PSAM_API void psam_attention_set_ablation(int32_t a) { g_attn_abl = a; }
#else
#define FA_ABL(bit) false
#endif

template <int HD8, int DT>
__global__ __launch_bounds__(256) void flash_attn_f32_kernel(const FlashArgs p) {
    constexpr int HD = HD8 * 8;
    constexpr int KLD = HD + 4;
    constexpr int VLD = DT * 32;
    constexpr int ROW4 = HD / 4;                       // float4 per K/V row
    constexpr int NF4 = (FA_BKV * ROW4 + 255) / 256;   // float4 per thread per operand tile
    __shared__ __attribute__((aligned(16))) float sK[2][FA_BKV * KLD];
    __shared__ __attribute__((aligned(16))) float sV[2][FA_BKV * VLD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * FA_BQ + wave * 32;
    const float* Q = p.q + b * p.sq + head * HD;
    const float* K = p.k + b * p.sk + head * HD;
    const float* V = p.v + b * p.sv + head * HD;

    // zero the V padding columns once (hd not a multiple of 32)
    if (VLD > HD) {
        for (int i = tid; i < 2 * FA_BKV * VLD; i += 256) (&sV[0][0])[i] = 0.f;
        __syncthreads();
    }

    // this lane's query row, k-permuted float4s: dims 8s+4h .. 8s+4h+3
    f32x4 qf[HD8];
    {
        const int qrow = q0 + r32;
        const bool ok = qrow < p.Lq;
        const float* qp = Q + (int64_t)(ok ? qrow : 0) * p.ldq + h * 4;
#pragma unroll
        for (int s = 0; s < HD8; ++s) {
            f32x4 t = *reinterpret_cast<const f32x4*>(qp + s * 8);
            qf[s] = ok ? t * p.scale_log2e : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    f32x16 oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // Loads go through bounds-checked buffer descriptors (keys >= Lk read as zeros) and are UNCONDITIONAL: with exec-masked
    // loads the compiler waits vmcnt(0) at the join, i.e. before the tile's compute, and the ~2 us latency is fully exposed.
    f32x4 rk[NF4], rv[NF4];
    constexpr int OOB = 0x7ffffff0;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (int)((((int64_t)p.Lk - 1) * p.ldk + HD) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, (int)((((int64_t)p.Lk - 1) * p.ldv + HD) * 4), 0x00020000);
    auto load_tile = [&](int kv0) {
        const int kbase = (int)((int64_t)kv0 * p.ldk * 4), vbase = (int)((int64_t)kv0 * p.ldv * 4);
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = i * 256 + tid;
            const int row = f / ROW4, c4 = f % ROW4;
            const bool ok = (f < FA_BKV * ROW4) && kv0 + row < p.Lk;
            rk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsK, ok ? (int)(((int64_t)row * p.ldk + c4 * 4) * 4) : OOB, kbase, 0));
            rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, ok ? (int)(((int64_t)row * p.ldv + c4 * 4) * 4) : OOB, vbase, 0));
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = i * 256 + tid;
            if (f < FA_BKV * ROW4) {
                const int row = f / ROW4, c4 = f % ROW4;
                *reinterpret_cast<f32x4*>(&sK[buf][row * KLD + c4 * 4]) = rk[i];
                *reinterpret_cast<f32x4*>(&sV[buf][row * VLD + c4 * 4]) = rv[i];
            }
        }
    };

    const int ntiles = (p.Lk + FA_BKV - 1) / FA_BKV;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) load_tile((t + 1) * FA_BKV);
#pragma unroll
        for (int kt = 0; kt < FA_BKV / 32; ++kt) {
            const int key_base = t * FA_BKV + kt * 32;
            if (key_base >= p.Lk) break;  // uniform: whole sub-tile past the end
            // ---- S^T (32 keys x 32 queries) = K_sub Q^T
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            const float* kb = &sK[buf][(kt * 32 + r32) * KLD + h * 4];
#pragma unroll
            for (int s = 0; s < HD8; ++s) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(kb + s * 8);
#pragma unroll
                for (int c = 0; c < 4; ++c) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c], qf[s][c], st, 0, 0, 0);
            }
            // ---- online softmax for this lane's query row (keys (r&3)+8*(r>>2)+4*h of the sub-tile)
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key_base + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (key >= p.Lk) st[r] = -INFINITY;
                mx = fmaxf(mx, st[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = __builtin_amdgcn_exp2f(st[r] - m_new);
                ps += st[r];
            }
            l_run = l_run * alpha + ps;  // per-half partial sum; halves are added once at the end
            m_run = m_new;
#pragma unroll
            for (int d = 0; d < DT; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) oacc[d][r] *= alpha;
            // ---- O^T += V_sub^T P^T
            const float* vb = &sV[buf][(kt * 32 + 4 * h) * VLD + r32];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int d = 0; d < DT; ++d)
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[krow * VLD + d * 32], st[r], oacc[d], 0, 0, 0);
            }
        }
        if (t + 1 < ntiles) store_tile(buf ^ 1);
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + r32;
    if (qrow < p.Lq) {
        float* op = p.o + b * p.so + (int64_t)qrow * p.ldo + head * HD;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = d * 32 + 8 * g + 4 * h;  // regs 4g..4g+3 hold channels d0..d0+3 of this query row
                if (d0 < HD) {
                    f32x4 o4 = {oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv};
                    *reinterpret_cast<f32x4*>(op + d0) = o4;
                }
            }
    }
}

// q/k/v/o: [B, L, H*hd] views with row strides ld* and batch strides s* (elements); head h lives at columns
// [h*hd, (h+1)*hd).  Works on the fused qkv buffer of a ViT block (ldq = ldk = ldv = 3*D).
PSAM_API int32_t psam_attention_f32(const float* q, int64_t ldq, int64_t sq, const float* k, int64_t ldk, int64_t sk, const float* v, int64_t ldv,
                                    int64_t sv, float* o, int64_t ldo, int64_t so, int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd,
                                    float scale, hipStream_t stream) {
    PSAM_REQUIRE(q && k && v && o, PSAM_EINVAL, "psam_attention_f32: null pointer");
    PSAM_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0, PSAM_EINVAL, "psam_attention_f32: bad shape");
    PSAM_REQUIRE(B <= 65535 && H <= 65535, PSAM_EINVAL, "psam_attention_f32: B/H too large");
    PSAM_REQUIRE(((ldq | ldk | ldv | ldo | sq | sk | sv | so) & 3) == 0 && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0,
                 PSAM_EALIGN, "psam_attention_f32: strides must be multiples of 4 floats and pointers 16-byte aligned");
    PSAM_REQUIRE((int64_t)Lk * ldk < ((int64_t)1 << 29) && (int64_t)Lk * ldv < ((int64_t)1 << 29), PSAM_EINVAL,
                 "psam_attention_f32: one (batch) K/V slice must span < 2 GiB (32-bit buffer offsets)");
    FlashArgs p;
    p.q = q; p.k = k; p.v = v; p.o = o;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.sq = sq; p.sk = sk; p.sv = sv; p.so = so;
    p.H = H; p.Lq = Lq; p.Lk = Lk; p.B = B;
    p.scale_log2e = scale * 1.4426950408889634f;
    p.a_scale = nullptr; p.o_scale = nullptr; p.k1 = p.k2 = 0.f;
    const dim3 grid((unsigned)psam_cdiv(Lq, FA_BQ), H, B), block(256);
#define FA_LAUNCH(HD8, DT) hipLaunchKernelGGL((flash_attn_f32_kernel<HD8, DT>), grid, block, 0, stream, p)
    switch (hd) {
        case 16: FA_LAUNCH(2, 1); break;
        case 24: FA_LAUNCH(3, 1); break;
        case 32: FA_LAUNCH(4, 1); break;
        case 48: FA_LAUNCH(6, 2); break;
        case 64: FA_LAUNCH(8, 2); break;
        case 88: FA_LAUNCH(11, 3); break;
        case 96: FA_LAUNCH(12, 3); break;
        case 128: FA_LAUNCH(16, 4); break;
        default:
            psam_set_error("psam_attention_f32: head_dim must be one of 16,24,32,48,64,88,96,128");
            return PSAM_EINVAL;
    }
#undef FA_LAUNCH
    return psam_launch_status("psam_attention_f32: launch failed");
}

// ================================================================================================================
// The same attention on the fp16 matrix pipe with fp32-grade products ("f16x3", see gemm_f16x3.hip): every operand is
// scaled by a power of two and split into hi + lo fp16; hi*hi + hi*lo + lo*hi run on v_mfma_f32_32x32x16_f16 (fp32
// accumulation).  3/16 of the matrix-pipe time of the f32 kernel above; the softmax VALU work becomes the longer pole.
//   Q: scaled per query row (lane-local), split once into registers (B operand of S^T = K Q^T).
//   K, V: scaled per 64-key tile (workgroup max through LDS), split while the tile is staged; K planes row-major
//         [key][d], V planes TRANSPOSED [d][key] (the A operand of O^T = V^T P^T needs 8 consecutive keys per channel).
//   P = exp2(s - m + 14): produced already scaled by 2^14, split in registers; because a 32x32x16 operand holds 8
//         consecutive k-slots per lane while the S^T accumulator holds keys {0-3, 8-11} (+4 for the upper half-wave),
//         the k-slots of the second product are simply DEFINED in that order (slot e of half h = key (e&3)+8(e>>2)+4h):
//         the probabilities never leave their registers and V^T is stored with the matching key permutation.
//   O is accumulated in the scaled domain of the current V tile; the tile-to-tile ratio of V scales (a power of two)
//         rides on the online-softmax rescale that exists anyway.
// LDS rows are 128 bytes with the 16-byte chunk XOR-swizzled by (row>>1)&7: conflict-free ds_read_b128 fragments.
typedef _Float16 fa_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 fa_f16x2 __attribute__((ext_vector_type(2)));
typedef float fa_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned fa_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned fa_u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void fa_split2(const fa_f32x2 xs, unsigned& hi, unsigned& lo) {   // xs already scaled
    const fa_f16x2 h = __builtin_convertvector(xs, fa_f16x2);
    const fa_f32x2 r = xs - __builtin_convertvector(h, fa_f32x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, fa_f16x2));
}
__device__ __forceinline__ float fa_inv_pow2(float s) { return __builtin_bit_cast(float, (254u << 23) - __builtin_bit_cast(unsigned, s)); }

template <int HD>   // head dim: 64 or 128
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HD == 64 ? 2 : 1))) void flash_attn_f16x3_kernel(const FlashArgs p) {
    static_assert(HD == 64 || HD == 128, "head dim");
    constexpr int KS = HD / 16;                         // k16 steps of S^T = K Q^T
    constexpr int DT = HD / 32;                         // 32-channel tiles of O^T
    constexpr int ROWB = 128;                           // bytes per LDS row (64 fp16)
    constexpr int KROW = HD * 2;                        // bytes per K-plane row
    constexpr int KSW = (KROW / 16 - 1) < 7 ? (KROW / 16 - 1) : 7;   // swizzle mask: stays inside the row's chunks
    constexpr int KPLANE = FA_BKV * KROW, VPLANE = HD * ROWB;   // K: [64 keys][HD], V^T: [HD][64 keys]
    constexpr int STAGE = 2 * KPLANE + 2 * VPLANE;
    constexpr int ROW4 = HD / 4;
    constexpr int NF4 = FA_BKV * ROW4 / 256;            // float4 per thread per operand tile (2, 4, 8)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
    __shared__ float smax[2][4][2];                     // per stage, per wave: max|K tile|, max|V tile|

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, h = lane >> 5;
    // 1-D grid: consecutive workgroup ids go to different XCDs (own L2 each), so the nq query blocks of one (batch, head) -- which all read the
    // same K/V slice -- take ids 8 apart and share an XCD's L2 (rocprofv3 FETCH_SIZE at B=8, H=16, L=512 with the query block as the fastest
    // grid dimension: 151 MB per launch against 50 MB of q/k/v -- four XCDs each fetched every slice)
    const int nq = (p.Lq + FA_BQ - 1) / FA_BQ, HB = p.H * p.B;
    int qb, hb;
    if ((HB & 7) == 0) { const int id = blockIdx.x, grp = id / (8 * nq), r = id - grp * 8 * nq; hb = grp * 8 + (r & 7); qb = r >> 3; }
    else { qb = blockIdx.x % nq; hb = blockIdx.x / nq; }
    const int head = hb % p.H, b = hb / p.H;
    const int q0 = qb * FA_BQ + wave * 32;
    const float* Q = p.q + b * p.sq + head * HD;
    const float* K = p.k + b * p.sk + head * HD;
    const float* V = p.v + b * p.sv + head * HD;

    f32x16 oacc[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;   // l_run sums the 2^14-scaled probabilities
    float sv_acc = 1.f;                     // V scale of the domain oacc is in

    // ---- staging: thread f = i*256 + tid owns float4 c4 of tile row `row`
    f32x4 rk[NF4], rv[NF4];
    // K: thread f = i*256 + tid owns float4 c4 of tile row `row`.  V: thread owns float4 column vc4 of the 4 CONSECUTIVE keys
    // 4*vj .. 4*vj+3 (+ 64/VJ-row blocks for HD < 64), so that the transposed planes are written 4 keys (8 bytes) at a time.
    // Loads go through bounds-checked buffer descriptors (keys >= Lk read as zeros) and are UNCONDITIONAL: with exec-masked
    // loads the compiler waits vmcnt(0) at the join, i.e. before the tile's compute, and the ~2 us latency is fully exposed.
    constexpr int VJ = 256 / ROW4;                       // key groups covered per pass (16 for HD = 64)
    const int vc4 = tid % ROW4, vj = tid / ROW4;
    constexpr int OOB = 0x7ffffff0;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (int)((((int64_t)p.Lk - 1) * p.ldk + HD) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, (int)((((int64_t)p.Lk - 1) * p.ldv + HD) * 4), 0x00020000);
    int koffs[NF4], voffs[NF4];                          // byte offsets inside a tile; the tile base goes in the scalar offset
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
        const int f = i * 256 + tid;
        koffs[i] = (int)(((int64_t)(f / ROW4) * p.ldk + (f % ROW4) * 4) * 4);
        voffs[i] = (int)(((int64_t)(4 * (vj + VJ * (i >> 2)) + (i & 3)) * p.ldv + vc4 * 4) * 4);
    }
    auto load_tile = [&](int kv0) {
        const int kbase = (int)((int64_t)kv0 * p.ldk * 4), vbase = (int)((int64_t)kv0 * p.ldv * 4);
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int key = kv0 + (i * 256 + tid) / ROW4;
            rk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsK, key < p.Lk ? koffs[i] : OOB, kbase, 0));
        }
#pragma unroll
        for (int i = 0; i < NF4; ++i) {                  // i = (pass, e): key = 4 * (vj + VJ * pass) + e
            const int key = kv0 + 4 * (vj + VJ * (i >> 2)) + (i & 3);
            rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, key < p.Lk ? voffs[i] : OOB, vbase, 0));
        }
    };
    auto publish_max = [&](int buf) {
        float km = 0.f, vm = 0.f;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            km = fmaxf(fmaxf(km, fmaxf(fabsf(rk[i][0]), fabsf(rk[i][1]))), fmaxf(fabsf(rk[i][2]), fabsf(rk[i][3])));
            vm = fmaxf(fmaxf(vm, fmaxf(fabsf(rv[i][0]), fabsf(rv[i][1]))), fmaxf(fabsf(rv[i][2]), fabsf(rv[i][3])));
        }
        km = wave_max(km); vm = wave_max(vm);
        if (lane == 0) { smax[buf][wave][0] = km; smax[buf][wave][1] = vm; }
    };
    auto tile_scales = [&](int buf, float& sk, float& sv) {
        sk = f16_row_scale(fmaxf(fmaxf(smax[buf][0][0], smax[buf][1][0]), fmaxf(smax[buf][2][0], smax[buf][3][0])));
        sv = f16_row_scale(fmaxf(fmaxf(smax[buf][0][1], smax[buf][1][1]), fmaxf(smax[buf][2][1], smax[buf][3][1])));
    };
    auto store_tile = [&](int buf, float sk, float sv) {
        unsigned char* st = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = i * 256 + tid;
            const int row = f / ROW4, c4 = f % ROW4;
            unsigned h0, l0, h1, l1;
            // K planes: row-major, chunk (c4>>1) of the row swizzled
            fa_split2(fa_f32x2{rk[i][0], rk[i][1]} * sk, h0, l0);
            fa_split2(fa_f32x2{rk[i][2], rk[i][3]} * sk, h1, l1);
            const int koff = row * KROW + ((((c4 >> 1) ^ ((row >> 1) & KSW)) << 4) | ((c4 & 1) << 3));
            *reinterpret_cast<fa_u32x2*>(st + koff) = fa_u32x2{h0, h1};
            *reinterpret_cast<fa_u32x2*>(st + KPLANE + koff) = fa_u32x2{l0, l1};
        }
        // V planes: transposed; keys 4g..4g+3 go to slots pos..pos+3 with pos = (4g & ~12) | swap of bits 2,3 (see header)
#pragma unroll
        for (int ps = 0; ps < NF4 / 4; ++ps) {
            const int key0 = 4 * (vj + VJ * ps);
            const int pos = (key0 & ~12) | ((key0 & 4) << 1) | ((key0 & 8) >> 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = vc4 * 4 + e;
                unsigned h0, l0, h1, l1;
                fa_split2(fa_f32x2{rv[4 * ps][e], rv[4 * ps + 1][e]} * sv, h0, l0);
                fa_split2(fa_f32x2{rv[4 * ps + 2][e], rv[4 * ps + 3][e]} * sv, h1, l1);
                const int voff = 2 * KPLANE + d * ROWB + ((((pos >> 3) ^ ((d >> 1) & 7)) << 4) | ((pos & 7) << 1));
                *reinterpret_cast<fa_u32x2*>(st + voff) = fa_u32x2{h0, h1};
                *reinterpret_cast<fa_u32x2*>(st + VPLANE + voff) = fa_u32x2{l0, l1};
            }
        }
    };

    const int ntiles = (p.Lk + FA_BKV - 1) / FA_BKV;
    float sk_cur, sv_cur;
    load_tile(0);      // in flight while the query rows (and the packed output's scale) are fetched: one memory round trip, not three
    float out_scale = 0.f;     // packed output: one scale for every row of the cloud (see FlashArgs)
    if (p.o_scale) {
        float smin = INFINITY;
        for (int i = tid; i < p.Lk; i += 256) smin = fminf(smin, p.a_scale[(int64_t)b * p.Lk + i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) smin = fminf(smin, __shfl_xor(smin, o, 64));
        if (lane == 0) smax[0][wave][0] = smin;
        __syncthreads();
        smin = fminf(fminf(smax[0][0][0], smax[0][1][0]), fminf(smax[0][2][0], smax[0][3][0]));
        __syncthreads();       // smax is reused by the tile scales below
        out_scale = f16_row_scale(p.k1 / smin + p.k2);
    }

    // ---- this lane's query row: d-slots 16s + 8h .. +7 of every k16 step, scaled by the row's power of two, split
    fa_f16x8 qh[KS], ql[KS];
    float q_inv;   // 1 / row scale
    {
        const int qrow = q0 + r32;
        const bool ok = qrow < p.Lq;
        const float* qp = Q + (int64_t)(ok ? qrow : 0) * p.ldq + h * 8;
        f32x4 t[KS][2];
        float amax = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                t[s][e] = ok ? *reinterpret_cast<const f32x4*>(qp + s * 16 + e * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(t[s][e][0]), fabsf(t[s][e][1]))), fmaxf(fabsf(t[s][e][2]), fabsf(t[s][e][3])));
            }
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const float sq = f16_row_scale(amax);
        q_inv = fa_inv_pow2(sq);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            unsigned hi[4], lo[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                fa_split2(fa_f32x2{t[s][e][0], t[s][e][1]} * sq, hi[2 * e], lo[2 * e]);
                fa_split2(fa_f32x2{t[s][e][2], t[s][e][3]} * sq, hi[2 * e + 1], lo[2 * e + 1]);
            }
            qh[s] = __builtin_bit_cast(fa_f16x8, fa_u32x4{hi[0], hi[1], hi[2], hi[3]});
            ql[s] = __builtin_bit_cast(fa_f16x8, fa_u32x4{lo[0], lo[1], lo[2], lo[3]});
        }
    }

    publish_max(0);
    __syncthreads();
    tile_scales(0, sk_cur, sv_cur);
    store_tile(0, sk_cur, sv_cur);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        const unsigned char* st_base = smem + buf * STAGE;
        if (t + 1 < ntiles && !FA_ABL(2)) load_tile((t + 1) * FA_BKV);
        const float c_s = p.scale_log2e * q_inv * fa_inv_pow2(sk_cur);     // S_scaled -> log2-domain logits
#pragma unroll
        for (int kt = 0; kt < FA_BKV / 32; ++kt) {
            const int key_base = t * FA_BKV + kt * 32;
            if (key_base >= p.Lk) break;  // uniform
            // ---- S^T (32 keys x 32 queries), scaled domain
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
            const int krow = kt * 32 + r32;
            const unsigned char* kb = st_base + krow * KROW;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int off = ((2 * s + h) ^ ((krow >> 1) & KSW)) << 4;
                const fa_f16x8 kh = *reinterpret_cast<const fa_f16x8*>(kb + off);
                const fa_f16x8 kl = *reinterpret_cast<const fa_f16x8*>(kb + KPLANE + off);
                if (FA_ABL(8)) { st[0] += (float)kh[0] + (float)kl[1]; continue; }
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], st, 0, 0, 0);
                st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], st, 0, 0, 0);
            }
            // ---- online softmax for this lane's query row (keys (r&3)+8*(r>>2)+4*h of the sub-tile)
            if (key_base + 32 > p.Lk) {   // uniform: only the ragged last sub-tile masks
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key_base + (r & 3) + 8 * (r >> 2) + 4 * h >= p.Lk) st[r] = -INFINITY;
            }
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c_s;        // c_s > 0: max commutes with the scaling
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            const float m14 = m_new - 14.f;
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[r] = FA_ABL(4) ? __builtin_fmaf(st[r], c_s, -m14) : __builtin_amdgcn_exp2f(__builtin_fmaf(st[r], c_s, -m14));   // probability * 2^14, in [0, 2^14]
                ps += st[r];
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
            // rescale O into the domain of this tile's V scale (ratio of powers of two) and the new running max
            const float fo = alpha * (sv_cur * fa_inv_pow2(sv_acc));
            sv_acc = sv_cur;
            if (__builtin_amdgcn_ballot_w64(fo != 1.f) != 0) {
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[d][r] *= fo;
            }
            // ---- split P (registers 0-7 = k-slots of step 0, 8-15 = step 1) and O^T += V_sub^T P^T
            fa_f16x8 ph[2], pl[2];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                unsigned hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (FA_ABL(4)) { hi[e] = __builtin_bit_cast(unsigned, st[8 * s2 + 2 * e]); lo[e] = __builtin_bit_cast(unsigned, st[8 * s2 + 2 * e + 1]); }
                    else fa_split2(fa_f32x2{st[8 * s2 + 2 * e], st[8 * s2 + 2 * e + 1]}, hi[e], lo[e]);
                }
                ph[s2] = __builtin_bit_cast(fa_f16x8, fa_u32x4{hi[0], hi[1], hi[2], hi[3]});
                pl[s2] = __builtin_bit_cast(fa_f16x8, fa_u32x4{lo[0], lo[1], lo[2], lo[3]});
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const int vrow = d * 32 + r32;
                    const unsigned char* vb = st_base + 2 * KPLANE + vrow * ROWB + (((kt * 4 + s2 * 2 + h) ^ ((vrow >> 1) & 7)) << 4);
                    const fa_f16x8 vh = *reinterpret_cast<const fa_f16x8*>(vb);
                    const fa_f16x8 vl = *reinterpret_cast<const fa_f16x8*>(vb + VPLANE);
                    if (FA_ABL(16)) { oacc[d][0] += (float)vh[0] + (float)vl[1] + (float)pl[s2][0] + (float)ph[s2][1]; continue; }
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[s2], oacc[d], 0, 0, 0);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[s2], oacc[d], 0, 0, 0);
                    oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[s2], oacc[d], 0, 0, 0);
                }
        }
        if (t + 1 < ntiles && !FA_ABL(1)) {
            publish_max(buf ^ 1);
            __syncthreads();
            tile_scales(buf ^ 1, sk_cur, sv_cur);
            store_tile(buf ^ 1, sk_cur, sv_cur);
        }
        if (!FA_ABL(32)) __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);          // 2^14 * sum of probabilities
    const float inv = fa_inv_pow2(sv_acc) / l_tot;                  // (oacc / (2^14 sv)) / (l_tot / 2^14)
    const int qrow = q0 + r32;
    if (p.o_scale) {
        // g8-packed rows: this lane holds channels d0..d0+3 of its query row, lane ^ 32 the other four of the same group of 8: the
        // lower half-wave collects the 16-byte hi chunk, the upper one the lo chunk (one exchange); both land at container d0
        if (head == 0 && h == 0 && qrow < p.Lq) p.o_scale[(int64_t)b * p.Lq + qrow] = out_scale;
        float* op = p.o + b * p.so + (int64_t)(qrow < p.Lq ? qrow : 0) * p.ldo + head * HD;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = d * 32 + 8 * g + 4 * h;
                unsigned h0, l0, h1, l1;
                fa_split2(fa_f32x2{oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv} * out_scale, h0, l0);
                fa_split2(fa_f32x2{oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv} * out_scale, h1, l1);
                const unsigned r0 = __shfl_xor(h ? h0 : l0, 32, 64), r1 = __shfl_xor(h ? h1 : l1, 32, 64);
                if (qrow < p.Lq) *reinterpret_cast<fa_u32x4*>(op + d0) = h ? fa_u32x4{r0, r1, l0, l1} : fa_u32x4{h0, h1, r0, r1};
            }
        return;
    }
    if (qrow < p.Lq) {
        float* op = p.o + b * p.so + (int64_t)qrow * p.ldo + head * HD;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = d * 32 + 8 * g + 4 * h;
                f32x4 o4 = {oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(op + d0) = o4;
            }
    }
}

// Same contract as psam_attention_f32; head_dim in {64, 128}.  a_scale != NULL: packed output (FlashArgs), o_scale [B*Lq] receives the row
// scales; o must then be 32-byte aligned with ldo % 8 == 0 and H*hd % 8 == 0.
PSAM_API int32_t psam_attention_f16x3_ex(const float* q, int64_t ldq, int64_t sq, const float* k, int64_t ldk, int64_t sk, const float* v, int64_t ldv,
                                         int64_t sv, float* o, int64_t ldo, int64_t so, int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd,
                                         float scale, const float* a_scale, float k1, float k2, float* o_scale, hipStream_t stream) {
    PSAM_REQUIRE(q && k && v && o, PSAM_EINVAL, "psam_attention_f16x3: null pointer");
    PSAM_REQUIRE((a_scale == nullptr) == (o_scale == nullptr), PSAM_EINVAL, "psam_attention_f16x3: packed output needs both a_scale and o_scale");
    PSAM_REQUIRE(!o_scale || ((ldo & 7) == 0 && ((uintptr_t)o & 31) == 0 && (so & 7) == 0 && Lq == Lk), PSAM_EINVAL,
                 "psam_attention_f16x3: packed output needs 32-byte aligned rows and self-attention (Lq == Lk)");
    PSAM_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0, PSAM_EINVAL, "psam_attention_f16x3: bad shape");
    PSAM_REQUIRE((int64_t)psam_cdiv(Lq, FA_BQ) * H * B < ((int64_t)1 << 31), PSAM_EINVAL, "psam_attention_f16x3: too many workgroups");
    PSAM_REQUIRE(((ldq | ldk | ldv | ldo | sq | sk | sv | so) & 3) == 0 && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0,
                 PSAM_EALIGN, "psam_attention_f16x3: strides must be multiples of 4 floats and pointers 16-byte aligned");
    PSAM_REQUIRE((int64_t)Lk * ldk < ((int64_t)1 << 29) && (int64_t)Lk * ldv < ((int64_t)1 << 29), PSAM_EINVAL,
                 "psam_attention_f16x3: one (batch) K/V slice must span < 2 GiB (32-bit buffer offsets)");
    FlashArgs p;
    p.q = q; p.k = k; p.v = v; p.o = o;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.sq = sq; p.sk = sk; p.sv = sv; p.so = so;
    p.H = H; p.Lq = Lq; p.Lk = Lk; p.B = B;
    p.scale_log2e = scale * 1.4426950408889634f;
    p.a_scale = a_scale; p.o_scale = o_scale; p.k1 = k1; p.k2 = k2;
#ifdef PSAM_ATTN_ABLATE
    p.abl = g_attn_abl;
#endif
    const dim3 grid((unsigned)(psam_cdiv(Lq, FA_BQ) * H * B)), block(256);      // 1-D over (query block, head, batch), see the kernel
    switch (hd) {
        case 64: hipLaunchKernelGGL((flash_attn_f16x3_kernel<64>), grid, block, 0, stream, p); break;
        case 128: hipLaunchKernelGGL((flash_attn_f16x3_kernel<128>), grid, block, 0, stream, p); break;
        default:
            psam_set_error("psam_attention_f16x3: head_dim must be 64 or 128");
            return PSAM_EINVAL;
    }
    return psam_launch_status("psam_attention_f16x3: launch failed");
}

PSAM_API int32_t psam_attention_f16x3(const float* q, int64_t ldq, int64_t sq, const float* k, int64_t ldk, int64_t sk, const float* v, int64_t ldv,
                                      int64_t sv, float* o, int64_t ldo, int64_t so, int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd,
                                      float scale, hipStream_t stream) {
    return psam_attention_f16x3_ex(q, ldq, sq, k, ldk, sk, v, ldv, sv, o, ldo, so, B, H, Lq, Lk, hd, scale, nullptr, 0.f, 0.f, nullptr, stream);
}
