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
#include <utility>
#include <cstdlib>

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
    int hd;       // flash_attn_f16x3_kernel: real head dim <= the kernel's HD (% 8 == 0); channels hd .. HD-1 are read as zeros, never written
    // flash_attn_f16x3_kernel, few workgroups (one cloud): `ksplit` workgroups per (query block, head) share the key tiles; each parks its unnormalised
    // output, running maximum and sum (device-coherent stores), and the last arrival combines them in split order (fixed: results do not depend on timing)
    int ksplit; float* sk_part; int* sk_count;
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

// HD: head dim of the LDS layout, 64 or 128.  HDA <= HD (a multiple of 32): the channels that can be non-zero -- a head dim of 88 runs on the 128-wide layout
// (power-of-two index maps) but issues the MFMAs, fragment reads and conversions of 96 channels only (round 6: the products with the zero padding add exact
// zeros, so the results are the same bits as the full 128-channel instance; 6 of 8 k16 steps of S^T, 3 of 4 channel tiles of O^T).
template <int HD, int HDA = HD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(HD == 64 ? 2 : 1))) void flash_attn_f16x3_kernel(const FlashArgs p) {
    static_assert(HD == 64 || HD == 128, "head dim");
    static_assert(HDA % 32 == 0 && HDA > 0 && HDA <= HD, "active channels");
    constexpr int KS = HD / 16;                         // k16 steps of S^T = K Q^T
    constexpr int DT = HD / 32;                         // 32-channel tiles of O^T
    constexpr int KSA = HDA / 16, DTA = HDA / 32;       // ... that can hold non-zero channels
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
    const int split = p.ksplit > 1 ? (int)(blockIdx.x % (unsigned)p.ksplit) : 0, bid = p.ksplit > 1 ? (int)(blockIdx.x / (unsigned)p.ksplit) : (int)blockIdx.x;
    int qb, hb;
    if ((HB & 7) == 0) { const int id = bid, grp = id / (8 * nq), r = id - grp * 8 * nq; hb = grp * 8 + (r & 7); qb = r >> 3; }
    else { qb = bid % nq; hb = bid / nq; }
    const int head = hb % p.H, b = hb / p.H;
    const int q0 = qb * FA_BQ + wave * 32;
    const int hd = p.hd;      // == HD, or smaller (88 under HD = 128): the missing channels are zeros
    const float* Q = p.q + b * p.sq + head * hd;
    const float* K = p.k + b * p.sk + head * hd;
    const float* V = p.v + b * p.sv + head * hd;

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
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)K, 0, (int)((((int64_t)p.Lk - 1) * p.ldk + hd) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)V, 0, (int)((((int64_t)p.Lk - 1) * p.ldv + hd) * 4), 0x00020000);
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
            rk[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsK, (key < p.Lk && ((i * 256 + tid) % ROW4) * 4 < hd) ? koffs[i] : OOB, kbase, 0));
        }
#pragma unroll
        for (int i = 0; i < NF4; ++i) {                  // i = (pass, e): key = 4 * (vj + VJ * pass) + e
            const int key = kv0 + 4 * (vj + VJ * (i >> 2)) + (i & 3);
            rv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsV, (key < p.Lk && vc4 * 4 < hd) ? voffs[i] : OOB, vbase, 0));
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
            if (HDA < HD && c4 * 4 >= HDA) continue;      // padding channels: never read (the k16 steps beyond HDA are not issued)
            unsigned h0, l0, h1, l1;
            // K planes: row-major, chunk (c4>>1) of the row swizzled
            fa_split2(fa_f32x2{rk[i][0], rk[i][1]} * sk, h0, l0);
            fa_split2(fa_f32x2{rk[i][2], rk[i][3]} * sk, h1, l1);
            const int koff = row * KROW + ((((c4 >> 1) ^ ((row >> 1) & KSW)) << 4) | ((c4 & 1) << 3));
            *reinterpret_cast<fa_u32x2*>(st + koff) = fa_u32x2{h0, h1};
            *reinterpret_cast<fa_u32x2*>(st + KPLANE + koff) = fa_u32x2{l0, l1};
        }
        // V planes: transposed; keys 4g..4g+3 go to slots pos..pos+3 with pos = (4g & ~12) | swap of bits 2,3 (see header)
        if (HDA == HD || vc4 * 4 < HDA)
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

    const int ntiles_all = (p.Lk + FA_BKV - 1) / FA_BKV;
    const int t_first = p.ksplit > 1 ? (int)((int64_t)split * ntiles_all / p.ksplit) : 0;
    const int ntiles = p.ksplit > 1 ? (int)((int64_t)(split + 1) * ntiles_all / p.ksplit) : ntiles_all;      // this workgroup's key tiles: [t_first, ntiles)
    float sk_cur, sv_cur;
    load_tile(t_first * FA_BKV);      // in flight while the query rows (and the packed output's scale) are fetched: one memory round trip, not three
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
        for (int s = 0; s < KSA; ++s)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                t[s][e] = (ok && s * 16 + h * 8 + e * 4 < hd) ? *reinterpret_cast<const f32x4*>(qp + s * 16 + e * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(t[s][e][0]), fabsf(t[s][e][1]))), fmaxf(fabsf(t[s][e][2]), fabsf(t[s][e][3])));
            }
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        const float sq = f16_row_scale(amax);
        q_inv = fa_inv_pow2(sq);
#pragma unroll
        for (int s = 0; s < KSA; ++s) {
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
    for (int t = t_first; t < ntiles; ++t) {
        const int buf = (t - t_first) & 1;
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
            for (int s = 0; s < KSA; ++s) {
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
                for (int d = 0; d < DTA; ++d) {
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

    float l_tot = l_run + __shfl_xor(l_run, 32, 64);                // 2^14 * sum of probabilities
    if (p.ksplit > 1) {
        // ---- key-split combine.  Partials: the accumulators in units of V (raw MFMA layout: [register quad][thread] float4, coalesced), then
        // (m, l) per thread.  sc1 = device-coherent accesses (written through / not served from another XCD's stale L2 line), as in the GEMM's
        // split-K fix-up (gemm_f16x3p.hip).  The last arrival re-reads ALL partials, its own too, and adds them in split order.
        constexpr int SC1 = 16, QUADS = DT * 4, SPLIT_BYTES = (QUADS * 256 + 128) * 16;      // + 256 x (m, l) pairs
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.sk_part + (int64_t)bid * p.ksplit * (SPLIT_BYTES / 4)), 0, 0x7fffffff, 0x00020000);
        const float isv = fa_inv_pow2(sv_acc);
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const float a0 = oacc[d][4 * r4] * isv, a1 = oacc[d][4 * r4 + 1] * isv, a2 = oacc[d][4 * r4 + 2] * isv, a3 = oacc[d][4 * r4 + 3] * isv;
                const f32x4 vf = {a0, a1, a2, a3};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fa_u32x4, vf), rs, split * SPLIT_BYTES + ((d * 4 + r4) * 256 + tid) * 16, 0, SC1);
            }
        {
            const fa_f32x2 ml = {m_run, l_tot};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fa_u32x2, ml), rs, split * SPLIT_BYTES + QUADS * 256 * 16 + tid * 8, 0, SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(&smax[0][0][0]);
        if (tid == 0) *flag = (int)__hip_atomic_fetch_add(p.sk_count + bid, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*flag != p.ksplit - 1) return;
        if (tid == 0) __hip_atomic_store(p.sk_count + bid, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float ms[4], ls[4], m_all = -INFINITY;
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            ms[sp] = -INFINITY; ls[sp] = 0.f;
            if (sp < p.ksplit) {
                const fa_f32x2 ml = __builtin_bit_cast(fa_f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, sp * SPLIT_BYTES + QUADS * 256 * 16 + tid * 8, 0, SC1));
                ms[sp] = ml[0]; ls[sp] = ml[1];
                m_all = fmaxf(m_all, ml[0]);
            }
        }
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[d][r] = 0.f;
        l_tot = 0.f;
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) {
            if (sp < p.ksplit) {      // uniform
                const float w = __builtin_amdgcn_exp2f(ms[sp] - m_all);
                l_tot = __builtin_fmaf(w, ls[sp], l_tot);
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, sp * SPLIT_BYTES + ((d * 4 + r4) * 256 + tid) * 16, 0, SC1));
#pragma unroll
                        for (int e = 0; e < 4; ++e) oacc[d][4 * r4 + e] = __builtin_fmaf(w, v[e], oacc[d][4 * r4 + e]);
                    }
            }
        }
        sv_acc = 1.f;
    }
    const float inv = fa_inv_pow2(sv_acc) / l_tot;                  // (oacc / (2^14 sv)) / (l_tot / 2^14)
    const int qrow = q0 + r32;
    if (p.o_scale) {
        // g8-packed rows: this lane holds channels d0..d0+3 of its query row, lane ^ 32 the other four of the same group of 8: the
        // lower half-wave collects the 16-byte hi chunk, the upper one the lo chunk (one exchange); both land at container d0
        if (head == 0 && h == 0 && qrow < p.Lq) p.o_scale[(int64_t)b * p.Lq + qrow] = out_scale;
        float* op = p.o + b * p.so + (int64_t)(qrow < p.Lq ? qrow : 0) * p.ldo + head * hd;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = d * 32 + 8 * g + 4 * h;
                unsigned h0, l0, h1, l1;
                fa_split2(fa_f32x2{oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv} * out_scale, h0, l0);
                fa_split2(fa_f32x2{oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv} * out_scale, h1, l1);
                const unsigned r0 = __shfl_xor(h ? h0 : l0, 32, 64), r1 = __shfl_xor(h ? h1 : l1, 32, 64);
                if (qrow < p.Lq && d0 < hd) *reinterpret_cast<fa_u32x4*>(op + d0) = h ? fa_u32x4{r0, r1, l0, l1} : fa_u32x4{h0, h1, r0, r1};
            }
        return;
    }
    if (qrow < p.Lq) {
        float* op = p.o + b * p.so + (int64_t)qrow * p.ldo + head * hd;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = d * 32 + 8 * g + 4 * h;
                f32x4 o4 = {oacc[d][4 * g] * inv, oacc[d][4 * g + 1] * inv, oacc[d][4 * g + 2] * inv, oacc[d][4 * g + 3] * inv};
                if (d0 < hd) *reinterpret_cast<f32x4*>(op + d0) = o4;
            }
    }
}

// Key split (see FlashArgs.ksplit): the partial softmax states go to the caller's scratch (psam_attention_f16x3_keysplit_ws_bytes) and the workgroups of a
// (query block, head) count in through the caller's arrival-counter block (PSAM_COUNTER_BYTES, include/pointsam_hip.h: words PSAM_CNT_ATTN ..); the
// library allocates nothing and keeps no per-stream state.  PSAM_ATTN_KEYSPLIT=0 / psam_attention_f16x3_force_keysplit(0) switch the split off.
constexpr int64_t FA_SK_MAX_UNITS = PSAM_CNT_ATTN_N;
static int g_fa_keysplit = -1;
static bool fa_keysplit_enabled() {
    if (g_fa_keysplit >= 0) return g_fa_keysplit != 0;
    static int on = -1;
    if (on < 0) { const char* e = getenv("PSAM_ATTN_KEYSPLIT"); on = e ? (atoi(e) != 0) : 1; }
    return on != 0;
}
PSAM_API void psam_attention_f16x3_force_keysplit(int32_t mode) { g_fa_keysplit = mode; }
// A/B hook: PSAM_ATTN_FULL_WIDTH=1 runs head dims in (64, 96] on the full 128-channel instance (the round-5 kernel; same bits)
static bool fa_full_width() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("PSAM_ATTN_FULL_WIDTH"); on = e ? (atoi(e) != 0) : 0; }
    return on != 0;
}
// the split factor the launch would use, given unlimited scratch (0 / 1: unsplit)
static int fa_keysplit_factor(int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd, int32_t max_keysplit) {
    if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || hd <= 64) return 1;      // (head dim 64 at this size runs on the packed-operand kernel)
    const int64_t units = (int64_t)psam_cdiv(Lq, FA_BQ) * H * B;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    const int ntiles = (int)psam_cdiv(Lk, FA_BKV);
    int ks = fa_keysplit_enabled() ? (int)(ncu / units) : 1;
    if (ks > 4) ks = 4;
    if (ks > max_keysplit) ks = max_keysplit;
    if (ks > ntiles / 2) ks = ntiles / 2;
    return (ks > 1 && units <= FA_SK_MAX_UNITS) ? ks : 1;
}
static int64_t fa_split_bytes(int32_t hd) { return ((int64_t)((hd == 64 ? 64 : 128) / 32) * 4 * 256 + 128) * 16; }      // one workgroup's partial state
PSAM_API size_t psam_attention_f16x3_keysplit_ws_bytes(int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd, int32_t max_keysplit) {
    const int ks = fa_keysplit_factor(B, H, Lq, Lk, hd, max_keysplit);
    return ks > 1 ? (size_t)((int64_t)psam_cdiv(Lq, FA_BQ) * H * B * ks * fa_split_bytes(hd)) : 0;
}

// Same contract as psam_attention_f32; head_dim 64, or a multiple of 8 in (64, 128] (computed zero-padded to 128: the giant encoder's 88).  a_scale != NULL: packed output (FlashArgs), o_scale [B*Lq] receives the row
// scales; o must then be 32-byte aligned with ldo % 8 == 0 and H*hd % 8 == 0.
PSAM_API int32_t psam_attention_f16x3_ex2(const float* q, int64_t ldq, int64_t sq, const float* k, int64_t ldk, int64_t sk, const float* v, int64_t ldv,
                                          int64_t sv, float* o, int64_t ldo, int64_t so, int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd,
                                          float scale, const float* a_scale, float k1, float k2, float* o_scale, int32_t max_keysplit, void* ks_ws,
                                          size_t ks_ws_bytes, int32_t* counters, hipStream_t stream) {
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
    p.a_scale = a_scale; p.o_scale = o_scale; p.k1 = k1; p.k2 = k2; p.hd = hd;
#ifdef PSAM_ATTN_ABLATE
    p.abl = g_attn_abl;
#endif
    // few workgroups (one cloud: 16 heads x 4 query blocks = 64 on 256 CUs, each walking all key tiles in sequence): split the keys over up to four
    // workgroups per (query block, head); the last arrival combines the partial softmax states in the kernel (no second launch)
    p.ksplit = 1; p.sk_part = nullptr; p.sk_count = nullptr;
    const int64_t units = (int64_t)psam_cdiv(Lq, FA_BQ) * H * B;
    if (ks_ws && counters) {
        int ks = fa_keysplit_factor(B, H, Lq, Lk, hd, max_keysplit);
        while (ks > 1 && (size_t)(units * ks * fa_split_bytes(hd)) > ks_ws_bytes) --ks;
        if (ks > 1) {
            PSAM_REQUIRE(((uintptr_t)ks_ws & 15) == 0, PSAM_EALIGN, "psam_attention_f16x3: ks_ws must be 16-byte aligned");
            p.ksplit = ks; p.sk_part = static_cast<float*>(ks_ws); p.sk_count = counters + PSAM_CNT_ATTN;
        }
    }
    const dim3 grid((unsigned)(units * p.ksplit)), block(256);      // 1-D over (key split, query block, head, batch), see the kernel
    if (hd == 64) hipLaunchKernelGGL((flash_attn_f16x3_kernel<64>), grid, block, 0, stream, p);
    else if (hd > 64 && hd <= 96 && (hd & 7) == 0 && !fa_full_width()) hipLaunchKernelGGL((flash_attn_f16x3_kernel<128, 96>), grid, block, 0, stream, p);   // 128-wide layout, 96 active channels (the giant encoder's 88)
    else if (hd > 64 && hd <= 128 && (hd & 7) == 0) hipLaunchKernelGGL((flash_attn_f16x3_kernel<128>), grid, block, 0, stream, p);   // zero-padded to 128
    else {
        psam_set_error("psam_attention_f16x3: head_dim must be 64 or a multiple of 8 in (64, 128]");
        return PSAM_EINVAL;
    }
    return psam_launch_status("psam_attention_f16x3: launch failed");
}

PSAM_API int32_t psam_attention_f16x3_ex(const float* q, int64_t ldq, int64_t sq, const float* k, int64_t ldk, int64_t sk, const float* v, int64_t ldv,
                                         int64_t sv, float* o, int64_t ldo, int64_t so, int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd,
                                         float scale, const float* a_scale, float k1, float k2, float* o_scale, hipStream_t stream) {
    return psam_attention_f16x3_ex2(q, ldq, sq, k, ldk, sk, v, ldv, sv, o, ldo, so, B, H, Lq, Lk, hd, scale, a_scale, k1, k2, o_scale, 1, nullptr, 0, nullptr, stream);
}

PSAM_API int32_t psam_attention_f16x3(const float* q, int64_t ldq, int64_t sq, const float* k, int64_t ldk, int64_t sk, const float* v, int64_t ldv,
                                      int64_t sv, float* o, int64_t ldo, int64_t so, int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd,
                                      float scale, hipStream_t stream) {
    return psam_attention_f16x3_ex(q, ldq, sq, k, ldk, sk, v, ldv, sv, o, ldo, so, B, H, Lq, Lk, hd, scale, nullptr, 0.f, 0.f, nullptr, stream);
}

// ================================================================================================================
// Attention on PRE-PACKED operands ("f16x3", head dim 64): the qkv GEMM's epilogue has already written every row of q | k | v in the
// g8-packed hi|lo fp16 form of gemm_f16x3p.hip ([hi d0..7 : 16 B][lo d0..7 : 16 B] per 8 channels) with ONE power-of-two scale for
// the whole tensor (psam_gemm_fuse_t pack_out with k1 = 0: the scale comes from an a-priori bound of |q|, |k|, |v| -- Cauchy-Schwarz on
// the LayerNorm output and the weight row norms; a loose bound costs nothing here: the lo plane resolves 2^-24 of the scaled domain
// whatever the magnitude).  So nothing is converted, scaled or reduced per tile any more:
//   * Q fragments: eight 16-byte global loads per lane, straight from the packed rows (B operand of S^T = K Q^T);
//   * K and V tiles (64 keys x 256 B each): global -> LDS by LDS-DMA, double buffered, chunk-swizzled on the source address;
//   * K fragments: ds_read_b128 of one chunk (A operand: lane = key, 8 consecutive channels);
//   * V^T fragments (A operand of O^T = V^T P^T: lane = channel, 8 consecutive KEYS): V stays row-major in LDS and is transposed on
//     the way out by ds_read_b64_tr_b16 -- within a 16-lane group, lane i receives element i%4 of the 8 bytes that lanes i/4, 4+i/4,
//     8+i/4, 12+i/4 point at (probed on gfx950, scripts/exp/tr_probe.hip); source lane s therefore points at key s/4, channels
//     4(s%4)..+3 of the group's 16 channels, and two such reads fill the 8 k-slots {0-3, 8-11} (+4 for the upper half wave) that the
//     S^T accumulator layout dictates for P;
//   * one workgroup = 8 waves = 256 query rows of one (cloud, head): a K/V tile is fetched once for all of them;
//   * online softmax in the log2 domain per lane (= per query row), P produced scaled by 2^14 and split in registers.
// The output leaves g8-packed for the projection GEMM with the constant scale f16_row_scale(v_bound) (attention outputs are convex
// combinations of V rows).  47 -> see profiles/r03/r03_attention.txt.
struct PackedAttnArgs {
    const unsigned char* qkv;      // packed rows: [B * L][ld containers]; q at column 0, k at column D, v at column 2 D (containers)
    const float* sc;               // the rows' (common) scale: sc[b * L] is read
    float* o; float* o_scale;
    int64_t ld, ldo;               // containers (4 bytes) per row
    int H, L, B, D;
    float scale_log2e, v_bound;
#ifdef PSAM_ATTN_ABLATE
    int abl;      // 1 no DMA after the prologue, 2 no barriers, 4 no softmax arithmetic, 8 no S MFMAs, 16 no PV MFMAs, 32 no V reads, 64 no K reads
#endif
};

typedef short fa_s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef short fa_s16x8 __attribute__((ext_vector_type(8)));
// ds_read_b64_tr_b16 through the compiler's builtin: the read is counted and scheduled like any other LDS load
__device__ __forceinline__ fa_s16x4 fa_ds_read_tr16(const unsigned char* lds_ptr) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fa_s16x4*)(lds_ptr));
#else
    (void)lds_ptr; return fa_s16x4{0, 0, 0, 0};
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
#define FA_DMA16(rsrc, dst, voff, soff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst), 16, voff, soff, 0, 0)
#else
#define FA_DMA16(rsrc, dst, voff, soff) ((void)(rsrc), (void)(dst), (void)(voff), (void)(soff))
#endif

constexpr int PA_BQ = 256;     // query rows per workgroup (8 waves x 32); the NW = 4 instance (128 rows) serves grids that would leave CUs idle
#ifndef PA_PRIO
#define PA_PRIO 1
#endif
constexpr int PA_TILE = FA_BKV * 256;      // bytes of one K (or V) tile: 64 keys x 64 channels x 4 B

// RING: K/V tiles in LDS.  3 (96 KiB): tile t + 2 is issued at the barrier of tile t -- two tile times for the data, one workgroup per CU.
// 2 (64 KiB; NW = 4): tile t + 1 is issued at the barrier of tile t -- one tile time for the data, TWO independent workgroups per CU, whose
// phases (S^T products, softmax, PV products) drift apart and cover each other instead of idling the matrix pipe together (round 4).
// QB = 2 (round 6): every wave owns TWO 32-query blocks.  The K and V^T fragments of a sub-tile are read from LDS once and feed both blocks' products
// (half the LDS bytes per MFMA), and the softmax arithmetic of one block is independent of the other block's MFMAs: an in-order wave can put them under each
// other.  Four waves x 64 queries = 256-row workgroups, one per CU, up to 512 registers per lane.
template <int NW, int RING = 3, int QB = 1>
__global__ __launch_bounds__(64 * NW, RING == 2 ? 2 : 1) void flash_attn_packed_kernel(const PackedAttnArgs p) {
    constexpr int HD = 64, KS = 4, DT = 2, ROWB = 256;
    static_assert(RING == 3 || (RING == 2 && NW == 4), "ring of three tiles, or two tiles with four waves (two workgroups per CU)");
    constexpr int BQ = NW * 32 * QB, NPC = 32 / NW;      // query rows per workgroup; DMA pieces (of the 16 K + 16 V per tile) per wave
    static_assert(QB == 1 || (QB == 2 && NW == 4 && RING == 3), "two query blocks per wave: four waves, three-tile ring");
    static_assert(NW == 8 || NW == 4, "waves per workgroup");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [3][K tile | V tile]: ring of three tiles, ONE barrier per tile
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r32 = lane & 31, h = lane >> 5;
    const int nq = (p.L + BQ - 1) / BQ, HB = p.H * p.B;
    int qb, hb;        // consecutive workgroup ids go to different XCDs: the query blocks of one (cloud, head) share an XCD's L2
    if ((HB & 7) == 0) { const int id = blockIdx.x, grp = id / (8 * nq), r = id - grp * 8 * nq; hb = grp * 8 + (r & 7); qb = r >> 3; }
    else { qb = blockIdx.x % nq; hb = blockIdx.x / nq; }
    const int head = hb % p.H, b = hb / p.H;
    const int64_t rowb = p.ld * 4;                               // bytes per packed row
    const unsigned char* base = p.qkv + (int64_t)b * p.L * rowb + (int64_t)head * HD * 4;
    const float s_u = p.sc[(int64_t)b * p.L];                   // the tensor's scale (a power of two)

    // ---- K / V tiles by LDS-DMA: piece pc (1 KiB) = rows 4 pc .. 4 pc + 3; lane -> (row 4 pc + lane / 16, slot lane % 16) receives chunk
    // slot ^ f(row): f = row & 15 for K (16 different keys of a ds_read_b128 lane group on 16 different slots), (row & 1) | (row & 2) << 2
    // for V (the 4 keys x 4 channel quads of a transposing read on 16 different slots).  8 waves x 4 pieces = K tile + V tile.
    const int nt = (p.L + FA_BKV - 1) / FA_BKV;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    int dvoff[NPC];        // byte offset of this lane's chunk inside a tile's rows (row * rowb + column block + chunk * 16), tile base in soffset
    int dkey[NPC];
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int pc = wave * NPC + i;                           // 0..15: K pieces, 16..31: V pieces
        const bool isv = pc >= 16;
        const int row = (pc & 15) * 4 + (lane >> 4), slot = lane & 15;
        const int chunk = slot ^ (isv ? ((row & 1) | ((row & 2) << 2)) : (row & 15));
        dkey[i] = row;
        dvoff[i] = (isv ? 2 : 1) * p.D * 4 + chunk * 16;
    }
    auto issue_tile = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int pc = wave * NPC + i;
            int key = t * FA_BKV + dkey[i];
            key = key < p.L ? key : p.L - 1;                     // rows past the end are clamped (their scores are masked)
            unsigned char* dst = smem + buf * 2 * PA_TILE + (pc >= 16 ? PA_TILE : 0) + (pc & 15) * 1024;
            FA_DMA16(rs, dst, (int)((int64_t)key * rowb) + dvoff[i], 0);
        }
    };
    issue_tile(0, 0);
    if (RING == 3 && nt > 1) issue_tile(1, 1);

    // ---- this lane's query row: the hi / lo chunks of channels 16 s + 8 h .. + 7, as stored
    const int q0 = qb * BQ + wave * 32 * QB;
    int qrow[QB];
    fa_f16x8 qh[QB][KS], ql[QB][KS];
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        qrow[u] = q0 + 32 * u + r32;
        const unsigned char* qp = base + (int64_t)(qrow[u] < p.L ? qrow[u] : p.L - 1) * rowb;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            qh[u][s] = *reinterpret_cast<const fa_f16x8*>(qp + (2 * s + h) * 32);
            ql[u][s] = *reinterpret_cast<const fa_f16x8*>(qp + (2 * s + h) * 32 + 16);
        }
    }
    const float inv_su = fa_inv_pow2(s_u);
    const float c_s = p.scale_log2e * inv_su * inv_su;          // scaled-domain S -> log2-domain logits

    f32x16 oacc[QB][DT];
    float m_run[QB], l_run[QB];   // l_run sums the 2^14-scaled probabilities
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        m_run[u] = -INFINITY; l_run[u] = 0.f;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[u][d][r] = 0.f;
    }

    // transposing V reads: this lane as SOURCE lane s = lane & 15 of its group g = lane >> 4: key (s >> 2) of the quad of keys, channels
    // (g & 1) * 16 + 4 (s & 3) .. + 3 of the 32-channel tile; as RESULT lane it is channel (g & 1) * 16 + s = r32, half h = g >> 1
    const int tr_key = (lane & 15) >> 2, tr_d = ((lane >> 4) & 1) * 16 + 4 * (lane & 3);
    const int tr_sw = (tr_key & 1) | ((tr_key & 2) << 2);       // f(row) of key rows kbase + tr_key, kbase % 4 == 0

    // the two waves of a SIMD run the same S^T -> softmax -> PV sequence: with equal priority they take the matrix pipe at the same time and
    // leave it idle at the same time; a standing priority for one of them lets it run unimpeded while the other fills its softmax gaps
    if (PA_PRIO && NW == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    int buf = 0;
    for (int t = 0; t < (FA_ABL(128) ? 0 : nt); ++t) {
        if (RING == 3 && t + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPC) : "memory");      // this wave's pieces of tile t landed (tile t+1's may fly)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!FA_ABL(2)) __builtin_amdgcn_s_barrier();                          // tile t visible to every wave; every wave is done with tile t-1
        if (RING == 3) { if (t + 2 < nt && !FA_ABL(1)) issue_tile(t + 2, buf == 0 ? 2 : buf - 1); }      // ... whose buffer takes tile t+2
        else if (t + 1 < nt && !FA_ABL(1)) issue_tile(t + 1, buf ^ 1);
        const unsigned char* kt0 = smem + buf * 2 * PA_TILE;
        const unsigned char* vt0 = kt0 + PA_TILE;
        // ---- the tile's two 32-key sub-tiles as ONE straight-line block (the second sub-tile of a ragged last tile may lie wholly past the
        // end: its scores are masked to -inf, which the running maximum of the first sub-tile absorbs), every LDS read a compiler-visible
        // operation: the scheduler is free to put the S^T products of sub-tile 1 under the softmax arithmetic of sub-tile 0, and the PV
        // products of sub-tile 0 under the softmax of sub-tile 1 -- an in-order wave overlaps matrix and vector work only instruction by
        // instruction.
        f32x16 st[QB][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 sa[QB], sb[QB];        // two accumulation chains (k16 steps 0,1 and 2,3): consecutive MFMAs never wait for one another's result
#pragma unroll
            for (int u = 0; u < QB; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) { sa[u][r] = 0.f; sb[u][r] = 0.f; }
            const int krow = kt * 32 + r32;
            const unsigned char* kb = kt0 + krow * ROWB;
#pragma unroll
            for (int sp = 0; sp < KS / 2; ++sp) {
                fa_f16x8 kh[2], kl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = 2 * (2 * (sp + 2 * u) + h);
                    kh[u] = *reinterpret_cast<const fa_f16x8*>(kb + (((c) ^ (krow & 15)) << 4));
                    kl[u] = *reinterpret_cast<const fa_f16x8*>(kb + (((c + 1) ^ (krow & 15)) << 4));
                }
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    sa[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[0], ql[u][sp], sa[u], 0, 0, 0);
                    sb[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[1], ql[u][sp + 2], sb[u], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    sa[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[0], qh[u][sp], sa[u], 0, 0, 0);
                    sb[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl[1], qh[u][sp + 2], sb[u], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    sa[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[0], qh[u][sp], sa[u], 0, 0, 0);
                    sb[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh[1], qh[u][sp + 2], sb[u], 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < QB; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[u][kt][r] = sa[u][r] + sb[u][r];
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int key_base = t * FA_BKV + kt * 32;
            // ---- V^T fragments: transposing reads (see the header of this kernel)
            fa_f16x8 vh[2][DT], vl[2][DT];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    // keys kbase + (0..3) [k-slots 0-3] and kbase + 8 + (0..3) [k-slots 4-7], kbase = kt*32 + 16*s2 + 4*h; channels d*32 + tr_d ..
                    const int kbase = kt * 32 + 16 * s2 + 4 * h;
                    const int ch = d * 32 + tr_d;                                   // first of this source lane's 4 channels
                    const int chunk = 2 * (ch >> 3), inb = (ch & 7) * 2;
                    // f(row): kbase % 4 == 0, so the low two bits of the row are tr_key's; rows kbase + 8 + tr_key have the same low bits
                    const unsigned char* a0 = vt0 + (kbase + tr_key) * ROWB + inb;
                    const unsigned char* a1 = a0 + 8 * ROWB;
                    const fa_s16x4 h0 = fa_ds_read_tr16(a0 + (((chunk) ^ tr_sw) << 4)), h1 = fa_ds_read_tr16(a1 + (((chunk) ^ tr_sw) << 4));
                    const fa_s16x4 l0 = fa_ds_read_tr16(a0 + (((chunk + 1) ^ tr_sw) << 4)), l1 = fa_ds_read_tr16(a1 + (((chunk + 1) ^ tr_sw) << 4));
                    vh[s2][d] = __builtin_bit_cast(fa_f16x8, fa_s16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]});
                    vl[s2][d] = __builtin_bit_cast(fa_f16x8, fa_s16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]});
                }
            // ---- online softmax for this lane's query row(s) (keys (r&3)+8*(r>>2)+4*h of the sub-tile)
#pragma unroll
            for (int u = 0; u < QB; ++u) {
            if (key_base + 32 > p.L) {   // uniform: only a ragged last tile masks
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key_base + (r & 3) + 8 * (r >> 2) + 4 * h >= p.L) st[u][kt][r] = -INFINITY;
            }
            float mxa = fmaxf(fmaxf(st[u][kt][0], st[u][kt][1]), fmaxf(st[u][kt][2], st[u][kt][3])), mxb = fmaxf(fmaxf(st[u][kt][4], st[u][kt][5]), fmaxf(st[u][kt][6], st[u][kt][7]));
            float mxc = fmaxf(fmaxf(st[u][kt][8], st[u][kt][9]), fmaxf(st[u][kt][10], st[u][kt][11])), mxd = fmaxf(fmaxf(st[u][kt][12], st[u][kt][13]), fmaxf(st[u][kt][14], st[u][kt][15]));
            float mx = fmaxf(fmaxf(mxa, mxb), fmaxf(mxc, mxd));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c_s;        // c_s > 0: max commutes with the scaling
            const float m_new = fmaxf(m_run[u], mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_new);
            const float m14 = m_new - 14.f;
            float psum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[u][kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(st[u][kt][r], c_s, -m14));   // probability * 2^14, in [0, 2^14]
                psum[r & 3] += st[u][kt][r];
            }
            l_run[u] = l_run[u] * alpha + ((psum[0] + psum[1]) + (psum[2] + psum[3]));
            m_run[u] = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[u][d][r] *= alpha;
            }
            // ---- split P (registers 0-7 = k-slots of step 0, 8-15 = step 1) and O^T += V_sub^T P^T
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                unsigned hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) fa_split2(fa_f32x2{st[u][kt][8 * s2 + 2 * e], st[u][kt][8 * s2 + 2 * e + 1]}, hi[e], lo[e]);
                const fa_f16x8 ph = __builtin_bit_cast(fa_f16x8, fa_u32x4{hi[0], hi[1], hi[2], hi[3]});
                const fa_f16x8 pl = __builtin_bit_cast(fa_f16x8, fa_u32x4{lo[0], lo[1], lo[2], lo[3]});
#pragma unroll
                for (int d = 0; d < DT; ++d) oacc[u][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[s2][d], pl, oacc[u][d], 0, 0, 0);      // the two channel tiles alternate
#pragma unroll
                for (int d = 0; d < DT; ++d) oacc[u][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[s2][d], ph, oacc[u][d], 0, 0, 0);
#pragma unroll
                for (int d = 0; d < DT; ++d) oacc[u][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[s2][d], ph, oacc[u][d], 0, 0, 0);
            }
            }
        }
        buf = RING == 2 ? (buf ^ 1) : (buf == 2 ? 0 : buf + 1);
    }

    if (FA_ABL(256)) { if (l_run[0] == 123.f) p.o_scale[0] = oacc[0][0][0] + oacc[0][1][5]; return; }
    const float out_scale = f16_row_scale(p.v_bound);
#pragma unroll
    for (int u = 0; u < QB; ++u) {
        const float l_tot = l_run[u] + __shfl_xor(l_run[u], 32, 64);          // 2^14 * sum of probabilities
        const float inv = inv_su / l_tot;                               // (oacc / (2^14 s_u)) / (l_tot / 2^14)
        if (head == 0 && h == 0 && qrow[u] < p.L) p.o_scale[(int64_t)b * p.L + qrow[u]] = out_scale;
        float* op = p.o + ((int64_t)b * p.L + (qrow[u] < p.L ? qrow[u] : 0)) * p.ldo + head * HD;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = d * 32 + 8 * g + 4 * h;
                unsigned h0, l0, h1, l1;
                fa_split2(fa_f32x2{oacc[u][d][4 * g] * inv, oacc[u][d][4 * g + 1] * inv} * out_scale, h0, l0);
                fa_split2(fa_f32x2{oacc[u][d][4 * g + 2] * inv, oacc[u][d][4 * g + 3] * inv} * out_scale, h1, l1);
                const unsigned r0 = __shfl_xor(h ? h0 : l0, 32, 64), r1 = __shfl_xor(h ? h1 : l1, 32, 64);
                if (qrow[u] < p.L) *reinterpret_cast<fa_u32x4*>(op + d0) = h ? fa_u32x4{r0, r1, l0, l1} : fa_u32x4{h0, h1, r0, r1};
            }
    }
}

static int g_attn_variant = -1;
static int attn_variant_env() {
    static int v = -2;
    if (v == -2) { const char* e = getenv("PSAM_ATTN_VARIANT"); v = e ? atoi(e) : 1; }
    return v;
}
// tuning hook: -1 = default (environment PSAM_ATTN_VARIANT, else 1), 0 = one 256-row workgroup per CU on a three-tile ring, 1 = two 128-row
// workgroups per CU on a two-tile ring, 2 = 256-row workgroups of four waves with two query blocks per wave
PSAM_API void psam_attention_packed_force_variant(int32_t v) { g_attn_variant = v; }

// qkv: g8-packed rows [B * L, ld] (containers of 4 bytes: q | k | v column blocks of D = H * 64 each, one scale for all rows in
// sc[...]); o [B * L, ldo]: g8-packed attention output for the projection GEMM, o_scale [B * L] its (constant) row scales
// f16_row_scale(v_bound), v_bound >= max |v| (an a-priori bound; attention outputs are convex combinations of V rows).
PSAM_API int32_t psam_attention_packed(const void* qkv, int64_t ld, const float* sc, float* o, int64_t ldo, float* o_scale, int32_t B, int32_t H,
                                       int32_t L, int32_t hd, float scale, float v_bound, hipStream_t stream) {
    PSAM_REQUIRE(qkv && sc && o && o_scale, PSAM_EINVAL, "psam_attention_packed: null pointer");
    PSAM_REQUIRE(B > 0 && H > 0 && L > 0 && hd == 64, PSAM_EINVAL, "psam_attention_packed: bad shape (head_dim 64)");
    PSAM_REQUIRE(ld >= 3 * (int64_t)H * hd && (ld & 7) == 0 && (ldo & 7) == 0 && ldo >= (int64_t)H * hd && (((uintptr_t)qkv | (uintptr_t)o) & 31) == 0, PSAM_EALIGN,
                 "psam_attention_packed: packed rows must be 32-byte aligned, ld >= 3 H hd");
    PSAM_REQUIRE((int64_t)L * ld * 4 < ((int64_t)1 << 31), PSAM_EINVAL, "psam_attention_packed: one cloud's qkv slice must span < 2 GiB (32-bit buffer offsets)");
    PSAM_REQUIRE(v_bound > 0.f && (int64_t)psam_cdiv(L, PA_BQ) * H * B < ((int64_t)1 << 31), PSAM_EINVAL, "psam_attention_packed: bad bound / too many workgroups");
    PackedAttnArgs p;
    p.qkv = (const unsigned char*)qkv; p.sc = sc; p.o = o; p.o_scale = o_scale; p.ld = ld; p.ldo = ldo; p.H = H; p.L = L; p.B = B; p.D = H * hd;
    p.scale_log2e = scale * 1.4426950408889634f; p.v_bound = v_bound;
#ifdef PSAM_ATTN_ABLATE
    p.abl = g_attn_abl;
#endif
    constexpr int lds = 3 * 2 * PA_TILE;        // 96 KiB
    static unsigned long long attr_done = 0;      // > 64 KiB of dynamic LDS: opt in per device (both instances at once)
    int dev = 0, ncu = 256;
    PSAM_REQUIRE(hipGetDevice(&dev) == hipSuccess, PSAM_EINVAL, "psam_attention_packed: no device");
    {
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(__atomic_load_n(&attr_done, __ATOMIC_ACQUIRE) & bit)) {
            PSAM_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_packed_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess &&
                         hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_packed_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess &&
                         hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_packed_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * PA_TILE) == hipSuccess
#ifdef PSAM_BUILD_EXPERIMENTS
                         && hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_attn_packed_kernel<4, 3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds) == hipSuccess
#endif
                         ,
                         PSAM_EINVAL, "psam_attention_packed: cannot reserve LDS");
            __atomic_fetch_or(&attr_done, bit, __ATOMIC_RELEASE);
        }
    }
    static int force_nw = -1;      // tuning hook (environment, read once): PSAM_ATTN_PACKED_NW = 4 | 8
    if (force_nw < 0) { const char* e = getenv("PSAM_ATTN_PACKED_NW"); force_nw = e ? atoi(e) : 0; }
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
    // 256-row workgroups (two waves per SIMD cover each other's softmax gaps) unless they would leave CUs without work: one cloud of 2048 tokens
    // and 16 heads is 128 of them on 256 CUs (99 us); 128-row workgroups fill the chip
    const int64_t wg8 = (int64_t)psam_cdiv(L, PA_BQ) * H * B;
    const bool small = force_nw ? force_nw == 4 : (wg8 < ncu && L > PA_BQ / 2);
    // two 128-row workgroups per CU on a two-tile ring where the 256-row grid is about one workgroup per CU (B = 8 clouds x 16 heads x 512 tokens: 256)
    const int variant = g_attn_variant >= 0 ? g_attn_variant : attn_variant_env();
    const int64_t wg4 = (int64_t)psam_cdiv(L, PA_BQ / 2) * H * B;
#ifdef PSAM_BUILD_EXPERIMENTS
    // two query blocks per wave (QB = 2): 256-row workgroups of four waves, one per CU.  Bitwise equal to the other variants and SLOWER (44.3 vs 35.8 us at
    // B = 8, L = 512; 124.5 vs 109-121 us at B = 2, L = 2048 -- profiles/r06/r06_attn_qb2.txt): at 254 + 192 registers the compiler parks values in AGPRs (576
    // v_accvgpr moves per tile) and the rescale branches split the two blocks' instruction streams instead of interleaving them.  Experiments builds only.
    if (variant == 2 && !force_nw && L > PA_BQ / 2) {
        hipLaunchKernelGGL((flash_attn_packed_kernel<4, 3, 2>), dim3((unsigned)wg8), dim3(256), lds, stream, p);
        return psam_launch_status("psam_attention_packed: launch failed");
    }
#endif
    if (variant == 1 && !force_nw && L > PA_BQ / 2 && wg4 <= (int64_t)4 * ncu) {      // (a forced workgroup shape -- PSAM_ATTN_PACKED_NW -- names the kernel: it wins)
        hipLaunchKernelGGL((flash_attn_packed_kernel<4, 2>), dim3((unsigned)wg4), dim3(256), 2 * 2 * PA_TILE, stream, p);
        if (psam_ablate_repeat() & 1) hipLaunchKernelGGL((flash_attn_packed_kernel<4, 2>), dim3((unsigned)wg4), dim3(256), 2 * 2 * PA_TILE, stream, p);
        return psam_launch_status("psam_attention_packed: launch failed");
    }
    if (small) hipLaunchKernelGGL(flash_attn_packed_kernel<4>, dim3((unsigned)(psam_cdiv(L, PA_BQ / 2) * H * B)), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL(flash_attn_packed_kernel<8>, dim3((unsigned)wg8), dim3(512), lds, stream, p);
    return psam_launch_status("psam_attention_packed: launch failed");
}
