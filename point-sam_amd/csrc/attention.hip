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
    int H, Lq, Lk;
    float scale_log2e;
};

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

    f32x4 rk[NF4], rv[NF4];
    auto load_tile = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < NF4; ++i) {
            const int f = i * 256 + tid;
            const int row = f / ROW4, c4 = f % ROW4;
            const int key = kv0 + row;
            const bool ok = (f < FA_BKV * ROW4) && key < p.Lk;
            rk[i] = ok ? *reinterpret_cast<const f32x4*>(K + (int64_t)key * p.ldk + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            rv[i] = ok ? *reinterpret_cast<const f32x4*>(V + (int64_t)key * p.ldv + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
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
    FlashArgs p;
    p.q = q; p.k = k; p.v = v; p.o = o;
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.sq = sq; p.sk = sk; p.sv = sv; p.so = so;
    p.H = H; p.Lq = Lq; p.Lk = Lk;
    p.scale_log2e = scale * 1.4426950408889634f;
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
