// Coarse C-ABI entry point: ONE EVA02 (SwiGLU) transformer block of the patch encoder -- timm's block as the reference calls it
// (pc_sam/model/pc_encoder.py:138-139: x = block(x), no rope):  x += proj(SDPA(q, k, v)(LN1 x));  x += fc2(LN(SiLU(fc1_g h) * fc1_x h)), h = LN2 x.
//
// psam_eva_block_prepare packs a block's weights ONCE (what the Python host did in PointCloudSAM._pack): q|k|v concatenated, fc1_g / fc1_x
// interleaved in 32-row blocks, fc2 pre-multiplied by the inner LayerNorm's gamma (the LayerNorm folded into the GEMM), every matrix split
// into g8-packed hi|lo fp16 with power-of-two row scales, and the a-priori bounds of the packed hand-overs.  psam_eva_block then runs the
// block as eight launches with nothing in between -- LayerNorm (packed output + per-row bound) | qkv GEMM (packed q|k|v) | attention on
// packed operands | projection + residual | LayerNorm | fc1 (SwiGLU gate, packed output, LayerNorm partials) | partial merge | fc2 with the
// folded LayerNorm + residual -- so that a caller in any language drives the dominant 83 % of the path's FLOPs with one call per layer.
// Host code only; every kernel is another entry point of this library.
#include <cmath>
#include <cstdlib>
#include <cstring>
#ifdef PSAM_BUILD_EXPERIMENTS
#include <map>
#include <mutex>
#endif
#include <utility>
#include <vector>
#include "common.h"      // brings in include/pointsam_hip.h

namespace {
constexpr int64_t align256(int64_t b) { return (b + 255) / 256 * 256; }
inline int kpad(int k) { return (k + 31) / 32 * 32; }

struct Carve {      // bump allocator over a caller-provided buffer
    char* base; int64_t off = 0;
    explicit Carve(void* p) : base(static_cast<char*>(p)) {}
    template <typename T> T* take(int64_t count) { T* r = reinterpret_cast<T*>(base + off); off += align256(count * (int64_t)sizeof(T)); return r; }
};

// A Linear on a few hundred to ~2000 rows (one or two clouds' patch tokens): the exact-fp32 32 x 64-tile kernel (psam_linear_rows_multi) beats the
// packing pass + packed-operand GEMM there (5-6 us against 5 + 8-10 us: tiles of 128 rows leave a handful of workgroups, all latency).
// PSAM_ROWS_MULTI=0 switches it off (A/B).
static bool rows_multi_enabled() {
    static int on = -1;
    if (on < 0) { const char* e = getenv("PSAM_ROWS_MULTI"); on = e ? (atoi(e) != 0) : 1; }
    return on != 0;
}
// The choice is made from the rows of ONE cloud (its G patch tokens), never from the batch: a cloud's results must not depend on how many clouds share
// the launch (bit-exact batch independence, tests/test_gpu_e2e.py::test_batch_independence_across_the_row_kernel_threshold).
static bool rows_multi_fits(int64_t rows_per_cloud, int K) { return rows_multi_enabled() && rows_per_cloud <= 2048 && (K & 15) == 0; }
static int32_t rows_linear(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* b, float* y, int64_t ldy, int64_t M, int N, int K, hipStream_t stream) {
    psam_skinny_jobs_t jobs;
    std::memset(&jobs, 0, sizeof(jobs));
    jobs.n = 1;
    jobs.job[0].x = x; jobs.job[0].W = w; jobs.job[0].bias = b; jobs.job[0].y = y; jobs.job[0].ldy = ldy; jobs.job[0].N = N;
    return psam_linear_rows_multi(&jobs, ldx, ldx, 1, 1, ldw, M, K, stream);
}


double row_norm(const float* w, int k) {
    double s = 0.0;
    for (int i = 0; i < k; ++i) s += (double)w[i] * (double)w[i];
    return std::sqrt(s);
}
}  // namespace

// bytes of the device blob psam_eva_block_prepare fills
PSAM_API size_t psam_eva_block_prepared_bytes(int32_t dim, int32_t hidden) {
    if (dim <= 0 || hidden <= 0) return 0;
    const int64_t D = dim, Dp = kpad(dim), Hp = (hidden + 31) / 32 * 32;
    int64_t b = 0;
    b += align256(3 * D * Dp * 4) + align256(3 * D * 4) + align256(3 * D * 4);      // wqkv packed, scales, bias
    b += align256(D * Dp * 4) + align256(D * 4);                                    // proj packed, scales
    b += align256(2 * Hp * Dp * 4) + align256(2 * Hp * 4) + align256(2 * Hp * 4);   // fc1 (g/x interleaved) packed, scales, bias
    b += align256(D * Hp * 4) + align256(D * 4) + align256(D * 4) + align256(D * 4);   // fc2 * gamma packed, scales, ln_c, ln_d
    return (size_t)b;
}

// Packs one block's weights into `prepared` (device, psam_eva_block_prepared_bytes) and fills `plan` (host).  Load-time call: it copies the
// weights to the host, builds the concatenated / interleaved / gamma-folded matrices and the bounds there in double precision, and uses a
// temporary device buffer for the fp32 staging of the largest matrix.  Synchronises `stream`.
PSAM_API int32_t psam_eva_block_prepare(const psam_eva_block_weights_t* wt, psam_eva_block_plan_t* plan, void* prepared, size_t prepared_bytes, hipStream_t stream) {
    PSAM_REQUIRE(wt && plan && prepared, PSAM_EINVAL, "psam_eva_block_prepare: null pointer");
    const int D = wt->dim, H = wt->hidden, heads = wt->heads;
    PSAM_REQUIRE(D > 0 && H > 0 && heads > 0 && D % heads == 0 && D / heads == 64 && D % 32 == 0, PSAM_EINVAL,
                 "psam_eva_block_prepare: head dim must be 64 and dim % 32 == 0 (the packed-operand attention)");
    const int Hp = (H + 31) / 32 * 32, Dp = kpad(D);
    PSAM_REQUIRE(Hp % 64 == 0, PSAM_EINVAL, "psam_eva_block_prepare: hidden rounded up to 32 must be a multiple of 64 (fused SwiGLU epilogue)");
    PSAM_REQUIRE(prepared_bytes >= psam_eva_block_prepared_bytes(D, H), PSAM_EWORKSPACE, "psam_eva_block_prepare: prepared buffer too small");
    const float* ptrs[] = {wt->norm1_w, wt->norm1_b, wt->q_w, wt->q_b, wt->k_w, wt->v_w, wt->v_b, wt->proj_w, wt->proj_b, wt->norm2_w, wt->norm2_b,
                           wt->fc1_g_w, wt->fc1_g_b, wt->fc1_x_w, wt->fc1_x_b, wt->mlp_norm_w, wt->mlp_norm_b, wt->fc2_w, wt->fc2_b};
    for (const float* p : ptrs) PSAM_REQUIRE(p, PSAM_EINVAL, "psam_eva_block_prepare: null weight pointer");
    PSAM_REQUIRE(hipStreamSynchronize(stream) == hipSuccess, PSAM_EINVAL, "psam_eva_block_prepare: stream error");      // the weights may still be arriving on `stream`

    auto fetch = [&](const float* dev, int64_t n) { std::vector<float> h((size_t)n); return hipMemcpy(h.data(), dev, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess ? h : std::vector<float>(); };
#define FETCH(name, dev, n) std::vector<float> name = fetch(dev, n); PSAM_REQUIRE((int64_t)name.size() == (int64_t)(n), PSAM_EINVAL, "psam_eva_block_prepare: cannot read a weight tensor")
    FETCH(qw, wt->q_w, (int64_t)D * D); FETCH(kw, wt->k_w, (int64_t)D * D); FETCH(vw, wt->v_w, (int64_t)D * D);
    FETCH(qb, wt->q_b, D); FETCH(vb, wt->v_b, D);
    FETCH(g1, wt->norm1_w, D); FETCH(b1n, wt->norm1_b, D);
    FETCH(gw, wt->fc1_g_w, (int64_t)H * D); FETCH(xw, wt->fc1_x_w, (int64_t)H * D); FETCH(gb, wt->fc1_g_b, H); FETCH(xb, wt->fc1_x_b, H);
    FETCH(mg, wt->mlp_norm_w, H); FETCH(mb, wt->mlp_norm_b, H);
    FETCH(w2, wt->fc2_w, (int64_t)D * H); FETCH(b2, wt->fc2_b, D);
#undef FETCH

    // ---- host-side matrices
    std::vector<float> wqkv((size_t)3 * D * D), bqkv((size_t)3 * D, 0.f);
    std::memcpy(wqkv.data(), qw.data(), (size_t)D * D * 4);
    std::memcpy(wqkv.data() + (size_t)D * D, kw.data(), (size_t)D * D * 4);
    std::memcpy(wqkv.data() + (size_t)2 * D * D, vw.data(), (size_t)D * D * 4);
    for (int i = 0; i < D; ++i) { bqkv[i] = qb[i]; bqkv[2 * D + i] = vb[i]; }      // k_proj has no bias (timm EvaAttention)
    // fc1: alternating 32-row blocks of fc1_g / fc1_x, hidden padded to Hp with zero rows (PSAM_ACT_SWIGLU)
    std::vector<float> w1((size_t)2 * Hp * D, 0.f), bias1((size_t)2 * Hp, 0.f);
    for (int n = 0; n < H; ++n) {
        const int64_t blk = n / 32, r = n % 32;
        std::memcpy(&w1[(size_t)((blk * 2) * 32 + r) * D], &gw[(size_t)n * D], (size_t)D * 4);
        std::memcpy(&w1[(size_t)((blk * 2 + 1) * 32 + r) * D], &xw[(size_t)n * D], (size_t)D * 4);
        bias1[(size_t)(blk * 2) * 32 + r] = gb[n];
        bias1[(size_t)(blk * 2 + 1) * 32 + r] = xb[n];
    }
    // fc2 with the inner LayerNorm folded in: fc2(LN(u)) = rstd (u (W2 gamma)^T - mean c) + d,  c = (W2 gamma) 1,  d = W2 beta + b2
    std::vector<float> w2g((size_t)D * Hp, 0.f), lnc(D), lnd(D);
    for (int n = 0; n < D; ++n) {
        double c = 0.0, d = (double)b2[n];
        for (int k = 0; k < H; ++k) {
            const double wg = (double)w2[(size_t)n * H + k] * (double)mg[k];
            w2g[(size_t)n * Hp + k] = (float)wg;
            c += wg;
            d += (double)w2[(size_t)n * H + k] * (double)mb[k];
        }
        lnc[n] = (float)c; lnd[n] = (float)d;
    }
    // ---- bounds: |W_n . h + b_n| <= ||W_n|| ||h|| + |b_n|;  a LayerNorm output has ||h|| <= max|gamma| sqrt(D) + ||beta||
    double gmax = 0.0, bnorm = 0.0;
    for (int i = 0; i < D; ++i) { gmax = std::fmax(gmax, std::fabs((double)g1[i])); bnorm += (double)b1n[i] * (double)b1n[i]; }
    const double hnorm = gmax * std::sqrt((double)D) + std::sqrt(bnorm);
    double nmax_all = 0.0, nmax_v = 0.0, bmax_all = 0.0, bmax_v = 0.0;
    for (int n = 0; n < 3 * D; ++n) {
        const double nn = row_norm(&wqkv[(size_t)n * D], D);
        nmax_all = std::fmax(nmax_all, nn); bmax_all = std::fmax(bmax_all, std::fabs((double)bqkv[n]));
        if (n >= 2 * D) { nmax_v = std::fmax(nmax_v, nn); bmax_v = std::fmax(bmax_v, std::fabs((double)bqkv[n])); }
    }
    // gated rows: |silu(g_n) x_n| <= (a_n t + b_n)(c_n t + d_n), t = ||h||_2 per row; coefficient-wise maxima over the hidden units
    double k2 = 0.0, k1 = 0.0, k0 = 0.0;
    for (int n = 0; n < H; ++n) {
        const double a = row_norm(&gw[(size_t)n * D], D), c = row_norm(&xw[(size_t)n * D], D), b = std::fabs((double)gb[n]), d = std::fabs((double)xb[n]);
        k2 = std::fmax(k2, a * c); k1 = std::fmax(k1, a * d + b * c); k0 = std::fmax(k0, b * d);
    }
    std::memset(plan, 0, sizeof(*plan));
    plan->dim = D; plan->heads = heads; plan->hidden = H; plan->hidden_pad = Hp; plan->eps = wt->eps;
    plan->qkv_bound = (float)(1.001 * (nmax_all * hnorm + bmax_all) + 1e-30);
    plan->v_bound = (float)(1.001 * (nmax_v * hnorm + bmax_v) + 1e-30);
    plan->u_c2 = (float)(1.002 * k2); plan->u_c1 = (float)(1.002 * k1); plan->u_c0 = (float)(1.002 * k0 + 1e-30);
    plan->norm1_w = wt->norm1_w; plan->norm1_b = wt->norm1_b; plan->norm2_w = wt->norm2_w; plan->norm2_b = wt->norm2_b; plan->proj_b = wt->proj_b;

    // ---- device blob: upload the fp32 matrix to a temporary, row scales + g8 packing by the library's own kernels
    Carve cv(prepared);
    float* p_wqkv = cv.take<float>((int64_t)3 * D * Dp); float* s_wqkv = cv.take<float>(3 * D); float* d_bqkv = cv.take<float>(3 * D);
    float* p_proj = cv.take<float>((int64_t)D * Dp); float* s_proj = cv.take<float>(D);
    float* p_w1 = cv.take<float>((int64_t)2 * Hp * Dp); float* s_w1 = cv.take<float>(2 * Hp); float* d_b1 = cv.take<float>(2 * Hp);
    float* p_w2g = cv.take<float>((int64_t)D * Hp); float* s_w2g = cv.take<float>(D); float* d_lnc = cv.take<float>(D); float* d_lnd = cv.take<float>(D);
    plan->o_wqkv = (char*)p_wqkv - cv.base; plan->o_sqkv = (char*)s_wqkv - cv.base; plan->o_bqkv = (char*)d_bqkv - cv.base;
    plan->o_wproj = (char*)p_proj - cv.base; plan->o_sproj = (char*)s_proj - cv.base;
    plan->o_w1 = (char*)p_w1 - cv.base; plan->o_s1 = (char*)s_w1 - cv.base; plan->o_b1 = (char*)d_b1 - cv.base;
    plan->o_w2g = (char*)p_w2g - cv.base; plan->o_s2g = (char*)s_w2g - cv.base; plan->o_lnc = (char*)d_lnc - cv.base; plan->o_lnd = (char*)d_lnd - cv.base;

    float* tmp = nullptr;
    const int64_t tmp_floats = std::max<int64_t>({(int64_t)3 * D * D, (int64_t)2 * Hp * D, (int64_t)D * Hp});
    PSAM_REQUIRE(hipMalloc(&tmp, (size_t)tmp_floats * 4) == hipSuccess, PSAM_EINVAL, "psam_eva_block_prepare: cannot allocate the staging buffer");
    int32_t rc = PSAM_OK;
    auto pack = [&](const float* host, const float* dev_src, int rows, int K, float* packed, float* scales) {
        if (rc != PSAM_OK) return;
        const float* src = dev_src;
        if (host) {
            if (hipMemcpyAsync(tmp, host, (size_t)rows * K * 4, hipMemcpyHostToDevice, stream) != hipSuccess) { psam_set_error("psam_eva_block_prepare: upload failed"); rc = PSAM_EINVAL; return; }
            src = tmp;
        }
        rc = psam_row_scale_f16(src, K, rows, K, scales, stream);
        if (rc == PSAM_OK) rc = psam_pack_rows_f16x2_g8(src, K, scales, rows, K, packed, kpad(K), stream);
        if (rc == PSAM_OK && hipStreamSynchronize(stream) != hipSuccess) { psam_set_error("psam_eva_block_prepare: packing failed"); rc = PSAM_EINVAL; }
    };
    pack(wqkv.data(), nullptr, 3 * D, D, p_wqkv, s_wqkv);
    pack(nullptr, wt->proj_w, D, D, p_proj, s_proj);
    pack(w1.data(), nullptr, 2 * Hp, D, p_w1, s_w1);
    pack(w2g.data(), nullptr, D, Hp, p_w2g, s_w2g);
    auto up = [&](float* dst, const std::vector<float>& src) {
        if (rc == PSAM_OK && hipMemcpy(dst, src.data(), src.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { psam_set_error("psam_eva_block_prepare: upload failed"); rc = PSAM_EINVAL; }
    };
    up(d_bqkv, bqkv); up(d_b1, bias1); up(d_lnc, lnc); up(d_lnd, lnd);
    hipFree(tmp);
    return rc;
}

// workspace of one psam_eva_block call on M token rows
PSAM_API size_t psam_eva_block_ws_bytes(int64_t M, int32_t dim, int32_t hidden) {
    if (M <= 0 || dim <= 0 || hidden <= 0) return 0;
    const int64_t D = dim, Dp = kpad(dim), Hp = (hidden + 31) / 32 * 32, segs = psam_gemm_f16x3p_stat_segs((int32_t)(2 * Hp));
    int64_t b = 0;
    b += align256(M * Dp * 4);            // h: LayerNorm output, packed
    b += 3 * align256(M * 4);             // its row scales, the gated rows' bound, attention output scales
    b += align256(M * 3 * D * 4);         // q | k | v packed
    b += align256(M * 4);                 // their scale
    b += align256(M * Dp * 4);            // attention output, packed
    b += align256(M * Hp * 4);            // gated hidden rows, packed
    b += align256(M * 4);                 // their scales
    b += align256(M * segs * 2 * 4);      // LayerNorm partials
    b += 2 * align256(M * 4);             // mean, rstd
    return (size_t)b;
}

// x [B*L, dim] (fp32, updated in place) <- block(x).  M = B * L must be a multiple of 256 (the fused GEMM epilogues work on whole tiles).
PSAM_API int32_t psam_eva_block(const psam_eva_block_plan_t* plan, const void* prepared, float* x, int32_t B, int32_t L, void* ws, size_t ws_bytes, hipStream_t stream) {
    PSAM_REQUIRE(plan && prepared && x && ws, PSAM_EINVAL, "psam_eva_block: null pointer");
    const int D = plan->dim, H = plan->hidden, Hp = plan->hidden_pad, heads = plan->heads, Dp = kpad(D);
    const int64_t M = (int64_t)B * L;
    PSAM_REQUIRE(B > 0 && L > 0 && M % 256 == 0 && M < ((int64_t)1 << 31), PSAM_EINVAL, "psam_eva_block: B * L must be a positive multiple of 256");
    PSAM_REQUIRE(ws_bytes >= psam_eva_block_ws_bytes(M, D, H), PSAM_EWORKSPACE, "psam_eva_block: workspace too small");
    const char* pb = static_cast<const char*>(prepared);
    auto P = [&](int64_t off) { return reinterpret_cast<const float*>(pb + off); };
    const int segs = psam_gemm_f16x3p_stat_segs(2 * Hp);
    Carve cv(ws);
    float* h = cv.take<float>(M * Dp);
    float* rs = cv.take<float>(M); float* ub = cv.take<float>(M); float* so = cv.take<float>(M);
    float* qkv = cv.take<float>(M * 3 * D);
    float* sq = cv.take<float>(M);
    float* o = cv.take<float>(M * Dp);
    float* u = cv.take<float>(M * Hp);
    float* su = cv.take<float>(M);
    float* st = cv.take<float>(M * segs * 2);
    float* mean = cv.take<float>(M); float* rstd = cv.take<float>(M);
    int32_t rc;
    // attention half
    rc = psam_layernorm_ex(x, D, nullptr, 0, plan->norm1_w, plan->norm1_b, h, Dp, M, D, plan->eps, 0, rs, 1, stream);
    if (rc) return rc;
    psam_gemm_fuse_t f;
    std::memset(&f, 0, sizeof(f));
    f.pack_out = 1; f.out_scale = sq; f.out_k1 = 0.f; f.out_k2 = plan->qkv_bound;
    rc = psam_gemm_f16x3p_ex(h, Dp, rs, P(plan->o_wqkv), Dp, P(plan->o_sqkv), qkv, 3 * D, P(plan->o_bqkv), nullptr, 0, nullptr, 0, 0, (int32_t)M, 3 * D, Dp, 1.f, 0, &f, stream);
    if (rc) return rc;
    rc = psam_attention_packed(qkv, 3 * D, sq, o, Dp, so, B, heads, L, D / heads, 1.0f / std::sqrt((float)(D / heads)), plan->v_bound, stream);
    if (rc) return rc;
    rc = psam_gemm_f16x3p_ex(o, Dp, so, P(plan->o_wproj), Dp, P(plan->o_sproj), x, D, plan->proj_b, x, D, nullptr, 0, 0, (int32_t)M, D, Dp, 1.f, 0, nullptr, stream);
    if (rc) return rc;
    // MLP half
    rc = psam_layernorm_ex2(x, D, nullptr, 0, plan->norm2_w, plan->norm2_b, h, Dp, M, D, plan->eps, 0, rs, 1, ub, plan->u_c2, plan->u_c1, plan->u_c0, stream);
    if (rc) return rc;
    std::memset(&f, 0, sizeof(f));
    f.pack_out = 1; f.out_scale = su; f.out_bound = ub; f.stats = st; f.stat_cols = H;
    rc = psam_gemm_f16x3p_ex(h, Dp, rs, P(plan->o_w1), Dp, P(plan->o_s1), u, Hp, P(plan->o_b1), nullptr, 0, nullptr, 0, 0, (int32_t)M, 2 * Hp, Dp, 1.f, PSAM_ACT_SWIGLU, &f, stream);
    if (rc) return rc;
    rc = psam_ln_stats_finalize(st, (int32_t)M, segs, H, plan->eps, mean, rstd, stream);
    if (rc) return rc;
    std::memset(&f, 0, sizeof(f));
    f.ln_mean = mean; f.ln_rstd = rstd; f.ln_c = P(plan->o_lnc);
    return psam_gemm_f16x3p_ex(u, Hp, su, P(plan->o_w2g), Hp, P(plan->o_s2g), x, D, P(plan->o_lnd), x, D, nullptr, 0, 0, (int32_t)M, D, Hp, 1.f, 0, &f, stream);
}

// ================================================================================================================================
// psam_eva_gelu_block: one block of the giant encoder (timm eva_giant_patch14_560: fused qkv with q / v bias, GELU MLP, head dim 88) -- the launch
// sequence of PointCloudSAM._block for `not vit.swiglu` (point_sam_amd/model.py), decisions included: the same split-K factors (the library's own
// psam_gemm_f16x3p_splitk), the same packed hand-overs, hence the same bits.
// ================================================================================================================================
PSAM_API size_t psam_eva_gelu_block_prepared_bytes(int32_t dim, int32_t hidden) {
    if (dim <= 0 || hidden <= 0) return 0;
    const int64_t D = dim, Dp = kpad(dim), H = hidden, Hp = kpad(hidden);
    return (size_t)(align256(3 * D * Dp * 4) + 2 * align256(3 * D * 4) + align256(D * Dp * 4) + align256(D * 4) + align256(H * Dp * 4) + align256(H * 4) +
                    align256(D * Hp * 4) + align256(D * 4));
}

PSAM_API int32_t psam_eva_gelu_block_prepare(const psam_eva_gelu_block_weights_t* wt, psam_eva_gelu_block_plan_t* plan, void* prepared, size_t prepared_bytes,
                                             hipStream_t stream) {
    PSAM_REQUIRE(wt && plan && prepared, PSAM_EINVAL, "psam_eva_gelu_block_prepare: null pointer");
    const int D = wt->dim, H = wt->hidden, heads = wt->heads;
    PSAM_REQUIRE(wt->precision == PSAM_PRECISION_F16X3, PSAM_EINVAL, "psam_eva_gelu_block_prepare: the coarse entry is built for PSAM_PRECISION_F16X3");
    PSAM_REQUIRE(D >= 256 && D <= 4096 && D % 32 == 0 && H > 0 && H % 32 == 0 && heads > 0 && D % heads == 0, PSAM_EINVAL,
                 "psam_eva_gelu_block_prepare: need 256 <= dim <= 4096, dim % 32 == 0, hidden % 32 == 0");
    const int hd = D / heads;
    PSAM_REQUIRE(hd == 64 || (hd > 64 && hd <= 128 && hd % 8 == 0), PSAM_EINVAL, "psam_eva_gelu_block_prepare: head dim must be 64 or a multiple of 8 in (64, 128]");
    PSAM_REQUIRE(prepared_bytes >= psam_eva_gelu_block_prepared_bytes(D, H), PSAM_EWORKSPACE, "psam_eva_gelu_block_prepare: prepared buffer too small");
    const float* ptrs[] = {wt->norm1_w, wt->norm1_b, wt->qkv_w, wt->q_bias, wt->v_bias, wt->proj_w, wt->proj_b, wt->norm2_w, wt->norm2_b, wt->fc1_w, wt->fc1_b, wt->fc2_w, wt->fc2_b};
    for (const float* q : ptrs) PSAM_REQUIRE(q, PSAM_EINVAL, "psam_eva_gelu_block_prepare: null weight pointer");
    PSAM_REQUIRE(hipStreamSynchronize(stream) == hipSuccess, PSAM_EINVAL, "psam_eva_gelu_block_prepare: stream error");
    auto fetch = [&](const float* dev, int64_t n) { std::vector<float> h((size_t)n); return hipMemcpy(h.data(), dev, (size_t)n * 4, hipMemcpyDeviceToHost) == hipSuccess ? h : std::vector<float>(); };
#define FETCH(name, dev, n) std::vector<float> name = fetch(dev, n); PSAM_REQUIRE((int64_t)name.size() == (int64_t)(n), PSAM_EINVAL, "psam_eva_gelu_block_prepare: cannot read a weight tensor")
    FETCH(vw, wt->qkv_w + (int64_t)2 * D * D, (int64_t)D * D); FETCH(qb, wt->q_bias, D); FETCH(vb, wt->v_bias, D);
    FETCH(w1, wt->fc1_w, (int64_t)H * D); FETCH(b1, wt->fc1_b, H);
#undef FETCH
    std::vector<float> bqkv((size_t)3 * D, 0.f);
    for (int i = 0; i < D; ++i) { bqkv[i] = qb[i]; bqkv[2 * D + i] = vb[i]; }
    // bound of |V| from the scale of the LayerNorm row that produced it (psam_attention_f16x3_ex): k1 = 2^15 sqrt(D) max ||W_v[n]||, k2 = max |b_v|
    double nv = 0.0, bv = 0.0, n1 = 0.0, bm1 = 0.0;
    for (int n = 0; n < D; ++n) { nv = std::fmax(nv, row_norm(&vw[(size_t)n * D], D)); bv = std::fmax(bv, std::fabs((double)vb[n])); }
    for (int n = 0; n < H; ++n) { n1 = std::fmax(n1, row_norm(&w1[(size_t)n * D], D)); bm1 = std::fmax(bm1, std::fabs((double)b1[n])); }
    std::memset(plan, 0, sizeof(*plan));
    plan->dim = D; plan->heads = heads; plan->hidden = H; plan->precision = wt->precision; plan->eps = wt->eps;
    plan->attn_keysplit = 4;
    plan->vk1 = (float)(32768.0 * std::sqrt((double)D) * nv); plan->vk2 = (float)bv;
    plan->u_c1 = (float)(1.002 * n1); plan->u_c0 = (float)(1.002 * bm1 + 1e-30);      // |GELU(W_n . h + b_n)| <= ||W_n|| t + |b_n|, t = ||h||_2
    plan->norm1_w = wt->norm1_w; plan->norm1_b = wt->norm1_b; plan->norm2_w = wt->norm2_w; plan->norm2_b = wt->norm2_b; plan->proj_b = wt->proj_b;
    plan->fc1_b = wt->fc1_b; plan->fc2_b = wt->fc2_b;
    const int Dp = kpad(D), Hp = kpad(H);
    Carve cv(prepared);
    float* p_wqkv = cv.take<float>((int64_t)3 * D * Dp); float* s_wqkv = cv.take<float>(3 * D); float* d_bqkv = cv.take<float>(3 * D);
    float* p_proj = cv.take<float>((int64_t)D * Dp); float* s_proj = cv.take<float>(D);
    float* p_w1 = cv.take<float>((int64_t)H * Dp); float* s_w1 = cv.take<float>(H);
    float* p_w2 = cv.take<float>((int64_t)D * Hp); float* s_w2 = cv.take<float>(D);
    plan->o_wqkv = (char*)p_wqkv - cv.base; plan->o_sqkv = (char*)s_wqkv - cv.base; plan->o_bqkv = (char*)d_bqkv - cv.base;
    plan->o_wproj = (char*)p_proj - cv.base; plan->o_sproj = (char*)s_proj - cv.base; plan->o_w1 = (char*)p_w1 - cv.base; plan->o_s1 = (char*)s_w1 - cv.base;
    plan->o_w2 = (char*)p_w2 - cv.base; plan->o_s2 = (char*)s_w2 - cv.base;
    int32_t rc = PSAM_OK;
    auto pack = [&](const float* src, int rows, int K, float* packed, float* scales) {
        if (rc != PSAM_OK) return;
        rc = psam_row_scale_f16(src, K, rows, K, scales, stream);
        if (rc == PSAM_OK) rc = psam_pack_rows_f16x2_g8(src, K, scales, rows, K, packed, kpad(K), stream);
    };
    pack(wt->qkv_w, 3 * D, D, p_wqkv, s_wqkv);
    pack(wt->proj_w, D, D, p_proj, s_proj);
    pack(wt->fc1_w, H, D, p_w1, s_w1);
    pack(wt->fc2_w, D, H, p_w2, s_w2);
    if (rc == PSAM_OK && hipMemcpy(d_bqkv, bqkv.data(), bqkv.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { psam_set_error("psam_eva_gelu_block_prepare: upload failed"); rc = PSAM_EINVAL; }
    if (rc == PSAM_OK && hipStreamSynchronize(stream) != hipSuccess) { psam_set_error("psam_eva_gelu_block_prepare: packing failed"); rc = PSAM_EINVAL; }
    return rc;
}

// scratch of the key-split attention inside psam_eva_gelu_block: the largest single-cloud case (4 splits x 64 units x 66 KiB = 16.5 MiB) with room to spare;
// a shape that would need more runs with a smaller split factor (psam_attention_f16x3_ex2 caps it to the scratch it is given)
constexpr int64_t GELU_KS_WS_BYTES = (int64_t)32 << 20;
PSAM_API size_t psam_eva_gelu_block_ws_bytes(int64_t M, int32_t dim, int32_t hidden) {
    if (M <= 0 || dim <= 0 || hidden <= 0) return 0;
    const int64_t D = dim, Dp = kpad(dim), Hp = kpad(hidden), widest = 3 * D > Hp ? 3 * D : Hp;
    int64_t b = 0;
    b += align256(M * Dp * 4) + 4 * align256(M * 4);      // h (packed LayerNorm rows), its scales, the fc1 bound, attention-output scales, fc2-input scales
    b += align256(M * 3 * D * 4);                         // qkv, fp32
    b += align256(M * Dp * 4);                            // attention output, packed
    b += 2 * align256(M * Hp * 4);                        // GELU(fc1) rows (fp32 or packed) and their packed copy
    b += align256(4 * M * widest * 4);                    // split-K partial planes (<= 4 per launch)
    b += GELU_KS_WS_BYTES;                                // key-split attention: partial softmax states (used by single-cloud shapes only)
    return (size_t)b;
}

PSAM_API int32_t psam_eva_gelu_block(const psam_eva_gelu_block_plan_t* plan, const void* prepared, float* x, int32_t B, int32_t L, void* ws, size_t ws_bytes,
                                     int32_t* counters, hipStream_t stream) {
    PSAM_REQUIRE(plan && prepared && x && ws, PSAM_EINVAL, "psam_eva_gelu_block: null pointer");
    const int D = plan->dim, H = plan->hidden, heads = plan->heads, Dp = kpad(D), Hp = kpad(H), hd = D / heads;
    const int64_t M = (int64_t)B * L;
    PSAM_REQUIRE(B > 0 && L > 0 && M >= 256 && M < ((int64_t)1 << 31), PSAM_EINVAL, "psam_eva_gelu_block: B * L must be at least 256 (the packed-operand GEMMs)");
    PSAM_REQUIRE(ws_bytes >= psam_eva_gelu_block_ws_bytes(M, D, H), PSAM_EWORKSPACE, "psam_eva_gelu_block: workspace too small");
    const char* pb = static_cast<const char*>(prepared);
    auto P = [&](int64_t off) { return reinterpret_cast<const float*>(pb + off); };
    const int64_t widest = 3 * (int64_t)D > Hp ? 3 * (int64_t)D : Hp;
    Carve cv(ws);
    float* h = cv.take<float>(M * Dp);
    float* rs = cv.take<float>(M); float* ub = cv.take<float>(M); float* so = cv.take<float>(M); float* sg = cv.take<float>(M);
    float* qkv = cv.take<float>(M * 3 * D);
    float* o = cv.take<float>(M * Dp);
    float* g = cv.take<float>(M * Hp); float* gp = cv.take<float>(M * Hp);
    float* planes = cv.take<float>(4 * M * widest);
    char* ks_ws = cv.take<char>(GELU_KS_WS_BYTES);
    // a plain GEMM as the host's ops.linear issues it: the library's split-K factor, partial planes in the workspace
    auto gemm = [&](const float* A, int64_t lda, const float* sA, int64_t ow, int64_t os, int N, int K, float* C, int64_t ldc, const float* bias, const float* res, int act) -> int32_t {
        psam_gemm_fuse_t f;
        std::memset(&f, 0, sizeof(f));
        const int ks = ((N & 3) == 0 && (ldc & 3) == 0) ? psam_gemm_f16x3p_splitk((int32_t)M, N, K, act) : 1;
        if (ks > 1) { f.splitk = ks; f.splitk_ws = planes; f.splitk_plane = M * (int64_t)N; f.counters = counters; }
        return psam_gemm_f16x3p_ex(A, lda, sA, P(ow), K, P(os), C, ldc, bias, res, res ? ldc : 0, nullptr, 0, 0, (int32_t)M, N, K, 1.f, act, ks > 1 ? &f : nullptr, stream);
    };
    int32_t rc;
    // attention half
    rc = psam_layernorm_ex(x, D, nullptr, 0, plan->norm1_w, plan->norm1_b, h, Dp, M, D, plan->eps, 0, rs, 1, stream);
    if (rc) return rc;
    rc = gemm(h, Dp, rs, plan->o_wqkv, plan->o_sqkv, 3 * D, Dp, qkv, 3 * D, P(plan->o_bqkv), nullptr, 0);
    if (rc) return rc;
    rc = psam_attention_f16x3_ex2(qkv, 3 * D, (int64_t)L * 3 * D, qkv + D, 3 * D, (int64_t)L * 3 * D, qkv + 2 * D, 3 * D, (int64_t)L * 3 * D, o, Dp, (int64_t)L * Dp, B, heads, L, L,
                                  hd, (float)std::pow((double)hd, -0.5), rs, plan->vk1, plan->vk2, so, plan->attn_keysplit > 0 ? plan->attn_keysplit : 1, ks_ws, (size_t)GELU_KS_WS_BYTES, counters, stream);      // float(hd ** -0.5), as the host computes it
    if (rc) return rc;
    rc = gemm(o, Dp, so, plan->o_wproj, plan->o_sproj, D, Dp, x, D, plan->proj_b, x, 0);
    if (rc) return rc;
    // MLP half
    const bool fused_gelu = (M % 256 == 0) && (H % 128 == 0) && psam_gemm_f16x3p_splitk((int32_t)M, H, Dp, PSAM_ACT_GELU) == 1;
    if (fused_gelu) {
        rc = psam_layernorm_ex2(x, D, nullptr, 0, plan->norm2_w, plan->norm2_b, h, Dp, M, D, plan->eps, 0, rs, 1, ub, 0.f, plan->u_c1, plan->u_c0, stream);
        if (rc) return rc;
        psam_gemm_fuse_t f;
        std::memset(&f, 0, sizeof(f));
        f.pack_out = 1; f.out_scale = sg; f.out_bound = ub;
        rc = psam_gemm_f16x3p_ex(h, Dp, rs, P(plan->o_w1), Dp, P(plan->o_s1), g, Hp, plan->fc1_b, nullptr, 0, nullptr, 0, 0, (int32_t)M, H, Dp, 1.f, PSAM_ACT_GELU, &f, stream);
        if (rc) return rc;
        return gemm(g, Hp, sg, plan->o_w2, plan->o_s2, D, Hp, x, D, plan->fc2_b, x, 0);
    }
    rc = psam_layernorm_ex(x, D, nullptr, 0, plan->norm2_w, plan->norm2_b, h, Dp, M, D, plan->eps, 0, rs, 1, stream);
    if (rc) return rc;
    rc = gemm(h, Dp, rs, plan->o_w1, plan->o_s1, H, Dp, g, H, plan->fc1_b, nullptr, PSAM_ACT_GELU);
    if (rc) return rc;
    rc = psam_scale_pack_rows_g8(g, H, (int32_t)M, H, gp, Hp, sg, stream);
    if (rc) return rc;
    return gemm(gp, Hp, sg, plan->o_w2, plan->o_s2, D, Hp, x, D, plan->fc2_b, x, 0);
}

// ================================================================================================================================
// psam_patch_encoder: PatchEncoder.forward on kNN groups (pc_sam/model/common.py:477-506 after KNNGrouper / group_with_centers_and_knn,
// :99-120 / :126-187) -- the mini-PointNet of the patch embedding (features = rgb) and of the mask encoder (features = mask logits) -- as the
// Python host runs it in "f16x3": gather + Linear + LayerNorm + GELU in one kernel (packed rows) | conv1.3 GEMM with the group maximum and a
// packed output in its epilogue | the pooled half of conv2.0 once per group | conv2.0 on the rows with that as a row bias | LayerNorm + GELU
// (packed) | conv2.3 GEMM whose epilogue keeps only the group maximum.  hidden_dims[0] == 128, group size 32 or a multiple of 64
// (above 64: the epilogues pool 64-row parts, psam_group_max pools the parts), rows % 256 == 0.
// ================================================================================================================================
PSAM_API size_t psam_patch_encoder_prepared_bytes(int32_t h0, int32_t h1, int32_t cout) {
    if (h0 <= 0 || h1 <= 0 || cout <= 0) return 0;
    const int64_t a = kpad(h0), b = kpad(h1);
    return (size_t)(align256((int64_t)h0 * a * 4) + align256(h0 * 4) + 2 * (align256((int64_t)h1 * a * 4) + align256(h1 * 4)) + align256((int64_t)cout * b * 4) + align256(cout * 4));
}

PSAM_API int32_t psam_patch_encoder_prepare(const psam_patch_encoder_weights_t* wt, psam_patch_encoder_plan_t* plan, void* prepared, size_t prepared_bytes,
                                            hipStream_t stream) {
    PSAM_REQUIRE(wt && plan && prepared, PSAM_EINVAL, "psam_patch_encoder_prepare: null pointer");
    const int h0 = wt->h0, h1 = wt->h1, cout = wt->cout, cin = wt->cin;
    PSAM_REQUIRE(h0 == 128 && h1 >= 256 && h1 % 128 == 0 && h1 <= 4096 && cout >= 128 && cout % 128 == 0 && cin >= 4, PSAM_EINVAL,
                 "psam_patch_encoder_prepare: hidden_dims[0] must be 128, hidden_dims[1] a multiple of 128 in [256, 4096], out_channels a multiple of 128");
    PSAM_REQUIRE(wt->c10_w && wt->c10_b && wt->c11_w && wt->c11_b && wt->c13_w && wt->c13_b && wt->c20_w && wt->c20_b && wt->c21_w && wt->c21_b && wt->c23_w && wt->c23_b,
                 PSAM_EINVAL, "psam_patch_encoder_prepare: null weight pointer");
    PSAM_REQUIRE(prepared_bytes >= psam_patch_encoder_prepared_bytes(h0, h1, cout), PSAM_EWORKSPACE, "psam_patch_encoder_prepare: prepared buffer too small");
    std::vector<float> w13((size_t)h0 * h0), b13(h0);
    PSAM_REQUIRE(hipStreamSynchronize(stream) == hipSuccess, PSAM_EINVAL, "psam_patch_encoder_prepare: stream error");      // the weights may still be arriving on `stream`
    PSAM_REQUIRE(hipMemcpy(w13.data(), wt->c13_w, w13.size() * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(b13.data(), wt->c13_b, b13.size() * 4, hipMemcpyDeviceToHost) == hipSuccess,
                 PSAM_EINVAL, "psam_patch_encoder_prepare: cannot read conv1.3");
    double nmax = 0.0, bmax = 0.0;
    for (int n = 0; n < h0; ++n) { nmax = std::fmax(nmax, row_norm(&w13[(size_t)n * h0], h0)); bmax = std::fmax(bmax, std::fabs((double)b13[n])); }
    std::memset(plan, 0, sizeof(*plan));
    plan->cin = cin; plan->h0 = h0; plan->h1 = h1; plan->cout = cout; plan->eps = wt->eps;
    plan->k1 = (float)(32768.0 * std::sqrt((double)h0) * nmax);      // |conv1.3 row| <= k1 / scale(input row) + k2 (psam_gemm_fuse_t)
    plan->k2 = (float)bmax;
    {   // conv2.1's LayerNorm output: |(x - mean) rstd gamma + beta| <= max |gamma| sqrt(h1) + max |beta| (GELU only shrinks it)
        std::vector<float> g21(h1), b21(h1);
        PSAM_REQUIRE(hipMemcpy(g21.data(), wt->c21_w, (size_t)h1 * 4, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(b21.data(), wt->c21_b, (size_t)h1 * 4, hipMemcpyDeviceToHost) == hipSuccess,
                     PSAM_EINVAL, "psam_patch_encoder_prepare: cannot read conv2.1");
        double gm = 0.0, bm = 0.0;
        for (int n = 0; n < h1; ++n) { gm = std::fmax(gm, std::fabs((double)g21[n])); bm = std::fmax(bm, std::fabs((double)b21[n])); }
        plan->ln21_bound = (float)(1.001 * (gm * std::sqrt((double)h1) + bm) + 1e-30);
    }
    plan->c10_w = wt->c10_w; plan->c10_b = wt->c10_b; plan->c11_w = wt->c11_w; plan->c11_b = wt->c11_b; plan->c13_b = wt->c13_b; plan->c20_w = wt->c20_w;
    plan->c20_b = wt->c20_b; plan->c21_w = wt->c21_w; plan->c21_b = wt->c21_b; plan->c23_b = wt->c23_b;
    Carve cv(prepared);
    const int a = kpad(h0), b = kpad(h1);
    float* p13 = cv.take<float>((int64_t)h0 * a); float* s13 = cv.take<float>(h0);
    float* p20m = cv.take<float>((int64_t)h1 * a); float* s20m = cv.take<float>(h1);
    float* p20x = cv.take<float>((int64_t)h1 * a); float* s20x = cv.take<float>(h1);
    float* p23 = cv.take<float>((int64_t)cout * b); float* s23 = cv.take<float>(cout);
    plan->o_w13 = (char*)p13 - cv.base; plan->o_s13 = (char*)s13 - cv.base; plan->o_w20m = (char*)p20m - cv.base; plan->o_s20m = (char*)s20m - cv.base;
    plan->o_w20x = (char*)p20x - cv.base; plan->o_s20x = (char*)s20x - cv.base; plan->o_w23 = (char*)p23 - cv.base; plan->o_s23 = (char*)s23 - cv.base;
    int32_t rc = psam_row_scale_f16(wt->c13_w, h0, h0, h0, s13, stream);
    if (!rc) rc = psam_pack_rows_f16x2_g8(wt->c13_w, h0, s13, h0, h0, p13, a, stream);
    // conv2.0 [h1, 2 h0] = [pooled half | per-row half]: cat([max, x]) W^T = max W[:, :h0]^T + x W[:, h0:]^T, each half packed with its own row scales
    if (!rc) rc = psam_row_scale_f16(wt->c20_w, 2 * h0, h1, h0, s20m, stream);
    if (!rc) rc = psam_pack_rows_f16x2_g8(wt->c20_w, 2 * h0, s20m, h1, h0, p20m, a, stream);
    if (!rc) rc = psam_row_scale_f16(wt->c20_w + h0, 2 * h0, h1, h0, s20x, stream);
    if (!rc) rc = psam_pack_rows_f16x2_g8(wt->c20_w + h0, 2 * h0, s20x, h1, h0, p20x, a, stream);
    if (!rc) rc = psam_row_scale_f16(wt->c23_w, h1, cout, h1, s23, stream);
    if (!rc) rc = psam_pack_rows_f16x2_g8(wt->c23_w, h1, s23, cout, h1, p23, b, stream);
    if (!rc && hipStreamSynchronize(stream) != hipSuccess) { psam_set_error("psam_patch_encoder_prepare: packing failed"); rc = PSAM_EINVAL; }
    return rc;
}

PSAM_API size_t psam_patch_encoder_ws_bytes(int64_t rows, int64_t groups, int32_t h0, int32_t h1) {
    if (rows <= 0 || groups <= 0 || h0 <= 0 || h1 <= 0) return 0;
    const int64_t a = kpad(h0), b = kpad(h1);
    return (size_t)(2 * align256(rows * a * 4) + 3 * align256(rows * 4) + 2 * align256(groups * a * 4) + align256(groups * 4) + align256(groups * h1 * 4) + align256(rows * b * 4));
}

// out [B*rep*G, cout] = max over the K group members of conv2(cat(max conv1(x), conv1(x))), x = [rel. xyz (/ radius) | feats (| feats - centre feats)]
PSAM_API int32_t psam_patch_encoder(const psam_patch_encoder_plan_t* plan, const void* prepared, const float* xyz, const float* feats, const float* centers,
                                    const int64_t* knn_idx, const int64_t* center_idx, int32_t B, int32_t rep, int32_t N, int32_t G, int32_t K, int32_t C,
                                    float radius, float* out, void* ws, size_t ws_bytes, hipStream_t stream) {
    PSAM_REQUIRE(plan && prepared && xyz && feats && centers && knn_idx && out && ws, PSAM_EINVAL, "psam_patch_encoder: null pointer");
    const int h0 = plan->h0, h1 = plan->h1, cout = plan->cout, a = kpad(h0), b = kpad(h1);
    const int64_t groups = (int64_t)B * rep * G, rows = groups * K;
    PSAM_REQUIRE(B > 0 && rep > 0 && N > 0 && G > 0 && (K == 32 || (K > 0 && K % 64 == 0)) && rows % 256 == 0 && rows < ((int64_t)1 << 31), PSAM_EINVAL,
                 "psam_patch_encoder: group size must be 32 or a multiple of 64, and B * rep * G * K a multiple of 256");
    const int Kp = K <= 64 ? K : 64, parts = K / Kp;      // the GEMM epilogues pool 64-row parts of a larger group; psam_group_max pools the parts
    PSAM_REQUIRE(plan->cin == 3 + C * (center_idx ? 2 : 1), PSAM_EINVAL, "psam_patch_encoder: feature channels do not match the first Linear");
    PSAM_REQUIRE(ws_bytes >= psam_patch_encoder_ws_bytes(rows, groups, h0, h1), PSAM_EWORKSPACE, "psam_patch_encoder: workspace too small");
    const char* pb = static_cast<const char*>(prepared);
    auto P = [&](int64_t off) { return reinterpret_cast<const float*>(pb + off); };
    Carve cv(ws);
    float* x1 = cv.take<float>(rows * a); float* x2 = cv.take<float>(rows * a);
    float* s1 = cv.take<float>(rows); float* s2 = cv.take<float>(rows); float* rs = cv.take<float>(rows);
    float* y1 = cv.take<float>(groups * a); float* y1p = cv.take<float>(groups * a); float* sy = cv.take<float>(groups);
    float* g1 = cv.take<float>(groups * h1);
    float* x3 = cv.take<float>(rows * b);
    float* part1 = parts > 1 ? x3 : y1;                     // [groups * parts, h0] partial maxima of conv1.3 (x3 is free until conv2.0 writes it)
    float* part2 = parts > 1 ? x1 : out;                    // [groups * parts, cout] partial maxima of conv2.3 (x1 is free after conv1.3)
    PSAM_REQUIRE(parts == 1 || ((int64_t)groups * parts * h0 <= rows * b && (int64_t)groups * parts * cout <= rows * a), PSAM_EINVAL,
                 "psam_patch_encoder: partial maxima do not fit the workspace");
    int32_t rc = psam_patch_l1_ex(xyz, feats, centers, knn_idx, center_idx, plan->c10_w, plan->c10_b, plan->c11_w, plan->c11_b, plan->eps, B, rep, N, G, K, C, radius, x1, s1, stream);
    if (rc) return rc;
    psam_gemm_fuse_t f;
    std::memset(&f, 0, sizeof(f));
    f.pack_out = 1; f.out_scale = s2; f.out_k1 = plan->k1; f.out_k2 = plan->k2; f.gmax_out = part1; f.gmax_ld = h0; f.gmax_k = Kp;
    rc = psam_gemm_f16x3p_ex(x1, a, s1, P(plan->o_w13), a, P(plan->o_s13), x2, a, plan->c13_b, nullptr, 0, nullptr, 0, 0, (int32_t)rows, h0, a, 1.f, 0, &f, stream);
    if (rc) return rc;
    if (parts > 1) { rc = psam_group_max(part1, h0, y1, h0, groups, parts, h0, stream); if (rc) return rc; }
    if (rows_multi_fits(G, h0)) {      // the pooled half of conv2.0, one row per group
        rc = rows_linear(y1, h0, plan->c20_w, 2 * h0, plan->c20_b, g1, h1, groups, h1, h0, stream);
    } else if (groups >= 256) {
        rc = psam_scale_pack_rows_g8(y1, h0, (int32_t)groups, h0, y1p, a, sy, stream);
        if (!rc) rc = psam_gemm_f16x3p_ex(y1p, a, sy, P(plan->o_w20m), a, P(plan->o_s20m), g1, h1, plan->c20_b, nullptr, 0, nullptr, 0, 0, (int32_t)groups, h1, a, 1.f, 0, nullptr, stream);
    } else if (groups <= 64) {      // the host's dispatch (point_sam_amd/ops.py linear): a handful of rows -> the skinny kernel
        rc = psam_linear_skinny(y1, h0, plan->c20_w, 2 * h0, plan->c20_b, nullptr, 0, g1, h1, (int32_t)groups, h1, h0, 0, stream);
    } else {
        rc = psam_linear(y1, h0, plan->c20_w, 2 * h0, plan->c20_b, nullptr, 0, g1, h1, (int32_t)groups, h1, h0, 0, stream);
    }
    if (rc) return rc;
    if (h1 == 512 && rows % 128 == 0 && psam_gemm_f16x3p_fused_row_ln(h1)) {
        // conv2.0 -> LayerNorm -> GELU in ONE GEMM on full-row 128x512 tiles (register epilogue): packed rows out, scaled by the LayerNorm's a-priori bound
        std::memset(&f, 0, sizeof(f));
        f.row_ln_g = plan->c21_w; f.row_ln_b = plan->c21_b; f.row_ln_eps = plan->eps; f.pack_out = 1; f.out_scale = rs; f.out_k1 = 0.f; f.out_k2 = plan->ln21_bound;
        rc = psam_gemm_f16x3p_ex(x2, a, s2, P(plan->o_w20x), a, P(plan->o_s20x), x3, b, nullptr, nullptr, 0, g1, h1, K, (int32_t)rows, h1, a, 1.f, PSAM_ACT_GELU, &f, stream);
        if (rc) return rc;
    } else {
        rc = psam_gemm_f16x3p_ex(x2, a, s2, P(plan->o_w20x), a, P(plan->o_s20x), x3, h1, nullptr, nullptr, 0, g1, h1, K, (int32_t)rows, h1, a, 1.f, 0, nullptr, stream);
        if (rc) return rc;
        rc = psam_layernorm_ex(x3, h1, nullptr, 0, plan->c21_w, plan->c21_b, x3, b, rows, h1, plan->eps, PSAM_ACT_GELU, rs, 1, stream);
        if (rc) return rc;
    }
    std::memset(&f, 0, sizeof(f));
    f.gmax_out = part2; f.gmax_ld = cout; f.gmax_k = Kp; f.no_store = 1;
    rc = psam_gemm_f16x3p_ex(x3, b, rs, P(plan->o_w23), b, P(plan->o_s23), out, cout, plan->c23_b, nullptr, 0, nullptr, 0, 0, (int32_t)rows, cout, b, 1.f, 0, &f, stream);
    if (rc || parts == 1) return rc;
    return psam_group_max(part2, cout, out, cout, groups, parts, cout, stream);
}

// ================================================================================================================================
// psam_upscale_masks: the mask decoder after its transformer (pc_sam/model/mask_decoder.py:146-176) -- 3-NN interpolation G -> N, output_upscaling
// (Linear, LayerNorm, GELU, Linear, GELU) and the hyper-network products -- as the Python host runs it in "f16x3": the first Linear on the G
// patch rows BEFORE the interpolation (an affine combination commutes with it), LayerNorm + GELU inside the interpolation kernel (packed rows),
// the second Linear + GELU as one GEMM whose epilogue takes the C products per row; the [N, 256] activations are never written.
// transformer_dim 256, Z * N % 256 == 0, N % 32 == 0, C <= 4.
// ================================================================================================================================
PSAM_API size_t psam_upscale_masks_prepared_bytes(int32_t dim) {
    if (dim <= 0) return 0;
    const int64_t a = kpad(dim);
    return (size_t)(2 * (align256((int64_t)dim * a * 4) + align256(dim * 4)));
}

PSAM_API int32_t psam_upscale_masks_prepare(const psam_upscale_weights_t* wt, psam_upscale_plan_t* plan, void* prepared, size_t prepared_bytes, hipStream_t stream) {
    PSAM_REQUIRE(wt && plan && prepared && wt->u0_w && wt->u0_b && wt->u1_w && wt->u1_b && wt->u3_w && wt->u3_b, PSAM_EINVAL, "psam_upscale_masks_prepare: null pointer");
    PSAM_REQUIRE(wt->dim == 256, PSAM_EINVAL, "psam_upscale_masks_prepare: transformer_dim must be 256");
    PSAM_REQUIRE(prepared_bytes >= psam_upscale_masks_prepared_bytes(wt->dim), PSAM_EWORKSPACE, "psam_upscale_masks_prepare: prepared buffer too small");
    const int E = wt->dim, a = kpad(E);
    std::memset(plan, 0, sizeof(*plan));
    plan->dim = E; plan->eps = wt->eps; plan->u0_w = wt->u0_w; plan->u0_b = wt->u0_b; plan->u1_w = wt->u1_w; plan->u1_b = wt->u1_b; plan->u3_b = wt->u3_b;
    Carve cv(prepared);
    float* p0 = cv.take<float>((int64_t)E * a); float* s0 = cv.take<float>(E); float* p3 = cv.take<float>((int64_t)E * a); float* s3 = cv.take<float>(E);
    plan->o_w0 = (char*)p0 - cv.base; plan->o_s0 = (char*)s0 - cv.base; plan->o_w3 = (char*)p3 - cv.base; plan->o_s3 = (char*)s3 - cv.base;
    int32_t rc = psam_row_scale_f16(wt->u0_w, E, E, E, s0, stream);
    if (!rc) rc = psam_pack_rows_f16x2_g8(wt->u0_w, E, s0, E, E, p0, a, stream);
    if (!rc) rc = psam_row_scale_f16(wt->u3_w, E, E, E, s3, stream);
    if (!rc) rc = psam_pack_rows_f16x2_g8(wt->u3_w, E, s3, E, E, p3, a, stream);
    if (!rc && hipStreamSynchronize(stream) != hipSuccess) { psam_set_error("psam_upscale_masks_prepare: packing failed"); rc = PSAM_EINVAL; }
    return rc;
}

PSAM_API size_t psam_upscale_masks_ws_bytes(int64_t Z, int32_t N, int32_t G, int32_t C, int32_t dim) {
    if (Z <= 0 || N <= 0 || G <= 0 || C <= 0 || dim <= 0) return 0;
    const int64_t a = kpad(dim), planes = psam_gemm_f16x3p_hyper_planes(dim, 0);
    return (size_t)(2 * align256(Z * G * a * 4) + align256(Z * G * 4) + align256(Z * N * a * 4) + align256(Z * N * 4) + align256(planes * Z * C * N * 4));
}

// keys [Z*G, 256]: the patch tokens after the transformer; idx3 / w3 [Z / rep, N, 3]: compute_interp_weights (psam_three_nn); hyper [Z, C, 256]: the
// hyper-network outputs of the selected mask tokens (psam_mlp3); masks [Z, C, N] out.
PSAM_API int32_t psam_upscale_masks(const psam_upscale_plan_t* plan, const void* prepared, const float* keys, const int64_t* idx3, const float* w3,
                                    const float* hyper, int32_t rep, int64_t Z, int32_t N, int32_t G, int32_t C, float* masks, void* ws, size_t ws_bytes,
                                    hipStream_t stream) {
    PSAM_REQUIRE(plan && prepared && keys && idx3 && w3 && hyper && masks && ws, PSAM_EINVAL, "psam_upscale_masks: null pointer");
    const int E = plan->dim, a = kpad(E);
    PSAM_REQUIRE(rep > 0 && Z > 0 && Z % rep == 0 && N > 0 && G > 0 && C > 0 && C <= 4 && (Z * N) % 256 == 0 && N % 32 == 0 && Z * N < ((int64_t)1 << 31), PSAM_EINVAL,
                 "psam_upscale_masks: need Z * N % 256 == 0, N % 32 == 0, C <= 4");
    PSAM_REQUIRE(ws_bytes >= psam_upscale_masks_ws_bytes(Z, N, G, C, E), PSAM_EWORKSPACE, "psam_upscale_masks: workspace too small");
    const char* pb = static_cast<const char*>(prepared);
    auto P = [&](int64_t off) { return reinterpret_cast<const float*>(pb + off); };
    const int planes = psam_gemm_f16x3p_hyper_planes(E, 0);
    const int64_t count = Z * C * N;
    Carve cv(ws);
    float* kp = cv.take<float>(Z * G * a); float* k1 = cv.take<float>(Z * G * a); float* sk = cv.take<float>(Z * G);
    float* up = cv.take<float>(Z * N * a); float* s1 = cv.take<float>(Z * N);
    float* parts = cv.take<float>(planes * count);
    int32_t rc;
    if (rows_multi_fits(G, E)) {
        rc = rows_linear(keys, E, plan->u0_w, E, plan->u0_b, k1, E, Z * G, E, E, stream);
    } else if (Z * G >= 256) {
        rc = psam_scale_pack_rows_g8(keys, E, (int32_t)(Z * G), E, kp, a, sk, stream);
        if (!rc) rc = psam_gemm_f16x3p_ex(kp, a, sk, P(plan->o_w0), a, P(plan->o_s0), k1, E, plan->u0_b, nullptr, 0, nullptr, 0, 0, (int32_t)(Z * G), E, a, 1.f, 0, nullptr, stream);
    } else if (Z * G <= 64) {
        rc = psam_linear_skinny(keys, E, plan->u0_w, E, plan->u0_b, nullptr, 0, k1, E, (int32_t)(Z * G), E, E, 0, stream);
    } else {
        rc = psam_linear(keys, E, plan->u0_w, E, plan->u0_b, nullptr, 0, k1, E, (int32_t)(Z * G), E, E, 0, stream);
    }
    if (rc) return rc;
    rc = psam_interp3_ex(k1, idx3, w3, up, rep, Z, N, G, E, s1, plan->u1_w, plan->u1_b, plan->eps, PSAM_ACT_GELU, stream);
    if (rc) return rc;
    psam_gemm_fuse_t f;
    std::memset(&f, 0, sizeof(f));
    f.hyper = hyper; f.masks = planes > 1 ? parts : masks; f.hyper_c = C; f.hyper_rows = N; f.hyper_pstride = count; f.no_store = 1;
    rc = psam_gemm_f16x3p_ex(up, a, s1, P(plan->o_w3), a, P(plan->o_s3), masks, E, plan->u3_b, nullptr, 0, nullptr, 0, 0, (int32_t)(Z * N), E, a, 1.f, PSAM_ACT_GELU, &f, stream);
    if (rc || planes <= 1) return rc;
    return psam_sum_planes(parts, planes, count, count, masks, stream);
}

// ================================================================================================================================
// psam_twoway_decoder: TwoWayTransformer.forward (pc_sam/model/transformer.py:61-100; blocks :144-176, Attention :214-236) -- the decoder's
// transformer over the patch tokens (`keys`, Z*G rows) and the output / prompt tokens (Z*T rows) -- sequenced by the library exactly as the Python
// host sequences it: per nn.Linear the kernel the host would pick (<= 64 rows: psam_linear_skinny; >= 256 rows and N, K >= 128: the packed-operand
// "f16x3" GEMM on a weight packed at prepare time; otherwise psam_linear), psam_attention_small for every attention, psam_layernorm with the
// residual folded in, psam_add_bcast for the positional additions.  (The one-launch token-side kernel psam_twoway_tokens is slower than these
// launches, DESIGN.md 4.4, so it is not used here.)
// ================================================================================================================================
namespace {
struct TwLin { const float* w; const float* b; const float* packed; const float* scales; int N, K; };

int64_t tw_count_linears(int depth) { return (int64_t)depth * 14 + 4; }

// y [M, N] = act(x W^T + b): the host's dispatch (point_sam_amd/ops.py linear)
int32_t tw_lin(const TwLin& l, const float* x, int64_t M, float* y, int act, float* pack_buf, float* scale_buf, hipStream_t stream) {
    if (M >= 256 && l.packed) {
        const int kp = kpad(l.K);
        int32_t rc = psam_scale_pack_rows_g8(x, l.K, (int32_t)M, l.K, pack_buf, kp, scale_buf, stream);
        if (rc) return rc;
        return psam_gemm_f16x3p_ex(pack_buf, kp, scale_buf, l.packed, kp, l.scales, y, l.N, l.b, nullptr, 0, nullptr, 0, 0, (int32_t)M, l.N, kp, 1.f, act, nullptr, stream);
    }
    if (M <= 64 && (l.K & 15) == 0) return psam_linear_skinny(x, l.K, l.w, l.K, l.b, nullptr, 0, y, l.N, (int32_t)M, l.N, l.K, act, stream);
    return psam_linear(x, l.K, l.w, l.K, l.b, nullptr, 0, y, l.N, (int32_t)M, l.N, l.K, act, stream);
}

// The two-way transformer has two chains per layer that do not depend on each other: the token side (self-attention, norm1, the q projection of
// the token -> patch attention; later the token MLP) and the projections of the PATCH rows (keys + key_pe packed once, k / v of token -> patch, q of
// patch -> token).  psam_twoway_decoder issues the patch-side chain on a side stream forked from the caller's stream by an event and joins it (two
// events) where its results are consumed -- inside a graph capture the side stream joins the capture and the graph gets two parallel branches.
// Same kernels on the same data: same bits.  One side stream + three events per (device, caller stream), created on first use outside a capture
// (a captured call on a stream that has none yet runs the serial sequence).
// MEASURED SLOWER, so OFF by default (PSAM_TWOWAY_FORK=1 / psam_twoway_decoder_force_fork(1) switch it on): the event hand-overs cost more than the
// six short launches they take off the critical path -- eager two-way stage 0.43 -> 0.43 ms (no gain), and inside captured graphs every fork / join
// splits the graph into segments: cfg #2 792 -> 704 clouds/s, cfg #5 131 -> 94 sessions/s, a replayed click 0.61 -> 1.27 ms (profiles/r04/r04_twoway_fork.txt).
struct TwSide { hipStream_t side; hipEvent_t fork, join1, join2; };
static int g_tw_fork = -1;
static bool tw_fork_enabled() {
#ifndef PSAM_BUILD_EXPERIMENTS
    return false;      // the forked form lost (profiles/r04/r04_twoway_fork.txt): reachable in experiments builds only
#endif
    if (g_tw_fork >= 0) return g_tw_fork != 0;
    static int on = -1;
    if (on < 0) { const char* e = getenv("PSAM_TWOWAY_FORK"); on = e ? (atoi(e) != 0) : 0; }
    return on != 0;
}
static const TwSide* tw_side(hipStream_t stream) {
#ifndef PSAM_BUILD_EXPERIMENTS
    (void)stream;
    return nullptr;      // the default build creates no streams or events and keeps no per-stream state
#else
    static std::mutex mu;
    static std::map<std::pair<int, hipStream_t>, TwSide> table;
    if (!tw_fork_enabled()) return nullptr;
    int dev = 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipGetDevice(&dev) != hipSuccess || hipStreamIsCapturing(stream, &cs) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    auto it = table.find({dev, stream});
    if (it != table.end()) return &it->second;
    if (cs != hipStreamCaptureStatusNone) return nullptr;
    TwSide t;
    if (hipStreamCreateWithFlags(&t.side, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&t.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&t.join1, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&t.join2, hipEventDisableTiming) != hipSuccess) return nullptr;
    return &(table[{dev, stream}] = t);
#endif
}
}  // namespace

#ifdef PSAM_BUILD_EXPERIMENTS
PSAM_API void psam_twoway_decoder_force_fork(int32_t mode) { g_tw_fork = mode; }
#endif
// The regrouped launch sequence of psam_twoway_decoder (see there): -1 = environment PSAM_TWOWAY_FAST (default on), 0 = the operator-by-operator
// sequence, 1 = on.
static int g_tw_fast = -1;
PSAM_API void psam_twoway_decoder_force_fast(int32_t mode) { g_tw_fast = mode; }
static int tw_fast_mode() {      // 0 operator sequence, 1 regrouped with packed-operand GEMMs on the patch side, 2 (default) regrouped with the exact-fp32 row kernel there
    if (g_tw_fast >= 0) return g_tw_fast;
    static int on = -1;
    if (on < 0) { const char* e = getenv("PSAM_TWOWAY_FAST"); on = e ? atoi(e) : 2; }
    return on;
}
static bool tw_fast_enabled() { return tw_fast_mode() != 0; }

PSAM_API size_t psam_twoway_decoder_prepared_bytes(int32_t depth, int32_t dim, int32_t mlp, int32_t downsample) {
    if (depth <= 0 || dim <= 0 || mlp <= 0 || downsample <= 0) return 0;
    const int64_t E = dim, IX = dim / downsample;
    auto one = [](int64_t n, int64_t k) { return align256(n * kpad((int)k) * 4) + align256(n * 4); };
    const int64_t self_attn = 4 * one(E, E), cross = 3 * one(IX, E) + one(E, IX), mlpb = one(mlp, E) + one(E, mlp);
    const int64_t cat = one(2 * IX, E) + align256(2 * IX * 4);      // [k of tokens -> patches | q of patches -> tokens] as one weight (they share their input rows)
    return (size_t)(depth * (self_attn + 2 * cross + mlpb + cat) + cross);
}

PSAM_API int32_t psam_twoway_decoder_prepare(const psam_twoway_weights_t* wt, psam_twoway_plan_t* plan, void* prepared, size_t prepared_bytes, hipStream_t stream) {
    PSAM_REQUIRE(wt && plan && prepared && wt->layers, PSAM_EINVAL, "psam_twoway_decoder_prepare: null pointer");
    PSAM_REQUIRE(wt->depth > 0 && wt->depth <= PSAM_TWOWAY_MAX_DEPTH && wt->dim > 0 && wt->dim % 32 == 0 && wt->heads > 0 && wt->mlp > 0 && wt->mlp % 32 == 0 && wt->downsample > 0 &&
                     wt->dim % wt->downsample == 0 && (wt->dim / wt->downsample) % wt->heads == 0 && wt->dim % wt->heads == 0,
                 PSAM_EINVAL, "psam_twoway_decoder_prepare: bad dimensions");
    PSAM_REQUIRE(prepared_bytes >= psam_twoway_decoder_prepared_bytes(wt->depth, wt->dim, wt->mlp, wt->downsample), PSAM_EWORKSPACE,
                 "psam_twoway_decoder_prepare: prepared buffer too small");
    std::memset(plan, 0, sizeof(*plan));
    plan->weights = *wt;
    for (int i = 0; i < wt->depth; ++i) plan->layers[i] = wt->layers[i];
    plan->weights.layers = nullptr;      // the plan carries its own copy of the per-layer pointer blocks
    const int E = wt->dim, IX = wt->dim / wt->downsample;
    Carve cv(prepared);
    int32_t rc = PSAM_OK;
    int slot = 0;
    auto pack = [&](const float* w, int n, int k) {
        float* p = cv.take<float>((int64_t)n * kpad(k)); float* s = cv.take<float>(n);
        plan->o_packed[slot] = (char*)p - cv.base; plan->o_scales[slot] = (char*)s - cv.base; ++slot;
        if (rc || !w) { if (!w && !rc) { psam_set_error("psam_twoway_decoder_prepare: null weight pointer"); rc = PSAM_EINVAL; } return; }
        rc = psam_row_scale_f16(w, k, n, k, s, stream);
        if (!rc) rc = psam_pack_rows_f16x2_g8(w, k, s, n, k, p, kpad(k), stream);
    };
    auto pack_attn = [&](const psam_attn_weights_t& a, int inner) { pack(a.q_w, inner, E); pack(a.k_w, inner, E); pack(a.v_w, inner, E); pack(a.o_w, E, inner); };
    for (int i = 0; i < wt->depth; ++i) {
        const psam_twoway_layer_weights_t& L = wt->layers[i];
        pack_attn(L.self_attn, E); pack_attn(L.t2i, IX); pack(L.m1_w, wt->mlp, E); pack(L.m2_w, E, wt->mlp); pack_attn(L.i2t, IX);
    }
    pack_attn(wt->final_attn, IX);
    // k_proj of cross_attn_token_to_image and q_proj of cross_attn_image_to_token both read keys + key_pe (transformer.py:160-175): one [2 IX, E] weight
    for (int i = 0; i < wt->depth && !rc; ++i) {
        const psam_twoway_layer_weights_t& L = wt->layers[i];
        const int kp = kpad(E);
        float* p = cv.take<float>((int64_t)2 * IX * kp); float* sc = cv.take<float>(2 * IX); float* bc = cv.take<float>(2 * IX);
        plan->o_cat_packed[i] = (char*)p - cv.base; plan->o_cat_scales[i] = (char*)sc - cv.base; plan->o_cat_bias[i] = (char*)bc - cv.base;
        const float* ws_[2] = {L.t2i.k_w, L.i2t.q_w};
        const float* bs_[2] = {L.t2i.k_b, L.i2t.q_b};
        for (int h = 0; h < 2 && !rc; ++h) {
            if (!ws_[h] || !bs_[h]) { psam_set_error("psam_twoway_decoder_prepare: null weight pointer"); rc = PSAM_EINVAL; break; }
            rc = psam_row_scale_f16(ws_[h], E, IX, E, sc + h * IX, stream);
            if (!rc) rc = psam_pack_rows_f16x2_g8(ws_[h], E, sc + h * IX, IX, E, p + (int64_t)h * IX * kp, kp, stream);
            if (!rc && hipMemcpyAsync(bc + h * IX, bs_[h], (size_t)IX * 4, hipMemcpyDeviceToDevice, stream) != hipSuccess) { psam_set_error("psam_twoway_decoder_prepare: bias copy failed"); rc = PSAM_EINVAL; }
        }
    }
    if (!rc && hipStreamSynchronize(stream) != hipSuccess) { psam_set_error("psam_twoway_decoder_prepare: packing failed"); rc = PSAM_EINVAL; }
    return rc;
}

PSAM_API size_t psam_twoway_decoder_ws_bytes(int64_t Z, int32_t T, int32_t G, int32_t dim, int32_t mlp) {
    if (Z <= 0 || T <= 0 || G <= 0 || dim <= 0 || mlp <= 0) return 0;
    const int64_t R = Z * T, I = Z * G, E = dim, big = R > I ? R : I, wide = mlp > E ? mlp : E;
    return (size_t)(6 * align256(R * E * 4) + align256(R * mlp * 4) + 6 * align256(I * E * 4) + align256(big * kpad((int)wide) * 4) + align256(big * 4) +
                    align256(I * kpad((int)E) * 4) + align256(I * 4) +      // + the packed keys + key_pe rows (shared by two projections per layer) and their scales
                    3 * align256(R * E * 4) + align256(8 * R * 256 * 4) + align256(I * kpad((int)E) * 4) + align256(I * 4));      // fast sequence: next projections, Linear + LN scratch, packed keys
}

// tokens [Z*T, dim] (output tokens + sparse prompt embeddings = the point embedding `query_pe`), keys [Z*G, dim] (src = image embedding + dense
// prompt; UPDATED IN PLACE to the transformer's second output), pos [Z / rep, G, dim] (image positional encoding, shared by the rep prompt sets of
// a cloud) -> queries [Z*T, dim] (the transformer's first output)
PSAM_API int32_t psam_twoway_decoder(const psam_twoway_plan_t* plan, const void* prepared, const float* tokens, float* keys, const float* pos, int32_t rep,
                                     int64_t Z, int32_t T, int32_t G, float* queries, void* ws, size_t ws_bytes, int32_t* counters, hipStream_t stream) {
    PSAM_REQUIRE(plan && prepared && tokens && keys && pos && queries && ws, PSAM_EINVAL, "psam_twoway_decoder: null pointer");
    const psam_twoway_weights_t& W = plan->weights;
    const int E = W.dim, IX = W.dim / W.downsample, H = W.heads, mlp = W.mlp;
    const int64_t R = Z * T, I = Z * G;
    PSAM_REQUIRE(rep > 0 && Z > 0 && Z % rep == 0 && T > 0 && G > 0 && R < ((int64_t)1 << 31) && I < ((int64_t)1 << 31), PSAM_EINVAL, "psam_twoway_decoder: bad shape");
    PSAM_REQUIRE(ws_bytes >= psam_twoway_decoder_ws_bytes(Z, T, G, E, mlp), PSAM_EWORKSPACE, "psam_twoway_decoder: workspace too small");
    const char* pb = static_cast<const char*>(prepared);
    int slot = 0;
    auto L = [&](const float* w, const float* b, int n, int k) {
        TwLin l{w, b, nullptr, nullptr, n, k};
        if (n >= 128 && k >= 128) { l.packed = reinterpret_cast<const float*>(pb + plan->o_packed[slot]); l.scales = reinterpret_cast<const float*>(pb + plan->o_scales[slot]); }
        ++slot;
        return l;
    };
    Carve cv(ws);
    float* q = cv.take<float>(R * E); float* pq = cv.take<float>(R * E); float* pk = cv.take<float>(R * E); float* pv = cv.take<float>(R * E);
    float* ta = cv.take<float>(R * E); float* ty = cv.take<float>(R * E); float* tm = cv.take<float>(R * mlp);
    float* k = cv.take<float>(I * E); float* ik = cv.take<float>(I * E); float* iv = cv.take<float>(I * E); float* iq = cv.take<float>(I * E);
    float* ia = cv.take<float>(I * E); float* iy = cv.take<float>(I * E);
    float* pack_buf = cv.take<float>((R > I ? R : I) * kpad(mlp > E ? mlp : E)); float* scale_buf = cv.take<float>(R > I ? R : I);
    float* kin_p = cv.take<float>(I * kpad(E)); float* kin_s = cv.take<float>(I);
    float* nq = cv.take<float>(R * E); float* nk = cv.take<float>(R * E); float* nv = cv.take<float>(R * E); float* lntmp = cv.take<float>(8 * R * 256);
    float* kv_p = cv.take<float>(I * kpad(E)); float* kv_s = cv.take<float>(I);
    int32_t rc = PSAM_OK;
    // Round 4, fewer and wider launches with the same arithmetic (the same bits as the per-operator sequence the Python host issues):
    //  * token side (R <= 64 rows): the q / k / v projections of an attention are ONE psam_linear_skinny_multi launch, `queries + query_pe` is added
    //    while the rows are loaded -- no psam_add_bcast, no separate launches;
    //  * patch side: `keys + key_pe` is added inside the pass that scales and packs the rows (psam_scale_pack_rows_g8_add), and the packed rows feed
    //    BOTH projections that read them in a layer (k of tokens -> patches, q of patches -> tokens): one pack instead of an add and two packs.
    const bool tok_fast = R <= 64 && (E & 15) == 0;
    auto skinny = [&](int n, const TwLin* l, const float* const* xs, const float* const* xadds, float* const* ys) -> int32_t {
        psam_skinny_jobs_t jobs;
        std::memset(&jobs, 0, sizeof(jobs));
        jobs.n = n;
        for (int i = 0; i < n; ++i) { jobs.job[i].x = xs[i]; jobs.job[i].xadd = xadds[i]; jobs.job[i].W = l[i].w; jobs.job[i].bias = l[i].b; jobs.job[i].y = ys[i]; jobs.job[i].ldy = l[i].N; jobs.job[i].N = l[i].N; jobs.job[i].act = 0; }
        return psam_linear_skinny_multi(&jobs, E, E, l[0].K, (int32_t)R, l[0].K, stream);
    };
    auto gemm_packed = [&](const TwLin& l, const float* xp, const float* sx, int64_t M, float* y, hipStream_t s) -> int32_t {
        const int kp = kpad(l.K);
        return psam_gemm_f16x3p_ex(xp, kp, sx, l.packed, kp, l.scales, y, l.N, l.b, nullptr, 0, nullptr, 0, 0, (int32_t)M, l.N, kp, 1.f, 0, nullptr, s);
    };
    // k_in = keys + pos, packed: shared by every projection of the layer that reads it (needs I >= 256 and packed weights)
    auto pack_kin = [&](hipStream_t s) -> int32_t { return psam_scale_pack_rows_g8_add(keys, E, pos, E, G, rep, (int32_t)I, E, kin_p, kpad(E), kin_s, s); };
#define TWCK(call) do { rc = (call); if (rc) return rc; } while (0)
#define TWHIP(call) do { if ((call) != hipSuccess) { psam_set_error("psam_twoway_decoder: stream fork / join failed"); return PSAM_EINVAL; } } while (0)
    const TwSide* sd = tok_fast ? tw_side(stream) : nullptr;      // patch-side projections on a forked side stream (see TwSide)
    // Attention.forward up to out_proj: projections + softmax(q k^T / sqrt(hd)) v
    auto attn = [&](const TwLin& lq, const TwLin& lk, const TwLin& lv, const float* q_in, int64_t Mq, const float* k_in, const float* v_in, int64_t Mk, float* oq, float* ok,
                    float* ov, float* out, int Lq, int Lk) -> int32_t {
        int32_t r = tw_lin(lq, q_in, Mq, oq, 0, pack_buf, scale_buf, stream);
        if (!r) r = tw_lin(lk, k_in, Mk, ok, 0, pack_buf, scale_buf, stream);
        if (!r) r = tw_lin(lv, v_in, Mk, ov, 0, pack_buf, scale_buf, stream);
        const int inner = lq.N, hd = inner / H;
        if (!r) r = psam_attention_small(oq, inner, (int64_t)Lq * inner, ok, inner, (int64_t)Lk * inner, ov, inner, (int64_t)Lk * inner, out, inner, (int64_t)Lq * inner, Z, H, Lq,
                                         Lk, hd, 1.0f / std::sqrt((float)hd), stream);
        return r;
    };
    // ---- Round 5: the short sequence (36 launches for depth 2 against 50; profiles/r05/r05_click_kernels_*.txt).  Same operators, regrouped along their
    // data dependences:
    //  * every `queries = norm(queries + Linear(..))` of the token side is ONE launch (psam_linear_skinny_ln: the last workgroup of the Linear
    //    normalises the rows; lin2 of the MLP, K = 2048, split over 8 workgroup rows instead of walking K in 16 dependent rounds);
    //  * the token projections that read the layer's final `queries` -- k / v of patches -> tokens AND the next layer's self-attention q / k / v (or
    //    the final attention's q) -- are one psam_linear_skinny_multi launch (the patch -> token attention in between only changes `keys`);
    //  * patch side: keys + key_pe and keys are packed in one pass (psam_scale_pack_rows_g8_add_dual), the k projection of tokens -> patches and the
    //    q projection of patches -> tokens are one GEMM on the concatenated weight (both read keys + key_pe).
    // Needs the caller's arrival-counter block (`counters`) and embedding_dim 256.
    const bool fast = tok_fast && !sd && tw_fast_enabled() && E == 256 && IX >= 128 && 2 * IX <= E && IX <= 512 && I >= 256 && (mlp & 15) == 0 && (IX & 15) == 0 && plan->o_cat_packed[0] != 0 &&
                      counters != nullptr;
    if (fast) {
        auto LS = [&](int sl, const float* w, const float* b, int n, int k) {
            TwLin l{w, b, nullptr, nullptr, n, k};
            if (n >= 128 && k >= 128) { l.packed = reinterpret_cast<const float*>(pb + plan->o_packed[sl]); l.scales = reinterpret_cast<const float*>(pb + plan->o_scales[sl]); }
            return l;
        };
        auto multi = [&](int n, const TwLin* ls, const float* x, const float* const* xadds, float* const* ys) -> int32_t {
            psam_skinny_jobs_t jobs;
            std::memset(&jobs, 0, sizeof(jobs));
            jobs.n = n;
            for (int j = 0; j < n; ++j) { jobs.job[j].x = x; jobs.job[j].xadd = xadds[j]; jobs.job[j].W = ls[j].w; jobs.job[j].bias = ls[j].b; jobs.job[j].y = ys[j]; jobs.job[j].ldy = ls[j].N; jobs.job[j].N = ls[j].N; jobs.job[j].act = 0; }
            return psam_linear_skinny_multi(&jobs, E, E, E, (int32_t)R, E, stream);
        };
        auto lin_ln = [&](const TwLin& l, const float* x, const float* res, const float* nw, const float* nb) -> int32_t {
            return psam_linear_skinny_ln(x, l.K, l.w, l.K, l.b, res, E, nw, nb, W.eps, lntmp, queries, E, (int32_t)R, E, l.K, counters, stream);
        };
        auto sattn = [&](int inner, const float* oq, int64_t ldq, const float* ok, int64_t ldk, const float* ov, int64_t ldv, float* out, int Lq, int Lk) -> int32_t {
            return psam_attention_small(oq, ldq, Lq * ldq, ok, ldk, Lk * ldk, ov, ldv, Lk * ldv, out, inner, (int64_t)Lq * inner, Z, H, Lq, Lk, inner / H,
                                        1.0f / std::sqrt((float)(inner / H)), stream);
        };
        const bool small_rows = tw_fast_mode() >= 2 && rows_multi_fits(G, E);
        auto rows_multi = [&](int n, const TwLin* ls, const float* const* xadds, float* const* ys, const int64_t* ldys) -> int32_t {
            psam_skinny_jobs_t jobs;
            std::memset(&jobs, 0, sizeof(jobs));
            jobs.n = n;
            for (int j = 0; j < n; ++j) { jobs.job[j].x = keys; jobs.job[j].xadd = xadds[j]; jobs.job[j].W = ls[j].w; jobs.job[j].bias = ls[j].b; jobs.job[j].y = ys[j]; jobs.job[j].ldy = ldys[j]; jobs.job[j].N = ls[j].N; }
            return psam_linear_rows_multi(&jobs, E, E, G, rep, E, I, E, stream);
        };
        auto dual_pack = [&]() -> int32_t { return psam_scale_pack_rows_g8_add_dual(keys, E, pos, E, G, rep, (int32_t)I, E, kin_p, kin_s, kv_p, kv_s, kpad(E), stream); };
        float *aq = pq, *ak = pk, *av = pv;      // self-attention projections of the layer at hand
        float *cqo = nq, *jko = nk, *jvo = nv;   // q of tokens -> patches (and of the final attention); k / v of patches -> tokens
        {   // first layer: q = k = v = the tokens, no positional encoding (skip_first_layer_pe)
            const psam_twoway_layer_weights_t& L0 = plan->layers[0];
            const TwLin ls[3] = {LS(0, L0.self_attn.q_w, L0.self_attn.q_b, E, E), LS(1, L0.self_attn.k_w, L0.self_attn.k_b, E, E), LS(2, L0.self_attn.v_w, L0.self_attn.v_b, E, E)};
            const float* xa[3] = {nullptr, nullptr, nullptr};
            float* ys[3] = {aq, ak, av};
            TWCK(multi(3, ls, tokens, xa, ys));
        }
        for (int i = 0; i < W.depth; ++i) {
            const psam_twoway_layer_weights_t& Lw = plan->layers[i];
            const int b0 = 14 * i;
            const TwLin so = LS(b0 + 3, Lw.self_attn.o_w, Lw.self_attn.o_b, E, E);
            const TwLin cq = LS(b0 + 4, Lw.t2i.q_w, Lw.t2i.q_b, IX, E), cvl = LS(b0 + 6, Lw.t2i.v_w, Lw.t2i.v_b, IX, E), co = LS(b0 + 7, Lw.t2i.o_w, Lw.t2i.o_b, E, IX);
            const TwLin m1 = LS(b0 + 8, Lw.m1_w, Lw.m1_b, mlp, E), m2 = LS(b0 + 9, Lw.m2_w, Lw.m2_b, E, mlp);
            const TwLin jk = LS(b0 + 11, Lw.i2t.k_w, Lw.i2t.k_b, IX, E), jv = LS(b0 + 12, Lw.i2t.v_w, Lw.i2t.v_b, IX, E), jo = LS(b0 + 13, Lw.i2t.o_w, Lw.i2t.o_b, E, IX);
            const TwLin cat{nullptr, reinterpret_cast<const float*>(pb + plan->o_cat_bias[i]), reinterpret_cast<const float*>(pb + plan->o_cat_packed[i]),
                            reinterpret_cast<const float*>(pb + plan->o_cat_scales[i]), 2 * IX, E};
            // patch side: k of tokens -> patches | q of patches -> tokens (both from keys + key_pe; ik: row stride 2 IX) and v (from keys)
            if (small_rows) {      // one launch, exact fp32 products, key_pe added on load
                const TwLin ck = LS(b0 + 5, Lw.t2i.k_w, Lw.t2i.k_b, IX, E), jq = LS(b0 + 10, Lw.i2t.q_w, Lw.i2t.q_b, IX, E);
                const TwLin ls[3] = {ck, jq, cvl};
                const float* xa[3] = {pos, pos, nullptr};
                float* ys[3] = {ik, ik + IX, iv};
                const int64_t lds_[3] = {2 * IX, 2 * IX, IX};
                TWCK(rows_multi(3, ls, xa, ys, lds_));
            } else {               // both packed forms of the keys in one pass, [k | q] as one GEMM on the concatenated weight
                TWCK(dual_pack());
                TWCK(gemm_packed(cat, kin_p, kin_s, I, ik, stream));
                TWCK(gemm_packed(cvl, kv_p, kv_s, I, iv, stream));
            }
            // self-attention of the tokens + norm1 (the first layer replaces the queries: no residual)
            TWCK(sattn(E, aq, E, ak, E, av, E, ta, T, T));
            TWCK(lin_ln(so, ta, i == 0 ? nullptr : queries, Lw.n1_w, Lw.n1_b));
            // tokens attend to the patch tokens + norm2
            {
                const TwLin ls[1] = {cq};
                const float* xa[1] = {tokens};
                float* ys[1] = {cqo};
                TWCK(multi(1, ls, queries, xa, ys));
            }
            TWCK(sattn(IX, cqo, IX, ik, 2 * IX, iv, IX, ta, T, G));
            TWCK(lin_ln(co, ta, queries, Lw.n2_w, Lw.n2_b));
            // token MLP + norm3
            TWCK(psam_linear_skinny(queries, E, m1.w, E, m1.b, nullptr, 0, tm, mlp, (int32_t)R, mlp, E, PSAM_ACT_RELU, stream));
            TWCK(lin_ln(m2, tm, queries, Lw.n3_w, Lw.n3_b));
            // every projection of the finished queries: k / v of patches -> tokens, and what the NEXT attention of the token side needs
            if (i + 1 < W.depth) {
                const psam_twoway_layer_weights_t& Ln = plan->layers[i + 1];
                const TwLin ls[5] = {jk, jv, LS(b0 + 14, Ln.self_attn.q_w, Ln.self_attn.q_b, E, E), LS(b0 + 15, Ln.self_attn.k_w, Ln.self_attn.k_b, E, E),
                                     LS(b0 + 16, Ln.self_attn.v_w, Ln.self_attn.v_b, E, E)};
                const float* xa[5] = {tokens, nullptr, tokens, tokens, nullptr};
                float* ys[5] = {jko, jvo, aq, ak, av};
                TWCK(multi(5, ls, queries, xa, ys));
            } else {
                const TwLin ls[3] = {jk, jv, LS(b0 + 14, W.final_attn.q_w, W.final_attn.q_b, IX, E)};
                const float* xa[3] = {tokens, nullptr, tokens};
                float* ys[3] = {jko, jvo, cqo};
                TWCK(multi(3, ls, queries, xa, ys));
            }
            // patch tokens attend to the tokens + norm4
            TWCK(sattn(IX, ik + IX, 2 * IX, jko, IX, jvo, IX, ia, G, T));
            TWCK(psam_linear_ln256(ia, IX, jo.w, IX, jo.b, keys, E, Lw.n4_w, Lw.n4_b, W.eps, keys, E, I, E, IX, stream));      // out_proj + residual + norm4: one launch, 16 whole rows per workgroup
        }
        const int bf = 14 * W.depth;
        const TwLin fk = LS(bf + 1, W.final_attn.k_w, W.final_attn.k_b, IX, E), fv = LS(bf + 2, W.final_attn.v_w, W.final_attn.v_b, IX, E),
                    fo = LS(bf + 3, W.final_attn.o_w, W.final_attn.o_b, E, IX);
        if (small_rows) {
            const TwLin ls[2] = {fk, fv};
            const float* xa[2] = {pos, nullptr};
            float* ys[2] = {ik, iv};
            const int64_t lds_[2] = {IX, IX};
            TWCK(rows_multi(2, ls, xa, ys, lds_));
        } else {
            TWCK(dual_pack());
            TWCK(gemm_packed(fk, kin_p, kin_s, I, ik, stream));
            TWCK(gemm_packed(fv, kv_p, kv_s, I, iv, stream));
        }
        TWCK(sattn(IX, cqo, IX, ik, IX, iv, IX, ta, T, G));
        TWCK(lin_ln(fo, ta, queries, W.nf_w, W.nf_b));
        return PSAM_OK;
    }
    const float* cur = tokens;      // queries start as the tokens themselves (transformer.py:84)
    for (int i = 0; i < W.depth; ++i) {
        const psam_twoway_layer_weights_t& Lw = plan->layers[i];
        const TwLin sq = L(Lw.self_attn.q_w, Lw.self_attn.q_b, E, E), sk = L(Lw.self_attn.k_w, Lw.self_attn.k_b, E, E), sv = L(Lw.self_attn.v_w, Lw.self_attn.v_b, E, E),
                    so = L(Lw.self_attn.o_w, Lw.self_attn.o_b, E, E);
        const TwLin cq = L(Lw.t2i.q_w, Lw.t2i.q_b, IX, E), ck = L(Lw.t2i.k_w, Lw.t2i.k_b, IX, E), cvl = L(Lw.t2i.v_w, Lw.t2i.v_b, IX, E), co = L(Lw.t2i.o_w, Lw.t2i.o_b, E, IX);
        const TwLin m1 = L(Lw.m1_w, Lw.m1_b, mlp, E), m2 = L(Lw.m2_w, Lw.m2_b, E, mlp);
        const TwLin jq = L(Lw.i2t.q_w, Lw.i2t.q_b, IX, E), jk = L(Lw.i2t.k_w, Lw.i2t.k_b, IX, E), jv = L(Lw.i2t.v_w, Lw.i2t.v_b, IX, E), jo = L(Lw.i2t.o_w, Lw.i2t.o_b, E, IX);
        // self-attention of the tokens (the first layer without positional encoding and without residual: skip_first_layer_pe)
        const bool key_fast = I >= 256 && ck.packed && jq.packed && cvl.packed;
        auto small_attn = [&](const TwLin& lq, const float* oq, const float* ok, const float* ov, float* out, int Lq, int Lk) -> int32_t {
            const int inner = lq.N, hd = inner / H;
            return psam_attention_small(oq, inner, (int64_t)Lq * inner, ok, inner, (int64_t)Lk * inner, ov, inner, (int64_t)Lk * inner, out, inner, (int64_t)Lq * inner, Z, H, Lq, Lk,
                                        hd, 1.0f / std::sqrt((float)hd), stream);
        };
        const bool fork = sd && key_fast;
        if (fork) {      // keys (as the previous layer left them) -> k_in packed, k / v for token -> patch, q for patch -> token: beside the token chain below
            TWHIP(hipEventRecord(sd->fork, stream));
            TWHIP(hipStreamWaitEvent(sd->side, sd->fork, 0));
            TWCK(pack_kin(sd->side));
            TWCK(gemm_packed(ck, kin_p, kin_s, I, ik, sd->side));
            TWCK(tw_lin(cvl, keys, I, iv, 0, pack_buf, scale_buf, sd->side));
            TWHIP(hipEventRecord(sd->join1, sd->side));
            TWCK(gemm_packed(jq, kin_p, kin_s, I, iq, sd->side));
            TWHIP(hipEventRecord(sd->join2, sd->side));
        }
        if (tok_fast) {
            const TwLin ls[3] = {sq, sk, sv};
            const float* xs[3] = {cur, cur, cur};
            const float* xa[3] = {i == 0 ? nullptr : tokens, i == 0 ? nullptr : tokens, nullptr};      // first layer: no positional encoding (skip_first_layer_pe)
            float* ys[3] = {pq, pk, pv};
            TWCK(skinny(3, ls, xs, xa, ys));
            TWCK(small_attn(sq, pq, pk, pv, ta, T, T));
        } else if (i == 0) {
            TWCK(attn(sq, sk, sv, cur, R, cur, cur, R, pq, pk, pv, ta, T, T));
        } else {
            TWCK(psam_add_bcast(queries, (int64_t)T * E, 1, tokens, (int64_t)T * E, E, q, (int64_t)T * E, Z, T, E, stream));
            TWCK(attn(sq, sk, sv, q, R, q, queries, R, pq, pk, pv, ta, T, T));
        }
        TWCK(tw_lin(so, ta, R, ty, 0, pack_buf, scale_buf, stream));
        TWCK(psam_layernorm(ty, E, i == 0 ? nullptr : queries, i == 0 ? 0 : E, Lw.n1_w, Lw.n1_b, queries, E, R, E, W.eps, 0, stream));
        cur = queries;
        // tokens attend to the patch tokens
        if (tok_fast) {
            const TwLin ls[1] = {cq};
            const float* xs[1] = {queries};
            const float* xa[1] = {tokens};
            float* ys[1] = {pq};
            TWCK(skinny(1, ls, xs, xa, ys));
        } else {
            TWCK(psam_add_bcast(queries, (int64_t)T * E, 1, tokens, (int64_t)T * E, E, q, (int64_t)T * E, Z, T, E, stream));
            TWCK(tw_lin(cq, q, R, pq, 0, pack_buf, scale_buf, stream));
        }
        if (fork) {
            TWHIP(hipStreamWaitEvent(stream, sd->join1, 0));
        } else {
            if (key_fast) {
                TWCK(pack_kin(stream));
                TWCK(gemm_packed(ck, kin_p, kin_s, I, ik, stream));
            } else {
                TWCK(psam_add_bcast(pos, (int64_t)G * E, rep, keys, (int64_t)G * E, E, k, (int64_t)G * E, Z, G, E, stream));
                TWCK(tw_lin(ck, k, I, ik, 0, pack_buf, scale_buf, stream));
            }
            TWCK(tw_lin(cvl, keys, I, iv, 0, pack_buf, scale_buf, stream));
        }
        TWCK(small_attn(cq, pq, ik, iv, ta, T, G));
        TWCK(tw_lin(co, ta, R, ty, 0, pack_buf, scale_buf, stream));
        TWCK(psam_layernorm(ty, E, queries, E, Lw.n2_w, Lw.n2_b, queries, E, R, E, W.eps, 0, stream));
        // token MLP
        TWCK(tw_lin(m1, queries, R, tm, PSAM_ACT_RELU, pack_buf, scale_buf, stream));
        TWCK(tw_lin(m2, tm, R, ty, 0, pack_buf, scale_buf, stream));
        TWCK(psam_layernorm(ty, E, queries, E, Lw.n3_w, Lw.n3_b, queries, E, R, E, W.eps, 0, stream));
        // patch tokens attend to the tokens
        if (fork) TWHIP(hipStreamWaitEvent(stream, sd->join2, 0));
        else if (key_fast) TWCK(gemm_packed(jq, kin_p, kin_s, I, iq, stream));      // keys + key_pe: packed once above, unchanged since
        else TWCK(tw_lin(jq, k, I, iq, 0, pack_buf, scale_buf, stream));
        if (tok_fast) {
            const TwLin ls[2] = {jk, jv};
            const float* xs[2] = {queries, queries};
            const float* xa[2] = {tokens, nullptr};
            float* ys[2] = {pk, pv};
            TWCK(skinny(2, ls, xs, xa, ys));
        } else {
            TWCK(psam_add_bcast(queries, (int64_t)T * E, 1, tokens, (int64_t)T * E, E, q, (int64_t)T * E, Z, T, E, stream));
            TWCK(tw_lin(jk, q, R, pk, 0, pack_buf, scale_buf, stream));
            TWCK(tw_lin(jv, queries, R, pv, 0, pack_buf, scale_buf, stream));
        }
        TWCK(small_attn(jq, iq, pk, pv, ia, G, T));
        TWCK(tw_lin(jo, ia, I, iy, 0, pack_buf, scale_buf, stream));
        TWCK(psam_layernorm(iy, E, keys, E, Lw.n4_w, Lw.n4_b, keys, E, I, E, W.eps, 0, stream));
    }
    const TwLin fq = L(W.final_attn.q_w, W.final_attn.q_b, IX, E), fk = L(W.final_attn.k_w, W.final_attn.k_b, IX, E), fv = L(W.final_attn.v_w, W.final_attn.v_b, IX, E),
                fo = L(W.final_attn.o_w, W.final_attn.o_b, E, IX);
    {
        const int inner = fq.N, hd = inner / H;
        const bool fork = sd && I >= 256 && fk.packed && fv.packed;
        if (fork) {
            TWHIP(hipEventRecord(sd->fork, stream));
            TWHIP(hipStreamWaitEvent(sd->side, sd->fork, 0));
            TWCK(pack_kin(sd->side));
            TWCK(gemm_packed(fk, kin_p, kin_s, I, ik, sd->side));
            TWCK(tw_lin(fv, keys, I, iv, 0, pack_buf, scale_buf, sd->side));
            TWHIP(hipEventRecord(sd->join1, sd->side));
        }
        if (tok_fast) {
            const TwLin ls[1] = {fq};
            const float* xs[1] = {queries};
            const float* xa[1] = {tokens};
            float* ys[1] = {pq};
            TWCK(skinny(1, ls, xs, xa, ys));
        } else {
            TWCK(psam_add_bcast(queries, (int64_t)T * E, 1, tokens, (int64_t)T * E, E, q, (int64_t)T * E, Z, T, E, stream));
            TWCK(tw_lin(fq, q, R, pq, 0, pack_buf, scale_buf, stream));
        }
        if (fork) {
            TWHIP(hipStreamWaitEvent(stream, sd->join1, 0));
        } else {
            if (I >= 256 && fk.packed) {
                TWCK(pack_kin(stream));
                TWCK(gemm_packed(fk, kin_p, kin_s, I, ik, stream));
            } else {
                TWCK(psam_add_bcast(pos, (int64_t)G * E, rep, keys, (int64_t)G * E, E, k, (int64_t)G * E, Z, G, E, stream));
                TWCK(tw_lin(fk, k, I, ik, 0, pack_buf, scale_buf, stream));
            }
            TWCK(tw_lin(fv, keys, I, iv, 0, pack_buf, scale_buf, stream));
        }
        TWCK(psam_attention_small(pq, inner, (int64_t)T * inner, ik, inner, (int64_t)G * inner, iv, inner, (int64_t)G * inner, ta, inner, (int64_t)T * inner, Z, H, T, G, hd,
                                  1.0f / std::sqrt((float)hd), stream));
    }
    TWCK(tw_lin(fo, ta, R, ty, 0, pack_buf, scale_buf, stream));
    TWCK(psam_layernorm(ty, E, queries, E, W.nf_w, W.nf_b, queries, E, R, E, W.eps, 0, stream));
#undef TWCK
#undef TWHIP
    return PSAM_OK;
}
