/*
 * pointsam_hip.h -- C ABI of libpointsam_hip.so (gfx950 / MI355X), the drop-in boundary for the Point-SAM
 * inference hot path (encode + prompt decode).
 *
 * The reference (zyc00/Point-SAM) has no FFI of its own: its "operator API" for this path is the set of Python
 * call sites into third-party native code and ATen.  Each entry point below names the reference interface it
 * replaces (file:line under the reference checkout).  The Python host (point_sam_amd/ops.py) binds these with
 * ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer valid on `stream`; calls are asynchronous and never allocate;
 *   - tensors are fp32 row-major unless stated; indices are int64 (torch.long) as in the reference;
 *   - return value: 0 = ok, < 0 = invalid argument (PSAM_E*), > 0 = hipError_t of the failed launch;
 *     psam_last_error_string() describes the last failure on the calling thread;
 *   - thread-safe for distinct streams / workspaces.
 */
#ifndef POINTSAM_HIP_H
#define POINTSAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* psam_stream_t; /* == hipStream_t */

#define PSAM_OK 0
#define PSAM_EINVAL (-1)
#define PSAM_EALIGN (-2)
#define PSAM_EWORKSPACE (-3)

#define PSAM_ACT_NONE 0
#define PSAM_ACT_GELU 1 /* exact erf GELU (torch.nn.GELU default) */
#define PSAM_ACT_RELU 2
#define PSAM_ACT_SWIGLU 3 /* GEMM only: W packs alternating 32-row blocks of fc1_g / fc1_x; C gets N/2 columns silu(g)*x */

int32_t psam_version(void);
const char* psam_last_error_string(void);

/* ---------------------------------------------------------------- point tokenizer */

/* Farthest point sampling + center gather.
 * Replaces torkit3d.ops.sample_farthest_points(points, num_samples) and torkit3d batch_index_select:
 * pc_sam/model/common.py:91-92 (also :22-23, :199-200).  Start index 0, fp32 squared distances without FMA,
 * arg-max with lowest index on ties (bit-exact vs oracle/tokenizer_oracle.c).
 *   xyz [B,N,3] -> fps_idx [B,G] int64, centers [B,G,3];  ws: psam_fps_workspace_bytes() bytes, 16B aligned. */
size_t psam_fps_workspace_bytes(int32_t B, int32_t N, int32_t G);
int32_t psam_fps(const float* xyz, int32_t B, int32_t N, int32_t G, int64_t* fps_idx, float* centers, void* ws, size_t ws_bytes,
                 psam_stream_t stream);
void psam_fps_set_cooperative(int32_t on); /* test hook: 0 = never use the multi-workgroup kernel for N > 32768, 2 = use it without the one-XCD placement */
void psam_fps_set_pruning(int32_t mode);   /* A/B and test hook: 0 = the multi-workgroup kernel scans every point in every iteration, 1 = it buckets the cloud by
                                            * grid cell and skips, exactly, the waves whose bounding box the new centre cannot reach; -1 = default
                                            * (environment PSAM_FPS_PRUNE, else 1) */

/* K nearest points of each center, ascending by (squared distance, index); the [G,N] distance matrix is never
 * materialised.  Replaces knn_points(centers, xyz, K) = torch.cdist + torch.topk: pc_sam/model/common.py:27-56,97.
 *   centers [B,G,3], xyz [B,N,3] -> knn_idx [B,G,K] int64.  K <= 1024. */
int32_t psam_knn(const float* centers, const float* xyz, int32_t B, int32_t G, int32_t N, int32_t K, int64_t* knn_idx, psam_stream_t stream);
/* tuning / test hook: 1 = the band kernel (one or two distance evaluations per pair; default), 0 = the four-pass kernel, -1 = default (environment PSAM_KNN_BAND) */
void psam_knn_force_band(int32_t mode);

/* 3 nearest centers of every point + normalised 1/max(d^2, eps) weights.
 * Replaces compute_interp_weights(query, key): pc_sam/model/common.py:238-255 (called from mask_decoder.py:151-156).
 *   xyz [B,N,3], centers [B,G,3] -> idx3 [B,N,3] int64, w3 [B,N,3]. */
int32_t psam_three_nn(const float* xyz, const float* centers, int32_t B, int32_t N, int32_t G, float eps, int64_t* idx3, float* w3,
                      psam_stream_t stream);

/* Neighbourhood gather out[bf,g,k,:] = [xyz[idx]-center, feats[bf,idx,:]]  (bf = b*rep + m).
 * Replaces the gather/centre/concat of KNNGrouper.forward: pc_sam/model/common.py:99-120, and of
 * group_with_centers_and_knn: pc_sam/model/common.py:126-187 (rep > 1: several feature sets per cloud).
 *   feats [B*rep,N,C] -> out [B*rep,G,K,3+C]. */
int32_t psam_group_gather(const float* xyz, const float* feats, const float* centers, const int64_t* knn_idx, int32_t B, int32_t rep,
                          int32_t N, int32_t G, int32_t K, int32_t C, float* out, psam_stream_t stream);

/* Same gather fused with the first mini-PointNet layer: GELU(LayerNorm_128(Linear(3+C -> 128))).
 * Replaces PatchEncoder.conv1[0:3] on the grouped features: pc_sam/model/common.py:486-489,499. C in {1,3}.
 *   out [B*rep*G*K, 128]. */
int32_t psam_patch_l1(const float* xyz, const float* feats, const float* centers, const int64_t* knn_idx, const float* W, const float* bias,
                      const float* lnw, const float* lnb, float eps, int32_t B, int32_t rep, int32_t N, int32_t G, int32_t K, int32_t C,
                      float* out, psam_stream_t stream);
/* Both with the grouper's `radius` option (KNNGrouper.radius, common.py:107-108; MaskEncoder.radius, common.py:161-164;
 * configs/model/enc_with_radius.yaml): relative coordinates divided by radius; radius <= 0 = none. */
int32_t psam_group_gather_r(const float* xyz, const float* feats, const float* centers, const int64_t* knn_idx, int32_t B, int32_t rep,
                            int32_t N, int32_t G, int32_t K, int32_t C, float radius, float* out, psam_stream_t stream);
/* the same with a row stride ldo >= 3 + C; the tail of every row is written as zeros (K of the following Linear % 4 == 0) */
int32_t psam_group_gather_ld(const float* xyz, const float* feats, const float* centers, const int64_t* knn_idx, int32_t B, int32_t rep, int32_t N,
                             int32_t G, int32_t K, int32_t C, float radius, float* out, int64_t ldo, psam_stream_t stream);
int32_t psam_patch_l1_r(const float* xyz, const float* feats, const float* centers, const int64_t* knn_idx, const float* W, const float* bias,
                        const float* lnw, const float* lnb, float eps, int32_t B, int32_t rep, int32_t N, int32_t G, int32_t K, int32_t C,
                        float radius, float* out, psam_stream_t stream);
/* The same with the remaining grouper options and a packed output:
 *   center_idx [B,G] (the groups' FPS indices) != NULL: `centralize_features` (KNNGrouper / group_with_centers_and_knn,
 *   common.py:116-118, 183-186): input = [rel xyz, features, features - centre features], W [128, 3 + 2C];
 *   scale_out [rows] != NULL: out receives the g8-packed rows (A operand of psam_gemm_f16x3p for conv1.3) and scale_out their scales. */
int32_t psam_patch_l1_ex(const float* xyz, const float* feats, const float* centers, const int64_t* knn_idx, const int64_t* center_idx,
                         const float* W, const float* bias, const float* lnw, const float* lnb, float eps, int32_t B, int32_t rep, int32_t N,
                         int32_t G, int32_t K, int32_t C, float radius, float* out, float* scale_out, psam_stream_t stream);

/* Click simulation of the evaluation protocol.  psam_error_regions: fn = gt & !(logit > 0), fp = !gt & (logit > 0)
 * (logits == NULL: fn = gt, fp = 0) -- sample_fixed_points, pc_sam/model/common.py:388-405.  psam_border_farthest: per
 * region the member point farthest from the region's complement (squared distance; -1/-1 when either is empty) --
 * sample_furthest_points_from_border + torkit3d chamfer_distance, pc_sam/model/common.py:443-474.
 *   gt/fn/fp/region [Z,N] uint8, xyz [B,N,3], Z = B*rep. */
int32_t psam_error_regions(const uint8_t* gt, const float* logits, uint8_t* fn, uint8_t* fp, int64_t total, psam_stream_t stream);
size_t psam_border_farthest_workspace_bytes(int32_t Z, int32_t N);
int32_t psam_border_farthest(const float* xyz, const uint8_t* region, int32_t B, int32_t rep, int32_t N, int64_t* out_idx, float* out_dist,
                             void* ws, size_t ws_bytes, psam_stream_t stream);

/* Max over the K members of each group: x [groups*K, C] -> y [groups, C].
 * Replaces torch.max(x, dim=-2): pc_sam/model/common.py:502,505. */
int32_t psam_group_max(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t groups, int32_t K, int32_t C, psam_stream_t stream);

/* ---------------------------------------------------------------- dense layers (fp32-exact MFMA) */

/* C[z] = act(alpha * A[z] @ W[z]^T + bias + rowbias[row/rowgroup]) + residual[z];  A [M,K], W [N,K] (nn.Linear layout).
 * Replaces every nn.Linear / F.linear / matmul on the path (cuBLAS in the reference): common.py:486-497,
 * pc_encoder.py:99-116, timm Eva blocks, transformer.py:199-202,248-249, mask_decoder.py:53-59,176,201-203.
 * Batch index z = z1*batch2 + z2 with element strides s?1 / s?2.  K, lda, ldw and batch strides multiples of 4. */
int32_t psam_gemm_f32(const float* A, int64_t lda, int64_t sA1, int64_t sA2, const float* W, int64_t ldw, int64_t sW1, int64_t sW2, float* C,
                      int64_t ldc, int64_t sC1, int64_t sC2, const float* bias, const float* residual, int64_t ldr, int64_t sR1, int64_t sR2,
                      const float* rowbias, int64_t ldrb, int32_t rowgroup, int32_t M, int32_t N, int32_t K, int32_t batch1, int32_t batch2,
                      float alpha, int32_t act, psam_stream_t stream);
/* Same contract, fp32-accurate on the bf16 matrix pipe: each fp32 operand is split exactly into 3 bf16 terms while its
 * K slab is staged, 6 of the 9 partial products are accumulated in fp32 (dropped terms <= 2^-24 relative).  128x128
 * tiles: use for M, N >= 128. */
int32_t psam_gemm_bf16x6(const float* A, int64_t lda, int64_t sA1, int64_t sA2, const float* W, int64_t ldw, int64_t sW1, int64_t sW2, float* C,
                         int64_t ldc, int64_t sC1, int64_t sC2, const float* bias, const float* residual, int64_t ldr, int64_t sR1, int64_t sR2,
                         const float* rowbias, int64_t ldrb, int32_t rowgroup, int32_t M, int32_t N, int32_t K, int32_t batch1, int32_t batch2,
                         float alpha, int32_t act, psam_stream_t stream);
/* fp32-grade GEMM on the fp16 matrix pipe ("f16x3"; 2-D form of the psam_gemm_f32 contract, same reference call sites).
 * Each operand row is scaled by a power of two (row maximum into [2^14, 2^15): psam_row_scale_f16; a static weight's scales are
 * computed once at load) and split into hi + lo fp16; hi*hi + hi*lo + lo*hi are accumulated in fp32 (dropped lo*lo and split
 * residual <= 3*2^-22 relative per product) and the epilogue multiplies by 1/(scaleA[row] scaleW[col]) exactly. */
int32_t psam_row_scale_f16(const float* X, int64_t ldx, int32_t rows, int32_t cols, float* scale, psam_stream_t stream);
/* Production form of the large GEMMs ("f16x3p", csrc/gemm_f16x3p.hip): BOTH operands pre-packed in the "g8" form -- per group of 8
 * consecutive k: [hi k0..k7 (8 x fp16) | lo k0..k7] in the same 32-bit-per-element container -- so that one 16-byte chunk is one
 * matrix-instruction operand and a K slab moves global -> LDS by LDS-DMA with no staging registers or arithmetic.  K % 32 == 0 (pack
 * pads with zeros up to the next multiple of 32: ldp >= that), K >= 128, packed rows 32-byte aligned.  Replaces the same reference
 * call sites as psam_gemm_f32 (every nn.Linear of pc_sam/model/ *.py and of the timm Eva blocks). */
int32_t psam_pack_rows_f16x2_g8(const float* X, int64_t ldx, const float* scale, int32_t rows, int32_t K, void* P, int64_t ldp, psam_stream_t stream);
int32_t psam_gemm_f16x3p(const void* A, int64_t lda, const float* scaleA, const void* W, int64_t ldw, const float* scaleW, float* C, int64_t ldc,
                         const float* bias, const float* residual, int64_t ldr, const float* rowbias, int64_t ldrb, int32_t rowgroup, int32_t M,
                         int32_t N, int32_t K, float alpha, int32_t act, psam_stream_t stream);
void psam_gemm_f16x3p_force_config(int32_t cfg); /* tuning hook: tile / ring configuration index, -1 = auto */
/* Epilogue of the packed-operand GEMMs: 1 = register-only epilogue on transposed accumulator tiles (csrc/gemm_epilogue_t.h) wherever the
 * launch's options allow it, 0 = always the LDS-transposition epilogue (csrc/gemm_epilogue.h), -1 = default (environment PSAM_GEMM_TR, else 1).
 * Both give the same bits; the hook exists for A/B measurements and the bitwise test. */
void psam_gemm_f16x3p_force_epilogue(int32_t mode);
/* split-K launches of psam_gemm_f16x3p_ex: 1 = in-kernel fix-up (the last workgroup of a tile adds the partial accumulators in split order and runs the
 * epilogue; no reduction launch), 0 = partial planes + reduction launch, -1 = the default (fix-up wherever psam_gemm_fuse_t.counters is given and the workspace allows;
 * environment PSAM_GEMM_SPLITK_FIXUP=0 switches it off).  Tuning / test hook: both forms give the same bits for power-of-two scales. */
void psam_gemm_f16x3p_force_splitk_fixup(int32_t mode);
#ifdef PSAM_BUILD_EXPERIMENTS
/* Batch-sized launches (>= 2048 rows) of the packed-operand GEMM on the 128x128 register-epilogue configuration: the persistent kernel (csrc/gemm_f16x3c.hip: resident
 * workgroups draw whole tiles from per-XCD queues and keep one continuous stream of K slabs going across tile boundaries; the same bits, measured the same
 * time: profiles/r05/r05_continuous_sweep.txt) -- -1 = default (environment PSAM_GEMM_CONTINUOUS, else off), 0 = never, 1 = wherever it applies. */
void psam_gemm_f16x3p_force_continuous(int32_t mode);
#endif
/* ARRIVAL-COUNTER BLOCK (round 6: the library allocates nothing and keeps no per-stream state).  Kernels with an in-kernel fix-up -- the split-K GEMM
 * (the last workgroup of a tile adds the partial accumulators), the key-split attention (the last arrival combines the partial softmax states), the
 * skinny Linear + LayerNorm of the decoder's token side (the last workgroup normalises the rows) -- count their workgroups in through device memory the
 * CALLER owns: PSAM_COUNTER_BYTES bytes, zero before the first use (one hipMemsetAsync), left zero by every launch that completes.  Launches that
 * share a block must be ordered with respect to each other (one stream, or one graph); streams or graphs that run CONCURRENTLY need a block each.
 * A captured launch carries the block's address like any other buffer: a graph replays on any stream.  After a FAILED launch re-zero the block.
 * NULL where an entry takes `counters`: the entry runs without the in-kernel fix-up (reduction pass, unsplit attention, two launches). */
#define PSAM_COUNTER_BYTES 65536
/* psam_attention_f16x3(_ex) with few workgroups (one cloud, head dim in (64, 128]): up to four workgroups per (query block, head) share the key tiles and
 * the last arrival combines their partial softmax states in split order.  0 = never split, 1 / -1 = the default (environment PSAM_ATTN_KEYSPLIT=0: off). */
void psam_attention_f16x3_force_keysplit(int32_t mode);
/* psam_twoway_decoder: 1 = the patch-side projections of a layer run on a side stream forked from (and joined back into) the caller's stream -- also
 * inside a graph capture --, 0 / -1 = everything in sequence on the caller's stream (the default: the fork measured slower, csrc/blocks.hip TwSide;
 * environment PSAM_TWOWAY_FORK=1 switches it on).  Same kernels, same bits. */
#ifdef PSAM_BUILD_EXPERIMENTS      /* measured-and-rejected paths: built only with PSAM_BUILD_EXPERIMENTS=1 (point_sam_amd/build.py) */
void psam_twoway_decoder_force_fork(int32_t mode);
#endif
/* 1 when psam_gemm_f16x3p_ex accepts psam_gemm_fuse_t.row_ln_* for N output columns (Linear -> LayerNorm -> activation in one GEMM; common.py:493-496,
 * mask_decoder.py:53-59): N == 256 always, N == 512 with the register epilogue (packed output scaled by the a-priori bound out_k2, out_k1 == 0). */
int32_t psam_gemm_f16x3p_fused_row_ln(int32_t N);
/* The same GEMM with fused extras (all optional; M % 256 == 0 and N % 128 == 0 required when any is used) -- what lets the EVA02 MLP
 * `fc2(LayerNorm(SiLU(fc1_g x) * fc1_x x))` (timm SwiGLU with scale_mlp) run as two GEMMs and nothing in between:
 *   pack_out : C receives the g8-packed output rows (the next GEMM's A operand), scaled per row by out_scale[row] (written here) =
 *              the power of two that puts the BOUND B = out_k1 / scaleA[row] + out_k2 (B^2 with the SwiGLU gate) of the row's magnitude
 *              into [2^14, 2^15) -- out_k1 = 2^15 sqrt(K) max_n ||W[n]||_2, out_k2 = max |bias| (Cauchy-Schwarz);
 *   stats    : [M, psam_gemm_f16x3p_stat_segs(N), 2] (mean, centred sum of squares) of every 32-column segment of the gated rows over
 *              the columns < stat_cols; psam_ln_stats_finalize merges them (fixed order) into the LayerNorm's mean / rstd per row;
 *   gmax_*   : gmax_out [M / gmax_k, >= N] (row stride gmax_ld) = per-column maximum over every group of gmax_k (32 or 64) consecutive
 *              output rows -- PatchEncoder's max-pool over the group members (common.py:491,497); no_store: C is not written;
 *   row_ln_* : (N == 256) LayerNorm over each output row before the activation -- Linear -> LayerNorm -> GELU of the decoder's upscaling
 *              MLP (mask_decoder.py:53-59) in one epilogue (with pack_out pass out_k1 = 0, out_k2 >= sqrt(255) max|gamma| + max|beta|);
 *   hyper    : (N == 256) masks[z, c, n] = <hyper[z, c, :], out[z * hyper_rows + n, :]>, c < hyper_c <= 4, hyper_rows % 32 == 0 -- the hyper-network product
 *              of mask_decoder.py:171-176 taken from the rows as they are finished (with no_store the [rows, 256] output is not written);
 *   ln_*     : LayerNorm of the A rows folded into the GEMM: C = rstd[row] (A W'^T - mean[row] c[col]) + bias (+ residual), with the
 *              caller's W' = W * gamma (per column), c = W' 1, bias = W beta + b. */
typedef struct {
    float* out_scale; float out_k1, out_k2; int32_t pack_out;
    float* stats; int32_t stat_cols;
    const float* ln_mean; const float* ln_rstd; const float* ln_c;
    float* gmax_out; int64_t gmax_ld; int32_t gmax_k; int32_t no_store;
    const float* row_ln_g; const float* row_ln_b; float row_ln_eps;
    const float* hyper; float* masks; int32_t hyper_c; int32_t hyper_rows;
    int64_t hyper_pstride;
    /* split-K: splitk > 1 workgroups per output tile share the K loop; partial products go to splitk planes of splitk_ws (splitk_plane >=
     * M * N floats apart), added in a fixed order with bias / activation / residual applied afterwards.  No other extras, act != SwiGLU,
     * no rowbias, N % 4 == 0, splitk <= K / 128.  psam_gemm_f16x3p_splitk() suggests the factor for a shape (1: none). */
    float* splitk_ws; int64_t splitk_plane; int32_t splitk;
    /* pack_out with a bound PER ROW: out_scale[row] = f16_row_scale(out_bound[row]) instead of the k1 / k2 form (out_bound[row] >= max |output
     * row|; psam_layernorm_ex2 writes it from the L2 norm of the A row).  Tighter than k1 / scaleA + k2 by sqrt(K) max|a| / ||a||_2. */
    const float* out_bound;
    /* split-K with the in-kernel fix-up: the caller's arrival-counter block (PSAM_COUNTER_BYTES, see above); NULL = partial planes + reduction launch */
    int32_t* counters;
} psam_gemm_fuse_t;
int32_t psam_gemm_f16x3p_splitk(int32_t M, int32_t N, int32_t K, int32_t act);
/* hyper without row_ln_*: any N % 128 == 0, M % 256 == 0; every 64-column wave tile contributes the partial products of its columns:
 * masks then holds psam_gemm_f16x3p_hyper_planes(N, 0) = N / 64 planes of [Z, C, hyper_rows], hyper_pstride elements apart, and
 * psam_sum_planes adds them in a fixed order.  With row_ln_* (full-row tile, N == 256) there is one plane. */
int32_t psam_gemm_f16x3p_hyper_planes(int32_t N, int32_t with_row_ln);
int32_t psam_sum_planes(const float* parts, int32_t P, int64_t pstride, int64_t count, float* out, psam_stream_t stream);
int32_t psam_gemm_f16x3p_stat_segs(int32_t N);
int32_t psam_gemm_f16x3p_ex(const void* A, int64_t lda, const float* scaleA, const void* W, int64_t ldw, const float* scaleW, float* C, int64_t ldc,
                            const float* bias, const float* residual, int64_t ldr, const float* rowbias, int64_t ldrb, int32_t rowgroup, int32_t M,
                            int32_t N, int32_t K, float alpha, int32_t act, const psam_gemm_fuse_t* fuse, psam_stream_t stream);
int32_t psam_ln_stats_finalize(const float* stats, int32_t rows, int32_t segs, int32_t cols, float eps, float* mean, float* rstd, psam_stream_t stream);
/* Row scales and g8 packing of fp32 rows in ONE pass (an activation no LayerNorm produced, e.g. the attention output). */
int32_t psam_scale_pack_rows_g8(const float* X, int64_t ldx, int32_t rows, int32_t K, void* P, int64_t ldp, float* scale, psam_stream_t stream);
int32_t psam_linear(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* residual, int64_t ldr, float* y,
                    int64_t ldy, int32_t M, int32_t N, int32_t K, int32_t act, psam_stream_t stream);
void psam_gemm_bf16x6_force_config(int32_t cfg); /* tuning hook: 0=128x128, 1=128x64 tiles, -1=auto */
void psam_gemm_force_config(int32_t cfg); /* tuning hook: 0=128x128, 1=128x64, 2=64x64 tiles, -1=auto */

/* y = act(LayerNorm(x (+ res))).  Replaces nn.LayerNorm / apex FusedLayerNorm (pc_sam/utils/torch_utils.py:28-38)
 * at common.py:488,494, timm blocks, transformer.py:59,128-138, mask_decoder.py:55. */
int32_t psam_layernorm(const float* x, int64_t ldx, const float* res, int64_t ldr, const float* w, const float* b, float* y, int64_t ldy,
                       int64_t rows, int32_t cols, float eps, int32_t act, psam_stream_t stream);
/* Same, plus row_scale[rows] (optional): the power-of-two f16x3 scale of every OUTPUT row (psam_row_scale_f16 fused in). */
/* psam_layernorm_ex: pack != 0 writes y as the g8-packed form of the row-scaled output, zero-padded to cols rounded up to 32 when ldy has
 * room (needs row_scale; 256 <= cols <= 4096; 32-byte aligned output rows): the A operand of psam_gemm_f16x3p. */
int32_t psam_layernorm_ex(const float* x, int64_t ldx, const float* res, int64_t ldr, const float* w, const float* b, float* y, int64_t ldy,
                          int64_t rows, int32_t cols, float eps, int32_t act, float* row_scale, int32_t pack, psam_stream_t stream);
/* psam_layernorm_ex2: row_bound (optional, with pack, [rows]) receives c2 t^2 + c1 t + c0, t = 1.0001 x the L2 norm of the output row: the
 * a-priori bound of the rows of the GEMM that consumes this output (|W_n . h + b_n| <= ||W_n|| t + |b_n|; psam_gemm_fuse_t.out_bound). */
int32_t psam_layernorm_ex2(const float* x, int64_t ldx, const float* res, int64_t ldr, const float* w, const float* b, float* y, int64_t ldy,
                           int64_t rows, int32_t cols, float eps, int32_t act, float* row_scale, int32_t pack, float* row_bound, float c2,
                           float c1, float c0, psam_stream_t stream);
int32_t psam_layernorm_rs(const float* x, int64_t ldx, const float* res, int64_t ldr, const float* w, const float* b, float* y, int64_t ldy,
                          int64_t rows, int32_t cols, float eps, int32_t act, float* row_scale, psam_stream_t stream);

/* out = LayerNorm_H(SiLU(gx[:,0:H]) * gx[:,xoff:xoff+H]), zero-padded to ldo columns.
 * Replaces timm SwiGLU (act, mul, norm) inside eva02 blocks (called through pc_encoder.py:138-139). */
int32_t psam_swiglu_ln(const float* gx, int64_t ldg, int32_t xoff, const float* w, const float* b, float* out, int64_t ldo, int64_t rows,
                       int32_t H, float eps, psam_stream_t stream);

/* softmax(q k^T * scale) v per (batch, head), flash-style on the matrix cores; q/k/v/o are [B,L,H*hd] views.
 * Replaces F.scaled_dot_product_attention inside timm EvaAttention (pc_encoder.py:138-139).
 * hd in {16,24,32,48,64,88,96,128}. */
int32_t psam_attention_f32(const float* q, int64_t ldq, int64_t sq, const float* k, int64_t ldk, int64_t sk, const float* v, int64_t ldv,
                           int64_t sv, float* o, int64_t ldo, int64_t so, int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd,
                           float scale, psam_stream_t stream);
/* Same contract on the fp16 matrix pipe with fp32-grade products (power-of-two scaling + hi/lo fp16 split of Q, K, V and of the
 * probabilities, 3 MFMA products each; csrc/attention.hip).  head_dim 64, or a multiple of 8 in (64, 128] (computed zero-padded to 128). */
int32_t psam_attention_f16x3(const float* q, int64_t ldq, int64_t sq, const float* k, int64_t ldk, int64_t sk, const float* v, int64_t ldv,
                           int64_t sv, float* o, int64_t ldo, int64_t so, int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd,
                           float scale, psam_stream_t stream);
/* The same with a PACKED output for the output projection (psam_gemm_f16x3p): a_scale [B*Lk] = the row scales of the qkv GEMM's A
 * operand (LayerNorm output), k1 = 2^15 sqrt(D) max_n ||W_v[n]||_2, k2 = max |b_v|: every |V| of a cloud -- hence every attention
 * output, a convex combination of V rows -- is below B = k1 / min_rows(a_scale) + k2; o receives the g8-packed rows scaled by the
 * power of two that puts B into [2^14, 2^15), o_scale [B*Lq] that scale.  a_scale == NULL: plain fp32 output. */
int32_t psam_attention_f16x3_ex(const float* q, int64_t ldq, int64_t sq, const float* k, int64_t ldk, int64_t sk, const float* v, int64_t ldv,
                                int64_t sv, float* o, int64_t ldo, int64_t so, int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd, float scale,
                                const float* a_scale, float k1, float k2, float* o_scale, psam_stream_t stream);
/* the same with the KEY SPLIT: few workgroups (one cloud, head dim in (64, 128]: 16 heads x 4 query blocks = 64 on 256 CUs) -- up to max_keysplit <= 4
 * workgroups per (query block, head) share the key tiles and the last arrival combines their partial softmax states in split order, inside the kernel.
 * Needs ks_ws (psam_attention_f16x3_keysplit_ws_bytes(...) bytes of scratch for the partial states) and the caller's arrival-counter block `counters`
 * (PSAM_COUNTER_BYTES, see above); with either NULL, or max_keysplit = 1, the launch is unsplit (what psam_attention_f16x3_ex does; a caller that keeps
 * the chip busy from several streams -- throughput -- is better off without the split's extra work). */
size_t psam_attention_f16x3_keysplit_ws_bytes(int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd, int32_t max_keysplit);
int32_t psam_attention_f16x3_ex2(const float* q, int64_t ldq, int64_t sq, const float* k, int64_t ldk, int64_t sk, const float* v, int64_t ldv,
                                int64_t sv, float* o, int64_t ldo, int64_t so, int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t hd, float scale,
                                const float* a_scale, float k1, float k2, float* o_scale, int32_t max_keysplit, void* ks_ws, size_t ks_ws_bytes,
                                int32_t* counters, psam_stream_t stream);

/* Self-attention on PRE-PACKED operands (head dim 64): qkv [B*L, ld] holds every row's q | k | v (column blocks of D = H*64 containers) in
 * the g8-packed hi|lo fp16 form of psam_gemm_f16x3p, all rows with ONE power-of-two scale (sc[b*L] is read) -- what the qkv GEMM writes with
 * psam_gemm_fuse_t {pack_out = 1, out_k1 = 0, out_k2 = an a-priori bound of |q|, |k|, |v|}.  Replaces the SDPA call of timm's EvaAttention
 * (pc_sam/model/pc_encoder.py:138-139) together with the conversions the fp32-input kernels above do per tile: K / V tiles move global -> LDS
 * by LDS-DMA, V is transposed on read (ds_read_b64_tr_b16), nothing is scaled or split in the kernel except the probabilities.
 * o [B*L, ldo]: g8-packed output for the projection GEMM, o_scale [B*L] = f16_row_scale(v_bound) for every row (v_bound >= max |v|). */
int32_t psam_attention_packed(const void* qkv, int64_t ld, const float* sc, float* o, int64_t ldo, float* o_scale, int32_t B, int32_t H, int32_t L,
                              int32_t hd, float scale, float v_bound, psam_stream_t stream);
void psam_attention_packed_force_variant(int32_t v); /* tuning hook: -1 default, 0 = 256-row workgroups / 3-tile ring, 1 = two 128-row workgroups per CU / 2-tile ring */

/* y [M, N] = act(x [M, K] W [N, K]^T + bias) + residual for M <= 64 rows (K % 16 == 0, rows 16-byte aligned; act: none / GELU / ReLU),
 * exact fp32 products.  The decoder's token-side nn.Linear calls (7 output tokens per prompt): transformer.py:109-236. */
int32_t psam_linear_skinny(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* residual, int64_t ldr,
                           float* y, int64_t ldy, int32_t M, int32_t N, int32_t K, int32_t act, psam_stream_t stream);

/* Up to PSAM_SKINNY_MAX_JOBS skinny Linears over the same M <= 64 input rows in one launch: y_i = act_i((x_i + xadd_i) W_i^T + bias_i), xadd optional (the
 * decoder's q = k = queries + query_pe, v = queries: transformer.py:153-170,214-236).  All jobs share ldx (x rows), ldxadd, ldw, M and K. */
typedef struct {
    const float* x; const float* xadd; const float* W; const float* bias; float* y;
    int64_t ldy;
    int32_t N, act;
} psam_skinny_job_t;
#define PSAM_SKINNY_MAX_JOBS 6
typedef struct {
    psam_skinny_job_t job[PSAM_SKINNY_MAX_JOBS];
    int32_t n;
} psam_skinny_jobs_t;
int32_t psam_linear_skinny_multi(const psam_skinny_jobs_t* jobs, int64_t ldx, int64_t ldxadd, int64_t ldw, int32_t M, int32_t K, psam_stream_t stream);
/* The same for ANY number of rows (a few hundred to a few thousand: the decoder's patch rows), exact fp32 products, 32 x 32 tiles: y_i [M, N_i] =
 * act_i((x_i + xadd_i) W_i^T + bias_i).  xadd_i (optional) holds sets of rows_per_set rows; row r adds row (r / (rep * rows_per_set)) * rows_per_set +
 * r % rows_per_set (key_pe of the cloud, shared by its rep prompt sets: transformer.py:160-175). */
int32_t psam_linear_rows_multi(const psam_skinny_jobs_t* jobs, int64_t ldx, int64_t ldxadd, int32_t rows_per_set, int32_t rep, int64_t ldw, int64_t M, int32_t K,
                               psam_stream_t stream);
/* Skinny Linear + residual + LayerNorm in one launch: y [M, 256] = LayerNorm(x W^T + bias + residual) * ln_w + ln_b, M <= 64 rows, N == 256 -- the
 * `queries = norm(queries + out_proj(attn))` / `norm3(queries + mlp(queries))` steps of the decoder's token side (transformer.py:153-176).  The last
 * workgroup to finish its columns normalises the rows (`counters`: the caller's arrival-counter block, PSAM_COUNTER_BYTES, see above; required).  K is
 * split over up to 8 workgroup rows (lin2 of the token MLP: K = 2048).  tmp: psam_linear_skinny_ln_tmp_floats(M, K) floats. */
size_t psam_linear_skinny_ln_tmp_floats(int32_t M, int32_t K);
int32_t psam_linear_skinny_ln(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* residual, int64_t ldr,
                              const float* ln_w, const float* ln_b, float eps, float* tmp, float* y, int64_t ldy, int32_t M, int32_t N, int32_t K,
                              int32_t* counters, psam_stream_t stream);
/* The same for ANY number of rows and a short K (K % 16 == 0, K <= 512): y [M, 256] = LayerNorm(x W^T + bias + residual) * ln_w + ln_b, a workgroup per
 * 16 whole rows, exact fp32 products -- `keys = norm4(keys + out_proj(attn))` of the decoder's patch side (transformer.py:170-175; K = 128). */
int32_t psam_linear_ln256(const float* x, int64_t ldx, const float* W, int64_t ldw, const float* bias, const float* residual, int64_t ldr, const float* ln_w,
                          const float* ln_b, float eps, float* y, int64_t ldy, int64_t M, int32_t N, int32_t K, psam_stream_t stream);
/* psam_scale_pack_rows_g8 of X + add[(row / (rep * rows_per_set)) * rows_per_set + row % rows_per_set]: the broadcast positional add of the decoder's
 * keys (k = keys + key_pe, transformer.py:160-170) folded into the pass that scales and packs the rows for the k / q projection GEMMs. */
int32_t psam_scale_pack_rows_g8_add(const float* X, int64_t ldx, const float* add, int64_t ldadd, int32_t rows_per_set, int32_t rep, int32_t rows, int32_t K,
                                    void* P, int64_t ldp, float* scale, psam_stream_t stream);

/* Both packed forms of the decoder's patch rows in one pass (K <= 256): P_sum / scale_sum = psam_scale_pack_rows_g8_add's output (keys + key_pe, the
 * operand of the k / q projections), P_x / scale_x = psam_scale_pack_rows_g8 of X alone (the operand of the v projection); transformer.py:160-170. */
int32_t psam_scale_pack_rows_g8_add_dual(const float* X, int64_t ldx, const float* add, int64_t ldadd, int32_t rows_per_set, int32_t rep, int32_t rows, int32_t K,
                                         void* P_sum, float* scale_sum, void* P_x, float* scale_x, int64_t ldp, psam_stream_t stream);

/* ONE EVA02 (SwiGLU) transformer block of the patch encoder in one call (csrc/blocks.hip) -- timm's block as the reference runs it
 * (pc_sam/model/pc_encoder.py:138-139, no rope): x += proj(SDPA(LN1 x)); x += fc2(LN(SiLU(fc1_g h) * fc1_x h)), h = LN2 x.  "f16x3" arithmetic with
 * every hand-over fused as the Python host does it: eight launches, nothing in between.  Head dim 64, dim % 32 == 0, B * L % 256 == 0.
 * psam_eva_block_prepare (load time; copies the weights to the host once, uses a temporary device buffer, synchronises) packs a block's weights
 * -- given in the reference's / timm's state-dict layout, [out, in] fp32 device pointers -- into `prepared` (psam_eva_block_prepared_bytes) and
 * fills the host-side `plan`; the LayerNorm parameters and the projection bias are read through the plan at run time and must stay alive. */
typedef struct {
    const float *norm1_w, *norm1_b, *q_w, *q_b, *k_w, *v_w, *v_b, *proj_w, *proj_b, *norm2_w, *norm2_b;      /* attn.k_proj has no bias */
    const float *fc1_g_w, *fc1_g_b, *fc1_x_w, *fc1_x_b, *mlp_norm_w, *mlp_norm_b, *fc2_w, *fc2_b;
    int32_t dim, heads, hidden;      /* hidden: SwiGLU width (2730 for eva02_large) */
    float eps;                       /* LayerNorm eps of the transformer (1e-6) */
} psam_eva_block_weights_t;
typedef struct {
    int32_t dim, heads, hidden, hidden_pad;
    float eps, qkv_bound, v_bound, u_c2, u_c1, u_c0;      /* a-priori bounds of the packed hand-overs */
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b, *proj_b;
    int64_t o_wqkv, o_sqkv, o_bqkv, o_wproj, o_sproj, o_w1, o_s1, o_b1, o_w2g, o_s2g, o_lnc, o_lnd;      /* byte offsets into `prepared` */
} psam_eva_block_plan_t;
size_t psam_eva_block_prepared_bytes(int32_t dim, int32_t hidden);
int32_t psam_eva_block_prepare(const psam_eva_block_weights_t* weights, psam_eva_block_plan_t* plan, void* prepared, size_t prepared_bytes, psam_stream_t stream);
size_t psam_eva_block_ws_bytes(int64_t M, int32_t dim, int32_t hidden);
/* x [B*L, dim] fp32, updated in place; ws: psam_eva_block_ws_bytes(B*L, dim, hidden) bytes of scratch */
int32_t psam_eva_block(const psam_eva_block_plan_t* plan, const void* prepared, float* x, int32_t B, int32_t L, void* ws, size_t ws_bytes, psam_stream_t stream);

/* ONE GELU-MLP transformer block with a fused qkv projection (timm eva_giant_patch14_560, configs/model/giant.yaml; called through
 * pc_sam/model/pc_encoder.py:138-139, no rope): x += proj(SDPA(qkv(LN1 x) + [q_bias | 0 | v_bias])); x += fc2(GELU(fc1(LN2 x))).  "f16x3" arithmetic,
 * sequenced as the Python host sequences it (bit-identical): LayerNorm (packed rows) | qkv GEMM | attention on the fp16 pipe for head dims that are
 * multiples of 8 in (64, 128] (88 runs zero-padded to 128) or 64, output packed for the projection | projection + residual | LayerNorm | fc1 + GELU |
 * fc2 + residual.  Few token rows (one cloud: M = 512) leave most of the chip idle on whole-tile launches: every plain GEMM of the block takes the
 * library's split-K factor (psam_gemm_f16x3p_splitk) with the partial planes in the caller's workspace; when fc1 needs no split its epilogue hands
 * GELU(.) to fc2 g8-packed (per-row bound from the LayerNorm's L2 norm) and the scale + pack pass disappears.  dim % 32 == 0, hidden % 32 == 0,
 * 256 <= dim <= 4096, B * L >= 256.  `precision`: PSAM_PRECISION_F16X3 (the f32 / bf16x6 arithmetic of the same block is reachable through the
 * per-operator entries only). */
#define PSAM_PRECISION_F16X3 2
typedef struct {
    const float *norm1_w, *norm1_b, *qkv_w, *q_bias, *v_bias, *proj_w, *proj_b, *norm2_w, *norm2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
    int32_t dim, heads, hidden, precision;
    float eps;
} psam_eva_gelu_block_weights_t;
typedef struct {
    int32_t dim, heads, hidden, precision;
    float eps, vk1, vk2, u_c1, u_c0;      /* attention-output bound from the LayerNorm row scale; fc1 row bound c1 t + c0 from t = ||LN2 x||_2 */
    int32_t attn_keysplit;                /* max key split of the attention (psam_attention_f16x3_ex2): _prepare sets 4; 1 = off (throughput callers) */
    const float *norm1_w, *norm1_b, *norm2_w, *norm2_b, *proj_b, *fc1_b, *fc2_b;
    int64_t o_wqkv, o_sqkv, o_bqkv, o_wproj, o_sproj, o_w1, o_s1, o_w2, o_s2;      /* byte offsets into `prepared` */
} psam_eva_gelu_block_plan_t;
size_t psam_eva_gelu_block_prepared_bytes(int32_t dim, int32_t hidden);
int32_t psam_eva_gelu_block_prepare(const psam_eva_gelu_block_weights_t* weights, psam_eva_gelu_block_plan_t* plan, void* prepared, size_t prepared_bytes,
                                    psam_stream_t stream);
size_t psam_eva_gelu_block_ws_bytes(int64_t M, int32_t dim, int32_t hidden);      /* includes the split-K planes and the key-split partial states */
/* x [B*L, dim] fp32, updated in place.  counters: the caller's arrival-counter block (PSAM_COUNTER_BYTES, see above) for the split-K fix-up and the
 * key-split attention of single-cloud shapes; NULL = reduction passes / unsplit attention (same bits for the GEMMs, round-off for the attention). */
int32_t psam_eva_gelu_block(const psam_eva_gelu_block_plan_t* plan, const void* prepared, float* x, int32_t B, int32_t L, void* ws, size_t ws_bytes,
                            int32_t* counters, psam_stream_t stream);

/* PatchEncoder.forward on kNN groups in one call (csrc/blocks.hip): the mini-PointNet of the patch embedding (features = rgb) and of the mask
 * encoder (features = mask logits) -- pc_sam/model/common.py:477-506 after the gather of :99-120 / :126-187 -- "f16x3", fused as the Python host
 * runs it (six launches; both max-pools inside GEMM epilogues -- for group sizes above 64 as 64-row parts + psam_group_max over the parts).
 * hidden_dims[0] == 128, group size 32 or a multiple of 64, B * rep * G * K % 256 == 0.
 * Weights: the reference's conv1.{0,1,3} / conv2.{0,1,3} tensors ([out, in] fp32 device pointers); the plan keeps pointers to the small ones. */
typedef struct {
    const float *c10_w, *c10_b, *c11_w, *c11_b, *c13_w, *c13_b, *c20_w, *c20_b, *c21_w, *c21_b, *c23_w, *c23_b;
    int32_t cin, h0, h1, cout;
    float eps;
} psam_patch_encoder_weights_t;
typedef struct {
    int32_t cin, h0, h1, cout;
    float eps, k1, k2, ln21_bound;      /* ln21_bound >= max |gamma| sqrt(h1) + max |beta| of conv2.1: scale of the rows the fused conv2.0 GEMM packs */
    const float *c10_w, *c10_b, *c11_w, *c11_b, *c13_b, *c20_w, *c20_b, *c21_w, *c21_b, *c23_b;
    int64_t o_w13, o_s13, o_w20m, o_s20m, o_w20x, o_s20x, o_w23, o_s23;
} psam_patch_encoder_plan_t;
size_t psam_patch_encoder_prepared_bytes(int32_t h0, int32_t h1, int32_t cout);
int32_t psam_patch_encoder_prepare(const psam_patch_encoder_weights_t* weights, psam_patch_encoder_plan_t* plan, void* prepared, size_t prepared_bytes,
                                   psam_stream_t stream);
size_t psam_patch_encoder_ws_bytes(int64_t rows, int64_t groups, int32_t h0, int32_t h1);
/* feats [B*rep, N, C] (rep mask sets per cloud share xyz / centers / knn_idx); center_idx [B, G] != NULL: centralize_features; out [B*rep*G, cout] */
int32_t psam_patch_encoder(const psam_patch_encoder_plan_t* plan, const void* prepared, const float* xyz, const float* feats, const float* centers,
                           const int64_t* knn_idx, const int64_t* center_idx, int32_t B, int32_t rep, int32_t N, int32_t G, int32_t K, int32_t C,
                           float radius, float* out, void* ws, size_t ws_bytes, psam_stream_t stream);

/* The mask decoder after its transformer in one call (csrc/blocks.hip; pc_sam/model/mask_decoder.py:146-176): 3-NN interpolation G -> N,
 * output_upscaling and the hyper-network products, "f16x3", fused as the Python host runs it (the first Linear on the G patch rows before the
 * interpolation, LayerNorm + GELU inside the interpolation kernel, the products inside the second GEMM's epilogue).  transformer_dim 256. */
typedef struct { const float *u0_w, *u0_b, *u1_w, *u1_b, *u3_w, *u3_b; int32_t dim; float eps; } psam_upscale_weights_t;
typedef struct { int32_t dim; float eps; const float *u0_w, *u0_b, *u1_w, *u1_b, *u3_b; int64_t o_w0, o_s0, o_w3, o_s3; } psam_upscale_plan_t;
size_t psam_upscale_masks_prepared_bytes(int32_t dim);
int32_t psam_upscale_masks_prepare(const psam_upscale_weights_t* weights, psam_upscale_plan_t* plan, void* prepared, size_t prepared_bytes, psam_stream_t stream);
size_t psam_upscale_masks_ws_bytes(int64_t Z, int32_t N, int32_t G, int32_t C, int32_t dim);
/* keys [Z*G, 256], idx3 / w3 [Z / rep, N, 3] (psam_three_nn), hyper [Z, C, 256] (psam_mlp3) -> masks [Z, C, N]; Z * N % 256 == 0, N % 32 == 0, C <= 4 */
int32_t psam_upscale_masks(const psam_upscale_plan_t* plan, const void* prepared, const float* keys, const int64_t* idx3, const float* w3, const float* hyper,
                           int32_t rep, int64_t Z, int32_t N, int32_t G, int32_t C, float* masks, void* ws, size_t ws_bytes, psam_stream_t stream);

/* TwoWayTransformer.forward in one call (csrc/blocks.hip; pc_sam/model/transformer.py:61-100 with TwoWayAttentionBlock :103-176 and Attention
 * :179-236): the decoder's transformer over the patch tokens and the output / prompt tokens, sequenced as the Python host sequences it (per
 * nn.Linear the kernel the host would pick by row count; psam_attention_small; LayerNorm with the residual folded in).  Weights in the
 * reference's state-dict layout ([out, in] fp32 device pointers); _prepare packs every matrix for the rows >= 256 case. */
#define PSAM_TWOWAY_MAX_DEPTH 4
typedef struct { const float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *o_w, *o_b; } psam_attn_weights_t;      /* q_proj, k_proj, v_proj, out_proj */
typedef struct {
    psam_attn_weights_t self_attn, t2i, i2t;      /* self_attn, cross_attn_token_to_image, cross_attn_image_to_token */
    const float *n1_w, *n1_b, *n2_w, *n2_b, *n3_w, *n3_b, *n4_w, *n4_b, *m1_w, *m1_b, *m2_w, *m2_b;      /* norm1..4, mlp.lin1, mlp.lin2 */
} psam_twoway_layer_weights_t;
typedef struct {
    int32_t depth, dim, heads, mlp, downsample;      /* 2, 256, 8, 2048, 2 in every reference config */
    float eps;
    const psam_twoway_layer_weights_t* layers;       /* [depth] (host array) */
    psam_attn_weights_t final_attn;                  /* final_attn_token_to_image */
    const float *nf_w, *nf_b;                        /* norm_final_attn */
} psam_twoway_weights_t;
typedef struct {
    psam_twoway_weights_t weights;
    psam_twoway_layer_weights_t layers[PSAM_TWOWAY_MAX_DEPTH];
    int64_t o_packed[14 * PSAM_TWOWAY_MAX_DEPTH + 4], o_scales[14 * PSAM_TWOWAY_MAX_DEPTH + 4];
    int64_t o_cat_packed[PSAM_TWOWAY_MAX_DEPTH], o_cat_scales[PSAM_TWOWAY_MAX_DEPTH], o_cat_bias[PSAM_TWOWAY_MAX_DEPTH];      /* per layer: [t2i.k_proj | i2t.q_proj] as one weight */
} psam_twoway_plan_t;
size_t psam_twoway_decoder_prepared_bytes(int32_t depth, int32_t dim, int32_t mlp, int32_t downsample);
int32_t psam_twoway_decoder_prepare(const psam_twoway_weights_t* weights, psam_twoway_plan_t* plan, void* prepared, size_t prepared_bytes, psam_stream_t stream);
size_t psam_twoway_decoder_ws_bytes(int64_t Z, int32_t T, int32_t G, int32_t dim, int32_t mlp);
/* tokens [Z*T, dim] (= query_pe), keys [Z*G, dim] in / out (src -> the transformer's second output), pos [Z / rep, G, dim] -> queries [Z*T, dim].
 * counters: the caller's arrival-counter block (PSAM_COUNTER_BYTES, see above) for the fused Linear + LayerNorm launches of the regrouped sequence;
 * NULL = the operator-by-operator sequence (the same operators, other launch boundaries: results equal to fp32 round-off). */
int32_t psam_twoway_decoder(const psam_twoway_plan_t* plan, const void* prepared, const float* tokens, float* keys, const float* pos, int32_t rep, int64_t Z,
                            int32_t T, int32_t G, float* queries, void* ws, size_t ws_bytes, int32_t* counters, psam_stream_t stream);
/* A/B and test hook: 0 = the operator-by-operator launch sequence (what the Python host issues), 1 = the regrouped sequence (fused Linear + LayerNorm
 * launches of the token side, merged projections; csrc/blocks.hip), -1 = default (environment PSAM_TWOWAY_FAST, else 1). */
void psam_twoway_decoder_force_fast(int32_t mode);

/* Token side of one TwoWayAttentionBlock in ONE launch (csrc/twoway.hip): self-attention + norm1, token -> image attention + norm2, the MLP
 * + norm3 on the Z * T <= 64 output-token rows, and the k / v projections of the image -> token attention that follows -- what
 * pc_sam/model/transformer.py:144-175 does for `queries`; mode 1: only the token -> image attention + LayerNorm of :91-99
 * (final_attn_token_to_image / norm_final_attn passed in the cq / co / n2 slots).  embedding_dim 256, attention_downsample_rate 2.
 * Weights are the reference's [out, in] matrices; kimg / vimg are the k / v projections of the patch tokens ([Z, G, 128] views, computed by the
 * image-side GEMMs); queries [Z*T, 256] is updated in place, ktok / vtok [Z*T, 128] receive the projections for the image -> token attention.
 * ws: psam_twoway_tokens_ws_floats(mlp) floats of scratch (mode 1: mlp = 0). */
typedef struct {
    int32_t Z, T, G, heads, mlp, mode, skip_pe, reserved;
    float eps;
    float* queries; const float* pe;
    const float* kimg; int64_t ldk, sk; const float* vimg; int64_t ldv, sv;
    const float *sq_w, *sq_b, *sk_w, *sk_b, *sv_w, *sv_b, *so_w, *so_b, *n1_g, *n1_b;
    const float *cq_w, *cq_b, *co_w, *co_b, *n2_g, *n2_b;
    const float *m1_w, *m1_b, *m2_w, *m2_b, *n3_g, *n3_b;
    const float *ik_w, *ik_b, *iv_w, *iv_b;
    float *ktok, *vtok;
    float* ws; int64_t ws_floats;
} psam_twoway_tokens_t;
#ifdef PSAM_BUILD_EXPERIMENTS      /* the one-launch token side measured slower than the launches it replaces: experiments build only */
int64_t psam_twoway_tokens_ws_floats(int32_t mlp);
int32_t psam_twoway_tokens(const psam_twoway_tokens_t* args, psam_stream_t stream);
#endif

/* Voronoi variant (PointCloudSAMNN, configs/model/voronoi.yaml).  psam_nn_group_feats: per-point features relative to the nearest centre --
 * mode 0 = NNGrouper.forward (pc_sam/model/common.py:203-211): [unit offset 3 | distance 1 | feats C]; mode 1 = MaskEncoderNN.forward
 * (prompt_encoder.py:281-287): [logit | offset 3 | distance] for mask set z = b * rep + r; rows ldo floats apart, zero-padded (ldo % 4 == 0
 * for the Linear that follows).  psam_scatter_amax: the max-pool of rows into their cells (torch scatter_reduce "amax": pc_encoder.py:190-193
 * with include_self = 0 -- cells without points end as 0 --, prompt_encoder.py:291-297 with include_self = 1 -- zeros take part);
 * dest(r) = idx[(r / rows_per_set / idx_rep) * rows_per_set + r % rows_per_set] + (r / rows_per_set) * set_stride; exact in any order. */
int32_t psam_nn_group_feats(const float* xyz, const float* centers, const int64_t* nn_idx, const float* feats, const float* logits, int32_t B, int32_t rep,
                            int32_t N, int32_t G, int32_t C, int32_t mode, float* out, int64_t ldo, psam_stream_t stream);
int32_t psam_scatter_amax(const float* x, int64_t ldx, const int64_t* idx, int64_t rows, int32_t C, int64_t rows_per_set, int64_t set_stride,
                          int32_t idx_rep, float* out, int64_t out_rows, int32_t include_self, psam_stream_t stream);

/* Three-layer ReLU MLP on a few rows -- mask_decoder.py:189-211 (MLP), the hyper-networks :171-176 and the IoU head :180 in one launch
 * each.  Weights stacked in the reference's [out, in] layout: w1 [M, dh, din], w2 [M, dh, dh], w3 [M, dout, dh]; biases [M, dh], [M, dh],
 * [M, dout].  MLP m reads the row x + z*ldx + m*sx (z < Z) and writes dout values at out + z*ldo + m*so.  din, dh <= 1024, % 4 == 0. */
int32_t psam_mlp3(const float* x, int64_t ldx, int64_t sx, const float* w1, const float* b1, const float* w2, const float* b2,
                  const float* w3, const float* b3, float* out, int64_t ldo, int64_t so, int32_t Z, int32_t M, int32_t din, int32_t dh,
                  int32_t dout, psam_stream_t stream);

/* Two psam_mlp3 stacks over the same Z rows in ONE launch -- the hyper-networks (mask tokens) and the IoU head (token 0) both read the transformer's
 * token output (mask_decoder.py:167-182).  Fields as the arguments of psam_mlp3; the same bits as two psam_mlp3 calls. */
typedef struct {
    const float *x, *w1, *b1, *w2, *b2, *w3, *b3;
    float* out;
    int64_t ldx, sx, ldo, so;
    int32_t M, din, dh, dout;
} psam_mlp3_args_t;
int32_t psam_mlp3_pair(const psam_mlp3_args_t* a, const psam_mlp3_args_t* b, int32_t Z, psam_stream_t stream);

/* Same contraction for the decoder's token-sized problems (any hd, few queries or few keys).
 * Replaces Attention.forward's matmul-softmax-matmul: pc_sam/model/transformer.py:226-233. */
int32_t psam_attention_small(const float* q, int64_t ldq, int64_t sq, const float* k, int64_t ldk, int64_t sk, const float* v, int64_t ldv,
                             int64_t sv, float* out, int64_t ldo, int64_t so, int64_t Z, int32_t H, int32_t Lq, int32_t Lk, int32_t hd,
                             float scale, psam_stream_t stream);
/* Test / A-B hook: 0 = psam_attention_small always runs one wave per query; 1 (default) = few queries against >= 128 keys run one workgroup per query,
 * the keys split over its four waves. */
void psam_attention_small_force_split(int32_t on);

/* ---------------------------------------------------------------- encodings, token assembly, upsampling */

/* y[r,0:128] = GELU(W[128,3] @ centers[r] + bias): first layer of PointCloudEncoder.pos_embed, pc_encoder.py:102-104,130. */
int32_t psam_pos_l1(const float* centers, const float* W, const float* bias, float* y, int64_t rows, psam_stream_t stream);

/* [sin, cos](2*pi * x @ gauss[3,F]) (+ point_embeddings[label]); row r -> out + (r / rows_per_batch)*batch_stride
 * + (r % rows_per_batch)*2F.  *flag |= 1 if a coordinate leaves [-1-1e-6, 1+1e-6] (the reference raises ValueError).
 * Replaces PositionEmbeddingRandom.forward and PointEncoder.forward: pc_sam/model/prompt_encoder.py:27-48,63-77. */
int32_t psam_fourier_pe(const float* coords, const float* gauss, int32_t F, const int64_t* labels, const float* emb0, const float* emb1,
                        float* out, int64_t rows, int32_t rows_per_batch, int64_t batch_stride, int32_t* flag, psam_stream_t stream);

/* out[z,r,:] = a[z/rep,r,:] + (b ? b[z*sb + r*ldb + :] : 0).  Replaces repeat_interleave + adds:
 * pc_sam/model/mask_decoder.py:136-139, pc_sam/model/transformer.py:153-170, prompt_encoder.py:119-122. */
int32_t psam_add_bcast(const float* a, int64_t sa, int32_t rep, const float* b, int64_t sb, int64_t ldb, float* out, int64_t so, int64_t Z,
                       int64_t R, int32_t C, psam_stream_t stream);

/* out[z,n,:] = sum_k w3[b,n,k] * src[z, idx3[b,n,k], :], b = z/rep.  Replaces interpolate_features:
 * pc_sam/model/common.py:258-274 (mask_decoder.py:163). */
int32_t psam_interp3(const float* src, const int64_t* idx3, const float* w3, float* out, int32_t rep, int64_t Z, int32_t N, int32_t G, int32_t C,
                     psam_stream_t stream);
/* scale_out [Z*N] != NULL (C == 256): out receives the g8-packed rows (A operand of psam_gemm_f16x3p) and scale_out their row scales.
 * ln_gamma / ln_beta != NULL (C == 256): LayerNorm(ln_eps) and the activation `act` (none / GELU / ReLU) are applied to every interpolated
 * row -- `interp -> Linear -> LayerNorm -> GELU` of mask_decoder.py:53-59,163 evaluated as `Linear (on the G patch rows) -> interp + LayerNorm
 * + GELU`: the interpolation is an affine combination (weights sum to 1), so it commutes with the Linear layer. */
int32_t psam_interp3_ex(const float* src, const int64_t* idx3, const float* w3, float* out, int32_t rep, int64_t Z, int32_t N, int32_t G, int32_t C,
                        float* scale_out, const float* ln_gamma, const float* ln_beta, float ln_eps, int32_t act, psam_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* POINTSAM_HIP_H */
