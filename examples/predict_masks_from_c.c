/* The WHOLE hot path -- evaluation/inference.py:41-68's flow (load weights, set the cloud, one click, predict masks) -- from plain C99 through the C
 * ABI (include/pointsam_hip.h), no Python anywhere: tokenizer (psam_fps, psam_knn, psam_three_nn), patch embedding (psam_patch_encoder), token
 * assembly (psam_linear, psam_pos_l1), transformer (psam_eva_block per layer), prompt / positional encodings (psam_fourier_pe, psam_add_bcast),
 * two-way decoder (psam_twoway_decoder), hyper-networks and IoU head (psam_mlp3), upscaling + mask products (psam_upscale_masks).
 *
 *   python examples/make_c_demo_blob.py /tmp/c_demo.blob        # state dict + one cloud + what the CPU oracle computes for it
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/predict_masks_from_c.c -Lpoint_sam_amd/csrc -lpointsam_hip \
 *       -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/point_sam_amd/csrc -Wl,-rpath,/opt/rocm/lib -o /tmp/predict_masks_from_c
 *   /tmp/predict_masks_from_c /tmp/c_demo.blob
 *
 * Checks: FPS indices identical to the oracle's, mask logits and IoU predictions within 1e-3.  tests/test_gpu_c_example.py runs all of it. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pointsam_hip.h"

#define MAXT 256
typedef struct { char name[160]; int dtype, ndim; unsigned dims[4]; size_t count; void* host; void* dev; } tensor_t;
static tensor_t g_t[MAXT];
static int g_nt = 0;

static void die(const char* what) { fprintf(stderr, "FAILED: %s (%s)\n", what, psam_last_error_string()); exit(1); }
#define CK(call) do { if ((call) != 0) die(#call); } while (0)
#define HCK(call) do { if ((call) != hipSuccess) { fprintf(stderr, "HIP FAILED: %s\n", #call); exit(2); } } while (0)

static void load_blob(const char* path) {
    FILE* f = fopen(path, "rb");
    char magic[8];
    unsigned n, i, d;
    if (!f || fread(magic, 1, 8, f) != 8 || memcmp(magic, "PSAMBLOB", 8) != 0 || fread(&n, 4, 1, f) != 1 || n > MAXT) { fprintf(stderr, "bad blob %s\n", path); exit(2); }
    for (i = 0; i < n; ++i) {
        tensor_t* t = &g_t[i];
        unsigned nl, dt, nd;
        size_t es;
        if (fread(&nl, 4, 1, f) != 1 || nl >= sizeof(t->name) || fread(t->name, 1, nl, f) != nl) { fprintf(stderr, "bad blob entry\n"); exit(2); }
        t->name[nl] = 0;
        if (fread(&dt, 4, 1, f) != 1 || fread(&nd, 4, 1, f) != 1 || nd > 4) { fprintf(stderr, "bad blob entry\n"); exit(2); }
        t->dtype = (int)dt; t->ndim = (int)nd; t->count = 1;
        for (d = 0; d < nd; ++d) { if (fread(&t->dims[d], 4, 1, f) != 1) exit(2); t->count *= t->dims[d]; }
        es = dt ? 8 : 4;
        t->host = malloc(t->count * es + 8);
        if (fread(t->host, es, t->count, f) != t->count) { fprintf(stderr, "short blob\n"); exit(2); }
        HCK(hipMalloc(&t->dev, t->count * es + 8));
        HCK(hipMemcpy(t->dev, t->host, t->count * es, hipMemcpyHostToDevice));
    }
    g_nt = (int)n;
    fclose(f);
}
static tensor_t* find(const char* name) {
    int i;
    for (i = 0; i < g_nt; ++i) if (strcmp(g_t[i].name, name) == 0) return &g_t[i];
    fprintf(stderr, "tensor %s not in the blob\n", name);
    exit(2);
}
static const float* W(const char* name) { return (const float*)find(name)->dev; }
static const float* Wf(const char* fmt, int i, const char* tail) { char b[200]; snprintf(b, sizeof(b), fmt, i); strncat(b, tail, sizeof(b) - strlen(b) - 1); return W(b); }
static void* dmalloc(size_t bytes) { void* p = NULL; HCK(hipMalloc(&p, bytes ? bytes : 16)); return p; }

static psam_attn_weights_t attn_w(const char* prefix) {
    psam_attn_weights_t a;
    char b[200];
#define AW(field, suffix) snprintf(b, sizeof(b), "%s.%s", prefix, suffix); a.field = W(b)
    AW(q_w, "q_proj.weight"); AW(q_b, "q_proj.bias"); AW(k_w, "k_proj.weight"); AW(k_b, "k_proj.bias");
    AW(v_w, "v_proj.weight"); AW(v_b, "v_proj.bias"); AW(o_w, "out_proj.weight"); AW(o_b, "out_proj.bias");
#undef AW
    return a;
}

int main(int argc, char** argv) {
    /* geometry of examples/make_c_demo_blob.py::demo_config */
    enum { B = 1, N = 4096, G = 256, K = 32, D = 256, HEADS = 4, HID = 682, DEPTH = 2, E = 256, POUT = 512, NMT = 4, PN = 1, T = 1 + NMT + PN, C = 3, Z = 1,
           DEC_DEPTH = 2, DEC_HEADS = 8, DEC_MLP = 2048, DEC_DOWN = 2 };
    const float LN_EPS = 1e-5f, VIT_EPS = 1e-6f;
    const float *xyz, *rgb, *prompt;
    const int64_t* labels;
    int64_t *fps_idx, *knn_idx, *idx3;
    float *centers, *emb, *x, *p1, *h, *pc_emb, *pc_pe, *tokens, *src, *hs, *hyper, *masks, *iou, *w3, *out_tokens, *hw[6], *iw[6];
    int32_t *flag, *counters;
    void* ws;
    size_t wsb, pb;
    int i, j;
    if (argc < 2) { fprintf(stderr, "usage: %s blob\n", argv[0]); return 2; }
    load_blob(argv[1]);
    xyz = W("in.xyz"); rgb = W("in.rgb"); prompt = W("in.prompt"); labels = (const int64_t*)find("in.labels")->dev;
    flag = (int32_t*)dmalloc(4); HCK(hipMemset(flag, 0, 4));

    /* ---- tokenizer: FPS -> centres, kNN groups, 3-NN interpolation weights (common.py:91-97, 238-255) */
    fps_idx = (int64_t*)dmalloc((size_t)B * G * 8); centers = (float*)dmalloc((size_t)B * G * 3 * 4); knn_idx = (int64_t*)dmalloc((size_t)B * G * K * 8);
    idx3 = (int64_t*)dmalloc((size_t)B * N * 3 * 8); w3 = (float*)dmalloc((size_t)B * N * 3 * 4);
    wsb = psam_fps_workspace_bytes(B, N, G); ws = dmalloc(wsb);
    CK(psam_fps(xyz, B, N, G, fps_idx, centers, ws, wsb, NULL));
    CK(psam_knn(centers, xyz, B, G, N, K, knn_idx, NULL));
    CK(psam_three_nn(xyz, centers, B, N, G, 1e-8f, idx3, w3, NULL));

    /* ---- patch embedding: mini-PointNet over the groups (common.py:477-506), then patch_proj + pos_embed (pc_encoder.py:118-137) */
    {
        psam_patch_encoder_weights_t pw;
        psam_patch_encoder_plan_t plan;
        void* prep;
        const char* P = "pc_encoder.patch_embed.patch_encoder";
        char b[200];
#define PW(field, suffix) snprintf(b, sizeof(b), "%s.%s", P, suffix); pw.field = W(b)
        PW(c10_w, "conv1.0.weight"); PW(c10_b, "conv1.0.bias"); PW(c11_w, "conv1.1.weight"); PW(c11_b, "conv1.1.bias"); PW(c13_w, "conv1.3.weight"); PW(c13_b, "conv1.3.bias");
        PW(c20_w, "conv2.0.weight"); PW(c20_b, "conv2.0.bias"); PW(c21_w, "conv2.1.weight"); PW(c21_b, "conv2.1.bias"); PW(c23_w, "conv2.3.weight"); PW(c23_b, "conv2.3.bias");
#undef PW
        pw.cin = 6; pw.h0 = 128; pw.h1 = 512; pw.cout = POUT; pw.eps = LN_EPS;
        pb = psam_patch_encoder_prepared_bytes(128, 512, POUT); prep = dmalloc(pb);
        CK(psam_patch_encoder_prepare(&pw, &plan, prep, pb, NULL));
        emb = (float*)dmalloc((size_t)B * G * POUT * 4);
        wsb = psam_patch_encoder_ws_bytes((int64_t)B * G * K, (int64_t)B * G, 128, 512); ws = dmalloc(wsb);
        CK(psam_patch_encoder(&plan, prep, xyz, rgb, centers, knn_idx, NULL, B, 1, N, G, K, 3, 0.f, emb, ws, wsb, NULL));
    }
    x = (float*)dmalloc((size_t)B * G * D * 4); p1 = (float*)dmalloc((size_t)B * G * 128 * 4);
    CK(psam_linear(emb, POUT, W("pc_encoder.patch_proj.weight"), POUT, W("pc_encoder.patch_proj.bias"), NULL, 0, x, D, B * G, D, POUT, PSAM_ACT_NONE, NULL));
    CK(psam_pos_l1(centers, W("pc_encoder.pos_embed.0.weight"), W("pc_encoder.pos_embed.0.bias"), p1, (int64_t)B * G, NULL));
    CK(psam_linear(p1, 128, W("pc_encoder.pos_embed.2.weight"), 128, W("pc_encoder.pos_embed.2.bias"), x, D, x, D, B * G, D, 128, PSAM_ACT_NONE, NULL));

    /* ---- transformer blocks (timm eva02 block; pc_encoder.py:138-139), final norm, out_proj, positional encoding of the centres */
    wsb = psam_eva_block_ws_bytes((int64_t)B * G, D, HID); ws = dmalloc(wsb);
    for (i = 0; i < DEPTH; ++i) {
        psam_eva_block_weights_t bw;
        psam_eva_block_plan_t plan;
        void* prep;
        const char* F = "pc_encoder.transformer.blocks.%d";
        bw.norm1_w = Wf(F, i, ".norm1.weight"); bw.norm1_b = Wf(F, i, ".norm1.bias");
        bw.q_w = Wf(F, i, ".attn.q_proj.weight"); bw.q_b = Wf(F, i, ".attn.q_proj.bias"); bw.k_w = Wf(F, i, ".attn.k_proj.weight");
        bw.v_w = Wf(F, i, ".attn.v_proj.weight"); bw.v_b = Wf(F, i, ".attn.v_proj.bias"); bw.proj_w = Wf(F, i, ".attn.proj.weight"); bw.proj_b = Wf(F, i, ".attn.proj.bias");
        bw.norm2_w = Wf(F, i, ".norm2.weight"); bw.norm2_b = Wf(F, i, ".norm2.bias");
        bw.fc1_g_w = Wf(F, i, ".mlp.fc1_g.weight"); bw.fc1_g_b = Wf(F, i, ".mlp.fc1_g.bias"); bw.fc1_x_w = Wf(F, i, ".mlp.fc1_x.weight"); bw.fc1_x_b = Wf(F, i, ".mlp.fc1_x.bias");
        bw.mlp_norm_w = Wf(F, i, ".mlp.norm.weight"); bw.mlp_norm_b = Wf(F, i, ".mlp.norm.bias"); bw.fc2_w = Wf(F, i, ".mlp.fc2.weight"); bw.fc2_b = Wf(F, i, ".mlp.fc2.bias");
        bw.dim = D; bw.heads = HEADS; bw.hidden = HID; bw.eps = VIT_EPS;
        pb = psam_eva_block_prepared_bytes(D, HID); prep = dmalloc(pb);
        CK(psam_eva_block_prepare(&bw, &plan, prep, pb, NULL));
        CK(psam_eva_block(&plan, prep, x, B, G, ws, wsb, NULL));
        HCK(hipDeviceSynchronize());      /* the plan lives on this stack frame: finish before it goes away */
    }
    h = (float*)dmalloc((size_t)B * G * D * 4); pc_emb = (float*)dmalloc((size_t)B * G * E * 4); pc_pe = (float*)dmalloc((size_t)B * G * E * 4);
    CK(psam_layernorm(x, D, NULL, 0, W("pc_encoder.transformer.fc_norm.weight"), W("pc_encoder.transformer.fc_norm.bias"), h, D, (int64_t)B * G, D, VIT_EPS, PSAM_ACT_NONE, NULL));
    CK(psam_linear(h, D, W("pc_encoder.out_proj.weight"), D, W("pc_encoder.out_proj.bias"), NULL, 0, pc_emb, E, B * G, E, D, PSAM_ACT_NONE, NULL));
    CK(psam_fourier_pe(centers, W("point_encoder.pe_layer.positional_encoding_gaussian_matrix"), E / 2, NULL, NULL, NULL, pc_pe, (int64_t)B * G, G, (int64_t)G * E, flag, NULL));

    /* ---- prompt tokens [iou token | mask tokens | click] and src = embeddings + no-mask embedding (mask_decoder.py:126-139, prompt_encoder.py:63-77,118-122) */
    out_tokens = (float*)dmalloc((size_t)(1 + NMT) * E * 4);
    HCK(hipMemcpy(out_tokens, W("mask_decoder.iou_token.weight"), (size_t)E * 4, hipMemcpyDeviceToDevice));
    HCK(hipMemcpy(out_tokens + E, W("mask_decoder.mask_tokens.weight"), (size_t)NMT * E * 4, hipMemcpyDeviceToDevice));
    tokens = (float*)dmalloc((size_t)Z * T * E * 4); src = (float*)dmalloc((size_t)Z * G * E * 4); hs = (float*)dmalloc((size_t)Z * T * E * 4);
    CK(psam_add_bcast(out_tokens, 0, Z, NULL, 0, 0, tokens, (int64_t)T * E, Z, 1 + NMT, E, NULL));
    CK(psam_fourier_pe(prompt, W("point_encoder.pe_layer.positional_encoding_gaussian_matrix"), E / 2, labels, W("point_encoder.point_embeddings.0.weight"),
                       W("point_encoder.point_embeddings.1.weight"), tokens + (1 + NMT) * E, (int64_t)Z * PN, PN, (int64_t)T * E, flag, NULL));
    CK(psam_add_bcast(pc_emb, (int64_t)G * E, Z / B, W("mask_encoder.no_mask_embed.weight"), 0, 0, src, (int64_t)G * E, Z, G, E, NULL));

    /* ---- two-way transformer (transformer.py:61-100) */
    {
        psam_twoway_weights_t tw;
        psam_twoway_layer_weights_t lay[DEC_DEPTH];
        static psam_twoway_plan_t plan;
        void* prep;
        char b[200];
        for (i = 0; i < DEC_DEPTH; ++i) {
            const char* F = "mask_decoder.transformer.layers.%d";
            snprintf(b, sizeof(b), "mask_decoder.transformer.layers.%d.self_attn", i); lay[i].self_attn = attn_w(b);
            snprintf(b, sizeof(b), "mask_decoder.transformer.layers.%d.cross_attn_token_to_image", i); lay[i].t2i = attn_w(b);
            snprintf(b, sizeof(b), "mask_decoder.transformer.layers.%d.cross_attn_image_to_token", i); lay[i].i2t = attn_w(b);
            lay[i].n1_w = Wf(F, i, ".norm1.weight"); lay[i].n1_b = Wf(F, i, ".norm1.bias"); lay[i].n2_w = Wf(F, i, ".norm2.weight"); lay[i].n2_b = Wf(F, i, ".norm2.bias");
            lay[i].n3_w = Wf(F, i, ".norm3.weight"); lay[i].n3_b = Wf(F, i, ".norm3.bias"); lay[i].n4_w = Wf(F, i, ".norm4.weight"); lay[i].n4_b = Wf(F, i, ".norm4.bias");
            lay[i].m1_w = Wf(F, i, ".mlp.lin1.weight"); lay[i].m1_b = Wf(F, i, ".mlp.lin1.bias"); lay[i].m2_w = Wf(F, i, ".mlp.lin2.weight"); lay[i].m2_b = Wf(F, i, ".mlp.lin2.bias");
        }
        tw.depth = DEC_DEPTH; tw.dim = E; tw.heads = DEC_HEADS; tw.mlp = DEC_MLP; tw.downsample = DEC_DOWN; tw.eps = LN_EPS; tw.layers = lay;
        tw.final_attn = attn_w("mask_decoder.transformer.final_attn_token_to_image");
        tw.nf_w = W("mask_decoder.transformer.norm_final_attn.weight"); tw.nf_b = W("mask_decoder.transformer.norm_final_attn.bias");
        pb = psam_twoway_decoder_prepared_bytes(DEC_DEPTH, E, DEC_MLP, DEC_DOWN); prep = dmalloc(pb);
        CK(psam_twoway_decoder_prepare(&tw, &plan, prep, pb, NULL));
        wsb = psam_twoway_decoder_ws_bytes(Z, T, G, E, DEC_MLP); ws = dmalloc(wsb);
        /* the caller owns the arrival-counter block of the fused Linear + LayerNorm launches: zeroed once, left zero by every launch */
        counters = (int32_t*)dmalloc(PSAM_COUNTER_BYTES); HCK(hipMemset(counters, 0, PSAM_COUNTER_BYTES));
        CK(psam_twoway_decoder(&plan, prep, tokens, src, pc_pe, Z / B, Z, T, G, hs, ws, wsb, counters, NULL));
    }

    /* ---- hyper-networks of mask tokens 1..3 (multimask output), upscaling + mask products, IoU head (mask_decoder.py:146-182) */
    for (j = 0; j < 3; ++j) {      /* stacked [M, out, in] / [M, out] copies of layers.{j} */
        const int rows = j < 2 ? E : E;
        hw[2 * j] = (float*)dmalloc((size_t)C * rows * E * 4); hw[2 * j + 1] = (float*)dmalloc((size_t)C * rows * 4);
        for (i = 0; i < C; ++i) {
            char b[200];
            snprintf(b, sizeof(b), "mask_decoder.output_hypernetworks_mlps.%d.layers.%d.weight", 1 + i, j);
            HCK(hipMemcpy(hw[2 * j] + (size_t)i * rows * E, W(b), (size_t)rows * E * 4, hipMemcpyDeviceToDevice));
            snprintf(b, sizeof(b), "mask_decoder.output_hypernetworks_mlps.%d.layers.%d.bias", 1 + i, j);
            HCK(hipMemcpy(hw[2 * j + 1] + (size_t)i * rows, W(b), (size_t)rows * 4, hipMemcpyDeviceToDevice));
        }
    }
    hyper = (float*)dmalloc((size_t)Z * C * E * 4); masks = (float*)dmalloc((size_t)Z * C * N * 4); iou = (float*)dmalloc((size_t)Z * NMT * 4);
    CK(psam_mlp3(hs + 2 * E, (int64_t)T * E, E, hw[0], hw[1], hw[2], hw[3], hw[4], hw[5], hyper, (int64_t)C * E, E, Z, C, E, E, E, NULL));
    {
        psam_upscale_weights_t uw;
        psam_upscale_plan_t plan;
        void* prep;
        uw.u0_w = W("mask_decoder.output_upscaling.0.weight"); uw.u0_b = W("mask_decoder.output_upscaling.0.bias");
        uw.u1_w = W("mask_decoder.output_upscaling.1.weight"); uw.u1_b = W("mask_decoder.output_upscaling.1.bias");
        uw.u3_w = W("mask_decoder.output_upscaling.3.weight"); uw.u3_b = W("mask_decoder.output_upscaling.3.bias");
        uw.dim = E; uw.eps = LN_EPS;
        pb = psam_upscale_masks_prepared_bytes(E); prep = dmalloc(pb);
        CK(psam_upscale_masks_prepare(&uw, &plan, prep, pb, NULL));
        wsb = psam_upscale_masks_ws_bytes(Z, N, G, C, E); ws = dmalloc(wsb);
        CK(psam_upscale_masks(&plan, prep, src, idx3, w3, hyper, Z / B, Z, N, G, C, masks, ws, wsb, NULL));
        HCK(hipDeviceSynchronize());
    }
    for (j = 0; j < 3; ++j) { char b[200]; snprintf(b, sizeof(b), "mask_decoder.iou_prediction_head.layers.%d.weight", j); iw[2 * j] = (float*)W(b);
                              snprintf(b, sizeof(b), "mask_decoder.iou_prediction_head.layers.%d.bias", j); iw[2 * j + 1] = (float*)W(b); }
    CK(psam_mlp3(hs, (int64_t)T * E, 0, iw[0], iw[1], iw[2], iw[3], iw[4], iw[5], iou, NMT, 0, Z, 1, E, E, NMT, NULL));
    HCK(hipDeviceSynchronize());

    /* ---- checks against what the oracle computed for the same blob */
    {
        tensor_t *wf = find("want.fps_idx"), *wm = find("want.masks"), *wi = find("want.iou");
        int64_t* hf = (int64_t*)malloc((size_t)B * G * 8);
        float* hm = (float*)malloc((size_t)Z * C * N * 4);
        float hi[NMT];
        int32_t hflag = 0;
        size_t k, nbad = 0;
        double em = 0.0, ei = 0.0, mmax = 0.0;
        HCK(hipMemcpy(hf, fps_idx, (size_t)B * G * 8, hipMemcpyDeviceToHost));
        HCK(hipMemcpy(hm, masks, (size_t)Z * C * N * 4, hipMemcpyDeviceToHost));
        HCK(hipMemcpy(hi, iou, sizeof(hi), hipMemcpyDeviceToHost));
        HCK(hipMemcpy(&hflag, flag, 4, hipMemcpyDeviceToHost));
        for (k = 0; k < (size_t)B * G; ++k) nbad += hf[k] != ((const int64_t*)wf->host)[k];
        for (k = 0; k < (size_t)Z * C * N; ++k) { const double d = fabs((double)hm[k] - ((const float*)wm->host)[k]); if (d > em || d != d) em = d; if (fabs(hm[k]) > mmax) mmax = fabs(hm[k]); }
        for (k = 0; k < (size_t)C; ++k) { const double d = fabs((double)hi[1 + k] - ((const float*)wi->host)[k]); if (d > ei || d != d) ei = d; }
        printf("fps indices differing from the oracle: %zu of %d\nmask logits: max |err| %.3e (|logit| max %.3f)\niou: max |err| %.3e\ncoordinate range flag %d\n", nbad, B * G, em, mmax, ei,
               hflag);
        if (nbad != 0 || !(em < 1e-3) || !(ei < 1e-3) || hflag != 0) { printf("MISMATCH\n"); return 1; }
        printf("ok\n");
    }
    return 0;
}
