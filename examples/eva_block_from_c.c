/* A non-Python caller: one EVA02 transformer block through the C ABI (include/pointsam_hip.h), plain C99 + the HIP runtime.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/eva_block_from_c.c \
 *       -Lpoint_sam_amd/csrc -lpointsam_hip -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/point_sam_amd/csrc -Wl,-rpath,/opt/rocm/lib -o examples/eva_block_from_c
 *
 * Seeded random weights of a small block (dim 256, 4 heads of 64, SwiGLU hidden 682), x [512 tokens] -> block(x); prints a checksum twice
 * (the second run must repeat the first bit for bit).  tests/test_gpu_c_example.py builds and runs it on the GPU box. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "pointsam_hip.h"

static unsigned g_seed = 12345u;
static float rnd(void) { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) & 0xffff) / 32768.0f - 1.0f; }

static float* dev_random(size_t n, float scale, float offset) {
    float* h = (float*)malloc(n * sizeof(float));
    float* d = NULL;
    size_t i;
    for (i = 0; i < n; ++i) h[i] = offset + scale * rnd();
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess || hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { fprintf(stderr, "hip alloc/copy failed\n"); exit(2); }
    free(h);
    return d;
}

int main(void) {
    const int D = 256, HEADS = 4, HID = 682, B = 2, L = 256, M = B * L;
    const float ws_ = 1.0f / 16.0f;
    psam_eva_block_weights_t w;
    psam_eva_block_plan_t plan;
    void *prepared = NULL, *ws = NULL;
    float* x;
    float* hx = (float*)malloc((size_t)M * D * sizeof(float));
    size_t pb, wb;
    int run, rc;
    double first = 0.0;
    w.dim = D; w.heads = HEADS; w.hidden = HID; w.eps = 1e-6f;
    w.norm1_w = dev_random(D, 0.1f, 1.0f); w.norm1_b = dev_random(D, 0.1f, 0.0f);
    w.q_w = dev_random((size_t)D * D, ws_, 0.f); w.q_b = dev_random(D, 0.02f, 0.f);
    w.k_w = dev_random((size_t)D * D, ws_, 0.f);
    w.v_w = dev_random((size_t)D * D, ws_, 0.f); w.v_b = dev_random(D, 0.02f, 0.f);
    w.proj_w = dev_random((size_t)D * D, ws_, 0.f); w.proj_b = dev_random(D, 0.02f, 0.f);
    w.norm2_w = dev_random(D, 0.1f, 1.0f); w.norm2_b = dev_random(D, 0.1f, 0.0f);
    w.fc1_g_w = dev_random((size_t)HID * D, ws_, 0.f); w.fc1_g_b = dev_random(HID, 0.02f, 0.f);
    w.fc1_x_w = dev_random((size_t)HID * D, ws_, 0.f); w.fc1_x_b = dev_random(HID, 0.02f, 0.f);
    w.mlp_norm_w = dev_random(HID, 0.1f, 1.0f); w.mlp_norm_b = dev_random(HID, 0.1f, 0.0f);
    w.fc2_w = dev_random((size_t)D * HID, 1.0f / 26.0f, 0.f); w.fc2_b = dev_random(D, 0.02f, 0.f);
    pb = psam_eva_block_prepared_bytes(D, HID);
    wb = psam_eva_block_ws_bytes(M, D, HID);
    if (hipMalloc(&prepared, pb) != hipSuccess || hipMalloc(&ws, wb) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 2; }
    rc = psam_eva_block_prepare(&w, &plan, prepared, pb, NULL);
    if (rc) { fprintf(stderr, "prepare: rc=%d %s\n", rc, psam_last_error_string()); return 1; }
    for (run = 0; run < 2; ++run) {
        double sum = 0.0, asum = 0.0;
        int i, bad = 0;
        g_seed = 777u;
        x = dev_random((size_t)M * D, 1.0f, 0.f);
        rc = psam_eva_block(&plan, prepared, x, B, L, ws, wb, NULL);
        if (rc || hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "psam_eva_block: rc=%d %s\n", rc, psam_last_error_string()); return 1; }
        if (hipMemcpy(hx, x, (size_t)M * D * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 2;
        for (i = 0; i < M * D; ++i) { sum += hx[i]; asum += fabs(hx[i]); bad += !isfinite(hx[i]); }
        printf("run %d: sum %.9e  sum|x| %.9e  non-finite %d\n", run, sum, asum, bad);
        if (bad) return 1;
        if (run == 0) first = sum; else if (sum != first) { fprintf(stderr, "not repeatable\n"); return 1; }
        hipFree(x);
    }
    printf("psam_version %d: ok\n", psam_version());
    return 0;
}
