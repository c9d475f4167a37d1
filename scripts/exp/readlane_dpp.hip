// Reproducer attempt for the round-2 observation (DESIGN.md 4.1 / common.h): a wave reduction that combines the 16-lane rows by v_readlane of
// DPP results gave run-to-run different LayerNorm statistics in a multi-stream test, the DPP + ds_bpermute form and the plain butterfly did not.
// Here: a LayerNorm-statistics kernel in three forms (readlane / bpermute / butterfly), run REP times on the same input while a second stream
// keeps every CU busy with MFMA + LDS work; each form's outputs must be bitwise equal to its own first run.
//   hipcc --offload-arch=gfx950 -O3 scripts/exp/readlane_dpp.hip -o scripts/exp/readlane_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v);
    return v;
}
template <int FORM> __device__ __forceinline__ float wave_sum(float v) {
    if (FORM == 2) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
    v = row16_sum(v);
    if (FORM == 1) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }
    const int b = __builtin_bit_cast(int, v);      // FORM 0: SGPR read-back of the four rows' results
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

template <int FORM>
__global__ __launch_bounds__(256) void ln_stats(const float* __restrict__ x, int rows, int cols, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (int64_t)row * cols);
    f32x4 v[4];
    for (int i = 0; i < 4; ++i) v[i] = xr[i * 64 + lane];
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum<FORM>(s) * (1.0f / cols);
    float q = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum<FORM>(q) * (1.0f / cols) + 1e-6f);
    if (lane == 0) { out[row * 2] = mean; out[row * 2 + 1] = rstd; }
}

__global__ __launch_bounds__(256) void busy(float* sink, int iters) {      // MFMA + LDS traffic on every CU
    __shared__ float lds[4096];
    f32x16 acc = {0};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        lds[(threadIdx.x * 17 + it) & 4095] = acc[it & 15];
        __syncthreads();
        a[it & 7] = (_Float16)lds[(threadIdx.x * 29 + it) & 4095];
    }
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

template <int FORM> static int run(const char* name, const float* dx, int rows, int cols, float* dout, float* dsink, hipStream_t s0, hipStream_t s1) {
    const int REP = 300;
    std::vector<float> first(rows * 2), cur(rows * 2);
    int bad_runs = 0; long bad_vals = 0;
    for (int rep = 0; rep < REP; ++rep) {
        hipLaunchKernelGGL(busy, dim3(512), dim3(256), 0, s1, dsink, 3000);
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(ln_stats<FORM>, dim3(rows / 4), dim3(256), 0, s0, dx, rows, cols, dout);
        CK(hipStreamSynchronize(s0));
        CK(hipMemcpy(cur.data(), dout, rows * 2 * 4, hipMemcpyDeviceToHost));
        if (rep == 0) first = cur;
        else if (memcmp(first.data(), cur.data(), rows * 2 * 4)) { ++bad_runs; for (int i = 0; i < rows * 2; ++i) bad_vals += memcmp(&first[i], &cur[i], 4) != 0; }
    }
    CK(hipDeviceSynchronize());
    printf("%-44s %d of %d runs differ from the first (%ld values)\n", name, bad_runs, REP - 1, bad_vals); fflush(stdout);
    return 0;
}

int main() {
    const int rows = 4096, cols = 1024;
    std::vector<float> hx((size_t)rows * cols);
    unsigned s = 1;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f * 4.f - 2.f; }
    float *dx, *dout, *dsink;
    CK(hipMalloc(&dx, hx.size() * 4)); CK(hipMalloc(&dout, rows * 2 * 4)); CK(hipMalloc(&dsink, 64));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    if (run<0>("DPP rows + v_readlane of lanes 0/16/32/48", dx, rows, cols, dout, dsink, s0, s1)) return 1;
    if (run<1>("DPP rows + ds_bpermute (xor 16, xor 32)", dx, rows, cols, dout, dsink, s0, s1)) return 1;
    if (run<2>("six-step ds_bpermute butterfly", dx, rows, cols, dout, dsink, s0, s1)) return 1;
    return 0;
}
