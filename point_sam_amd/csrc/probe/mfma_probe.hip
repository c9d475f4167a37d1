// Measurement aid, NOT part of libpointsam_hip.so or its ABI: the matrix pipe's sustained rate on this box, for bench.py's
// `roofline.measured_mfma_ceiling_tflops`.  Back-to-back v_mfma_f32_32x32x16_f16 on register-resident operands that change from instruction to
// instruction (six A and six B fragments per lane in rotating pairs, so the multiplier inputs toggle as in a real K loop), four accumulators per wave,
// four waves per SIMD, no memory traffic inside the loop.  What it reaches is what clock and power management leave of the nominal 2.5 PFLOP/s for
// THESE operand values (DESIGN.md "power"): the yardstick the GEMM's executed product rate is compared with.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_probe_kernel(const i32x4* __restrict__ frags, float* __restrict__ out, int iters) {
    i32x4 a[6], b[6];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 6; ++k) { a[k] = frags[k * 64 + lane]; b[k] = frags[(6 + k) * 64 + lane]; }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(u + i) % 6]), __builtin_bit_cast(f16x8, b[(u + 2 * i + 1) % 6]), acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// frags: 12 x 64 x 4 dwords (fp16 pairs); out: grid * 256 floats.  One launch executes grid * 4 waves * iters * 24 MFMAs of 2 * 32*32*16 flops.
extern "C" __attribute__((visibility("default"))) int psam_probe_mfma_f16(const void* frags, float* out, int grid, int iters, hipStream_t st) {
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(grid), dim3(256), 0, st, (const i32x4*)frags, out, iters);
    return (int)hipGetLastError();
}
