// Matrix-pipe power probe (round 5): back-to-back MFMAs of one type on register-resident operands that CHANGE from instruction to instruction (six A and
// six B fragments of random data per lane, taken in rotating pairs: the multiplier inputs toggle as in a real K loop), four accumulators per wave, four waves
// per SIMD.  Held for seconds per type while rocm-smi samples package power and clock (scripts/exp/r05_mfma_power.py): which operand type buys how many
// products per second under the 1400 W cap.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// T: 0 = f16 32x32x16, 1 = bf16 32x32x16, 2 = i8 32x32x32, 3 = fp8 (e4m3) 32x32x16
template <int T>
__global__ __launch_bounds__(256) void probe(const i32x4* __restrict__ frags, float* out, int iters) {
    i32x4 a[6], b[6];
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 6; ++k) { a[k] = frags[(k * 64 + lane)]; b[k] = frags[((6 + k) * 64 + lane)]; }
    f32x16 acc[4]; i32x16 iacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; iacc[i][r] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const i32x4 x = a[(u + i) % 6], y = b[(u + 2 * i + 1) % 6];
                if constexpr (T == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), acc[i], 0, 0, 0);
                else if constexpr (T == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc[i], 0, 0, 0);
                else if constexpr (T == 2) iacc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, iacc[i], 0, 0, 0);
                else {
                    const long xl = ((long)x[1] << 32) | (unsigned)x[0], yl = ((long)y[1] << 32) | (unsigned)y[0];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(xl, yl, acc[i], 0, 0, 0);
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r] + (float)iacc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" void run(int type, const void* frags, float* out, int grid, int iters, hipStream_t st) {
    const i32x4* f = (const i32x4*)frags;
    if (type == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(256), 0, st, f, out, iters);
    if (type == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(256), 0, st, f, out, iters);
    if (type == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, st, f, out, iters);
    if (type == 3) hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(256), 0, st, f, out, iters);
}
