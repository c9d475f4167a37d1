// Probe: which physical CUs (XCC, SE, CU) does a stream created with hipExtStreamCreateWithCUMask use for a given mask?
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void probe_kernel(unsigned* out, int spin) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    const long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}
extern "C" __attribute__((visibility("default"))) int cumask_stream_create(void** stream, const uint32_t* mask, int words) {
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    *stream = (void*)s;
    return (int)e;
}
extern "C" __attribute__((visibility("default"))) int cumask_probe(void* stream, unsigned* out, int blocks, int spin) {
    hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, out, spin);
    return (int)hipGetLastError();
}
