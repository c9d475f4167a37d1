// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds u16 value = its own element index; lane l passes byte address a(l); prints what each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void probe(const int* addr, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    unsigned a = (unsigned)(uintptr_t)lds + addr[threadIdx.x];
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main(int argc, char** argv) {
    int mode = argc > 1 ? atoi(argv[1]) : 0;
    int h_addr[64];
    for (int l = 0; l < 64; ++l) {
        if (mode == 0) h_addr[l] = l * 8;                 // consecutive 8-byte slots
        else if (mode == 1) h_addr[l] = l * 128;          // one 128-byte row per lane
        else h_addr[l] = (l & 15) * 128 + (l >> 4) * 8;   // 16 rows x 4 column groups
    }
    int* d_addr; unsigned short* d_out; unsigned short h_out[256];
    hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("mode %d (lane: byte address -> 4 received element indices)\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d addr %5d (elem %4d): %4d %4d %4d %4d\n", l, h_addr[l], h_addr[l] / 2, h_out[l * 4], h_out[l * 4 + 1], h_out[l * 4 + 2], h_out[l * 4 + 3]);
    return 0;
}
