// Hand-over latency between workgroups (what bounds the cooperative FPS): W workgroups, every iteration each stores a tagged 64-bit key
// into its slot and wave 0 polls all W slots.  Variants: placement (consecutive ids = spread over the XCDs | ids 8 apart = one XCD, if the
// dispatcher deals workgroups round-robin), coherence bits of the accesses (sc1 = agent scope | sc0 = bypass the CU's L1 only: coherent in
// the XCD's L2).  Also a dependent scalar-load chain (the P[last] fetch).   hipcc --offload-arch=gfx950 -O3 fabric_probe.hip -o fabric_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE> __device__ __forceinline__ void st64(unsigned long long* p, unsigned long long v) {
    if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int MODE> __device__ __forceinline__ unsigned long long ld64(const unsigned long long* p) {
    if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long v;
    if (MODE == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int MODE>
__global__ __launch_bounds__(1024) void handover(unsigned long long* slots, int W, int stride, int iters, unsigned* xcc, long long* cycles) {
    if (blockIdx.x % stride != 0) return;
    const int w = blockIdx.x / stride;
    if (w >= W) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ int s_last[2];
    if (threadIdx.x == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[w] = id; }
    const long long t0 = wall_clock64();
    int last = 0;
    for (int j = 1; j <= iters; ++j) {
        const int slot = j & 1;
        __syncthreads();
        if (wave == 0) {
            const unsigned tag = (unsigned)j & 0xFFFu;
            if (lane == 0) st64<MODE>(slots + slot * 64 + w, ((unsigned long long)(w * 7 + last) << 32) | (tag << 20) | (unsigned)w);
            unsigned long long k;
            int spins = 0;
            for (;;) {
                k = lane < W ? ld64<MODE>(slots + slot * 64 + lane) : ((unsigned long long)tag << 20);
                if (__all((((unsigned)k >> 20) & 0xFFFu) == tag)) break;
                if (++spins > 200000) { if (lane == 0) cycles[1] = j; break; }      // never became visible (not coherent at this placement)
                __builtin_amdgcn_s_sleep(1);
            }
            if (spins > 200000) { s_last[0] = s_last[1] = -2; }
            for (int o = 32; o > 0; o >>= 1) { const unsigned long long ok = __shfl_xor(k, o, 64); k = ok > k ? ok : k; }
            if (lane == 0) s_last[slot] = (int)(k & 0xFFFFF);
        }
        __syncthreads();
        last = s_last[slot];
        if (last == -2) break;
    }
    if (threadIdx.x == 0 && w == 0) cycles[0] = wall_clock64() - t0;
    if (last == -1) slots[1000] = 1;
}

__global__ void chase(const int* next, int steps, int* out, long long* cycles) {
    int i = 0;
    const long long t0 = wall_clock64();
    for (int s = 0; s < steps; ++s) i = __builtin_amdgcn_readfirstlane(next[i]);     // uniform address -> scalar load
    cycles[0] = wall_clock64() - t0;
    out[0] = i;
}

template <int MODE> static int run(const char* name, int W, int stride, unsigned long long* slots, unsigned* xcc, long long* cyc) {
    const int iters = 2000;
    CK(hipMemset(slots, 0xff, 4096 * 8));
    CK(hipMemset(cyc, 0, 64));
    hipLaunchKernelGGL(handover<MODE>, dim3(W * stride), dim3(1024), 0, 0, slots, W, stride, iters, xcc, cyc);
    CK(hipDeviceSynchronize());
    long long c2[2]; std::vector<unsigned> x(W);
    CK(hipMemcpy(c2, cyc, 16, hipMemcpyDeviceToHost));
    const long long c = c2[0];
    if (c2[1]) { printf("%-28s W=%2d stride=%d: NOT COHERENT (gave up at iteration %lld)\n", name, W, stride, c2[1]); fflush(stdout); return 0; }
    CK(hipMemcpy(x.data(), xcc, W * 4, hipMemcpyDeviceToHost));
    printf("%-28s W=%2d stride=%d: %.3f us/iter   xcc:", name, W, stride, c / 100.0 / iters);     // wall_clock64: 100 MHz
    for (int i = 0; i < W; ++i) printf(" %u", x[i] & 15);
    printf("\n"); fflush(stdout);
    return 0;
}

int main() {
    unsigned long long* slots; unsigned* xcc; long long* cyc;
    CK(hipMalloc(&slots, 4096 * 8)); CK(hipMalloc(&xcc, 4096)); CK(hipMalloc(&cyc, 64));
    for (int W : {8, 16, 32}) {
        for (int stride : {1, 8}) {
            if (W * stride > 256) continue;
            if (run<0>("agent-scope atomics (sc1)", W, stride, slots, xcc, cyc)) return 1;
            // sc0 alone (bypass the CU's L1 only) never becomes visible to the pollers, not even on one XCD: measured, see profiles/r03/r03_fabric_probe.txt
            if (W == 8 && getenv("PROBE_SC0") && run<1>("sc0 (L2 of the XCD)", W, stride, slots, xcc, cyc)) return 1;
        }
    }
    // scalar pointer chase over 1.5 MiB (the xyz of a 131072-point cloud), random permutation
    const int n = 393216;
    std::vector<int> nxt(n);
    { std::vector<int> perm(n); for (int i = 0; i < n; ++i) perm[i] = i;
      unsigned s = 12345; for (int i = n - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; int j = s % (i + 1); std::swap(perm[i], perm[j]); }
      for (int i = 0; i < n; ++i) nxt[perm[i]] = perm[(i + 1) % n]; }
    int* dn; int* out;
    CK(hipMalloc(&dn, n * 4)); CK(hipMalloc(&out, 4));
    CK(hipMemcpy(dn, nxt.data(), n * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, dn, 4000, out, cyc);
        CK(hipDeviceSynchronize());
        long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        printf("scalar load chain (1.5 MiB, random): %.3f us/load\n", c / 100.0 / 4000);
    }
    return 0;
}
