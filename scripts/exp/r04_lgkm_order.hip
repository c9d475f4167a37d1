// Round 4, VERDICT item 2: does a COUNTED s_waitcnt lgkmcnt(1) cover ds_bpermute / ds_read results when a ds_write was issued behind them?
// The round-3 failure (gemm_epilogue.h pass loop at 1179a69, ISA in profiles/r04/r04_hazard_isa.txt) consumed three ds_bpermute results and a
// ds_read_b128 after `ds_write_b128 ; s_waitcnt lgkmcnt(1)` -- correct only if DS operations retire in issue order.  Victim: that exact
// instruction mix in a loop, every result checked against its closed form.  Aggressor (other stream, co-resident on every CU): LDS-DMA
// (buffer_load ... lds) + ds_read traffic, the data path of the GEMM main loop that shared the CU when the failure was seen.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lgkm scripts/exp/r04_lgkm_order.hip && /tmp/lgkm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int WAITN, bool ALIAS = false>      // lgkmcnt value before the results are consumed: 1 = the failing form, 0 = drained; ALIAS: address registers as in the failing ISA
__global__ __launch_bounds__(256) void victim(unsigned* bad, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned char* lw = smem + wave * 8704;                          // 32 rows x 272 B, as the epilogue's stripe
    for (int r = 0; r < 32; ++r) if (lane < 17) *reinterpret_cast<f4*>(lw + r * 272 + lane * 16) = f4{(float)(r * 100 + lane), 1.f, 2.f, 3.f};
    __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();      // lgkmcnt(0)
    unsigned nbad = 0;
    const int rl0 = lane >> 4, c4 = lane & 15;                        // 4 rows per pass, 16 float4 per row (TN = 2 mapping)
    for (int it = 0; it < iters; ++it) {
        const int q = it & 7, rl = q * 4 + rl0;
        const unsigned s0 = lane * 3 + it, s1 = lane * 5 + 2 * it, s2 = lane * 7 + 3 * it;      // row scalars held lane -> row
        const unsigned baddr = (unsigned)(rl & 63) * 4, raddr = (unsigned)(size_t)(lw - smem) + rl * 272 + c4 * 16, waddr = (unsigned)(size_t)(lw - smem) + ((rl + 16) & 31) * 272 + 16 * 16;
        unsigned r0, r1, r2; f4 v; const f4 park = f4{(float)it, 0.f, 0.f, 0.f};
        if (ALIAS) {
            // the failing loop's register pattern: the address register is written by a VALU instruction right before the DS operation that uses it AND
            // is (part of) that operation's destination (`ds_bpermute_b32 v46, v46, ..`, `ds_read_b128 v[48:51], v48` in the ISA of 1179a69)
            float x0, x1, x2, x3;
            asm volatile("v_lshlrev_b32 v110, 0, %7\n\tds_bpermute_b32 %0, v110, %8\n\tds_bpermute_b32 %1, v110, %9\n\tds_bpermute_b32 v110, v110, %10\n\t"
                         "v_add_u32 v112, 0, %11\n\tds_read_b128 v[112:115], v112\n\tds_write_b128 %12, %13\n\ts_waitcnt lgkmcnt(%14)\n\t"
                         "v_mov_b32 %2, v110\n\tv_mov_b32 %3, v112\n\tv_mov_b32 %4, v113\n\tv_mov_b32 %5, v114\n\tv_mov_b32 %6, v115"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                         : "v"(baddr), "v"(s0), "v"(s1), "v"(s2), "v"(raddr), "v"(waddr), "v"(park), "n"(WAITN) : "memory", "v110", "v112", "v113", "v114", "v115");
            v = f4{x0, x1, x2, x3};
        } else
        asm volatile("ds_bpermute_b32 %0, %4, %5\n\tds_bpermute_b32 %1, %4, %6\n\tds_bpermute_b32 %2, %4, %7\n\tds_read_b128 %3, %8\n\t"
                     "ds_write_b128 %9, %10\n\ts_waitcnt lgkmcnt(%11)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(v) : "v"(baddr), "v"(s0), "v"(s1), "v"(s2), "v"(raddr), "v"(waddr), "v"(park), "n"(WAITN) : "memory");
        const unsigned e0 = (unsigned)rl * 3 + it, e1 = (unsigned)rl * 5 + 2 * it, e2 = (unsigned)rl * 7 + 3 * it;
        const bool ok = r0 == e0 && r1 == e1 && r2 == e2 && v[0] == (float)(rl * 100 + c4) && v[1] == 1.f && v[2] == 2.f && v[3] == 3.f;
        nbad += !ok;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (nbad) atomicAdd(bad + (lane >> 4), nbad);                     // per 16-lane quarter of the wave
}

__global__ __launch_bounds__(256) void aggressor(const f4* src, float* sink, int iters) {      // 64 KiB of LDS: one per CU beside the victim
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (wave * 4 + i) * 4096 + (it & 3) * 1024), 16,
                                                     (int)(((blockIdx.x * 64 + it * 7 + i) & 4095) * 1024 + lane * 16), 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += *reinterpret_cast<const f4*>(smem + ((wave * 8 + i) * 2048 + lane * 16 + it * 64) % 65536);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

int main() {
    unsigned* bad; f4* src; float* sink;
    CK(hipMalloc(&bad, 16)); CK(hipMalloc(&src, 4096 * 1024 + 4096)); CK(hipMalloc(&sink, 4)); CK(hipMemset(src, 0, 4096 * 1024));
    hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
    CK(hipFuncSetAttribute((const void*)aggressor, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int with_aggr = 0; with_aggr < 2; ++with_aggr)
        for (int form = 0; form < 4; ++form) {
            unsigned long long tot[4] = {0, 0, 0, 0};
            for (int rep = 0; rep < 20; ++rep) {
                CK(hipMemsetAsync(bad, 0, 16, s1)); CK(hipStreamSynchronize(s1));
                if (with_aggr) hipLaunchKernelGGL(aggressor, dim3(256), dim3(256), 65536, s2, src, sink, 6000);
                if (form == 0) hipLaunchKernelGGL(victim<1>, dim3(512), dim3(256), 4 * 8704, s1, bad, 20000);
                else if (form == 1) hipLaunchKernelGGL(victim<0>, dim3(512), dim3(256), 4 * 8704, s1, bad, 20000);
                else if (form == 2) hipLaunchKernelGGL((victim<1, true>), dim3(512), dim3(256), 4 * 8704, s1, bad, 20000);
                else hipLaunchKernelGGL((victim<0, true>), dim3(512), dim3(256), 4 * 8704, s1, bad, 20000);
                CK(hipDeviceSynchronize());
                unsigned h[4]; CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
                for (int k = 0; k < 4; ++k) tot[k] += h[k];
            }
            printf("aggressor %s, %s registers, lgkmcnt(%d) before use: wrong results per lane quarter [0-15 16-31 32-47 48-63] = %llu %llu %llu %llu (of %.1e checks each)\n",
                   with_aggr ? "ON " : "off", form >= 2 ? "ALIASED " : "distinct", (form & 1) == 0 ? 1 : 0, tot[0], tot[1], tot[2], tot[3], 20.0 * 512 * 4 * 16 * 20000);
        }
    return 0;
}
