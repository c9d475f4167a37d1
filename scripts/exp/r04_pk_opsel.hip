// Reproducer for the round-3 "cross-lane hazard" (VERDICT r03 item 2), bisected at ISA level in round 4 (profiles/r04/r04_hazard.txt): on gfx950 a packed-FP32
// VOP3P instruction whose op_sel makes the LOW result read the HIGH register of a source pair -- e.g. `v_pk_mul_f32 v[52:53], v[74:75], v[42:43] op_sel:[0,1]` --
// returns results computed from a wrong operand in lanes 48-63 while a wave of the LDS-DMA GEMM kernel (another stream) is resident on the SIMD.  Alone, or
// beside pure MFMA waves, it never fails.  Each form below runs the bare instruction on fixed registers and compares with v_mul_f32 / v_add_f32 / v_fma_f32.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o libr04_pk_opsel.so r04_pk_opsel.hip ; python r04_pk_opsel_beside_gemm.py   (GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#define SET "v_mov_b32 v74, %2\n\tv_mov_b32 v75, %3\n\tv_mov_b32 v42, %4\n\tv_mov_b32 v43, %5\n\tv_mov_b32 v60, %6\n\tv_mov_b32 v61, %7\n\ts_nop 7\n\t"
#define GET "\n\ts_nop 7\n\tv_mov_b32 %0, v52\n\tv_mov_b32 %1, v53"
#define REGS "v42", "v43", "v52", "v53", "v60", "v61", "v74", "v75", "s20", "s21"
#define FORM_LIST(X) \
    X(0,  "v_pk_mul_f32 v[52:53], v[74:75], v[42:43] op_sel:[0,1]",                 __fmul_rn(a0, b1), __fmul_rn(a1, b1)) \
    X(1,  "v_pk_mul_f32 v[52:53], v[74:75], v[42:43]",                              __fmul_rn(a0, b0), __fmul_rn(a1, b1)) \
    X(2,  "v_pk_mul_f32 v[52:53], v[74:75], v[42:43] op_sel_hi:[1,0]",              __fmul_rn(a0, b0), __fmul_rn(a1, b0)) \
    X(3,  "v_pk_mul_f32 v[52:53], v[74:75], v[42:43] op_sel:[1,0]",                 __fmul_rn(a1, b0), __fmul_rn(a1, b1)) \
    X(4,  "v_pk_mul_f32 v[52:53], v[74:75], v[42:43] op_sel:[1,1]",                 __fmul_rn(a1, b1), __fmul_rn(a1, b1)) \
    X(5,  "v_pk_mul_f32 v[52:53], v[74:75], v[42:43] op_sel_hi:[0,1]",              __fmul_rn(a0, b0), __fmul_rn(a0, b1)) \
    X(6,  "v_pk_mul_f32 v[52:53], v[74:75], v[42:43] op_sel:[0,1] op_sel_hi:[1,0]", __fmul_rn(a0, b1), __fmul_rn(a1, b0)) \
    X(7,  "v_pk_add_f32 v[52:53], v[74:75], v[42:43] op_sel:[0,1]",                 __fadd_rn(a0, b1), __fadd_rn(a1, b1)) \
    X(8,  "v_pk_fma_f32 v[52:53], v[74:75], v[42:43], v[60:61] op_sel:[0,1,0]",     __fmaf_rn(a0, b1, c0), __fmaf_rn(a1, b1, c1)) \
    X(9,  "v_pk_mov_b32 v[52:53], v[74:75], v[42:43] op_sel:[1,0]",                 a1, b0) \
    X(10, "v_mul_f32 v52, v74, v43\n\tv_mul_f32 v53, v75, v43",                       __fmul_rn(a0, b1), __fmul_rn(a1, b1)) \
    /* cross-lane operand forms the library's reductions and epilogues use (DPP adds, lane reads, the half-wave swap): do they share the behaviour? */ \
    X(11, "v_add_f32_dpp v52, v74, v42 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v53, v75, v43 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf", \
          __fadd_rn(__shfl_xor(a0, 1, 64), b0), __fadd_rn(__shfl_xor(a1, 2, 64), b1)) \
    X(12, "v_add_f32_dpp v52, v74, v42 row_half_mirror row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp v53, v75, v43 row_mirror row_mask:0xf bank_mask:0xf", \
          __fadd_rn(__shfl(a0, (lane & ~7) | (7 - (lane & 7)), 64), b0), __fadd_rn(__shfl(a1, (lane & ~15) | (15 - (lane & 15)), 64), b1)) \
    X(13, "v_readlane_b32 s20, v74, 55\n\tv_readlane_b32 s21, v75, 63\n\ts_nop 3\n\tv_add_f32 v52, s20, v42\n\tv_add_f32 v53, s21, v43", \
          __fadd_rn(__shfl(a0, 55, 64), b0), __fadd_rn(__shfl(a1, 63, 64), b1)) \
    X(14, "v_mov_b32 v52, v74\n\tv_mov_b32 v53, v42\n\ts_nop 1\n\tv_permlane32_swap_b32 v52, v53", \
          sw_lo, sw_hi)
template <int FORM> __global__ __launch_bounds__(512) void victim(unsigned* bad, int iters) {
    const int lane = threadIdx.x & 63;
    float a0 = 1.0f + lane * 0.37f, a1 = 2.0f + lane * 0.11f, b0 = 0.5f + lane * 0.07f, b1 = 3.0f + lane * 0.013f, c0 = 0.25f + lane, c1 = 7.0f - lane;
    unsigned nlo = 0, nhi = 0;
    for (int it = 0; it < iters; ++it) {
        float lo = 0.f, hi = 0.f, elo = 0.f, ehi = 0.f;
        b1 += 0.25f; a0 -= 0.125f; a1 += 0.5f; b0 -= 0.0625f;
        const float b_dn = __shfl(b0, (lane + 32) & 63, 64), a_up = __shfl(a0, (lane + 32) & 63, 64);      // every lane takes part (no divergent shuffle)
        const float sw_lo = lane < 32 ? a0 : b_dn, sw_hi = lane < 32 ? a_up : b0;      // v_permlane32_swap: lanes 32-63 of the first register <-> lanes 0-31 of the second
        (void)sw_lo; (void)sw_hi;
#define X(N, INSTR, ELO, EHI) if constexpr (FORM == N) { asm volatile(SET INSTR GET : "=v"(lo), "=v"(hi) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1) : REGS); elo = ELO; ehi = EHI; }
        FORM_LIST(X)
#undef X
        nlo += __float_as_uint(lo) != __float_as_uint(elo); nhi += __float_as_uint(hi) != __float_as_uint(ehi);
    }
    if (nlo) atomicAdd(&bad[(lane >> 4) * 2], nlo);
    if (nhi) atomicAdd(&bad[(lane >> 4) * 2 + 1], nhi);
}
// synthetic aggressors (which property of the GEMM wave matters?): MFMA loop with 0: a small register footprint  1: 251+ VGPRs, no AGPRs
// 2: ~232 VGPRs + 64 AGPRs (more than 256 registers in total, the GEMM kernel's allocation)  3: as 0, plus 64 KiB of LDS per workgroup
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int KIND> __global__ __launch_bounds__(256) void aggressor(int n, unsigned* sink) {
    extern __shared__ float dyn[];
    const int lane = threadIdx.x & 63;
    if (KIND == 3) dyn[threadIdx.x] = lane;
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * lane + i); b[i] = (_Float16)(0.5f - 0.003f * lane * i); }
    f16v c0 = {0}, c1 = {0};
    float keep = lane;
    if (KIND == 1) asm volatile("v_mov_b32 v250, %0" :: "v"(keep) : "v250");
    if (KIND == 2) asm volatile("v_mov_b32 v230, %0\n\tv_accvgpr_write_b32 a63, %0" :: "v"(keep) : "v230", "a63");
    for (int it = 0; it < n; ++it) { c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0); }
    if (KIND == 1) asm volatile("v_add_f32 %0, %0, v250" : "+v"(keep) :: "v250");
    if (KIND == 2) asm volatile("v_accvgpr_read_b32 %0, a63\n\ts_nop 1\n\tv_add_f32 %0, %0, v230" : "+v"(keep) :: "v230", "a63");
    if (c0[0] + c1[1] + keep == 12345.678f) sink[63] = 1;
}
extern "C" __attribute__((visibility("default"))) int pk_probe_aggressor(int kind, int n, int blocks, void* stream, unsigned* sink) {
    hipStream_t s = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(aggressor<0>, dim3(blocks), dim3(256), 0, s, n, sink);
    if (kind == 1) hipLaunchKernelGGL(aggressor<1>, dim3(blocks), dim3(256), 0, s, n, sink);
    if (kind == 2) hipLaunchKernelGGL(aggressor<2>, dim3(blocks), dim3(256), 0, s, n, sink);
    if (kind == 3) { hipFuncSetAttribute((const void*)aggressor<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); hipLaunchKernelGGL(aggressor<3>, dim3(blocks), dim3(256), 65536, s, n, sink); }
    return (int)hipGetLastError();
}
extern "C" __attribute__((visibility("default"))) const char* pk_probe_name(int form) {
#define X(N, INSTR, ELO, EHI) if (form == N) return INSTR;
    FORM_LIST(X)
#undef X
    return nullptr;
}
// the victim alone on `stream`; out8 = wrong LOW / HIGH results per lane quarter [lo q0, hi q0, lo q1, ...].  The caller keeps the aggressor busy on another stream.
extern "C" __attribute__((visibility("default"))) int pk_probe_run(int form, int iters, int blocks, void* stream, unsigned* out8) {
    static unsigned* d = nullptr;
    if (!d && hipMalloc(&d, 256) != hipSuccess) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(d, 0, 256, s) != hipSuccess) return -1;
#define X(N, INSTR, ELO, EHI) if (form == N) hipLaunchKernelGGL(victim<N>, dim3(blocks), dim3(512), 0, s, d, iters);
    FORM_LIST(X)
#undef X
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    return hipMemcpy(out8, d, 32, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
