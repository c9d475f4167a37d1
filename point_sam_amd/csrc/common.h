// Shared helpers for the gfx950 (CDNA4, wave64) kernels of libpointsam_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define PSAM_OK 0
#define PSAM_EINVAL (-1)     // bad shape / null pointer / unsupported size
#define PSAM_EALIGN (-2)     // pointer or leading dimension not aligned as the kernel requires
#define PSAM_EWORKSPACE (-3) // workspace too small

#define PSAM_API extern "C" __attribute__((visibility("default")))

// The public header is part of every translation unit: a definition whose signature drifts from its declaration in include/pointsam_hip.h is a
// compile error (C linkage cannot be overloaded), and the structs that cross the ABI exist once.
#pragma GCC visibility push(default)
#include "../../include/pointsam_hip.h"
#include <cstdlib>
#pragma GCC visibility pop

void psam_set_error(const char* msg);

#define PSAM_REQUIRE(cond, code, msg)                 \
    do {                                              \
        if (!(cond)) {                                \
            psam_set_error(msg);                      \
            return (code);                            \
        }                                             \
    } while (0)

// Returns 0 or the positive hipError_t of the launch just issued.
static inline int32_t psam_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        psam_set_error(what);
        return (int32_t)e;
    }
    return PSAM_OK;
}

static inline int64_t psam_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

#ifdef __HIPCC__
constexpr int WAVE = 64;

// Power-of-two row scale of the f16x3 GEMM: 2^(14 - e), e = floor(log2(row maximum)) clamped below at -112 (scale and its inverse stay
// normal fp32 numbers over the whole finite range: 2^-113 .. 2^126 and 2^-126 .. 2^113); 1 for an
// all-zero / non-finite row (gemm_f16x3.hip).
__device__ __forceinline__ float f16_row_scale(float amax) {
    const unsigned bits = __builtin_bit_cast(unsigned, amax);
    int e = (int)((bits >> 23) & 0xff) - 127;
    if (bits == 0 || e == 128) return 1.f;
    e = e < -112 ? -112 : e;
    return __builtin_bit_cast(float, (unsigned)(127 + 14 - e) << 23);
}

// (x0, x1) * s (s = power-of-two row scale) -> packed fp16 pairs hi = RNE(s*x), lo = RNE(s*x - hi): the f16x3 operand split
// (gemm_f16x3.hip); the residual is exact (one FMA).
typedef float psam_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 psam_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void psam_split2_f16(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
    const psam_f32x2 xs = psam_f32x2{x0, x1} * s;
    const psam_f16x2 h = __builtin_convertvector(xs, psam_f16x2);
    const psam_f32x2 r = xs - __builtin_convertvector(h, psam_f32x2);
    hi = __builtin_bit_cast(unsigned, h);
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, psam_f16x2));
}

// Reductions on the VALU's data-parallel-primitive path instead of ds_bpermute (an LDS round trip, ~100+ cycles per step, six steps for a
// wave): quad_perm xor 1 / xor 2, row_half_mirror, row_mirror leave every lane of a 16-lane row with the row's result in four
// few-cycle steps.  Whole wave must be active.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) { return __builtin_bit_cast(float, dpp_i32<CTRL>(__builtin_bit_cast(int, v))); }
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f32<DPP_XOR1>(v)); v = fmaxf(v, dpp_f32<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f32<DPP_HALF_MIRROR>(v)); v = fmaxf(v, dpp_f32<DPP_MIRROR>(v));
    return v;
}
__device__ __forceinline__ int row16_min(int v) {
    v = min(v, dpp_i32<DPP_XOR1>(v)); v = min(v, dpp_i32<DPP_XOR2>(v));
    v = min(v, dpp_i32<DPP_HALF_MIRROR>(v)); v = min(v, dpp_i32<DPP_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f32<DPP_XOR1>(v); v += dpp_f32<DPP_XOR2>(v);
    v += dpp_f32<DPP_HALF_MIRROR>(v); v += dpp_f32<DPP_MIRROR>(v);
    return v;
}
// Wave reductions: four DPP steps inside the 16-lane rows, then two cross-row exchanges through ds_bpermute (xor 16, xor 32): every lane
// ends up with the result, in a fixed order.  (A variant that combined the rows with v_readlane of lanes 0/16/32/48 was faster still but
// gave run-to-run different LayerNorm statistics in one multi-stream test -- 9 failures in 10 runs against 0 in 10 for this form and for
// the plain six-step shuffle butterfly -- so SGPR read-back of DPP results is not used.)
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float wave_sum(float v) {      // pairs, quads, 8s, 16s inside a row, then rows 16 apart, then 32 apart
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ int wave_min_dpp(int v) {
    v = row16_min(v);
    v = min(v, __shfl_xor(v, 16, 64));
    return min(v, __shfl_xor(v, 32, 64));
}
// exact (erf) GELU, as torch.nn.GELU() default
// (Round 4: a branch-free erf -- Abramowitz & Stegun 7.1.26, ~16 VALU instructions against the device library's ~45 with both branches taken --
// changed nothing in the benchmark: 796.6 / 797.6 against 796.7 / 795.9 clouds/s on one box; the GELUs sit in memory-bound passes.  Not kept.)
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// Squared distance with the oracle's operation order and NO fused multiply-add:
// ((dx*dx) + dy*dy) + dz*dz, every op rounded to fp32 (oracle/tokenizer_oracle.c: dist2).
// HIP's __fmul_rn/__fadd_rn are plain operators (contractable), so contraction is switched off by pragma
// here and by -ffp-contract=off for the whole tokenizer translation unit.
__device__ __forceinline__ float dist2_exact(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    float s = dx * dx;
    const float t = dy * dy;
    s = s + t;
    const float u = dz * dz;
    s = s + u;
    return s;
}

// Experiments builds only: PSAM_ABLATE_REPEAT (bit mask, environment, read once) makes an idempotent kernel launch TWICE -- the throughput lost to the second
// launch is the kernel's exposed time in the pipeline (profiles/r06/r06_refill.txt): 1 = psam_attention_packed, 2 = psam_layernorm (out of place),
// 4 = psam_ln_stats_finalize.
static inline int psam_ablate_repeat() {
#ifdef PSAM_BUILD_EXPERIMENTS
    static int m = -1;
    if (m < 0) { const char* e = getenv("PSAM_ABLATE_REPEAT"); m = e ? atoi(e) : 0; }
    return m;
#else
    return 0;
#endif
}

// Layout of the caller's arrival-counter block (PSAM_COUNTER_BYTES = 16384 ints, include/pointsam_hip.h), in ints.  Launches that use one block are ordered,
// so the users could share words; they have ranges of their own anyway (a fault in one must not poison the others).
constexpr int PSAM_CNT_GEMM = 0, PSAM_CNT_GEMM_N = 4096;          // split-K GEMM: one word per output tile
constexpr int PSAM_CNT_ATTN = 4096, PSAM_CNT_ATTN_N = 4096;       // key-split attention: one word per (query block, head, batch)
constexpr int PSAM_CNT_ROW = 8192;                                // skinny Linear + LayerNorm: one word
static_assert((PSAM_CNT_ROW + 1) * 4 <= PSAM_COUNTER_BYTES, "counter block layout");
#endif
