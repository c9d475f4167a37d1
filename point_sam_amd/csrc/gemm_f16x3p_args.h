// Shared by the two packed-operand GEMM translation units (gemm_f16x3p.hip: lock-step ring kernels; gemm_f16x3pp.hip: the
// ping-pong kernel): argument block, LDS-DMA macro, tile-order panel model.
#pragma once
#include <cstdlib>
#include "common.h"

typedef float pf32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 pf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));

struct F16PArgs {
    const unsigned char* A; const unsigned char* W; float* C;
    const float* bias; const float* residual; const float* rowbias;
    const float* scaleA; const float* scaleW;
    int64_t lda, ldw, ldc, ldr, ldrb;     // lda / ldw in 32-bit containers
    int M, N, K, rowgroup, act;
    float alpha;
    int tiles_m, tiles_n, panel;      // panel: width (in column tiles) of the column panels the tile order walks row-major (f16x3p_panel)
    // fused extras (psam_gemm_fuse_t, see gemm_epilogue.h): all null / 0 for the plain GEMM
    float* out_scale; float out_k1, out_k2; int pack_out;
    const float* out_bound;           // pack_out: per-row bound of the output (replaces the k1 / k2 form)
    float* stats; int stat_cols, stat_segs;
    const float* ln_mean; const float* ln_rstd; const float* ln_c;
    float* gmax_out; int64_t gmax_ld; int gmax_k, no_store;
    const float* row_ln_g; const float* row_ln_b; float row_ln_eps;
    const float* hyper; float* masks; int hyper_c, hyper_rows; int64_t hyper_pstride;
    int ksplit; int64_t plane;        // split-K (lock-step kernel): workgroup (tile, split) sums its share of the K slabs into C + split * plane
    float* sk_part; int* sk_count;    // split-K with the in-kernel fix-up: raw accumulator tiles [tile][split] and one arrival counter per tile (zero between launches);
                                      // the LAST workgroup of a tile adds the ksplit partials in split order and runs the whole epilogue on the sum
    int epi_abl;      // measurement builds (-DPSAM_GEMM_ABLATE): parts of the epilogue switched off (gemm_epilogue.h)
    unsigned* dbg;    // measurement builds, ABL & 64: 16 words per wave (cycle budget of the K loop, gemm_f16x3p.hip); null otherwise
};

// s_memtime + the wait for it in ONE statement (hipcc does not count an asm SMEM operation; the stamps sit where no DS read is outstanding)
__device__ __forceinline__ unsigned long long gemm_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t)::"memory");
    return t;
}

#define P_LDS(ptr) ((__attribute__((address_space(3))) void*)(ptr))
// LDS-DMA of 16 bytes per lane: LDS[dst + lane * 16] = buffer[voff(lane) + soff].  The builtin exists only in the device compilation
// (the host pass of this translation unit must still parse the kernel template to emit its launch stub).
#if defined(__HIP_DEVICE_COMPILE__)
#define P_DMA16(rsrc, dst, voff, soff) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, P_LDS(dst), 16, voff, soff, 0, 0)
#else
#define P_DMA16(rsrc, dst, voff, soff) ((void)(rsrc), (void)(dst), (void)(voff), (void)(soff))
#endif

// Column-panel width of the tile order (see the kernel).  Fabric-side traffic model per XCD, which owns ntiles / 8 consecutive tiles and
// keeps about 2.5 MiB of operands in its L2: with panels of P column tiles the XCD's W panel (P * BN * K * 4 B) is fetched once per panel
// it touches if it fits, once per group of concurrently running row bands if it does not; every A row band (BM * K * 4 B) of the panel
// is fetched once.  The P with the least modelled traffic wins (ties: the widest).  PSAM_GEMM_PANEL overrides (0: plain row-major).
// Measured (profiles/r02/r02_gemm_panel_sweep.log): qkv 81.7 -> 79.6 us, fc1 143.4 -> 138.2 us, two-stream layer 303.9 -> 292 us.
static inline int f16x3p_panel(int tiles_m, int tiles_n, int BM, int BN, int K) {
    static int forced = -2;
    if (forced == -2) { const char* e = getenv("PSAM_GEMM_PANEL"); forced = e ? atoi(e) : -1; }
    if (forced == 0) return tiles_n;
    if (forced > 0) return forced < tiles_n ? forced : tiles_n;
    const double l2 = 2.5 * 1048576.0, a_band = (double)BM * K * 4, w_col = (double)BN * K * 4;
    const double chunk = (double)tiles_m * tiles_n / 8.0;
    int best = tiles_n;
    double best_cost = 1e300;
    for (int P = tiles_n; P >= 1; P = P > 1 ? (P + 1) / 2 : 0) {
        const double rows = chunk / P < tiles_m ? chunk / P : tiles_m;                 // row bands an XCD walks inside a panel
        const double panels = chunk / ((double)tiles_m * P) > 1.0 ? chunk / ((double)tiles_m * P) : 1.0;   // panels it touches
        const double wp = P * w_col, conc = 64.0 / P > 1.0 ? 64.0 / P : 1.0;           // ~64 tiles of an XCD in flight: conc row bands share a panel pass
        const double w_cost = (wp <= l2 || rows <= conc) ? wp * panels : wp * (rows / conc) * panels;
        const double cost = 8.0 * (w_cost + rows * a_band * panels);
        if (cost < best_cost * 0.999) { best_cost = cost; best = P; }
    }
    return best;
}


// > 64 KiB of dynamic LDS must be opted into per kernel AND per device (the attribute lives with the device's code object)
template <typename K>
static inline bool f16x3p_reserve_lds(K kernel, int lds, unsigned long long& done_mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&done_mask, __ATOMIC_ACQUIRE) & bit) return true;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return false;
    __atomic_fetch_or(&done_mask, bit, __ATOMIC_RELEASE);
    return true;
}

// gemm_f16x3pp.hip: ping-pong configurations (cfg 50 ..); returns PSAM_EINVAL for an unknown one
int32_t launch_f16x3pp(int cfg, F16PArgs& p, hipStream_t stream);
bool f16x3pp_supports(int cfg, int act, bool stats, bool gmax, bool hyper);
int f16x3pp_pick(int M, int N, int K, int act);      // -1: keep the lock-step kernel
bool f16x3p_use_register_epilogue(const F16PArgs& p);      // gemm_f16x3p.hip: whether this launch may run the register-only epilogue (gemm_epilogue_t.h)
// gemm_f16x3c.hip: persistent form of the 128x128 register-epilogue configuration -- whole tiles from a queue, one continuous stream of K slabs per
// workgroup; true = it took the launch (rc = status)
bool launch_f16x3c(F16PArgs& p, hipStream_t stream, int32_t& rc, int wgs_per_cu = 2);
void f16x3c_reset_state(hipStream_t stream);
// gemm_f16x3s.hip: persistent stream-K form of the 128x128 register-epilogue configuration; true = it took the launch (rc = status)
bool launch_f16x3s(F16PArgs& p, hipStream_t stream, int32_t& rc, int mode = 0);
void f16x3s_reset_state(hipStream_t stream);
// gemm_f16x3q.hip: lock-step kernel on a ring of k16 units (cfg 80 ..), register epilogue only
int32_t launch_f16x3q(int cfg, F16PArgs& p, hipStream_t stream);
bool f16x3q_supports(int cfg, const F16PArgs& p);
