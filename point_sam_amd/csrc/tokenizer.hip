// Point tokenizer kernels for gfx950: farthest point sampling, kNN grouping, 3-NN interpolation weights,
// neighbourhood gather and the first (tiny-K) layer of the per-group mini-PointNet.
//
// Reference semantics (file:line under /root/reference):
//   FPS            pc_sam/model/common.py:91  (torkit3d.sample_farthest_points; third party, absent)
//   center gather  pc_sam/model/common.py:92  (torkit3d batch_index_select)
//   kNN            pc_sam/model/common.py:27-56,97 (torch.cdist + topk)
//   grouping       pc_sam/model/common.py:99-120
//   3-NN weights   pc_sam/model/common.py:238-255
// Numerical spec = oracle/tokenizer_oracle.c: fp32 d = (dx*dx + dy*dy) + dz*dz without FMA contraction,
// ties broken by lowest index.  This file is compiled with -ffp-contract=off and additionally uses the
// explicitly rounded intrinsics in dist2_exact().
#include "common.h"
#include <math.h>

// ------------------------------------------------------------------------------------------------
// FPS.  One 1024-thread workgroup per cloud (G dependent iterations; per iteration one block-wide
// arg-max).  Coordinates are re-laid out once as planar x[] y[] z[] (float4-coalesced loads); the running
// min-distance lives in registers (N <= 32768) or in the workspace (larger N); for N <= 32768 the coordinates
// are held on chip as well (registers + LDS), see fps_kernel.
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int FPS_THREADS = 1024;
constexpr int FPS_WAVES = FPS_THREADS / WAVE;

static inline int64_t fps_npad(int64_t N) { return psam_cdiv(N, 4 * FPS_THREADS) * (4 * FPS_THREADS); }

__global__ void fps_soa_kernel(const float* __restrict__ xyz, int N, int64_t npad, float* __restrict__ ws) {
    const int b = blockIdx.y;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= npad) return;
    float x = 0.f, y = 0.f, z = 0.f;
    if (n < N) {
        const float* p = xyz + ((int64_t)b * N + n) * 3;
        x = p[0]; y = p[1]; z = p[2];
    }
    float* base = ws + (int64_t)b * 3 * npad;
    base[n] = x;
    base[npad + n] = y;
    base[2 * npad + n] = z;
}

// PPT4 = float4 groups (4 points each) per thread.  PPT4 > 0: the whole cloud lives on chip -- the running
// min-distances in VGPRs, the coordinates of the first RG <= 5 groups in VGPRs and of the next LG <= 3 groups in LDS
// (each thread only ever reads back its own 16-byte slots: LDS as spill space with ds_read_b128, no barrier involved).
// A 1024-thread workgroup owns the CU: 128 VGPRs/lane (32 min-dist + 60 xyz + temporaries) and 144 of the 160 KiB LDS
// hold a 32768-point cloud (512 KiB of state), so an iteration touches no memory besides the 12-byte winner broadcast.
// (SG = groups that would not fit and are re-read from L2 each iteration; 0 for every instantiated size.)
// PPT4 == 0: any N, min-distances and coordinates streamed from the workspace (L2-resident).
// x - c for four points as two packed subtractions (v_pk_add_f32 with the negate modifier on the SGPR-pair operand; a + (-c) is
// a - c bit for bit).  Written as asm because the compiler keeps subtractions of a scalar as four v_sub_f32.
typedef float fps_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 fps_sub_bcast(const f32x4 a, const fps_f32x2 c) {
    fps_f32x2 lo = {a.x, a.y}, hi = {a.z, a.w}, rl, rh;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(rl) : "v"(lo), "s"(c));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(rh) : "v"(hi), "s"(c));
    return f32x4{rl.x, rl.y, rh.x, rh.y};
}

// fminf() costs a second instruction per call (v_max_f32 x, x, x: NaN quieting of an operand the compiler cannot prove canonical);
// coordinates are finite by contract (the distances are then never NaN), so the bare v_min_f32 gives the same bits.
__device__ __forceinline__ float fps_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int PPT4>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(const float* __restrict__ xyz, const float* __restrict__ soa,
                                                          float* __restrict__ mdg, int N, int64_t npad, int G,
                                                          int64_t* __restrict__ idx_out, float* __restrict__ centers_out) {
    constexpr int RG = PPT4 > 5 ? 5 : (PPT4 > 0 ? PPT4 : 1);                    // groups with coordinates in registers
    constexpr int LG = PPT4 > 5 ? (PPT4 - 5 > 3 ? 3 : PPT4 - 5) : 0;          // groups with coordinates parked in LDS
    constexpr int SG = PPT4 > 0 ? PPT4 - RG - LG : 0;                          // highest group(s): re-read from L2 (48 KiB/iter)
    static_assert(SG <= 1, "register + LDS capacity of one CU");
    constexpr int NREG = PPT4 > 0 ? PPT4 : 1;
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const float* P = xyz + (int64_t)b * N * 3;
    const float* soa_b = soa + (int64_t)b * 3 * npad;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)soa_b, 0, (int)(3 * npad * 4), 0x00020000);
    const int plane_bytes = (int)(npad * 4);
    const int voff = tid * 16;
    float4* MD4 = reinterpret_cast<float4*>(mdg + (int64_t)b * npad);
    const int ngroups = (int)(npad / (4 * FPS_THREADS));

    __shared__ float s_val[2][FPS_WAVES];
    __shared__ int s_idx[2][FPS_WAVES];
    __shared__ __attribute__((aligned(16))) f32x4 s_xyz[LG > 0 ? LG * 3 * FPS_THREADS : 1];

    auto gload = [&](int g, int plane) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, g * (FPS_THREADS * 16) + plane * plane_bytes, 0));
    };

    f32x4 md[NREG];
    f32x4 rx[RG], ry[RG], rz[RG];
    if (PPT4 > 0) {
#pragma unroll
        for (int g = 0; g < NREG; ++g) {
            const int base = (g * FPS_THREADS + tid) * 4;
            md[g].x = base + 0 < N ? INFINITY : -1.0f;  // padding never wins: real min-distances are >= 0
            md[g].y = base + 1 < N ? INFINITY : -1.0f;
            md[g].z = base + 2 < N ? INFINITY : -1.0f;
            md[g].w = base + 3 < N ? INFINITY : -1.0f;
        }
#pragma unroll
        for (int g = 0; g < RG; ++g) { rx[g] = gload(g, 0); ry[g] = gload(g, 1); rz[g] = gload(g, 2); }
#pragma unroll
        for (int g = 0; g < LG; ++g) {
            s_xyz[(g * 3 + 0) * FPS_THREADS + tid] = gload(RG + g, 0);
            s_xyz[(g * 3 + 1) * FPS_THREADS + tid] = gload(RG + g, 1);
            s_xyz[(g * 3 + 2) * FPS_THREADS + tid] = gload(RG + g, 2);
        }
    } else {
        for (int g = 0; g < ngroups; ++g) {
            const int base = (g * FPS_THREADS + tid) * 4;
            float4 m;
            m.x = base + 0 < N ? INFINITY : -1.0f;
            m.y = base + 1 < N ? INFINITY : -1.0f;
            m.z = base + 2 < N ? INFINITY : -1.0f;
            m.w = base + 3 < N ? INFINITY : -1.0f;
            MD4[g * FPS_THREADS + tid] = m;
        }
    }

    int last = 0;
    if (tid == 0) idx_out[(int64_t)b * G] = 0;
    if (tid < 3) centers_out[(int64_t)b * G * 3 + tid] = P[tid];

    if constexpr (PPT4 > 0) {
        // On-chip cloud.  An iteration is (a) the scan: 5.5 VALU instructions per point -- packed fp32 subtract / multiply / add (v_pk_*,
        // two points per instruction, every operation individually rounded: the same bits as dist2_exact), one min per point, one
        // v_max3 per two points for the thread's maximum VALUE only; (b) the maximum over the workgroup (wave reduction, LDS, barrier);
        // (c) the index: only the lanes that hold the maximum look for its lowest slot, the lowest global index among them wins (wave
        // minimum, then the minimum over the waves' records), and that lane publishes index AND coordinates from its own registers /
        // LDS slots -- the next iteration starts from LDS instead of a dependent load through L2.
        // (The earlier form carried (value, slot) through the scan at 12 instructions per point and fetched P[last] from memory.)
        __shared__ float s_wmax[2][FPS_WAVES];
        __shared__ int s_ci[2][FPS_WAVES];                                      // each wave's candidate: index, coordinates
        __shared__ float s_cxyz[2][3][FPS_WAVES];
        float cx = P[0], cy = P[1], cz = P[2];
        for (int j = 1; j < G; ++j) {
            float best = -1.0f;
            const fps_f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};      // wave-uniform: SGPR pairs
            auto visit = [&](f32x4& m, const f32x4 x, const f32x4 y, const f32x4 z) {
                const f32x4 dx = fps_sub_bcast(x, c2x), dy = fps_sub_bcast(y, c2y), dz = fps_sub_bcast(z, c2z);
                const f32x4 d = (dx * dx + dy * dy) + dz * dz;      // -ffp-contract=off: no FMA, each op rounded (dist2_exact)
                m = f32x4{fps_min(m.x, d.x), fps_min(m.y, d.y), fps_min(m.z, d.z), fps_min(m.w, d.w)};
                best = fmaxf(fmaxf(best, m.x), m.y);
                best = fmaxf(fmaxf(best, m.z), m.w);
            };
            f32x4 sx, sy, sz;
            if (SG > 0) { sx = gload(RG + LG, 0); sy = gload(RG + LG, 1); sz = gload(RG + LG, 2); }
#pragma unroll
            for (int g = 0; g < RG; ++g) visit(md[g], rx[g], ry[g], rz[g]);
#pragma unroll
            for (int g = 0; g < LG; ++g) {
                __builtin_amdgcn_sched_barrier(0);  // one LDS group (12 VGPRs) in flight at a time: the register file is full
                visit(md[RG + g], s_xyz[(g * 3 + 0) * FPS_THREADS + tid], s_xyz[(g * 3 + 1) * FPS_THREADS + tid], s_xyz[(g * 3 + 2) * FPS_THREADS + tid]);
            }
            if (SG > 0) visit(md[RG + LG], sx, sy, sz);
            // (b) maximum value over the workgroup
            const int slot = j & 1;
            const float wmax = wave_max(best);
            if (lane == 0) s_wmax[slot][wave] = wmax;
            __syncthreads();
            static_assert(FPS_WAVES == 16, "one 16-lane row holds the waves' partial results");
            const float gmax = row16_max(s_wmax[slot][lane & 15]);
            // (c) lowest index holding it
            const bool mine = best == gmax;
            if (__builtin_amdgcn_ballot_w64(mine) != 0) {      // wave-uniform
                int cand = 0x7fffffff, bslot = 0;
                if (mine) {
#pragma unroll
                    for (int g = NREG - 1; g >= 0; --g) {      // descending: the lowest slot is assigned last
                        if (md[g].w == gmax) bslot = 4 * g + 3;
                        if (md[g].z == gmax) bslot = 4 * g + 2;
                        if (md[g].y == gmax) bslot = 4 * g + 1;
                        if (md[g].x == gmax) bslot = 4 * g;
                    }
                    cand = ((bslot >> 2) * FPS_THREADS + tid) * 4 + (bslot & 3);
                }
                const int wmin = wave_min_dpp(cand);
                if (mine && cand == wmin) {                     // one lane: global indices are unique
                    const int g = bslot >> 2, k = bslot & 3;
                    f32x4 X, Y, Z;
                    if (g < RG) {
                        X = rx[0]; Y = ry[0]; Z = rz[0];
#pragma unroll
                        for (int gg = 1; gg < RG; ++gg)
                            if (g == gg) { X = rx[gg]; Y = ry[gg]; Z = rz[gg]; }
                    } else if (g < RG + LG) {
                        X = s_xyz[((g - RG) * 3 + 0) * FPS_THREADS + tid]; Y = s_xyz[((g - RG) * 3 + 1) * FPS_THREADS + tid];
                        Z = s_xyz[((g - RG) * 3 + 2) * FPS_THREADS + tid];
                    } else { X = gload(g, 0); Y = gload(g, 1); Z = gload(g, 2); }
                    const float px = k == 0 ? X.x : (k == 1 ? X.y : (k == 2 ? X.z : X.w));
                    const float py = k == 0 ? Y.x : (k == 1 ? Y.y : (k == 2 ? Y.z : Y.w));
                    const float pz = k == 0 ? Z.x : (k == 1 ? Z.y : (k == 2 ? Z.z : Z.w));
                    s_ci[slot][wave] = cand;
                    s_cxyz[slot][0][wave] = px; s_cxyz[slot][1][wave] = py; s_cxyz[slot][2][wave] = pz;
                }
            } else if (lane == 0) {
                s_ci[slot][wave] = 0x7fffffff;
            }
            __syncthreads();
            last = __builtin_amdgcn_readfirstlane(row16_min(s_ci[slot][lane & 15]));
            const int wwin = ((last >> 2) & (FPS_THREADS - 1)) >> 6;      // the wave that owns point `last`
            cx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_cxyz[slot][0][wwin])));
            cy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_cxyz[slot][1][wwin])));
            cz = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, s_cxyz[slot][2][wwin])));
            asm volatile("s_nop 4");      // the SGPRs just written by the VALU feed packed-fp32 VALU operands (inline asm) at the top of the loop
            if (tid == 0) idx_out[(int64_t)b * G + j] = last;
            if (tid < 3) centers_out[((int64_t)b * G + j) * 3 + tid] = tid == 0 ? cx : (tid == 1 ? cy : cz);
        }
        return;
    }

    for (int j = 1; j < G; ++j) {
        const float cx = P[(int64_t)last * 3 + 0], cy = P[(int64_t)last * 3 + 1], cz = P[(int64_t)last * 3 + 2];
        float best = -1.0f;
        int bslot = -1;
        auto visit = [&](f32x4& m, const f32x4 x, const f32x4 y, const f32x4 z, int g) {
            float d;
            d = dist2_exact(x.x, y.x, z.x, cx, cy, cz); m.x = fminf(m.x, d); if (m.x > best) { best = m.x; bslot = 4 * g; }
            d = dist2_exact(x.y, y.y, z.y, cx, cy, cz); m.y = fminf(m.y, d); if (m.y > best) { best = m.y; bslot = 4 * g + 1; }
            d = dist2_exact(x.z, y.z, z.z, cx, cy, cz); m.z = fminf(m.z, d); if (m.z > best) { best = m.z; bslot = 4 * g + 2; }
            d = dist2_exact(x.w, y.w, z.w, cx, cy, cz); m.w = fminf(m.w, d); if (m.w > best) { best = m.w; bslot = 4 * g + 3; }
        };
        // any N: min-distances and coordinates streamed from the (L2-resident) workspace
        for (int g = 0; g < ngroups; ++g) {
            float4 m4 = MD4[g * FPS_THREADS + tid];
            f32x4 m = {m4.x, m4.y, m4.z, m4.w};
            visit(m, gload(g, 0), gload(g, 1), gload(g, 2), g);
            MD4[g * FPS_THREADS + tid] = make_float4(m.x, m.y, m.z, m.w);
        }
        int besti = bslot < 0 ? 0x7fffffff : ((bslot >> 2) * FPS_THREADS + tid) * 4 + (bslot & 3);
        // wave arg-max, lowest index on ties
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(besti, o, 64);
            if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
        }
        const int slot = j & 1;
        if (lane == 0) { s_val[slot][wave] = best; s_idx[slot][wave] = besti; }
        __syncthreads();
        float v = lane < FPS_WAVES ? s_val[slot][lane] : -2.0f;
        int vi = lane < FPS_WAVES ? s_idx[slot][lane] : 0x7fffffff;
#pragma unroll
        for (int o = FPS_WAVES / 2; o > 0; o >>= 1) {
            const float ov = __shfl_xor(v, o, 64);
            const int oi = __shfl_xor(vi, o, 64);
            if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
        }
        last = __builtin_amdgcn_readfirstlane(vi);
        if (tid == 0) idx_out[(int64_t)b * G + j] = last;
        if (tid < 3) centers_out[((int64_t)b * G + j) * 3 + tid] = P[(int64_t)last * 3 + tid];
    }
}

// 64-bit maximum over a 16-lane row (four DPP steps, every lane ends with the row's result) and over the wave (+ two cross-row exchanges)
template <int CTRL>
__device__ __forceinline__ unsigned long long fps_dpp_u64(unsigned long long v) {
    const unsigned lo = (unsigned)dpp_i32<CTRL>((int)(unsigned)v), hi = (unsigned)dpp_i32<CTRL>((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long fps_max_u64(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned long long fps_row16_max_u64(unsigned long long v) {
    v = fps_max_u64(v, fps_dpp_u64<DPP_XOR1>(v)); v = fps_max_u64(v, fps_dpp_u64<DPP_XOR2>(v));
    v = fps_max_u64(v, fps_dpp_u64<DPP_HALF_MIRROR>(v)); v = fps_max_u64(v, fps_dpp_u64<DPP_MIRROR>(v));
    return v;
}
__device__ __forceinline__ unsigned long long fps_wave_max_u64(unsigned long long v) {
    v = fps_row16_max_u64(v);
    v = fps_max_u64(v, __shfl_xor(v, 16, 64));
    return fps_max_u64(v, __shfl_xor(v, 32, 64));
}

// ------------------------------------------------------------------------------------------------
// Cooperative FPS for large clouds (N > 32768: cfg #3, N = 131072, G = 2048).  The single-workgroup kernel above streams
// such a cloud from L2 on ONE CU (22 us per iteration, 45 ms per cloud); here W = npad / (4096 PPT4) workgroups share a cloud,
// each keeping its 4096 PPT4 points and their running min-distances in registers.  Per iteration every workgroup publishes its
// local (max min-distance, lowest index) candidate as ONE 64-bit key that also carries the iteration number (12-bit tag), into its
// own slot; wave 0 of every workgroup polls the W slots (one per lane) until all carry the tag and reduces them itself (same winner
// everywhere: no second broadcast).  The key is its own arrival flag, so an iteration costs one store and one (polled) load across
// the fabric -- the first version (key store, acknowledged, then a counter barrier, then the key loads: four round trips) took 3.7 us per
// iteration.  Slots are double-buffered by iteration parity: a workgroup can only be one iteration ahead of the slowest one (it needs
// everybody's key to go on), so a slot is rewritten only after all have read it, and the tag of what it held before differs by 2.
// Keys are agent-scope relaxed atomics (device-coherent sc1 accesses on both sides, no bulk cache maintenance).
// All B*W workgroups must be resident at once: the host only takes this path when B*W <= number of CUs, i.e.
// half of the 2-per-CU capacity for 1024-thread workgroups, so two such launches may overlap (BatchPipeline issues all tokenizer
// work on ONE stream, so they never do); more than two concurrent cooperative launches from different streams are not supported.
// Same arithmetic and tie-break as fps_kernel -> bit-identical indices.
// ------------------------------------------------------------------------------------------------
template <int PPT4>
__global__ __launch_bounds__(FPS_THREADS) void fps_coop_kernel(const float* __restrict__ xyz, const float* __restrict__ soa, int N, int64_t npad,
                                                               int G, int W, int xcd_stride, unsigned long long* __restrict__ cand,
                                                               int64_t* __restrict__ idx_out, float* __restrict__ centers_out) {
    // xcd_stride == 8: the grid is 8 W wide and only the workgroups whose id is congruent to the cloud's XCD (ids are dealt to the eight
    // XCDs round-robin) take part, the others leave at once -- all W workgroups of a cloud then share one XCD and the hand-over is
    // ~0.3 us shorter (scripts/exp/fabric_probe.hip, profiles/r03/r03_fabric_probe.txt).  Placement is a speed matter only: the keys are
    // agent-scope atomics either way.
    const int b = blockIdx.y;
    int w = blockIdx.x;
    if (xcd_stride > 1) {
        // linear workgroup id = blockIdx.x + gridDim.x * blockIdx.y; gridDim.x is a multiple of 8, so id % 8 == blockIdx.x % 8
        if ((int)(blockIdx.x & 7) != (b & 7)) return;
        w = blockIdx.x >> 3;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* P = xyz + (int64_t)b * N * 3;
    const float* soa_b = soa + (int64_t)b * 3 * npad;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)soa_b, 0, (int)(3 * npad * 4), 0x00020000);
    const int plane_bytes = (int)(npad * 4);
    __shared__ unsigned long long s_key[2][FPS_WAVES];
    __shared__ int s_last[2];
    unsigned long long* cand_b = cand + (int64_t)b * 2 * 64;

    f32x4 md[PPT4], rx[PPT4], ry[PPT4], rz[PPT4];
#pragma unroll
    for (int g = 0; g < PPT4; ++g) {
        const int gg = w * PPT4 + g;
        const int base = (gg * FPS_THREADS + tid) * 4;
        md[g].x = base + 0 < N ? INFINITY : -1.0f;
        md[g].y = base + 1 < N ? INFINITY : -1.0f;
        md[g].z = base + 2 < N ? INFINITY : -1.0f;
        md[g].w = base + 3 < N ? INFINITY : -1.0f;
        const int off = gg * (FPS_THREADS * 16);
        rx[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, off, 0));
        ry[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, off + plane_bytes, 0));
        rz[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, tid * 16, off + 2 * plane_bytes, 0));
    }
    int last = 0;
    if (w == 0) {
        if (tid == 0) idx_out[(int64_t)b * G] = 0;
        if (tid < 3) centers_out[(int64_t)b * G * 3 + tid] = P[tid];
    }
    for (int j = 1; j < G; ++j) {
        // the centre just selected: three scalar loads (uniform address; ~0.1 us from L2, scripts/exp/fabric_probe.hip)
        const float cx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, P[(int64_t)last * 3 + 0])));
        const float cy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, P[(int64_t)last * 3 + 1])));
        const float cz = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, P[(int64_t)last * 3 + 2])));
        asm volatile("s_nop 4");      // SGPRs possibly written by the VALU feed packed-fp32 VALU operands (inline asm) right below
        // (a) the scan of fps_kernel: packed fp32, every operation individually rounded (the bits of dist2_exact), the thread's maximum VALUE
        // only -- 5.5 VALU instructions per point instead of 12 with a (value, slot) pair carried along; with 16 waves sharing four SIMDs
        // the scan of 16 points per thread was 1.4 us of a 2.5 us iteration
        float best = -1.0f;
        const fps_f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
        for (int g = 0; g < PPT4; ++g) {
            const f32x4 dx = fps_sub_bcast(rx[g], c2x), dy = fps_sub_bcast(ry[g], c2y), dz = fps_sub_bcast(rz[g], c2z);
            const f32x4 d = (dx * dx + dy * dy) + dz * dz;      // -ffp-contract=off: no FMA
            md[g] = f32x4{fps_min(md[g].x, d.x), fps_min(md[g].y, d.y), fps_min(md[g].z, d.z), fps_min(md[g].w, d.w)};
            best = fmaxf(fmaxf(best, md[g].x), md[g].y);
            best = fmaxf(fmaxf(best, md[g].z), md[g].w);
        }
        // (b) the wave's candidate: its maximum, then -- only in the lanes that hold it -- the lowest slot, the lowest global index among them.
        // key: [63:32] min-distance bits (larger wins), [31:20] iteration tag, [19:0] 0xFFFFF - index (then the LOWER index wins; all keys
        // of an iteration carry the same tag, so it never decides).  A wave of pure padding (maximum < 0) holds distance 0, index field 0.
        // The tag makes the published key its own arrival flag: one store per workgroup and iteration, nothing to order it against.
        const unsigned tag = (unsigned)j & 0xFFFu;
        const float wmax = wave_max(best);
        unsigned long long key = (unsigned long long)tag << 20;
        if (wmax >= 0.f) {      // wave-uniform
            int cand = 0x7fffffff;
            if (best == wmax) {
                int bslot = 0;
#pragma unroll
                for (int g = PPT4 - 1; g >= 0; --g) {      // descending: the lowest slot is assigned last
                    if (md[g].w == wmax) bslot = 4 * g + 3;
                    if (md[g].z == wmax) bslot = 4 * g + 2;
                    if (md[g].y == wmax) bslot = 4 * g + 1;
                    if (md[g].x == wmax) bslot = 4 * g;
                }
                cand = ((w * PPT4 + (bslot >> 2)) * FPS_THREADS + tid) * 4 + (bslot & 3);
            }
            const unsigned wmin = (unsigned)wave_min_dpp(cand);
            key = ((unsigned long long)__builtin_bit_cast(unsigned, wmax) << 32) | (tag << 20) | (0xFFFFFu - wmin);
        }
        const int slot = j & 1;
        if (lane == 0) s_key[slot][wave] = key;
        __syncthreads();
        if (wave == 0) {
            static_assert(FPS_WAVES == 16, "one 16-lane row holds the waves' keys");
            const unsigned long long wk = fps_row16_max_u64(s_key[slot][lane & 15]);
            if (lane == 0) __hip_atomic_store(cand_b + slot * 64 + w, wk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // every lane < W polls one workgroup's slot until all of them carry this iteration's tag
            unsigned long long k;
            for (;;) {
                k = lane < W ? __hip_atomic_load(cand_b + slot * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 20);
                if (__all((((unsigned)k >> 20) & 0xFFFu) == tag)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            k = W <= 16 ? fps_row16_max_u64(k) : fps_wave_max_u64(k);      // W <= 16: the keys sit in lanes 0..15, one row
            if (lane == 0) s_last[slot] = (int)(0xFFFFFu - ((unsigned)k & 0xFFFFFu));
        }
        __syncthreads();
        last = __builtin_amdgcn_readfirstlane(s_last[slot]);
        if (w == 0) {
            if (tid == 0) idx_out[(int64_t)b * G + j] = last;
            if (tid < 3) centers_out[((int64_t)b * G + j) * 3 + tid] = P[(int64_t)last * 3 + tid];
        }
    }
}

// every slot starts with tag 0xFFF (no iteration < 4095 waits for it; by then the slot has long been rewritten)
__global__ void fps_coop_reset_kernel(unsigned long long* cand, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) cand[i] = ~0ull;
}

// ------------------------------------------------------------------------------------------------
// Cooperative FPS with exact spatial pruning (round 5).  An iteration of fps_coop_kernel updates the running minimum of EVERY point against the new
// centre, although late in the run a new centre only lowers the minimum of the points around it.  Here the cloud is first bucketed by a 32^3 grid in
// Morton order (counting sort: histogram, scan, scatter; the order inside a cell is whatever the atomics give -- results do not depend on it), so that
// the 256 PPT4 consecutive points a wave owns are a compact patch of the surface; every wave keeps the bounding box of its points and its current
// candidate (maximum of the running minima, lowest ORIGINAL index among the points that hold it).  Per iteration a wave first evaluates the distance
// from the new centre to its box WITH THE SCAN'S OWN ARITHMETIC -- per axis max(lo - c, c - hi, 0), then (ax ax + ay ay) + az az, every operation
// rounded to fp32: rounding is monotone, so this is <= the distance the scan would compute for any point inside the box -- and, if that is >= the
// wave's current maximum, no running minimum of the wave can change (min(m, d) with d >= box distance >= maximum >= m): the wave re-publishes its
// cached candidate and skips the scan.  Same distances, same minima, same tie-break (lowest original index) as fps_kernel: bit-identical indices.
// On a surface-like cloud the number of waves that scan falls like (1 + 16 / sqrt(j))^2 of 256 at iteration j; what remains per iteration is the
// hand-over between the workgroups.
// ------------------------------------------------------------------------------------------------
constexpr int FPS_CELL_BITS = 5, FPS_NCELL = 1 << (3 * FPS_CELL_BITS);
__device__ __forceinline__ int fps_ordered(float f) { const int i = __builtin_bit_cast(int, f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float fps_unordered(int i) { return __builtin_bit_cast(float, i >= 0 ? i : i ^ 0x7fffffff); }
__device__ __forceinline__ unsigned fps_spread3(unsigned v) {      // 5 bits -> every third bit
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6) | ((v & 16u) << 8);
}
__device__ __forceinline__ int fps_cell_code(float x, float y, float z, const int* __restrict__ bbox) {
    const float lx = fps_unordered(bbox[0]), ly = fps_unordered(bbox[1]), lz = fps_unordered(bbox[2]);
    const float hx = fps_unordered(bbox[3]), hy = fps_unordered(bbox[4]), hz = fps_unordered(bbox[5]);
    const float n = (float)(1 << FPS_CELL_BITS);
    auto cell = [&](float v, float lo, float hi) {
        const float e = hi - lo;
        const int c = e > 0.f ? (int)((v - lo) * (n / e)) : 0;
        return (unsigned)(c < 0 ? 0 : (c > (1 << FPS_CELL_BITS) - 1 ? (1 << FPS_CELL_BITS) - 1 : c));
    };
    return (int)(fps_spread3(cell(x, lx, hx)) | (fps_spread3(cell(y, ly, hy)) << 1) | (fps_spread3(cell(z, lz, hz)) << 2));
}
// bbox [B][8] ints (ordered-int encoding of the floats: min x y z, max x y z), hist [B][FPS_NCELL]
__global__ void fps_prune_reset_kernel(int* __restrict__ bbox, int* __restrict__ hist, int B) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (int64_t)B * FPS_NCELL) hist[i] = 0;
    if (i < B * 8) bbox[i] = (i & 7) < 3 ? 0x7fffffff : (int)0x80000000;
}
__global__ __launch_bounds__(256) void fps_bbox_kernel(const float* __restrict__ xyz, int N, int* __restrict__ bbox) {
    const int b = blockIdx.y;
    const float* P = xyz + (int64_t)b * N * 3;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256)
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = P[n * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float l = -wave_max(-lo[a]), h = wave_max(hi[a]);
        if ((threadIdx.x & 63) == 0) { atomicMin(bbox + b * 8 + a, fps_ordered(l)); atomicMax(bbox + b * 8 + 3 + a, fps_ordered(h)); }
    }
}
__global__ __launch_bounds__(256) void fps_cell_hist_kernel(const float* __restrict__ xyz, int N, const int* __restrict__ bbox, int* __restrict__ hist) {
    const int b = blockIdx.y;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float* p = xyz + ((int64_t)b * N + n) * 3;
    atomicAdd(hist + (int64_t)b * FPS_NCELL + fps_cell_code(p[0], p[1], p[2], bbox + b * 8), 1);
}
// exclusive scan of a cloud's FPS_NCELL cell counts, in place: one 1024-thread workgroup per cloud, 32 cells per thread
__global__ __launch_bounds__(1024) void fps_cell_scan_kernel(int* __restrict__ hist) {
    constexpr int PT = FPS_NCELL / 1024;
    __shared__ int s_w[16];
    int* h = hist + (int64_t)blockIdx.x * FPS_NCELL + threadIdx.x * PT;
    int v[PT], sum = 0;
#pragma unroll
    for (int i = 0; i < PT; ++i) { v[i] = h[i]; sum += v[i]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int off = inc - sum;
    for (int w = 0; w < wave; ++w) off += s_w[w];
#pragma unroll
    for (int i = 0; i < PT; ++i) { h[i] = off; off += v[i]; }
}
// psoa [B][4][npad]: x, y, z planes of the bucketed cloud and the points' original indices (as int bits); positions >= N: zeros / -1
__global__ __launch_bounds__(256) void fps_cell_scatter_kernel(const float* __restrict__ xyz, int N, int64_t npad, const int* __restrict__ bbox, int* __restrict__ offs,
                                                               float* __restrict__ psoa) {
    const int b = blockIdx.y;
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= npad) return;
    float* base = psoa + (int64_t)b * 4 * npad;
    if (n >= N) { base[n] = 0.f; base[npad + n] = 0.f; base[2 * npad + n] = 0.f; base[3 * npad + n] = __builtin_bit_cast(float, -1); return; }
    const float* p = xyz + ((int64_t)b * N + n) * 3;
    const float x = p[0], y = p[1], z = p[2];
    const int pos = atomicAdd(offs + (int64_t)b * FPS_NCELL + fps_cell_code(x, y, z, bbox + b * 8), 1);
    base[pos] = x; base[npad + pos] = y; base[2 * npad + pos] = z; base[3 * npad + pos] = __builtin_bit_cast(float, (int)n);
}

// (Tried and dropped, round 5: publishing the candidate's COORDINATES with the key -- four self-validating 64-bit words per slot -- so that nobody fetches
// the winner's point after the exchange: exact, but four polled loads and four stores per hand-over cost more than the three scalar loads they replace,
// cfg #3 FPS 4.18 -> 4.91 ms, one cloud of 32768 points 0.97 -> 1.12 ms; profiles/r05/r05_fps_pruned.txt.)
template <int PPT4>
__global__ __launch_bounds__(FPS_THREADS) void fps_coop_pruned_kernel(const float* __restrict__ xyz, const float* __restrict__ psoa, int N, int64_t npad,
                                                                      int G, int W, int xcd_stride, unsigned long long* __restrict__ cand,
                                                                      int64_t* __restrict__ idx_out, float* __restrict__ centers_out) {
    const int b = blockIdx.y;
    int w = blockIdx.x;
    if (xcd_stride > 1) {      // one XCD per cloud, see fps_coop_kernel
        if ((int)(blockIdx.x & 7) != (b & 7)) return;
        w = blockIdx.x >> 3;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* P = xyz + (int64_t)b * N * 3;
    const float* soa_b = psoa + (int64_t)b * 4 * npad;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)soa_b, 0, (int)(4 * npad * 4), 0x00020000);
    const int plane_bytes = (int)(npad * 4);
    __shared__ unsigned long long s_key[2][FPS_WAVES];
    __shared__ int s_last[2];
    unsigned long long* cand_b = cand + (int64_t)b * 2 * 64;

    // this wave's points: positions ((w FPS_WAVES + wave) PPT4 + g) 256 + 4 lane .. + 3 of the bucketed order -- PPT4 * 256 consecutive positions
    f32x4 md[PPT4], rx[PPT4], ry[PPT4], rz[PPT4];
    u32x4 oi[PPT4];
    float blx = INFINITY, bly = INFINITY, blz = INFINITY, bhx = -INFINITY, bhy = -INFINITY, bhz = -INFINITY;
#pragma unroll
    for (int g = 0; g < PPT4; ++g) {
        const int base = (((w * FPS_WAVES + wave) * PPT4 + g) * 64 + lane) * 4;
        rx[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, base * 4, 0, 0));
        ry[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, base * 4, plane_bytes, 0));
        rz[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, base * 4, 2 * plane_bytes, 0));
        oi[g] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, base * 4, 3 * plane_bytes, 0));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = base + e < N;
            md[g][e] = ok ? INFINITY : -1.0f;
            if (ok) {
                blx = fminf(blx, rx[g][e]); bhx = fmaxf(bhx, rx[g][e]);
                bly = fminf(bly, ry[g][e]); bhy = fmaxf(bhy, ry[g][e]);
                blz = fminf(blz, rz[g][e]); bhz = fmaxf(bhz, rz[g][e]);
            }
        }
    }
    blx = -wave_max(-blx); bly = -wave_max(-bly); blz = -wave_max(-blz);      // an empty wave keeps lo = +inf, hi = -inf: its box distance is +inf, it never scans
    bhx = wave_max(bhx); bhy = wave_max(bhy); bhz = wave_max(bhz);
    float c_wmax = wave_max(md[0][0]) >= 0.f ? INFINITY : -1.0f;      // (position order: a wave with any valid point has a valid lane 0, slot 0)
    unsigned c_idx = 0;
    int last = 0;
    if (w == 0) {
        if (tid == 0) idx_out[(int64_t)b * G] = 0;
        if (tid < 3) centers_out[(int64_t)b * G * 3 + tid] = P[tid];
    }
    for (int j = 1; j < G; ++j) {
        const float cx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, P[(int64_t)last * 3 + 0])));
        const float cy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, P[(int64_t)last * 3 + 1])));
        const float cz = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, P[(int64_t)last * 3 + 2])));
        asm volatile("s_nop 4");      // SGPRs possibly written by the VALU feed packed-fp32 VALU operands (inline asm) right below
        // the centre's distance to the wave's box, in the scan's arithmetic (a - c and -(c - a) are the same bits; contraction is off in this file)
        const float ax = fmaxf(fmaxf(blx - cx, cx - bhx), 0.f), ay = fmaxf(fmaxf(bly - cy, cy - bhy), 0.f), az = fmaxf(fmaxf(blz - cz, cz - bhz), 0.f);
        const float dbox = (ax * ax + ay * ay) + az * az;
        const unsigned tag = (unsigned)j & 0xFFFu;
        if (!(dbox >= c_wmax)) {      // wave-uniform.  (c_wmax = -1: a wave of padding never scans; +inf in the first iteration: everybody does)
            float best = -1.0f;
            const fps_f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
            for (int g = 0; g < PPT4; ++g) {
                const f32x4 dx = fps_sub_bcast(rx[g], c2x), dy = fps_sub_bcast(ry[g], c2y), dz = fps_sub_bcast(rz[g], c2z);
                const f32x4 d = (dx * dx + dy * dy) + dz * dz;      // -ffp-contract=off: no FMA
                md[g] = f32x4{fps_min(md[g].x, d.x), fps_min(md[g].y, d.y), fps_min(md[g].z, d.z), fps_min(md[g].w, d.w)};
                best = fmaxf(fmaxf(best, md[g].x), md[g].y);
                best = fmaxf(fmaxf(best, md[g].z), md[g].w);
            }
            c_wmax = wave_max(best);
            unsigned cnd = 0x7fffffffu;      // the lowest ORIGINAL index among the points that hold the maximum
            if (best == c_wmax) {
#pragma unroll
                for (int g = 0; g < PPT4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (md[g][e] == c_wmax && oi[g][e] < cnd) cnd = oi[g][e];
            }
            c_idx = (unsigned)wave_min_dpp((int)cnd);
        }
        // key: [63:32] min-distance bits, [31:20] iteration tag, [19:0] 0xFFFFF - original index (fps_coop_kernel); a wave of padding: distance 0, index field 0
        unsigned long long key = (unsigned long long)tag << 20;
        if (c_wmax >= 0.f) key = ((unsigned long long)__builtin_bit_cast(unsigned, c_wmax) << 32) | (tag << 20) | (0xFFFFFu - c_idx);
        const int slot = j & 1;
        if (lane == 0) s_key[slot][wave] = key;
        __syncthreads();
        if (wave == 0) {
            static_assert(FPS_WAVES == 16, "one 16-lane row holds the waves' keys");
            const unsigned long long wk = fps_row16_max_u64(s_key[slot][lane & 15]);
            if (lane == 0) __hip_atomic_store(cand_b + slot * 64 + w, wk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long k;
            for (;;) {
                k = lane < W ? __hip_atomic_load(cand_b + slot * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 20);
                if (__all((((unsigned)k >> 20) & 0xFFFu) == tag)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            k = W <= 16 ? fps_row16_max_u64(k) : fps_wave_max_u64(k);
            if (lane == 0) s_last[slot] = (int)(0xFFFFFu - ((unsigned)k & 0xFFFFFu));
        }
        __syncthreads();
        last = __builtin_amdgcn_readfirstlane(s_last[slot]);
        if (w == 0) {
            if (tid == 0) idx_out[(int64_t)b * G + j] = last;
            if (tid < 3) centers_out[((int64_t)b * G + j) * 3 + tid] = P[(int64_t)last * 3 + tid];
        }
    }
}

static int fps_num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    return cus;
}

// cooperative layout: groups of 4096 points, PPT4 of them per workgroup, W <= 64 workgroups per cloud, all B*W resident.  The hand-over
// gets slower with W (1.0 us per iteration up to 16 workgroups on one XCD, 1.7 us at 32: profiles/r03/r03_fabric_probe.txt), the scan with
// PPT4 (16 waves share four SIMDs).  Measured per iteration (profiles/r03/r03_fps_coop.txt, N = 131072): W = 16 x 8 points per thread 1.98 us,
// W = 8 x 16 points 2.57 us, W = 32 x 4 points 2.98 us; N = 65536: W = 16 x 4 points 1.64 us, W = 8 x 8 points 1.85 us.  So: the fewest
// points per thread that bring W down to 16, else as few workgroups as the registers allow (PPT4 <= 4).
static int fps_coop_ppt4(int B, int N, int* W) {
    static int ppt4_max = 0, min_groups = 0;      // tuning hooks (environment, read once)
    if (!ppt4_max) {
        const char* e = getenv("PSAM_FPS_COOP_PPT4"); ppt4_max = e && atoi(e) > 0 ? atoi(e) : 4;
        const char* f = getenv("PSAM_FPS_COOP_MIN_GROUPS"); min_groups = f && atoi(f) > 0 ? atoi(f) : 8;
    }
    const int64_t groups = fps_npad(N) / (4 * FPS_THREADS);
    if (groups < min_groups || N > (1 << 20) || B > 1024) return 0;      // the hand-over key holds a 20-bit index
    if (groups == 8) {
        // N = 32768 (cfg #2 / #5): the single-workgroup kernel scans 32 points per thread (2.6 us per iteration).  Eight workgroups of 4 points
        // per thread: 1.72 us (0.88 instead of 1.3 ms for G = 512) -- but eight CUs per cloud instead of one, which a batch running beside the
        // dense stage of the previous one cannot afford (two workgroups of 16 points: 2.55 us, no gain; profiles/r03/r03_fps_coop.txt).  So
        // only for one or two clouds (the interactive case).
        if (B > 2 || ppt4_max < 1) return 0;
        *W = 8;
        return 1;
    }
    int best = 0;
    for (int ppt4 = 1; ppt4 <= 4 && ppt4 <= ppt4_max; ppt4 *= 2) {
        const int64_t w = groups / ppt4;
        if (groups % ppt4 != 0 || w > 64 || (int64_t)B * w > fps_num_cus()) continue;
        best = ppt4; *W = (int)w;
        if (w <= 16) break;
    }
    return best;
}

PSAM_API size_t psam_fps_workspace_bytes(int32_t B, int32_t N, int32_t G) {
    (void)G;
    if (B <= 0 || N <= 0) return 0;
    // planar xyz (3) + streamed min-distance (1) + cooperative hand-over: 2 x 64 candidate slots and one barrier counter per cloud
    // + the pruned cooperative kernel's counting sort: per cloud a bounding box (8 ints) and FPS_NCELL cell counters
    return (size_t)B * 4 * (size_t)fps_npad(N) * sizeof(float) + (size_t)B * (2 * 64 * sizeof(unsigned long long) + 16) + (size_t)B * (FPS_NCELL + 8) * sizeof(int);
}

static int g_fps_coop = 1;  // test hook: 0 forces the single-workgroup kernels, 2 the cooperative kernel without the one-XCD placement
PSAM_API void psam_fps_set_cooperative(int32_t on) { g_fps_coop = on; }
static int g_fps_prune = -1;      // -1: environment PSAM_FPS_PRUNE (default 1); 0 = the cooperative kernel scans every point in every iteration (A/B, tests)
PSAM_API void psam_fps_set_pruning(int32_t mode) { g_fps_prune = mode; }
static bool fps_prune_enabled() {
    if (g_fps_prune >= 0) return g_fps_prune != 0;
    static int on = -1;
    if (on < 0) { const char* e = getenv("PSAM_FPS_PRUNE"); on = e ? (atoi(e) != 0) : 1; }
    return on != 0;
}

// xyz [B,N,3] f32 -> fps_idx [B,G] i64 (start index 0), centers [B,G,3] f32 (fused batch_index_select).
PSAM_API int32_t psam_fps(const float* xyz, int32_t B, int32_t N, int32_t G, int64_t* fps_idx, float* centers, void* ws,
                          size_t ws_bytes, hipStream_t stream) {
    PSAM_REQUIRE(xyz && fps_idx && centers && ws, PSAM_EINVAL, "psam_fps: null pointer");
    PSAM_REQUIRE(B > 0 && N > 0 && G > 0 && G <= N, PSAM_EINVAL, "psam_fps: need B>0, 0<G<=N");
    PSAM_REQUIRE((int64_t)N <= (int64_t)1 << 30, PSAM_EINVAL, "psam_fps: N too large");
    PSAM_REQUIRE(ws_bytes >= psam_fps_workspace_bytes(B, N, G), PSAM_EWORKSPACE, "psam_fps: workspace too small");
    PSAM_REQUIRE(((uintptr_t)ws & 15) == 0, PSAM_EALIGN, "psam_fps: workspace must be 16-byte aligned");
    const int64_t npad = fps_npad(N);
    float* soa = (float*)ws;
    float* mdg = soa + (int64_t)B * 3 * npad;
    int W = 0;
    const int coop = g_fps_coop ? fps_coop_ppt4(B, N, &W) : 0;
    if (!(coop && fps_prune_enabled()))      // (the pruned cooperative kernel builds its own, bucketed planes)
        hipLaunchKernelGGL(fps_soa_kernel, dim3((unsigned)psam_cdiv(npad, 256), B), dim3(256), 0, stream, xyz, N, npad, soa);
    if (coop) {
        unsigned long long* cand = (unsigned long long*)(mdg + (int64_t)B * npad);
        hipLaunchKernelGGL(fps_coop_reset_kernel, dim3((unsigned)psam_cdiv(B * 128, 256)), dim3(256), 0, stream, cand, B * 128);
        // one XCD (32 CUs) per cloud when its workgroups fit beside those of the other clouds dealt to the same XCD
        const int xs = ((int64_t)W * psam_cdiv(B, 8) <= fps_num_cus() / 8 && g_fps_coop != 2) ? 8 : 1;
        if (fps_prune_enabled()) {
            // counting sort of every cloud by grid cell (bounding box, histogram, scan, scatter) into the four planes of the workspace (x, y, z,
            // original index: the cooperative kernels keep the running minima in registers, the fourth plane is free), then the pruned kernel
            int* bbox = (int*)(cand + (int64_t)B * 128 + 2 * (int64_t)B);
            int* hist = bbox + (int64_t)B * 8;
            float* psoa = soa;
            const unsigned nb = (unsigned)psam_cdiv(N, 256);
            hipLaunchKernelGGL(fps_prune_reset_kernel, dim3((unsigned)psam_cdiv((int64_t)B * FPS_NCELL, 256)), dim3(256), 0, stream, bbox, hist, B);
            hipLaunchKernelGGL(fps_bbox_kernel, dim3(nb < 256 ? nb : 256, B), dim3(256), 0, stream, xyz, N, bbox);
            hipLaunchKernelGGL(fps_cell_hist_kernel, dim3(nb, B), dim3(256), 0, stream, xyz, N, bbox, hist);
            hipLaunchKernelGGL(fps_cell_scan_kernel, dim3(B), dim3(1024), 0, stream, hist);
            hipLaunchKernelGGL(fps_cell_scatter_kernel, dim3((unsigned)psam_cdiv(npad, 256), B), dim3(256), 0, stream, xyz, N, npad, bbox, hist, psoa);
#define FPS_COOP_PRUNED(P) \
    hipLaunchKernelGGL(fps_coop_pruned_kernel<P>, dim3(W * xs, B), dim3(FPS_THREADS), 0, stream, xyz, psoa, N, npad, G, W, xs, cand, fps_idx, centers)
            if (coop == 1) FPS_COOP_PRUNED(1); else if (coop == 2) FPS_COOP_PRUNED(2); else FPS_COOP_PRUNED(4);
#undef FPS_COOP_PRUNED
            return psam_launch_status("psam_fps: launch failed");
        }
#define FPS_COOP(P) \
    hipLaunchKernelGGL(fps_coop_kernel<P>, dim3(W * xs, B), dim3(FPS_THREADS), 0, stream, xyz, soa, N, npad, G, W, xs, cand, fps_idx, centers)
        if (coop == 1) FPS_COOP(1); else if (coop == 2) FPS_COOP(2); else FPS_COOP(4);
#undef FPS_COOP
        return psam_launch_status("psam_fps: launch failed");
    }
    const int groups = (int)(npad / (4 * FPS_THREADS));
#define FPS_LAUNCH(P) \
    hipLaunchKernelGGL(fps_kernel<P>, dim3(B), dim3(FPS_THREADS), 0, stream, xyz, soa, mdg, N, npad, G, fps_idx, centers)
    switch (groups) {
        case 1: FPS_LAUNCH(1); break;
        case 2: FPS_LAUNCH(2); break;
        case 3: FPS_LAUNCH(3); break;
        case 4: FPS_LAUNCH(4); break;
        case 5: FPS_LAUNCH(5); break;
        case 6: FPS_LAUNCH(6); break;
        case 7: FPS_LAUNCH(7); break;
        case 8: FPS_LAUNCH(8); break;
        default: FPS_LAUNCH(0); break;
    }
#undef FPS_LAUNCH
    return psam_launch_status("psam_fps: launch failed");
}

// ------------------------------------------------------------------------------------------------
// kNN: for each center the K nearest points, ascending by (d2, index).  One 256-thread workgroup per
// center: 3-pass radix select on the fp32 bit pattern of d2 (11+11+10 bits) finds the exact K-th smallest
// value, a 4th pass collects, a bitonic sort orders.  The distance matrix is never materialised.
// ------------------------------------------------------------------------------------------------
constexpr int KNN_THREADS = 256;
constexpr int KNN_MAXK = 1024;

__device__ __forceinline__ int block_excl_scan_256(int v, int* s_wave, int& total) {
    // exclusive scan over 256 threads (4 waves); s_wave: 4 ints of LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int off = 0;
    total = 0;
#pragma unroll
    for (int w = 0; w < KNN_THREADS / 64; ++w) {
        const int t = s_wave[w];
        if (w < wave) off += t;
        total += t;
    }
    __syncthreads();
    return off + inc - v;
}

// The whole selection for one center with every pass over the whole cloud (four distance evaluations per point): the general path -- any band size, ties
// at the K-th value cut by index.  LDS arrays are the caller's (hist[2048], keys[KNN_MAXK], s_wave[4], sm[4] = bin / below / count-less / count-equal).
__device__ __forceinline__ void knn_select_full(const float* __restrict__ P, float cx, float cy, float cz, int N, int K, int64_t* __restrict__ out, unsigned* hist,
                                                unsigned long long* keys, int* s_wave, int* sm) {
    const int tid = threadIdx.x;
    int& s_bin = sm[0]; int& s_below = sm[1]; int& s_cnt_less = sm[2]; int& s_cnt_eq = sm[3];

    unsigned prefix_mask = 0, prefix_val = 0;
    int remaining = K;  // rank (1-based) of the K-th smallest inside the current candidate set
    int eq_total = 0;
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
        const int nb = pass == 2 ? 1024 : 2048;
        for (int i = tid; i < 2048; i += KNN_THREADS) hist[i] = 0;
        __syncthreads();
        for (int n = tid; n < N; n += KNN_THREADS) {
            const float d = dist2_exact(cx, cy, cz, P[n * 3 + 0], P[n * 3 + 1], P[n * 3 + 2]);
            const unsigned u = __float_as_uint(d);
            if ((u & prefix_mask) == prefix_val) atomicAdd(&hist[(u >> shift) & (nb - 1)], 1u);
        }
        __syncthreads();
        // locate the bin that holds rank `remaining`
        int loc[8], sum = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { loc[i] = (int)hist[tid * 8 + i]; sum += loc[i]; }
        int total;
        int before = block_excl_scan_256(sum, s_wave, total);
        if (remaining > before && remaining <= before + sum) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (remaining > before && remaining <= before + loc[i]) { s_bin = tid * 8 + i; s_below = before; }
                before += loc[i];
            }
        }
        __syncthreads();
        const int bin = s_bin;
        remaining -= s_below;
        eq_total = (int)hist[bin];
        prefix_val |= (unsigned)bin << shift;
        prefix_mask |= (unsigned)(nb - 1) << shift;
        __syncthreads();
    }
    const unsigned T = prefix_val;     // bit pattern of the K-th smallest d2
    const int need_eq = remaining;     // how many of the points with d2 == T belong to the answer (lowest indices)
    const int n_less = K - need_eq;
    if (tid == 0) { s_cnt_less = 0; s_cnt_eq = 0; }
    __syncthreads();
    if (eq_total == need_eq) {
        // no truncation among ties: unordered collection
        for (int n = tid; n < N; n += KNN_THREADS) {
            const float d = dist2_exact(cx, cy, cz, P[n * 3 + 0], P[n * 3 + 1], P[n * 3 + 2]);
            const unsigned u = __float_as_uint(d);
            if (u < T) {
                const int p = atomicAdd(&s_cnt_less, 1);
                keys[p] = ((unsigned long long)u << 32) | (unsigned)n;
            } else if (u == T) {
                const int p = atomicAdd(&s_cnt_eq, 1);
                keys[n_less + p] = ((unsigned long long)u << 32) | (unsigned)n;
            }
        }
    } else {
        // ties at the K-th distance must be cut: take the lowest indices -> index-ordered sweep
        int eq_taken = 0;
        for (int n0 = 0; n0 < N; n0 += KNN_THREADS) {
            const int n = n0 + tid;
            unsigned u = 0xffffffffu;
            if (n < N) u = __float_as_uint(dist2_exact(cx, cy, cz, P[n * 3 + 0], P[n * 3 + 1], P[n * 3 + 2]));
            if (u < T) {
                const int p = atomicAdd(&s_cnt_less, 1);
                keys[p] = ((unsigned long long)u << 32) | (unsigned)n;
            }
            int total;
            const int is_eq = (u == T) ? 1 : 0;
            const int rank = block_excl_scan_256(is_eq, s_wave, total);
            if (is_eq && eq_taken + rank < need_eq) keys[n_less + eq_taken + rank] = ((unsigned long long)u << 32) | (unsigned)n;
            eq_taken += total;
        }
    }
    // bitonic sort of K keys padded to a power of two
    int P2 = 1;
    while (P2 < K) P2 <<= 1;
    __syncthreads();
    for (int i = K + tid; i < P2; i += KNN_THREADS) keys[i] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= P2; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int i = tid; i < P2; i += KNN_THREADS) {
                const int ixj = i ^ jj;
                if (ixj > i) {
                    const unsigned long long a = keys[i], bb = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > bb) == up) { keys[i] = bb; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < K; i += KNN_THREADS) out[i] = (int64_t)(keys[i] & 0xffffffffull);
}

__global__ __launch_bounds__(KNN_THREADS) void knn_kernel(const float* __restrict__ centers, const float* __restrict__ xyz,
                                                          int G, int N, int K, int64_t* __restrict__ out_idx) {
    const int g = blockIdx.x, b = blockIdx.y;
    const float* c = centers + ((int64_t)b * G + g) * 3;
    __shared__ unsigned hist[2048];
    __shared__ unsigned long long keys[KNN_MAXK];
    __shared__ int s_wave[4];
    __shared__ int sm[4];
    knn_select_full(xyz + (int64_t)b * N * 3, c[0], c[1], c[2], N, K, out_idx + ((int64_t)b * G + g) * K, hist, keys, s_wave, sm);
}

// Round 5: the same selection with TWO distance evaluations per (center, point) pair instead of four.
// knn_kernel above evaluates every distance in each of its three radix passes and once more to collect.  But after the first pass (the top 11 bits of the
// fp32 pattern of d2) the answer is known up to a BAND: every point whose top bits lie below the selected bin belongs to it (fewer than K of them), no point
// above does, and the K-th distance lies among the bin's own points -- typically N / 2048 .. a few hundred.  So: sweep 1 builds the histogram; sweep 2
// appends the points below the bin to the answer and the bin's points (pattern, index) to a candidate list in LDS; the two remaining radix passes, the
// cut at the K-th value and the collection run on that list.  (Keeping a thread's sweep-1 distances in registers for sweep 2 -- one evaluation per pair --
// needs 128 registers at N = 32768 and left one workgroup per CU: slower than recomputing.)  Points are read four at a time as three 16-byte loads.
// Same comparisons on the same fp32 patterns, same final bitonic sort by (d2, index): bit-identical output.
// A band that overflows the list (more than KNN_CAND points share the top 11 bits: clouds of duplicates) or a tie at the K-th value that has to be cut by
// index sends the WORKGROUP (uniformly) through knn_select_full: the general path, same LDS, same result.
constexpr int KNN_CAND = 2048;      // candidate capacity (16 KiB of LDS: a band holds roughly 0.2 K .. 0.4 K points of a uniform cloud)
typedef float knn_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(KNN_THREADS) void knn_band_kernel(const float* __restrict__ centers, const float* __restrict__ xyz, int G, int N, int K,
                                                               int64_t* __restrict__ out_idx) {
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* P = xyz + (int64_t)b * N * 3;
    const float* c = centers + ((int64_t)b * G + g) * 3;
    const float cx = c[0], cy = c[1], cz = c[2];
    __shared__ unsigned hist[2048];
    __shared__ unsigned long long keys[KNN_MAXK];
    __shared__ unsigned cand_u[KNN_CAND];
    __shared__ int cand_n[KNN_CAND];
    __shared__ int s_wave[4];
    __shared__ int sm[4];
    __shared__ int s_cnt_cand;
    int& s_bin = sm[0]; int& s_below = sm[1]; int& s_cnt_less = sm[2];
    int64_t* const out = out_idx + ((int64_t)b * G + g) * K;

    // points in groups of four (48 contiguous bytes = three 16-byte loads when the cloud is 16-byte aligned); group q = i * 256 + tid
    const int ngroups = (N + 3) >> 2;
    const bool vec = (((uintptr_t)P) & 15) == 0;
    auto dist4 = [&](int q, unsigned (&u)[4]) {
        const int n0 = q * 4;
        if (vec && n0 + 4 <= N) {
            const knn_f32x4* p4 = reinterpret_cast<const knn_f32x4*>(P + (int64_t)n0 * 3);
            const knn_f32x4 a = p4[0], bb = p4[1], cc = p4[2];
            u[0] = __float_as_uint(dist2_exact(cx, cy, cz, a[0], a[1], a[2]));
            u[1] = __float_as_uint(dist2_exact(cx, cy, cz, a[3], bb[0], bb[1]));
            u[2] = __float_as_uint(dist2_exact(cx, cy, cz, bb[2], bb[3], cc[0]));
            u[3] = __float_as_uint(dist2_exact(cx, cy, cz, cc[1], cc[2], cc[3]));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + e;
                u[e] = n < N ? __float_as_uint(dist2_exact(cx, cy, cz, P[n * 3 + 0], P[n * 3 + 1], P[n * 3 + 2])) : 0xffffffffu;      // past the end: never selected (d2 patterns are < 0x7f800000)
            }
        }
    };
    // ---- sweep 1: histogram of the top 11 bits
    for (int i = tid; i < 2048; i += KNN_THREADS) hist[i] = 0;
    if (tid == 0) { s_cnt_less = 0; s_cnt_cand = 0; }
    __syncthreads();
    for (int q = tid; q < ngroups; q += KNN_THREADS) {
        unsigned u[4];
        dist4(q, u);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (u[e] != 0xffffffffu) atomicAdd(&hist[u[e] >> 21], 1u);
    }
    __syncthreads();
    int remaining = K;
    {
        int loc[8], sum = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { loc[i] = (int)hist[tid * 8 + i]; sum += loc[i]; }
        int total;
        int before = block_excl_scan_256(sum, s_wave, total);
        if (remaining > before && remaining <= before + sum) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (remaining > before && remaining <= before + loc[i]) { s_bin = tid * 8 + i; s_below = before; }
                before += loc[i];
            }
        }
        __syncthreads();
    }
    const unsigned bin0 = (unsigned)s_bin;
    const int n_band = (int)hist[bin0];
    remaining -= s_below;      // rank of the K-th smallest inside the band
    if (n_band > KNN_CAND) {      // uniform: the band does not fit the list
        __syncthreads();
        knn_select_full(P, cx, cy, cz, N, K, out, hist, keys, s_wave, sm);
        return;
    }
    // ---- sweep 2: below the band -> the answer; in the band -> the candidate list
    auto place = [&](unsigned u, int n) {
        const unsigned top = u >> 21;
        if (top < bin0) { const int p = atomicAdd(&s_cnt_less, 1); keys[p] = ((unsigned long long)u << 32) | (unsigned)n; }
        else if (top == bin0) { const int p = atomicAdd(&s_cnt_cand, 1); cand_u[p] = u; cand_n[p] = n; }
    };
    for (int q = tid; q < ngroups; q += KNN_THREADS) {
        unsigned u[4];
        dist4(q, u);
#pragma unroll
        for (int e = 0; e < 4; ++e) place(u[e], q * 4 + e);
    }
    __syncthreads();
    // ---- the two lower radix passes on the candidates
    unsigned prefix_mask = 0x7ffu << 21, prefix_val = bin0 << 21;
    int eq_total = n_band;
    for (int pass = 1; pass < 3; ++pass) {
        const int shift = pass == 1 ? 10 : 0;
        const int nb = pass == 2 ? 1024 : 2048;
        for (int i = tid; i < 2048; i += KNN_THREADS) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < n_band; i += KNN_THREADS) {
            const unsigned u = cand_u[i];
            if ((u & prefix_mask) == prefix_val) atomicAdd(&hist[(u >> shift) & (nb - 1)], 1u);
        }
        __syncthreads();
        int loc[8], sum = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { loc[i] = (int)hist[tid * 8 + i]; sum += loc[i]; }
        int total;
        int before = block_excl_scan_256(sum, s_wave, total);
        if (remaining > before && remaining <= before + sum) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (remaining > before && remaining <= before + loc[i]) { s_bin = tid * 8 + i; s_below = before; }
                before += loc[i];
            }
        }
        __syncthreads();
        const int bin = s_bin;
        remaining -= s_below;
        eq_total = (int)hist[bin];
        prefix_val |= (unsigned)bin << shift;
        prefix_mask |= (unsigned)(nb - 1) << shift;
        __syncthreads();
    }
    const unsigned T = prefix_val;     // bit pattern of the K-th smallest d2
    const int need_eq = remaining;     // how many of the points with d2 == T belong to the answer (lowest indices)
    if (eq_total != need_eq) {         // uniform: a tie at the K-th value must be cut by index
        __syncthreads();
        knn_select_full(P, cx, cy, cz, N, K, out, hist, keys, s_wave, sm);
        return;
    }
    for (int i = tid; i < n_band; i += KNN_THREADS) {
        const unsigned u = cand_u[i];
        if (u <= T) { const int p = atomicAdd(&s_cnt_less, 1); keys[p] = ((unsigned long long)u << 32) | (unsigned)cand_n[i]; }
    }
    // ---- bitonic sort of K keys padded to a power of two (as knn_kernel)
    int P2 = 1;
    while (P2 < K) P2 <<= 1;
    __syncthreads();
    for (int i = K + tid; i < P2; i += KNN_THREADS) keys[i] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= P2; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int i = tid; i < P2; i += KNN_THREADS) {
                const int ixj = i ^ jj;
                if (ixj > i) {
                    const unsigned long long a = keys[i], bb = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > bb) == up) { keys[i] = bb; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < K; i += KNN_THREADS) out[i] = (int64_t)(keys[i] & 0xffffffffull);
}

static int g_knn_band = -1;      // -1: environment PSAM_KNN_BAND (default 1); 0 = always the four-pass kernel (A/B, tests)
PSAM_API void psam_knn_force_band(int32_t mode) { g_knn_band = mode; }

// centers [B,G,3], xyz [B,N,3] -> knn_idx [B,G,K] i64 ascending by (squared distance, index).
PSAM_API int32_t psam_knn(const float* centers, const float* xyz, int32_t B, int32_t G, int32_t N, int32_t K, int64_t* knn_idx,
                          hipStream_t stream) {
    PSAM_REQUIRE(centers && xyz && knn_idx, PSAM_EINVAL, "psam_knn: null pointer");
    PSAM_REQUIRE(B > 0 && G > 0 && N > 0 && K > 0 && K <= N, PSAM_EINVAL, "psam_knn: need 0<K<=N");
    PSAM_REQUIRE(K <= KNN_MAXK, PSAM_EINVAL, "psam_knn: K > 1024 unsupported");
    PSAM_REQUIRE(B <= 65535, PSAM_EINVAL, "psam_knn: B > 65535 unsupported");
    int band = g_knn_band;
    if (band < 0) {
        static int env = -1;
        if (env < 0) { const char* e = getenv("PSAM_KNN_BAND"); env = e ? atoi(e) : 1; }
        band = env;
    }
    if (!band) hipLaunchKernelGGL(knn_kernel, dim3(G, B), dim3(KNN_THREADS), 0, stream, centers, xyz, G, N, K, knn_idx);
    else hipLaunchKernelGGL(knn_band_kernel, dim3(G, B), dim3(KNN_THREADS), 0, stream, centers, xyz, G, N, K, knn_idx);
    return psam_launch_status("psam_knn: launch failed");
}

// ------------------------------------------------------------------------------------------------
// 3-NN of every point among the G centers + inverse-squared-distance weights (common.py:238-255).
// Centers are staged in LDS (broadcast reads); one thread per point.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void three_nn_kernel(const float* __restrict__ xyz, const float* __restrict__ centers, int N, int G,
                                                       float eps, int64_t* __restrict__ idx3, float* __restrict__ w3) {
    extern __shared__ __attribute__((aligned(16))) float s_c[];
    const int b = blockIdx.y;
    const float* C = centers + (int64_t)b * G * 3;
    for (int i = threadIdx.x; i < G * 3; i += blockDim.x) s_c[i] = C[i];
    __syncthreads();
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* p = xyz + ((int64_t)b * N + n) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    float d0 = INFINITY, d1 = INFINITY, d2 = INFINITY;
    int i0 = -1, i1 = -1, i2 = -1;
    for (int g = 0; g < G; ++g) {
        const float d = dist2_exact(px, py, pz, s_c[g * 3 + 0], s_c[g * 3 + 1], s_c[g * 3 + 2]);
        if (d < d2) {  // strict: equal distance keeps the earlier (lower) center index
            if (d < d1) {
                d2 = d1; i2 = i1;
                if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = g; }
                else { d1 = d; i1 = g; }
            } else { d2 = d; i2 = g; }
        }
    }
    // dist = sqrt(d2); w = 1/max(dist*dist, eps), normalised: correctly rounded sqrt and divide (hipcc default),
    // no contraction (translation unit is built with -ffp-contract=off) => bit-identical to the C oracle.
    float s0 = sqrtf(d0), s1 = sqrtf(d1), s2 = sqrtf(d2);
    s0 = s0 * s0; s1 = s1 * s1; s2 = s2 * s2;
    const float v0 = 1.0f / fmaxf(s0, eps), v1 = 1.0f / fmaxf(s1, eps), v2 = 1.0f / fmaxf(s2, eps);
    float s = v0 + v1;
    s = s + v2;
    const int64_t o = ((int64_t)b * N + n) * 3;
    idx3[o + 0] = i0; idx3[o + 1] = i1; idx3[o + 2] = i2;
    w3[o + 0] = v0 / s; w3[o + 1] = v1 / s; w3[o + 2] = v2 / s;
}

PSAM_API int32_t psam_three_nn(const float* xyz, const float* centers, int32_t B, int32_t N, int32_t G, float eps, int64_t* idx3,
                               float* w3, hipStream_t stream) {
    PSAM_REQUIRE(xyz && centers && idx3 && w3, PSAM_EINVAL, "psam_three_nn: null pointer");
    PSAM_REQUIRE(B > 0 && N > 0 && G >= 3, PSAM_EINVAL, "psam_three_nn: need G>=3");
    PSAM_REQUIRE((size_t)G * 12 <= 144 * 1024, PSAM_EINVAL, "psam_three_nn: G too large for LDS staging");
    PSAM_REQUIRE(B <= 65535, PSAM_EINVAL, "psam_three_nn: B > 65535 unsupported");
    hipLaunchKernelGGL(three_nn_kernel, dim3((unsigned)psam_cdiv(N, 256), B), dim3(256), (size_t)G * 12, stream, xyz, centers, N, G,
                       eps, idx3, w3);
    return psam_launch_status("psam_three_nn: launch failed");
}

// ------------------------------------------------------------------------------------------------
// Neighbourhood gather: out[b,g,k,:] = [xyz[idx]-center, feats[idx]]   (common.py:99-120, radius=None)
// `feats` is [BF, N, C] with BF = B*rep (rep>1: several per-cloud feature sets share one knn_idx,
// as group_with_centers_and_knn does for mask logits, common.py:126-187).
// ------------------------------------------------------------------------------------------------
__global__ void group_gather_kernel(const float* __restrict__ xyz, const float* __restrict__ feats, const float* __restrict__ centers,
                                    const int64_t* __restrict__ knn_idx, int rep, int N, int G, int K, int C, int64_t total,
                                    float inv_radius, float* __restrict__ out, int64_t ldo) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t row = t;  // (bf, g, k)
    const int k = (int)(row % K);
    const int g = (int)((row / K) % G);
    const int64_t bf = row / ((int64_t)K * G);
    const int64_t b = bf / rep;
    const int64_t n = knn_idx[(b * G + g) * K + k];
    const float* p = xyz + (b * N + n) * 3;
    const float* c = centers + (b * G + g) * 3;
    float* o = out + row * ldo;
    o[0] = (p[0] - c[0]) * inv_radius; o[1] = (p[1] - c[1]) * inv_radius; o[2] = (p[2] - c[2]) * inv_radius;   // inv_radius == 1: exact
    const float* f = feats + (bf * N + n) * C;
    for (int i = 0; i < C; ++i) o[3 + i] = f[i];
    for (int i = 3 + C; i < ldo; ++i) o[i] = 0.f;      // zero padding up to the row stride (the K of the Linear that follows must be % 4)
}

// radius > 0: relative coordinates are divided by it (KNNGrouper.radius / MaskEncoder.radius, configs/model/enc_with_radius.yaml;
// as ATen does for a scalar divisor: multiply by the fp32 reciprocal); radius <= 0: none.
// ldo: floats between output rows (>= 3 + C; the tail of a row is written as zeros)
PSAM_API int32_t psam_group_gather_ld(const float* xyz, const float* feats, const float* centers, const int64_t* knn_idx, int32_t B,
                                      int32_t rep, int32_t N, int32_t G, int32_t K, int32_t C, float radius, float* out, int64_t ldo, hipStream_t stream) {
    PSAM_REQUIRE(xyz && feats && centers && knn_idx && out, PSAM_EINVAL, "psam_group_gather: null pointer");
    PSAM_REQUIRE(B > 0 && rep > 0 && N > 0 && G > 0 && K > 0 && C > 0 && ldo >= 3 + C, PSAM_EINVAL, "psam_group_gather: bad shape");
    const int64_t total = (int64_t)B * rep * G * K;
    hipLaunchKernelGGL(group_gather_kernel, dim3((unsigned)psam_cdiv(total, 256)), dim3(256), 0, stream, xyz, feats, centers, knn_idx,
                       rep, N, G, K, C, total, radius > 0.f ? 1.0f / radius : 1.0f, out, ldo);
    return psam_launch_status("psam_group_gather: launch failed");
}

PSAM_API int32_t psam_group_gather_r(const float* xyz, const float* feats, const float* centers, const int64_t* knn_idx, int32_t B,
                                     int32_t rep, int32_t N, int32_t G, int32_t K, int32_t C, float radius, float* out, hipStream_t stream) {
    return psam_group_gather_ld(xyz, feats, centers, knn_idx, B, rep, N, G, K, C, radius, out, 3 + C, stream);
}

PSAM_API int32_t psam_group_gather(const float* xyz, const float* feats, const float* centers, const int64_t* knn_idx, int32_t B,
                                   int32_t rep, int32_t N, int32_t G, int32_t K, int32_t C, float* out, hipStream_t stream) {
    return psam_group_gather_r(xyz, feats, centers, knn_idx, B, rep, N, G, K, C, 0.f, out, stream);
}

// ------------------------------------------------------------------------------------------------
// First mini-PointNet layer fused with the gather: rows = (bf,g,k);
//   h = GELU(LayerNorm_128(W[128,3+C] @ [xyz[idx]-center, feats[idx]] + bias))       (common.py:486-489,499)
// One wave per row, two channels per lane; weights live in registers.
// ------------------------------------------------------------------------------------------------
// CIN = 3 + C (+ C more with `centralize`: the neighbour's features minus the group centre's, common.py:116-118 / 183-186).  Lane l owns
// the output channels 2l, 2l+1.  PACK: the row leaves as the g8-packed A operand of the f16x3p GEMM (csrc/gemm_f16x3p.hip) -- per 8
// channels [hi x8 | lo x8] fp16 of the row scaled by scale_out[row] (its maximum into [2^14, 2^15)) -- instead of fp32.
template <int CIN, bool CENTRAL, bool PACK>
__global__ __launch_bounds__(256) void patch_l1_kernel(const float* __restrict__ xyz, const float* __restrict__ feats,
                                                       const float* __restrict__ centers, const int64_t* __restrict__ knn_idx,
                                                       const int64_t* __restrict__ center_idx, const float* __restrict__ W,
                                                       const float* __restrict__ bias, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                       float eps, int rep, int N, int G, int K, int64_t rows, float inv_radius,
                                                       float* __restrict__ out, float* __restrict__ scale_out) {
    constexpr int C = CENTRAL ? (CIN - 3) / 2 : CIN - 3;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float w0[CIN], w1[CIN];
#pragma unroll
    for (int i = 0; i < CIN; ++i) { w0[i] = W[(2 * lane) * CIN + i]; w1[i] = W[(2 * lane + 1) * CIN + i]; }
    const float b0 = bias[2 * lane], b1 = bias[2 * lane + 1];
    const float g0 = lnw[2 * lane], g1 = lnw[2 * lane + 1], e0 = lnb[2 * lane], e1 = lnb[2 * lane + 1];
    // R rows per wave and iteration, the R dependent chains (index -> gather -> two reductions -> maximum) interleaved: the kernel's time
    // is that chain's latency times the iterations per resident wave
    constexpr int R = 2;
    for (int64_t row0 = wave * R; row0 < rows; row0 += nwaves * R) {
        float y0[R], y1[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int64_t row = row0 + q < rows ? row0 + q : rows - 1;
            const int k = (int)(row % K);
            const int g = (int)((row / K) % G);
            const int64_t bf = row / ((int64_t)K * G);
            const int64_t b = bf / rep;
            const int64_t n = knn_idx[(b * G + g) * K + k];
            const float* p = xyz + (b * N + n) * 3;
            const float* c = centers + (b * G + g) * 3;
            float in[CIN];
            in[0] = (p[0] - c[0]) * inv_radius; in[1] = (p[1] - c[1]) * inv_radius; in[2] = (p[2] - c[2]) * inv_radius;
            const float* f = feats + (bf * N + n) * C;
#pragma unroll
            for (int i = 0; i < C; ++i) in[3 + i] = f[i];
            if (CENTRAL) {
                const float* fc = feats + (bf * N + center_idx[b * G + g]) * C;
#pragma unroll
                for (int i = 0; i < C; ++i) in[3 + C + i] = f[i] - fc[i];
            }
            y0[q] = b0; y1[q] = b1;
#pragma unroll
            for (int i = 0; i < CIN; ++i) { y0[q] = fmaf(w0[i], in[i], y0[q]); y1[q] = fmaf(w1[i], in[i], y1[q]); }
        }
        float mean[R], var[R];
#pragma unroll
        for (int q = 0; q < R; ++q) mean[q] = wave_sum(y0[q] + y1[q]) * (1.0f / 128.0f);
#pragma unroll
        for (int q = 0; q < R; ++q) { y0[q] -= mean[q]; y1[q] -= mean[q]; var[q] = wave_sum(y0[q] * y0[q] + y1[q] * y1[q]) * (1.0f / 128.0f); }
        float o0[R], o1[R], sc[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const float r = 1.0f / sqrtf(var[q] + eps);
            o0[q] = gelu_erf(y0[q] * r * g0 + e0); o1[q] = gelu_erf(y1[q] * r * g1 + e1);
        }
        if (PACK) {
#pragma unroll
            for (int q = 0; q < R; ++q) sc[q] = f16_row_scale(wave_max(fmaxf(fabsf(o0[q]), fabsf(o1[q]))));
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const int64_t row = row0 + q;
            if (row >= rows) break;
            if (PACK) {
                if (lane == 0) scale_out[row] = sc[q];
                unsigned hi, lo;
                psam_split2_f16(o0[q], o1[q], sc[q], hi, lo);
                unsigned* orow = reinterpret_cast<unsigned*>(out) + row * 128 + (lane >> 2) * 8 + (lane & 3);   // group of 8 channels = 4 lanes
                orow[0] = hi;
                orow[4] = lo;
            } else {
                *reinterpret_cast<psam_f32x2*>(out + row * 128 + 2 * lane) = psam_f32x2{o0[q], o1[q]};
            }
        }
    }
}

// W [128, Cin] (nn.Linear layout), Cin = 3 + C or 3 + 2C (centralize: center_idx [B, G] = the groups' FPS indices), bias/lnw/lnb [128];
// out [B*rep*G*K, 128] fp32, or (scale_out != null) the g8-packed rows + their scales.  C in {1, 3}.
PSAM_API int32_t psam_patch_l1_ex(const float* xyz, const float* feats, const float* centers, const int64_t* knn_idx, const int64_t* center_idx,
                                  const float* W, const float* bias, const float* lnw, const float* lnb, float eps, int32_t B, int32_t rep,
                                  int32_t N, int32_t G, int32_t K, int32_t C, float radius, float* out, float* scale_out, hipStream_t stream) {
    PSAM_REQUIRE(xyz && feats && centers && knn_idx && W && bias && lnw && lnb && out, PSAM_EINVAL, "psam_patch_l1: null pointer");
    PSAM_REQUIRE(B > 0 && rep > 0 && N > 0 && G > 0 && K > 0, PSAM_EINVAL, "psam_patch_l1: bad shape");
    PSAM_REQUIRE(C == 1 || C == 3, PSAM_EINVAL, "psam_patch_l1: C must be 1 (mask logit) or 3 (rgb)");
    PSAM_REQUIRE(!scale_out || ((uintptr_t)out & 31) == 0, PSAM_EALIGN, "psam_patch_l1: packed output rows must be 32-byte aligned");
    const int64_t rows = (int64_t)B * rep * G * K;
    const int64_t blocks = rows / 4 < 8192 ? (rows + 3) / 4 : 8192;
    const float inv_r = radius > 0.f ? 1.0f / radius : 1.0f;
#define L1_LAUNCH(CIN, CEN, PK)                                                                                                          \
    hipLaunchKernelGGL((patch_l1_kernel<CIN, CEN, PK>), dim3((unsigned)blocks), dim3(256), 0, stream, xyz, feats, centers, knn_idx, center_idx, W, \
                       bias, lnw, lnb, eps, rep, N, G, K, rows, inv_r, out, scale_out)
#define L1_PICK(PK)                                                          \
    do {                                                                     \
        if (center_idx) { if (C == 3) L1_LAUNCH(9, true, PK); else L1_LAUNCH(5, true, PK); } \
        else { if (C == 3) L1_LAUNCH(6, false, PK); else L1_LAUNCH(4, false, PK); }          \
    } while (0)
    if (scale_out) L1_PICK(true); else L1_PICK(false);
#undef L1_PICK
#undef L1_LAUNCH
    return psam_launch_status("psam_patch_l1: launch failed");
}

PSAM_API int32_t psam_patch_l1_r(const float* xyz, const float* feats, const float* centers, const int64_t* knn_idx, const float* W,
                                 const float* bias, const float* lnw, const float* lnb, float eps, int32_t B, int32_t rep, int32_t N,
                                 int32_t G, int32_t K, int32_t C, float radius, float* out, hipStream_t stream) {
    return psam_patch_l1_ex(xyz, feats, centers, knn_idx, nullptr, W, bias, lnw, lnb, eps, B, rep, N, G, K, C, radius, out, nullptr, stream);
}

PSAM_API int32_t psam_patch_l1(const float* xyz, const float* feats, const float* centers, const int64_t* knn_idx, const float* W,
                               const float* bias, const float* lnw, const float* lnb, float eps, int32_t B, int32_t rep, int32_t N,
                               int32_t G, int32_t K, int32_t C, float* out, hipStream_t stream) {
    return psam_patch_l1_r(xyz, feats, centers, knn_idx, W, bias, lnw, lnb, eps, B, rep, N, G, K, C, 0.f, out, stream);
}

// ------------------------------------------------------------------------------------------------
// Click simulation of the evaluation protocol (pc_sam/model/common.py:287-316,368-474, called from pc_sam.py:139-145).
//   psam_error_regions: fn = gt & !(logit > 0), fp = !gt & (logit > 0)   (common.py:399-405; logits == null: fn = gt, fp = 0)
//   psam_border_farthest: per region, the member point whose nearest NON-member point is farthest (squared fp32
//   distance, same arithmetic as FPS; first maximum) -- sample_furthest_points_from_border, whose nearest-neighbour
//   distances come from torkit3d's chamfer_distance in the reference (absent third party).  Brute force N_fg x N_bg like
//   the reference: background points are staged through LDS in tiles, members keep a running minimum.
// ------------------------------------------------------------------------------------------------
__global__ void error_regions_kernel(const unsigned char* __restrict__ gt, const float* __restrict__ logits, unsigned char* __restrict__ fn,
                                     unsigned char* __restrict__ fp, int64_t total) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const bool g = gt[t] != 0;
    const bool p = logits ? logits[t] > 0.0f : false;
    fn[t] = (g && !p) ? 1 : 0;
    fp[t] = (!g && p) ? 1 : 0;
}

PSAM_API int32_t psam_error_regions(const uint8_t* gt, const float* logits, uint8_t* fn, uint8_t* fp, int64_t total, hipStream_t stream) {
    PSAM_REQUIRE(gt && fn && fp && total > 0, PSAM_EINVAL, "psam_error_regions: bad argument");
    hipLaunchKernelGGL(error_regions_kernel, dim3((unsigned)psam_cdiv(total, 256)), dim3(256), 0, stream, gt, logits, fn, fp, total);
    return psam_launch_status("psam_error_regions: launch failed");
}

constexpr int BF_THREADS = 256;

__global__ __launch_bounds__(BF_THREADS) void border_partial_kernel(const float* __restrict__ xyz, const unsigned char* __restrict__ region,
                                                                   int rep, int N, int nblk, float* __restrict__ pval, int* __restrict__ pidx) {
    __shared__ float sx[BF_THREADS], sy[BF_THREADS], sz[BF_THREADS];
    __shared__ float s_val[BF_THREADS / 64];
    __shared__ int s_idx[BF_THREADS / 64];
    const int z = blockIdx.y, b = z / rep, tid = threadIdx.x;
    const float* P = xyz + (int64_t)b * N * 3;
    const unsigned char* R = region + (int64_t)z * N;
    const int n = blockIdx.x * BF_THREADS + tid;
    const bool member = n < N && R[n] != 0;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (member) { px = P[n * 3]; py = P[n * 3 + 1]; pz = P[n * 3 + 2]; }
    float m = INFINITY;
    if (__syncthreads_or(member)) {
        for (int k0 = 0; k0 < N; k0 += BF_THREADS) {
            const int k = k0 + tid;
            const bool bg = k < N && R[k] == 0;
            sx[tid] = bg ? P[k * 3] : INFINITY;  // members / padding: infinite distance, never the minimum
            sy[tid] = bg ? P[k * 3 + 1] : 0.f;
            sz[tid] = bg ? P[k * 3 + 2] : 0.f;
            __syncthreads();
            if (member) {
#pragma unroll 8
                for (int j = 0; j < BF_THREADS; ++j) m = fminf(m, dist2_exact(px, py, pz, sx[j], sy[j], sz[j]));
            }
            __syncthreads();
        }
    }
    // block arg-max over members with a finite nearest-background distance; lowest index on ties
    float v = (member && m < INFINITY) ? m : -1.0f;
    int vi = (member && m < INFINITY) ? n : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(vi, o, 64);
        if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
    }
    if ((tid & 63) == 0) { s_val[tid >> 6] = v; s_idx[tid >> 6] = vi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < BF_THREADS / 64; ++w)
            if (s_val[w] > v || (s_val[w] == v && s_idx[w] < vi)) { v = s_val[w]; vi = s_idx[w]; }
        pval[(int64_t)z * nblk + blockIdx.x] = v;
        pidx[(int64_t)z * nblk + blockIdx.x] = vi;
    }
}

__global__ __launch_bounds__(256) void border_final_kernel(const float* __restrict__ pval, const int* __restrict__ pidx, int nblk,
                                                           int64_t* __restrict__ out_idx, float* __restrict__ out_dist) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    const int z = blockIdx.x, tid = threadIdx.x;
    float v = -1.0f;
    int vi = 0x7fffffff;
    for (int i = tid; i < nblk; i += 256) {
        const float ov = pval[(int64_t)z * nblk + i];
        const int oi = pidx[(int64_t)z * nblk + i];
        if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(vi, o, 64);
        if (ov > v || (ov == v && oi < vi)) { v = ov; vi = oi; }
    }
    if ((tid & 63) == 0) { s_val[tid >> 6] = v; s_idx[tid >> 6] = vi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (s_val[w] > v || (s_val[w] == v && s_idx[w] < vi)) { v = s_val[w]; vi = s_idx[w]; }
        const bool none = v < 0.0f;  // empty region or empty complement (common.py:458-460 returns dist -1)
        out_idx[z] = none ? -1 : vi;
        out_dist[z] = none ? -1.0f : v;
    }
}

PSAM_API size_t psam_border_farthest_workspace_bytes(int32_t Z, int32_t N) {
    if (Z <= 0 || N <= 0) return 0;
    return (size_t)Z * (size_t)psam_cdiv(N, BF_THREADS) * 8;
}

// xyz [B,N,3], region [Z,N] uint8 (Z = B*rep) -> out_idx [Z] int64 (-1 if none), out_dist [Z] squared distance (-1 if none)
PSAM_API int32_t psam_border_farthest(const float* xyz, const uint8_t* region, int32_t B, int32_t rep, int32_t N, int64_t* out_idx,
                                      float* out_dist, void* ws, size_t ws_bytes, hipStream_t stream) {
    PSAM_REQUIRE(xyz && region && out_idx && out_dist && ws, PSAM_EINVAL, "psam_border_farthest: null pointer");
    PSAM_REQUIRE(B > 0 && rep > 0 && N > 0 && (int64_t)B * rep <= 65535, PSAM_EINVAL, "psam_border_farthest: bad shape");
    const int Z = B * rep, nblk = (int)psam_cdiv(N, BF_THREADS);
    PSAM_REQUIRE(ws_bytes >= psam_border_farthest_workspace_bytes(Z, N), PSAM_EWORKSPACE, "psam_border_farthest: workspace too small");
    float* pval = (float*)ws;
    int* pidx = (int*)(pval + (size_t)Z * nblk);
    hipLaunchKernelGGL(border_partial_kernel, dim3(nblk, Z), dim3(BF_THREADS), 0, stream, xyz, region, rep, N, nblk, pval, pidx);
    hipLaunchKernelGGL(border_final_kernel, dim3(Z), dim3(256), 0, stream, pval, pidx, nblk, out_idx, out_dist);
    return psam_launch_status("psam_border_farthest: launch failed");
}
