// ls_common.h -- shared device/host helpers for liblivingscenes_hip.so (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <float.h>
#include <limits.h>

#include "../../include/livingscenes_hip.h"

namespace ls {

void set_error(const char* fmt, ...);

#define LS_HIP_CHECK(expr)                                                                        \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            ls::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return LS_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define LS_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            ls::set_error(__VA_ARGS__); \
            return LS_ERR_INVALID;     \
        }                              \
    } while (0)

#define LS_LAUNCH_CHECK()                                                            \
    do {                                                                             \
        hipError_t _e = hipGetLastError();                                           \
        if (_e != hipSuccess) {                                                      \
            ls::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return LS_ERR_HIP;                                                       \
        }                                                                            \
    } while (0)

// Development A/B switches: a library built with -DLS_DEV_KNOBS (scripts/dev/build_variants.py) reads them from the environment; in the RELEASE
// library every knob is its default, a compile-time constant -- the release library reads nothing from the environment beyond what
// ls_model_create documents (LS_ENCODE_GRAPH, LS_EDGE_STAGED, LS_SDF_BF16X2) and the process-wide arithmetic mode LS_GEMM_MODE (gemm.hip).
// Latency-bound kernels with one workgroup (or wave) per instance -- FPS, the 32-point k-NN, the heads, matcher, Kabsch -- share their CUs with the chip-filling
// kernels of other steps in flight (and, inside one step, FPS runs beside layers 0 - 1): a raised wave priority makes the SIMD arbiter issue them first.
#ifndef LS_PRIO
#define LS_PRIO 0
#endif
#if LS_PRIO > 0
#define LS_LATENCY_CRITICAL() __builtin_amdgcn_s_setprio(LS_PRIO)
#else
#define LS_LATENCY_CRITICAL()
#endif
#ifdef LS_DEV_KNOBS
inline int dev_knob(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#else
constexpr int dev_knob(const char*, int dflt) { return dflt; }
#endif

constexpr int kWave = 64;
constexpr int kXcds = 8;

// XCD-aware block remap (MI355X: block b is dispatched to XCD b % 8, each XCD has a private 4 MiB L2).
// Returns a logical block id such that the blocks resident on one XCD cover a CONTIGUOUS range of logical
// ids, so consecutive logical ids (tiles of the same instance) share an L2.  Bijective for any nblocks.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks / kXcds, r = nblocks % kXcds;
    const int xcd = bid % kXcds, slot = bid / kXcds;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// wave maximum on the DPP network (no LDS crossbar): the result is valid in LANE 63 only
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_fmax_rm(float v) {
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROWMASK, 0xF, false)));
}
__device__ __forceinline__ float wave_max_lane63(float v) {
    v = dpp_fmax_rm<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
    v = dpp_fmax_rm<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
    v = dpp_fmax_rm<0x141, 0xF>(v);   // row_half_mirror
    v = dpp_fmax_rm<0x140, 0xF>(v);   // row_mirror
    v = dpp_fmax_rm<0x142, 0xA>(v);   // row_bcast15 -> rows 1, 3
    v = dpp_fmax_rm<0x143, 0xC>(v);   // row_bcast31 -> rows 2, 3
    return v;
}
// reductions inside aligned groups of 16 lanes (one attention head = 16 channels = one DPP row)
__device__ __forceinline__ float row16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// canonical squared-difference accumulation (oracle/ls_oracle.c acc_sq)
template <bool FMA>
__device__ __forceinline__ float acc_sq(float d, float diff) {
    if constexpr (FMA) return __fmaf_rn(diff, diff, d);
    else return __fadd_rn(d, __fmul_rn(diff, diff));
}

// VN activation closed form (vec_layers.py:241-268): y - (1-slope) * min(<y,k^>,0) * k^,  k^ = k / max(|k|,1e-12).
// With p = <y,k> un-normalised this is  y - (1-slope) * min(p,0) / max(|k|^2, 1e-24) * k : one v_rcp_f32 instead of a
// correctly rounded sqrt and division (~14 instead of ~40 VALU operations per 3-vector; the edge kernels apply it per edge and
// channel).  Differs from the reference's operation order at the 1e-7 level, like the rest of the folded edge-conv.
// Every multiply-add is SPELLED as an fma (round 3): with -ffp-contract=fast the compiler chose which products to fuse per call site, and
// two kernels that must agree bit for bit (the fused-destination attention kernel and the table path) stopped agreeing in the last bit
// once their loops were restructured differently.
__device__ __forceinline__ void vn_act(float& y0, float& y1, float& y2, float k0, float k1, float k2, float one_minus_slope) {
    const float n2 = __builtin_fmaf(k2, k2, __builtin_fmaf(k1, k1, k0 * k0));
    const float p = __builtin_fmaf(y2, k2, __builtin_fmaf(y1, k1, y0 * k0));
    const float f = one_minus_slope * fminf(p, 0.0f) * __builtin_amdgcn_rcpf(fmaxf(n2, 1e-24f));
    y0 = __builtin_fmaf(-f, k0, y0); y1 = __builtin_fmaf(-f, k1, y1); y2 = __builtin_fmaf(-f, k2, y2);
}

// operand range of the f16-split GEMMs (gemm.hip, "operand range of the f16 split"): optional caller-supplied row maxima
struct GemmAux {
    const float* a_rowmax = nullptr;   // [rows of A][a_parts]: max over the parts bounds max|A[row, :]|; indexed by the SOURCE row when a_rows gathers
    int a_parts = 0;
    const float* w_rowmax = nullptr;   // [N]
    const void* w_planes = nullptr;    // pre-split W (gemm_presplit_w_launch): row n = K / 32 lines [hi: 32 f16 | lo: 32 f16] of s_n W[n, :], s_n from w_rowmax (required)
    float* out_rowmax = nullptr;       // [M][2 * cdiv(N, 128)]: max|out[row, 64-column block]| written by the epilogue (un-split launches only)
    int noscale = 0;                   // LS_GEMM_RANGE=0: the round-2 arithmetic (no row scaling; |a| < 65504 required), A/B timing
    // SLICE-MAJOR output (edge_staged.hip): the M rows are instances of slice_rows rows; element (m, n) goes to
    // out[(m / slice_rows) * slice_rows * N + (n / slice_cols) * slice_rows * slice_cols + (m % slice_rows) * slice_cols + n % slice_cols] -- per instance
    // N / slice_cols contiguous slices of [slice_rows][slice_cols] floats.  slice_cols = 4 | 8, slice_rows % 32 == 0; K = 32 / 64 GEMMs only (gemm_h2_smallk_kernel)
    int slice_cols = 0, slice_rows = 0;
    const float* cs = nullptr;         // gemm_vn_dispatch only: partial column sums of A ([instance][cs_rows][3][C], edge.hip: attn_colsum) -- the kernel forms the
    int cs_rows = 0;                   // mean part of the conv itself instead of reading G (gemm.hip: gemm_vn_direct_kernel); ignored where that kernel is not taken
};

inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace ls
