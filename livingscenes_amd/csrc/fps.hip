// fps.hip -- farthest point sampling, bit-exact w.r.t. oracle/ls_oracle.c (lso_fps).
//
// Replaces pytorch3d.ops.sample_farthest_points(points, K=.., random_start_point=False) as called at
//   /root/reference/lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:169   (1024->512->128->32 inside the encoder)
//   /root/reference/model_utils.py:205, lib_more/more_solver.py:67,107,108      (raw instance cloud -> 1024)
//
// FPS is a chain of K dependent arg-max steps: latency-bound, no data reuse to exploit.  Two shapes:
//   * fps_wave_kernel: ONE WAVE per instance, the cloud and its running min-distance live in VGPRs
//     (<= 64 points per lane), so a step is ~PPL*8 VALU ops + a 6-stage shuffle arg-max + one broadcast
//     LDS read and needs NO barrier at all.  Four instances share a CU (one per SIMD).  Used for N <= 4096.
//   * fps_block_kernel: 1024 threads per instance, min-distances in VGPRs (<= 64 per thread), points
//     re-read from L2 each step, two-level (wave shuffle + LDS) arg-max.  Used for raw clouds up to 65536 pts.
// Tie rule everywhere: larger value wins, equal values -> smaller index wins (== first arg-max).
#include "ls_common.h"

namespace ls {

template <bool FMA>
__device__ __forceinline__ float dist3(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
    float d = 0.0f;
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    if constexpr (FMA) {
        d = __builtin_fmaf(dx, dx, d); d = __builtin_fmaf(dy, dy, d); d = __builtin_fmaf(dz, dz, d);
    } else {
        float p = dx * dx; d = d + p;
        p = dy * dy; d = d + p;
        p = dz * dz; d = d + p;
    }
    return d;
}

__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(i, o, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

// Wave arg-max on the DPP network (no LDS crossbar round trips): the pair (value >= 0, index) is packed into a
// 64-bit key = float bits << 32 | ~index, so "larger value, then smaller index" is a plain unsigned max.  Stages:
// quad_perm x2, row_half_mirror, row_mirror (16-lane row max), row_bcast15 (rows 1,3), row_bcast31 (rows 2,3);
// lane 63 ends up with the wave maximum.  Padding slots must be given the key 0.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ void dpp_max_u64(unsigned& hi, unsigned& lo) {
    const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, ROWMASK, 0xF, false);
    const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROWMASK, 0xF, false);
    const bool take = ohi > hi || (ohi == hi && olo > lo);
    hi = take ? ohi : hi;
    lo = take ? olo : lo;
}
// old value = own value: lanes a row_bcast does not reach (or masks out) keep theirs, so max / min stay correct
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROWMASK, 0xF, false));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_u(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xF, false);
}
__device__ __forceinline__ int wave_argmax_dpp(float v, int i, bool valid) {
    unsigned hi = valid ? __float_as_uint(v) + 1u : 0u;  // +1: a valid 0.0 still beats padding
    unsigned lo = valid ? ~(unsigned)i : 0u;
    dpp_max_u64<0xB1, 0xF>(hi, lo);   // quad_perm [1,0,3,2]
    dpp_max_u64<0x4E, 0xF>(hi, lo);   // quad_perm [2,3,0,1]
    dpp_max_u64<0x141, 0xF>(hi, lo);  // row_half_mirror
    dpp_max_u64<0x140, 0xF>(hi, lo);  // row_mirror
    dpp_max_u64<0x142, 0xA>(hi, lo);  // row_bcast15 -> rows 1, 3
    dpp_max_u64<0x143, 0xC>(hi, lo);  // row_bcast31 -> rows 2, 3
    return (int)~(unsigned)__builtin_amdgcn_readlane((int)lo, 63);
}

template <int PPL, bool FMA>
__global__ __launch_bounds__(64) void fps_wave_kernel(const float* __restrict__ pts, const int32_t* __restrict__ lengths,
                                                      int N, int K, int32_t* __restrict__ idx_out,
                                                      float* __restrict__ pts_out) {
    extern __shared__ __attribute__((aligned(16))) float lp[];  // [N][3]
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* p = pts + (size_t)b * N * 3;
    const int n = lengths ? min(lengths[b], N) : N;
    for (int t = lane; t < N * 3; t += 64) lp[t] = p[t];
    __syncthreads();
    float px[PPL], py[PPL], pz[PPL], md[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int j = i * 64 + lane;
        const bool ok = j < n;
        px[i] = ok ? lp[j * 3 + 0] : 0.f;
        py[i] = ok ? lp[j * 3 + 1] : 0.f;
        pz[i] = ok ? lp[j * 3 + 2] : 0.f;
        md[i] = ok ? INFINITY : -INFINITY;  // padding can never win an arg-max
    }
    int32_t* out = idx_out + (size_t)b * K;
    float* po = pts_out ? pts_out + (size_t)b * K * 3 : nullptr;
    int last = 0;
    const int kk = min(K, n);
    if (lane == 0 && n > 0) {
        out[0] = 0;
        if (po) { po[0] = lp[0]; po[1] = lp[1]; po[2] = lp[2]; }
    }
    for (int k = 1; k < kk; ++k) {
        const float lx = lp[last * 3 + 0], ly = lp[last * 3 + 1], lz = lp[last * 3 + 2];
        float bv = -INFINITY;
        int bi = INT_MAX;
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            const float d = dist3<FMA>(lx, ly, lz, px[i], py[i], pz[i]);
            const float m = fminf(md[i], d);
            md[i] = (md[i] == -INFINITY) ? md[i] : m;
            if (md[i] > bv) { bv = md[i]; bi = i * 64 + lane; }  // ascending index inside the lane: strict '>'
        }
        bi = wave_argmax_dpp(bv, bi, bi != INT_MAX);
        last = bi;
        if (lane == 0) {
            out[k] = bi;
            if (po) { po[k * 3 + 0] = lp[bi * 3 + 0]; po[k * 3 + 1] = lp[bi * 3 + 1]; po[k * 3 + 2] = lp[bi * 3 + 2]; }
        }
    }
    for (int k = max(kk, 0) + lane; k < K; k += 64) {
        out[k] = -1;
        if (po) { po[k * 3 + 0] = 0.f; po[k * 3 + 1] = 0.f; po[k * 3 + 2] = 0.f; }
    }
}

// Four waves per instance (N = 256 .. 2048): cloud and running min-distance in VGPRs (<= 8 points per thread), per-wave DPP
// arg-max, the four wave maxima exchanged through a double-buffered LDS slot with ONE workgroup barrier per step.  A step is
// ~PPT*8 VALU ops + the DPP network + ~2 LDS round trips: 4x shorter than the one-wave kernel's 16 points per lane at N = 1024
// (the encoder's first down-sampling: 0.52 -> 0.18 ms), same arithmetic and tie rule.
template <int PPT, bool FMA>
__global__ __launch_bounds__(256) void fps_quad_kernel(const float* __restrict__ pts, const int32_t* __restrict__ lengths,
                                                       int N, int K, int32_t* __restrict__ idx_out, float* __restrict__ pts_out) {
    extern __shared__ __attribute__((aligned(16))) float lp[];  // [N][3]
    __shared__ unsigned lmax[2][4][2];                          // [buffer][wave][hi, lo]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = pts + (size_t)b * N * 3;
    const int n = lengths ? min(lengths[b], N) : N;
    for (int t = tid; t < N * 3; t += 256) lp[t] = p[t];
    __syncthreads();
    float px[PPT], py[PPT], pz[PPT], md[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int j = i * 256 + tid;
        const bool ok = j < n;
        px[i] = ok ? lp[j * 3 + 0] : 0.f;
        py[i] = ok ? lp[j * 3 + 1] : 0.f;
        pz[i] = ok ? lp[j * 3 + 2] : 0.f;
        md[i] = ok ? INFINITY : -INFINITY;  // padding can never win an arg-max
    }
    int32_t* out = idx_out + (size_t)b * K;
    float* po = pts_out ? pts_out + (size_t)b * K * 3 : nullptr;
    int last = 0;
    const int kk = min(K, n);
    if (tid == 0 && n > 0) {
        out[0] = 0;
        if (po) { po[0] = lp[0]; po[1] = lp[1]; po[2] = lp[2]; }
    }
    for (int k = 1; k < kk; ++k) {
        const float lx = lp[last * 3 + 0], ly = lp[last * 3 + 1], lz = lp[last * 3 + 2];
        float bv = -INFINITY;
        int bi = INT_MAX;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const float d = dist3<FMA>(lx, ly, lz, px[i], py[i], pz[i]);
            const float m = fminf(md[i], d);
            md[i] = (md[i] == -INFINITY) ? md[i] : m;
            if (md[i] > bv) { bv = md[i]; bi = i * 256 + tid; }  // ascending index inside the thread: strict '>'
        }
        // wave arg-max in two single-instruction-per-stage DPP reductions (the compiler folds the DPP move into v_max_f32 /
        // v_min_u32): the maximum value, then the smallest index among the lanes that hold it.  (The 64-bit key network of the
        // one-wave kernel costs ~8 dependent instructions per stage.)
        float wm = bv;
        wm = fmaxf(wm, dpp_f<0xB1, 0xF>(wm)); wm = fmaxf(wm, dpp_f<0x4E, 0xF>(wm));
        wm = fmaxf(wm, dpp_f<0x141, 0xF>(wm)); wm = fmaxf(wm, dpp_f<0x140, 0xF>(wm));
        wm = fmaxf(wm, dpp_f<0x142, 0xA>(wm)); wm = fmaxf(wm, dpp_f<0x143, 0xC>(wm));
        const float wmax = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(wm), 63));
        unsigned ci = (bv == wmax) ? (unsigned)bi : 0xFFFFFFFFu;
        ci = min(ci, dpp_u<0xB1, 0xF>(ci)); ci = min(ci, dpp_u<0x4E, 0xF>(ci));
        ci = min(ci, dpp_u<0x141, 0xF>(ci)); ci = min(ci, dpp_u<0x140, 0xF>(ci));
        ci = min(ci, dpp_u<0x142, 0xA>(ci)); ci = min(ci, dpp_u<0x143, 0xC>(ci));
        if (lane == 63) { lmax[k & 1][wave][0] = __float_as_uint(wmax); lmax[k & 1][wave][1] = ci; }
        __syncthreads();
        float bh = __uint_as_float(lmax[k & 1][0][0]);
        unsigned bl = lmax[k & 1][0][1];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float oh = __uint_as_float(lmax[k & 1][w][0]);
            const unsigned ol = lmax[k & 1][w][1];
            const bool take = oh > bh || (oh == bh && ol < bl);
            bh = take ? oh : bh;
            bl = take ? ol : bl;
        }
        last = (int)bl;
        if (tid == 0) {
            out[k] = last;
            if (po) { po[k * 3 + 0] = lp[last * 3 + 0]; po[k * 3 + 1] = lp[last * 3 + 1]; po[k * 3 + 2] = lp[last * 3 + 2]; }
        }
    }
    for (int k = max(kk, 0) + tid; k < K; k += 256) {
        out[k] = -1;
        if (po) { po[k * 3 + 0] = 0.f; po[k * 3 + 1] = 0.f; po[k * 3 + 2] = 0.f; }
    }
}

template <int PPT, bool FMA>
__global__ __launch_bounds__(1024) void fps_block_kernel(const float* __restrict__ pts, const int32_t* __restrict__ lengths,
                                                         int N, int K, int32_t* __restrict__ idx_out,
                                                         float* __restrict__ pts_out) {
    __shared__ float sv[16];
    __shared__ int si[16];
    __shared__ int s_last;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = pts + (size_t)b * N * 3;
    const int n = lengths ? min(lengths[b], N) : N;
    float md[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) md[i] = (i * 1024 + tid) < n ? INFINITY : -INFINITY;
    int32_t* out = idx_out + (size_t)b * K;
    float* po = pts_out ? pts_out + (size_t)b * K * 3 : nullptr;
    const int kk = min(K, n);
    if (tid == 0 && n > 0) {
        out[0] = 0;
        if (po) { po[0] = p[0]; po[1] = p[1]; po[2] = p[2]; }
    }
    int last = 0;
    for (int k = 1; k < kk; ++k) {
        const float lx = p[(size_t)last * 3 + 0], ly = p[(size_t)last * 3 + 1], lz = p[(size_t)last * 3 + 2];
        float bv = -INFINITY;
        int bi = INT_MAX;
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int j = i * 1024 + tid;
            if (j < n) {
                const float d = dist3<FMA>(lx, ly, lz, p[(size_t)j * 3 + 0], p[(size_t)j * 3 + 1], p[(size_t)j * 3 + 2]);
                md[i] = fminf(md[i], d);
                if (md[i] > bv) { bv = md[i]; bi = j; }
            }
        }
        wave_argmax(bv, bi);
        if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
        __syncthreads();
        if (wave == 0) {
            float v = lane < 16 ? sv[lane] : -INFINITY;
            int i2 = lane < 16 ? si[lane] : INT_MAX;
            wave_argmax(v, i2);
            if (lane == 0) {
                s_last = i2;
                out[k] = i2;
                if (po) { po[k * 3 + 0] = p[(size_t)i2 * 3 + 0]; po[k * 3 + 1] = p[(size_t)i2 * 3 + 1]; po[k * 3 + 2] = p[(size_t)i2 * 3 + 2]; }
            }
        }
        __syncthreads();
        last = s_last;
    }
    for (int k = max(kk, 0) + tid; k < K; k += 1024) {
        out[k] = -1;
        if (po) { po[k * 3 + 0] = 0.f; po[k * 3 + 1] = 0.f; po[k * 3 + 2] = 0.f; }
    }
}

template <int PPL, bool FMA>
static int launch_wave(const float* pts, const int32_t* lengths, int B, int N, int K, int32_t* idx, float* po, hipStream_t st) {
    hipLaunchKernelGGL((fps_wave_kernel<PPL, FMA>), dim3(B), dim3(64), (size_t)N * 3 * sizeof(float), st, pts, lengths, N, K, idx, po);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
template <int PPT, bool FMA>
static int launch_quad(const float* pts, const int32_t* lengths, int B, int N, int K, int32_t* idx, float* po, hipStream_t st) {
    hipLaunchKernelGGL((fps_quad_kernel<PPT, FMA>), dim3(B), dim3(256), (size_t)N * 3 * sizeof(float), st, pts, lengths, N, K, idx, po);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
template <int PPT, bool FMA>
static int launch_block(const float* pts, const int32_t* lengths, int B, int N, int K, int32_t* idx, float* po, hipStream_t st) {
    hipLaunchKernelGGL((fps_block_kernel<PPT, FMA>), dim3(B), dim3(1024), 0, st, pts, lengths, N, K, idx, po);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

template <bool FMA>
static int fps_mode(const float* pts, const int32_t* lengths, int B, int N, int K, int32_t* idx, float* po, hipStream_t st) {
    static const bool one_wave = getenv("LS_FPS_ONE_WAVE") && atoi(getenv("LS_FPS_ONE_WAVE")) != 0;   // A/B: the one-wave kernel
    if (N <= 128) return launch_wave<2, FMA>(pts, lengths, B, N, K, idx, po, st);
    if (!one_wave) {
        if (N <= 256) return launch_quad<1, FMA>(pts, lengths, B, N, K, idx, po, st);
        if (N <= 512) return launch_quad<2, FMA>(pts, lengths, B, N, K, idx, po, st);
        if (N <= 1024) return launch_quad<4, FMA>(pts, lengths, B, N, K, idx, po, st);
        if (N <= 2048) return launch_quad<8, FMA>(pts, lengths, B, N, K, idx, po, st);
    }
    if (N <= 512) return launch_wave<8, FMA>(pts, lengths, B, N, K, idx, po, st);
    if (N <= 1024) return launch_wave<16, FMA>(pts, lengths, B, N, K, idx, po, st);
    if (N <= 2048) return launch_wave<32, FMA>(pts, lengths, B, N, K, idx, po, st);
    if (N <= 8192) return launch_block<8, FMA>(pts, lengths, B, N, K, idx, po, st);
    if (N <= 65536) return launch_block<64, FMA>(pts, lengths, B, N, K, idx, po, st);
    set_error("fps: N=%d too large (max 65536)", N);
    return LS_ERR_INVALID;
}

int fps_dispatch(const float* pts, const int32_t* lengths, int B, int N, int K, unsigned flags, int32_t* idx_out,
                 float* pts_out, hipStream_t st) {
    LS_REQUIRE(B > 0 && N > 0 && K > 0, "fps: empty problem (B=%d N=%d K=%d)", B, N, K);
    return (flags & LS_FLAG_CONTRACT_FMA) ? fps_mode<true>(pts, lengths, B, N, K, idx_out, pts_out, st)
                                          : fps_mode<false>(pts, lengths, B, N, K, idx_out, pts_out, st);
}

}  // namespace ls
