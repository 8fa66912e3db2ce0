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

typedef float fps_f2 __attribute__((ext_vector_type(2)));
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

// The same reduction on ready-made keys; both halves of the winning key come back wave-uniform (SGPRs).
__device__ __forceinline__ void wave_max_key_dpp(unsigned& hi, unsigned& lo) {
    dpp_max_u64<0xB1, 0xF>(hi, lo);
    dpp_max_u64<0x4E, 0xF>(hi, lo);
    dpp_max_u64<0x141, 0xF>(hi, lo);
    dpp_max_u64<0x140, 0xF>(hi, lo);
    dpp_max_u64<0x142, 0xA>(hi, lo);
    dpp_max_u64<0x143, 0xC>(hi, lo);
    hi = (unsigned)__builtin_amdgcn_readlane((int)hi, 63);
    lo = (unsigned)__builtin_amdgcn_readlane((int)lo, 63);
}

template <int PPL, bool FMA>
__global__ __launch_bounds__(64) void fps_wave_kernel(const float* __restrict__ pts, const int32_t* __restrict__ lengths,
                                                      int N, int K, int32_t* __restrict__ idx_out,
                                                      float* __restrict__ pts_out) {
    LS_LATENCY_CRITICAL();
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
// arg-max, the four wave results exchanged through a double-buffered LDS slot with ONE workgroup barrier per step.  Same arithmetic
// and tie rule as the one-wave kernel (the encoder's first down-sampling: 0.52 -> 0.18 ms when it replaced that kernel).
// Round 4: a step's critical path was FIVE dependent LDS round trips (the winner's coordinates lp[last], the slot write, then the
// four wave records read one after the other behind short-circuit branches) around ~120 VALU instructions: 0.61 - 0.65 us per step,
// 1 460 cycles, of which ~500 were LDS latency.  Now the winning lane of every wave writes its record WITH the point's coordinates
// {value, index, x, y | z}, every thread reads the four records in one batch of independent 16-byte broadcast reads after the barrier
// and merges them branch-free: two round trips per step (write -> barrier, read), and lp[] is only the first sample's source.
// One instruction per stage: the DPP permutation as the source modifier of v_max_f32 / v_min_u32 (hipcc emits v_mov_b32, s_nop, v_mov_b32_dpp,
// a canonicalising v_max and the v_max per stage of the update_dpp form: 5 dependent issue slots; here 2 -- the s_nop 1 are the two wait states a
// DPP read of a just-written VGPR needs, which nothing inserts inside an asm statement).  Lanes a row_bcast does not reach keep their value
// (row_mask / bound_ctrl 0), so lane 63 ends up with the wave's maximum / minimum.
__device__ __forceinline__ float wave_max_to_lane63(float v) {
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ unsigned wave_min_to_lane63(unsigned v) {
    asm("s_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"
        : "+v"(v));
    return v;
}
struct FpsRec { float v; unsigned i; float x, y; };
template <int PPT, bool FMA>
__global__ __launch_bounds__(256) void fps_quad_kernel(const float* __restrict__ pts, const int32_t* __restrict__ lengths,
                                                       int N, int K, int32_t* __restrict__ idx_out, float* __restrict__ pts_out) {
    LS_LATENCY_CRITICAL();
    extern __shared__ __attribute__((aligned(16))) float lp[];  // [N][3]
    __shared__ __attribute__((aligned(16))) FpsRec lrec[2][4];  // [buffer][wave] {value, index, x, y}
    __shared__ __attribute__((aligned(16))) float lz[2][4];     // [buffer][wave] z
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
    const int kk = min(K, n);
    float lx = lp[0], ly = lp[1], lz0 = lp[2];     // the first sample is point 0
    if (tid == 0 && n > 0) {
        out[0] = 0;
        if (po) { po[0] = lx; po[1] = ly; po[2] = lz0; }
    }
    for (int k = 1; k < kk; ++k) {
        float bv = -INFINITY, bx = 0.f, by = 0.f, bz = 0.f;
        int bi = INT_MAX;
        // Round 6: the distances of a thread's points two at a time on packed fp32 (v_pk_add / v_pk_mul: the same IEEE subtraction, product and
        // separately rounded additions per element as dist3<false> -- 0 + dx^2 is dx^2 exactly, so the first addition is dropped -- bit-identical
        // indices, tests/test_hip_parity.py::test_fps_bit_exact), and the running minimum without its padding test: fminf(-inf, d) is -inf for every d,
        // NaN included, so a padding slot can never leave -inf.  66 -> 48 VALU instructions in the per-step chain.
        float dd[PPT];
        if constexpr (!FMA && PPT % 2 == 0) {
#pragma clang fp contract(off)
            const fps_f2 l2x = {lx, lx}, l2y = {ly, ly}, l2z = {lz0, lz0};
#pragma unroll
            for (int i = 0; i < PPT; i += 2) {
                const fps_f2 dx = l2x - fps_f2{px[i], px[i + 1]}, dy = l2y - fps_f2{py[i], py[i + 1]}, dz = l2z - fps_f2{pz[i], pz[i + 1]};
                fps_f2 d = dx * dx;
                d = d + dy * dy;
                d = d + dz * dz;
                dd[i] = d.x; dd[i + 1] = d.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < PPT; ++i) dd[i] = dist3<FMA>(lx, ly, lz0, px[i], py[i], pz[i]);
        }
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            md[i] = fminf(md[i], dd[i]);
            const bool up = md[i] > bv;                          // ascending index inside the thread: strict '>'
            bv = up ? md[i] : bv; bi = up ? i * 256 + tid : bi;
            bx = up ? px[i] : bx; by = up ? py[i] : by; bz = up ? pz[i] : bz;
        }
        // wave arg-max in two single-instruction-per-stage DPP reductions: the maximum value, then the smallest index among the lanes that
        // hold it.  (The 64-bit key network of the one-wave kernel costs ~8 dependent instructions per stage.)
        const float wmax = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(wave_max_to_lane63(bv)), 63));
        // the lane that holds the wave's winner publishes it.  Almost always ONE lane holds the maximum (round 5: a ballot tells, and the second DPP
        // reduction -- six dependent stages + a v_readlane on the critical path of every step -- is skipped); only on an exact tie between lanes is the
        // smallest index among them reduced (a wave without a live point: every lane holds (-inf, INT_MAX), ties, and writes the same record, which
        // loses against any live point)
        bool mine = bv == wmax;
        if (__builtin_popcountll(__ballot(mine)) != 1) {     // (wave-uniform)
            const unsigned wci = (unsigned)__builtin_amdgcn_readlane((int)wave_min_to_lane63(mine ? (unsigned)bi : 0xFFFFFFFFu), 63);
            mine = mine && (unsigned)bi == wci;
        }
        if (mine) {
            lrec[k & 1][wave] = FpsRec{wmax, (unsigned)bi, bx, by};
            lz[k & 1][wave] = bz;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS only: the global stores below stay in flight across steps
        const FpsRec r0 = lrec[k & 1][0], r1 = lrec[k & 1][1], r2 = lrec[k & 1][2], r3 = lrec[k & 1][3];
        const float4 zz = *reinterpret_cast<const float4*>(lz[k & 1]);
        // branch-free merge, larger value first, then smaller index: (0, 1) and (2, 3) side by side, then the two winners
        const bool t01 = (r1.v > r0.v) | ((r1.v == r0.v) & (r1.i < r0.i));
        const bool t23 = (r3.v > r2.v) | ((r3.v == r2.v) & (r3.i < r2.i));
        const float av = t01 ? r1.v : r0.v, cv = t23 ? r3.v : r2.v;
        const unsigned ai = t01 ? r1.i : r0.i, cix = t23 ? r3.i : r2.i;
        const float ax = t01 ? r1.x : r0.x, ay = t01 ? r1.y : r0.y, az = t01 ? zz.y : zz.x;
        const float cx = t23 ? r3.x : r2.x, cy = t23 ? r3.y : r2.y, cz = t23 ? zz.w : zz.z;
        const bool tf = (cv > av) | ((cv == av) & (cix < ai));
        const int last = (int)(tf ? cix : ai);
        lx = tf ? cx : ax; ly = tf ? cy : ay; lz0 = tf ? cz : az;
        if (tid == 0) {
            out[k] = last;
            if (po) { po[k * 3 + 0] = lx; po[k * 3 + 1] = ly; po[k * 3 + 2] = lz0; }
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


// ---------------------------------------------------------------------------------------------------------------------
// Raw clouds (8 192 < N <= 65 536 points -> 1 024 samples: Shape_Prior.encode_fps, model_utils.py:199-215; more_solver.py:107-108).
// fps_block_kernel re-reads the whole cloud every step (720 KB through one CU for 60 000 points: 17.6 us per step, 18 ms per
// cloud).  But once a few dozen samples exist, a new sample can only lower the running minimum of points NEAR it: bucket the
// cloud on a uniform grid (counting sort inside the workgroup), keep per bucket the bounding box of its points and its largest
// running minimum (value, smallest original index), and per step
//   1. test every bucket: lb = canonical distance from the new sample to the bucket's box, evaluated with the SAME fp32 formula on
//      the per-axis gaps -- rounding is monotone, so lb <= the computed distance of every point in the box, exactly; a bucket with
//      lb >= its largest minimum cannot change (min(md, d) = md for all its points) and is skipped;
//   2. update the points of the remaining buckets (16 lanes per bucket, four buckets per wave step) and their maxima;
//   3. arg-max over the bucket maxima (larger value, then smaller ORIGINAL index: the first arg-max of the un-bucketed scan).
// Bit-identical to the full scan by construction (tests: ragged, duplicates, degenerate clouds, both rounding modes); the work per
// step falls from N to ~N / k point updates + one test per bucket.
constexpr int FB_MAXB = 4096;      // buckets
constexpr int FB_CELLS = 32768;    // finest grid: 32^3

__device__ __forceinline__ int fb_cell(const float* __restrict__ p, int i, const float (&lo)[3], const float (&sc)[3], int G) {
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = min(G - 1, max(0, (int)((p[(size_t)i * 3 + a] - lo[a]) * sc[a])));
    return (c[0] * G + c[1]) * G + c[2];
}

template <bool FMA, int NT>
__global__ __launch_bounds__(NT) void fps_bucket_kernel(const float* __restrict__ pts, const int32_t* __restrict__ lengths, int N, int K,
                                                          int32_t* __restrict__ idx_out, float* __restrict__ pts_out, char* __restrict__ ws,
                                                          size_t ws_stride) {
    __shared__ int s_cnt[FB_MAXB];        // bucket arg index (original index of its largest running minimum)
    __shared__ int s_start[FB_MAXB + 1];  // first point of a bucket in the bucket-contiguous copy
    __shared__ float s_max[FB_MAXB];      // largest running minimum of a bucket (-inf: empty bucket)
    __shared__ unsigned short s_work[FB_MAXB];
    __shared__ float s_red[(NT / 64) * 6];
    __shared__ int s_redi[NT / 64];
    __shared__ int s_nwork;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = pts + (size_t)b * N * 3;
    const int n = lengths ? min(lengths[b], N) : N;
    int32_t* out = idx_out + (size_t)b * K;
    float* po = pts_out ? pts_out + (size_t)b * K * 3 : nullptr;
    const int kk = min(K, n);
    // per-cloud scratch
    char* w = ws + (size_t)b * ws_stride;
    float* ppts = (float*)w;                                  // [N][3] bucket-contiguous copy of the cloud
    int32_t* pidx = (int32_t*)(ppts + (size_t)N * 3);         // [N] original index
    float* md = (float*)(pidx + N);                           // [N] running minimum
    int* hist = (int*)(md + N);                               // [FB_CELLS] points per grid cell -> running scatter offset (L2 atomics only)
    int32_t* bstart = hist + FB_CELLS;                        // [FB_MAXB + 1]

    if (n > 0) {
        // ---- bounding box of the cloud
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = tid; i < n; i += NT)
#pragma unroll
            for (int a = 0; a < 3; ++a) { const float v = p[(size_t)i * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64)); }
            if (lane == 0) { s_red[wave * 6 + a] = lo[a]; s_red[wave * 6 + 3 + a] = hi[a]; }
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 3; ++a)
            for (int ww = 0; ww < NT / 64; ++ww) { lo[a] = fminf(lo[a], s_red[ww * 6 + a]); hi[a] = fmaxf(hi[a], s_red[ww * 6 + 3 + a]); }
        // ---- counting sort on a uniform grid; the buckets are its NON-EMPTY cells.  Scanner clouds are surfaces: a 32^3 grid has a few
        //      thousand occupied cells of ~15-40 points; a cloud that fills the volume (more than FB_MAXB occupied cells) drops to 16^3.
        int G = 32, NB = 0;
        float sc[3];
        for (;; G = 16) {
            const int NC = G * G * G, per = NC / NT;
#pragma unroll
            for (int a = 0; a < 3; ++a) sc[a] = (float)G / fmaxf(hi[a] - lo[a], 1e-30f);
            for (int c = tid; c < NC; c += NT) __hip_atomic_store(&hist[c], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            for (int i = tid; i < n; i += NT) atomicAdd(&hist[fb_cell(p, i, lo, sc, G)], 1);
            __syncthreads();
            // exclusive scans over the cells (per consecutive cells per thread): point offset, and bucket id = rank among non-empty cells
            const int c0 = tid * per;
            int sum = 0, nz = 0;
            for (int e = 0; e < per; ++e) {
                const int h = __hip_atomic_load(&hist[c0 + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sum += h; nz += h > 0;
            }
            int isum = sum, inz = nz;   // inclusive wave scans
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t1 = __shfl_up(isum, o, 64), t2 = __shfl_up(inz, o, 64);
                if (lane >= o) { isum += t1; inz += t2; }
            }
            if (lane == 63) { s_redi[wave] = isum; s_cnt[wave] = inz; }
            __syncthreads();
            int run = isum - sum, bk = inz - nz, tot = 0;
            for (int ww = 0; ww < NT / 64; ++ww) { if (ww < wave) { run += s_redi[ww]; bk += s_cnt[ww]; } tot += s_cnt[ww]; }
            NB = tot;
            __syncthreads();            // s_redi / s_cnt are free again
            if (NB > FB_MAXB && G == 32) continue;
            for (int e = 0; e < per; ++e) {
                const int h = __hip_atomic_load(&hist[c0 + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (h > 0) bstart[bk++] = run;
                __hip_atomic_store(&hist[c0 + e], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                run += h;
            }
            if (tid == 0) bstart[NB] = n;
            break;
        }
        __syncthreads();
        for (int i = tid; i < n; i += NT) {
            const int pos = atomicAdd(&hist[fb_cell(p, i, lo, sc, G)], 1);   // order inside a bucket is arbitrary: nothing depends on it
            ppts[(size_t)pos * 3] = p[(size_t)i * 3]; ppts[(size_t)pos * 3 + 1] = p[(size_t)i * 3 + 1]; ppts[(size_t)pos * 3 + 2] = p[(size_t)i * 3 + 2];
            pidx[pos] = i;
            md[pos] = INFINITY;
        }
        for (int c = tid; c <= NB; c += NT) s_start[c] = bstart[c];
        __syncthreads();
        // thread t owns buckets t, t + 1024, ...: their exact boxes live in its registers for the whole run
        constexpr int E = FB_MAXB / NT, NW = NT / 64;   // buckets per thread, waves
        float bx[E][6];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int c = tid + NT * e;
#pragma unroll
            for (int a = 0; a < 3; ++a) { bx[e][a] = INFINITY; bx[e][3 + a] = -INFINITY; }
            if (c < NB) {
                const int s0 = s_start[c], e0 = s_start[c + 1];
                for (int j = s0; j < e0; ++j)
#pragma unroll
                    for (int a = 0; a < 3; ++a) { const float v = ppts[(size_t)j * 3 + a]; bx[e][a] = fminf(bx[e][a], v); bx[e][3 + a] = fmaxf(bx[e][3 + a], v); }
                s_max[c] = e0 > s0 ? INFINITY : -INFINITY;    // +inf: the first step visits every non-empty bucket
                s_cnt[c] = INT_MAX;                           // from here on: the bucket's arg index
            }
        }
        if (tid == 0) { s_nwork = 0; out[0] = 0; if (po) { po[0] = p[0]; po[1] = p[1]; po[2] = p[2]; } }
        __syncthreads();

        // ---- the K - 1 dependent steps
        int last = 0;
        const int sub = lane >> 4, sl = lane & 15;
        for (int k = 1; k < kk; ++k) {
            const float lx = p[(size_t)last * 3], ly = p[(size_t)last * 3 + 1], lz = p[(size_t)last * 3 + 2];
            // 1. which buckets can change?  per-axis gap to the box, then the canonical formula on the gaps: a lower bound of every
            //    point's COMPUTED distance
            bool hit[E];
            int cnt = 0;
            unsigned long long mask[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int c = tid + NT * e;
                hit[e] = false;
                if (c < NB) {
                    const float bm = s_max[c];
                    const float gx = lx > bx[e][3] ? lx - bx[e][3] : (lx < bx[e][0] ? bx[e][0] - lx : 0.f);
                    const float gy = ly > bx[e][4] ? ly - bx[e][4] : (ly < bx[e][1] ? bx[e][1] - ly : 0.f);
                    const float gz = lz > bx[e][5] ? lz - bx[e][5] : (lz < bx[e][2] ? bx[e][2] - lz : 0.f);
                    hit[e] = dist3<FMA>(gx, gy, gz, 0.f, 0.f, 0.f) < bm;   // empty bucket: bm = -inf, never
                }
                mask[e] = __ballot(hit[e]);
                cnt += __popcll(mask[e]);
            }
            if (cnt) {   // wave-uniform: one LDS atomic per wave
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_nwork, cnt);
                base = __shfl(base, 0, 64);
                const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (hit[e]) s_work[base + __popcll(mask[e] & below)] = (unsigned short)(tid + NT * e);
                    base += __popcll(mask[e]);
                }
            }
            __syncthreads();
            // 2. update them: 16 lanes per bucket
            const int nw = s_nwork;
            for (int it0 = wave * 4; it0 < nw; it0 += NW * 4) {   // wave-uniform bound: the four 16-lane groups stay in the loop together
                const int it = it0 + sub;
                float v = -INFINITY;
                int bi = INT_MAX, c = -1;
                if (it < nw) {
                    c = s_work[it];
                    const int s0 = s_start[c], e0 = s_start[c + 1];
                    for (int j = s0 + sl; j < e0; j += 16) {
                        const float d = dist3<FMA>(lx, ly, lz, ppts[(size_t)j * 3], ppts[(size_t)j * 3 + 1], ppts[(size_t)j * 3 + 2]);
                        const float mm = fminf(md[j], d);
                        md[j] = mm;
                        const int oi = pidx[j];
                        if (mm > v || (mm == v && oi < bi)) { v = mm; bi = oi; }
                    }
                }
                // 16-lane arg-max on the DPP network (keys as below; every lane of the row ends up with the row's maximum)
                unsigned gh = bi != INT_MAX ? __float_as_uint(v) + 1u : 0u, gl = bi != INT_MAX ? ~(unsigned)bi : 0u;
                dpp_max_u64<0xB1, 0xF>(gh, gl);
                dpp_max_u64<0x4E, 0xF>(gh, gl);
                dpp_max_u64<0x141, 0xF>(gh, gl);
                dpp_max_u64<0x140, 0xF>(gh, gl);
                if (c >= 0 && sl == 0) { s_max[c] = __uint_as_float(gh - 1u); s_cnt[c] = (int)~gl; }
            }
            __syncthreads();
            // 3. arg-max over the bucket maxima
            if (tid == 0) s_nwork = 0;
            float bv = -INFINITY;
            int bi = INT_MAX;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int c = tid + NT * e;
                if (c < NB) {
                    const float v = s_max[c];
                    const int oi = s_cnt[c];
                    if (v > bv || (v == bv && oi < bi)) { bv = v; bi = oi; }
                }
            }
            // (value >= 0, index) as one 64-bit key (see wave_argmax_dpp): larger value, then smaller index = unsigned max; key 0 = nothing
            unsigned khi = bi != INT_MAX ? __float_as_uint(bv) + 1u : 0u, klo = bi != INT_MAX ? ~(unsigned)bi : 0u;
            wave_max_key_dpp(khi, klo);
            if (lane == 0) { s_red[wave] = __uint_as_float(khi); s_redi[wave] = (int)klo; }
            __syncthreads();
            {   // every wave reduces the 16 partial results itself: no second barrier for a broadcast, and `last` is wave-uniform
                khi = lane < NW ? __float_as_uint(s_red[lane]) : 0u;
                klo = lane < NW ? (unsigned)s_redi[lane] : 0u;
                wave_max_key_dpp(khi, klo);
                last = (int)~klo;
            }
            if (tid == 0) {
                out[k] = last;
                if (po) { po[k * 3] = p[(size_t)last * 3]; po[k * 3 + 1] = p[(size_t)last * 3 + 1]; po[k * 3 + 2] = p[(size_t)last * 3 + 2]; }
            }
            // s_red / s_redi are rewritten after the next step's second barrier, s_nwork was reset before this step's third one
        }
    }
    for (int k = max(kk, 0) + tid; k < K; k += NT) {
        out[k] = -1;
        if (po) { po[k * 3 + 0] = 0.f; po[k * 3 + 1] = 0.f; po[k * 3 + 2] = 0.f; }
    }
}

size_t fps_scratch_bytes_per_cloud(int N) {
    if (N <= 8192) return 0;
    size_t b = (size_t)N * (12 + 4 + 4) + (size_t)(FB_CELLS + FB_MAXB + 1) * 4;
    return (b + 255) & ~(size_t)255;
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
static int fps_mode(const float* pts, const int32_t* lengths, int B, int N, int K, int32_t* idx, float* po, void* ws, size_t ws_bytes,
                    hipStream_t st) {
    if (N <= 128) return launch_wave<2, FMA>(pts, lengths, B, N, K, idx, po, st);
#ifdef LS_DEV_KNOBS      // dev A/B: the one-wave kernels for 256 .. 2048 points (same indices)
    static const bool one_wave = dev_knob("LS_FPS_ONE_WAVE", 0) != 0;
    if (one_wave && N <= 512) return launch_wave<8, FMA>(pts, lengths, B, N, K, idx, po, st);
    if (one_wave && N <= 1024) return launch_wave<16, FMA>(pts, lengths, B, N, K, idx, po, st);
    if (one_wave && N <= 2048) return launch_wave<32, FMA>(pts, lengths, B, N, K, idx, po, st);
#endif
    if (N <= 256) return launch_quad<1, FMA>(pts, lengths, B, N, K, idx, po, st);
    if (N <= 512) return launch_quad<2, FMA>(pts, lengths, B, N, K, idx, po, st);
    if (N <= 1024) return launch_quad<4, FMA>(pts, lengths, B, N, K, idx, po, st);
    if (N <= 2048) return launch_quad<8, FMA>(pts, lengths, B, N, K, idx, po, st);
    if (N <= 8192) return launch_block<8, FMA>(pts, lengths, B, N, K, idx, po, st);
    if (N <= 65536) {
        // 8 193 .. 65 536 points: exact bucket pruning in the caller's workspace (ls_fps_workspace_bytes).  (Until round 5 a call without workspace fell
        // back to a full scan per step -- fps_block_kernel<64>, 82 spilled registers, ~30x slower; it is an error now.)
        const size_t per = fps_scratch_bytes_per_cloud(N);
        if (!ws || ws_bytes < per * (size_t)B) {
            set_error("fps: N=%d needs a workspace of %zu bytes (ls_fps_workspace_bytes), got %zu", N, per * (size_t)B, ws ? ws_bytes : (size_t)0);
            return LS_ERR_WORKSPACE;
        }
        hipLaunchKernelGGL((fps_bucket_kernel<FMA, 1024>), dim3(B), dim3(1024), 0, st, pts, lengths, N, K, idx, po, (char*)ws, per);
        LS_LAUNCH_CHECK();
        return LS_OK;
    }
    set_error("fps: N=%d too large (max 65536)", N);
    return LS_ERR_INVALID;
}

int fps_dispatch(const float* pts, const int32_t* lengths, int B, int N, int K, unsigned flags, int32_t* idx_out,
                 float* pts_out, void* ws, size_t ws_bytes, hipStream_t st) {
    LS_REQUIRE(B > 0 && N > 0 && K > 0, "fps: empty problem (B=%d N=%d K=%d)", B, N, K);
    return (flags & LS_FLAG_CONTRACT_FMA) ? fps_mode<true>(pts, lengths, B, N, K, idx_out, pts_out, ws, ws_bytes, st)
                                          : fps_mode<false>(pts, lengths, B, N, K, idx_out, pts_out, ws, ws_bytes, st);
}

}  // namespace ls
