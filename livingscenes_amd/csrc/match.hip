// match.hip -- instance matching and closed-form registration.
//
//   cosine_scores_kernel   F.normalize + m0 @ m1^T            /root/reference/lib_more/matcher_new.py:110-120
//   greedy_match_kernel    the renormalise / first-arg-max / delete loop   matcher_new.py:121-136 (also :166-181, :212-227)
//   kabsch_kernel          weighted Kabsch + residuals         /root/reference/lib_more/pose_estimation.py:29-121
//
// All three are launch-latency-bound at the reference's sizes (32x32 scores, 256-point pseudo clouds), so each is a
// single launch over the whole batch: one workgroup for the matcher loop (no host round trips: the reference's
// .nonzero() syncs every iteration), one WAVE per Kabsch problem with the 3x3 SVD done in registers.
#include "ls_common.h"
#include "svd3.h"

namespace ls {

// ---------------------------------------------------------------------------------------------- scores
__global__ __launch_bounds__(256) void cosine_scores_kernel(const float* __restrict__ m0, const float* __restrict__ m1, int n,
                                                            int m, int D, float* __restrict__ inv_norm, float* __restrict__ S,
                                                            int phase) {
    LS_LATENCY_CRITICAL();
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (phase == 0) {  // one wave per row of [m0; m1]: 1 / max(|row|, 1e-12)
        if (w >= n + m) return;
        const float* r = w < n ? m0 + (size_t)w * D : m1 + (size_t)(w - n) * D;
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += r[c] * r[c];
        s = wave_sum(s);
        if (lane == 0) inv_norm[w] = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    } else {           // one wave per score entry
        if (w >= n * m) return;
        const int i = w / m, j = w % m;
        const float* a = m0 + (size_t)i * D;
        const float* b = m1 + (size_t)j * D;
        float ia, ib;
        if (phase == 2) {   // single-launch form (small problems): the entry's wave forms both norms itself -- the same sums in the same order as phase 0
            float sa = 0.f, sb = 0.f;
#pragma unroll 4
            for (int c = lane; c < D; c += 64) { sa += a[c] * a[c]; sb += b[c] * b[c]; }
            sa = wave_sum(sa); sb = wave_sum(sb);
            ia = 1.0f / fmaxf(sqrtf(sa), 1e-12f); ib = 1.0f / fmaxf(sqrtf(sb), 1e-12f);
        } else { ia = inv_norm[i]; ib = inv_norm[n + j]; }
        float s = 0.f;
#pragma unroll 4
        for (int c = lane; c < D; c += 64) s += (a[c] * ia) * (b[c] * ib);
        s = wave_sum(s);
        if (lane == 0) S[w] = s;
    }
}

// ---------------------------------------------------------------------------------------------- greedy loop
// S [n,m] in global (L2-resident), destroyed.  Deleted rows/columns are tracked with alive flags; "first
// row-major arg-max of the shrunken matrix" == lexicographically smallest (row, col) among alive maxima.
__global__ __launch_bounds__(1024) void greedy_match_kernel(float* __restrict__ S, int n, int m, long long* __restrict__ m0,
                                                            long long* __restrict__ m1) {
    extern __shared__ int alive[];  // [n] rows | [m] cols
    __shared__ float redv[16];
    __shared__ int redi[16];
    __shared__ float s_max;
    __shared__ int s_pos;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* ra = alive;
    int* ca = alive + n;
    for (int i = tid; i < n; i += 1024) { ra[i] = 1; m0[i] = -1; }
    for (int j = tid; j < m; j += 1024) { ca[j] = 1; m1[j] = -1; }
    __syncthreads();
    const int total = n * m;
    const int iters = n < m ? n : m;
    for (int it = 0; it < iters; ++it) {
        // max over alive entries
        float mx = -INFINITY;
        for (int e = tid; e < total; e += 1024) {
            const int i = e / m, j = e % m;
            if (ra[i] && ca[j]) mx = fmaxf(mx, S[e]);
        }
        mx = wave_max(mx);
        if (lane == 0) redv[wave] = mx;
        __syncthreads();
        if (tid == 0) { float v = redv[0]; for (int k = 1; k < 16; ++k) v = fmaxf(v, redv[k]); s_max = v; }
        __syncthreads();
        const float denom = s_max + 1e-5f;
        // S /= (max + 1e-5) (matcher_new.py:123), then max of the renormalised matrix
        float mx2 = -INFINITY;
        for (int e = tid; e < total; e += 1024) {
            const int i = e / m, j = e % m;
            if (ra[i] && ca[j]) { const float v = S[e] / denom; S[e] = v; mx2 = fmaxf(mx2, v); }
        }
        mx2 = wave_max(mx2);
        __syncthreads();
        if (lane == 0) redv[wave] = mx2;
        __syncthreads();
        if (tid == 0) { float v = redv[0]; for (int k = 1; k < 16; ++k) v = fmaxf(v, redv[k]); s_max = v; }
        __syncthreads();
        const float target = s_max;
        int pos = INT_MAX;
        for (int e = tid; e < total; e += 1024) {
            const int i = e / m, j = e % m;
            if (ra[i] && ca[j] && S[e] == target) { pos = e; break; }  // e ascending per thread
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) pos = min(pos, __shfl_xor(pos, o, 64));
        if (lane == 0) redi[wave] = pos;
        __syncthreads();
        if (tid == 0) {
            int p = redi[0];
            for (int k = 1; k < 16; ++k) p = min(p, redi[k]);
            s_pos = p;
            if (p != INT_MAX) {
                const int i = p / m, j = p % m;
                m0[i] = j; m1[j] = i;
                ra[i] = 0; ca[j] = 0;
            }
        }
        __syncthreads();
        if (s_pos == INT_MAX) break;  // NaN scores: the reference would raise here
    }
}

// n * m <= 1024 (the reference's scenes: up to 32 x 32): the WHOLE loop in one wave, the matrix in registers (16 entries per lane,
// entry e = lane + 64 k), wave reductions on the DPP network, no LDS and no barrier: 3 barrier-separated block reductions per
// iteration made the 1024-thread kernel above cost 2.4 us per assignment (77 us for 32 x 32).  Same arithmetic, same tie rule.
// wave reductions on the DPP network with the result broadcast through lane 63 (v_readlane): the __shfl_xor forms go through the LDS
// crossbar (ds_bpermute, ~100 cycles each; 18 of them per assignment made the 32 x 32 loop cost 56 us of pure latency)
template <int CTRL, int RM>
__device__ __forceinline__ int gm_dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, RM, 0xF, false); }
__device__ __forceinline__ float gm_wave_max(float v) {
    v = fmaxf(v, __int_as_float(gm_dpp<0xB1, 0xF>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(gm_dpp<0x4E, 0xF>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(gm_dpp<0x141, 0xF>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(gm_dpp<0x140, 0xF>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(gm_dpp<0x142, 0xA>(__float_as_int(v))));   // row_bcast15 -> rows 1, 3
    v = fmaxf(v, __int_as_float(gm_dpp<0x143, 0xC>(__float_as_int(v))));   // row_bcast31 -> rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int gm_wave_min(int v) {
    v = min(v, gm_dpp<0xB1, 0xF>(v));
    v = min(v, gm_dpp<0x4E, 0xF>(v));
    v = min(v, gm_dpp<0x141, 0xF>(v));
    v = min(v, gm_dpp<0x140, 0xF>(v));
    v = min(v, gm_dpp<0x142, 0xA>(v));
    v = min(v, gm_dpp<0x143, 0xC>(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__global__ __launch_bounds__(64) void greedy_match_wave_kernel(const float* __restrict__ S, int n, int m, long long* __restrict__ m0,
                                                              long long* __restrict__ m1) {
    LS_LATENCY_CRITICAL();
    const int lane = threadIdx.x;
    const int total = n * m;
    float v[16];
    int ri[16], ci[16];
    unsigned alive = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int e = lane + 64 * k;
        const bool in = e < total;
        v[k] = in ? S[e] : 0.f;
        ri[k] = in ? e / m : -1;
        ci[k] = in ? e - (e / m) * m : -1;
        if (in) alive |= 1u << k;
    }
    for (int i = lane; i < n; i += 64) m0[i] = -1;
    for (int j = lane; j < m; j += 64) m1[j] = -1;
    const int iters = n < m ? n : m;
    for (int it = 0; it < iters; ++it) {
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (alive >> k & 1) mx = fmaxf(mx, v[k]);
        mx = gm_wave_max(mx);
        const float denom = mx + 1e-5f;            // S /= (max + 1e-5)   (matcher_new.py:123)
        float mx2 = -INFINITY;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (alive >> k & 1) { v[k] = v[k] / denom; mx2 = fmaxf(mx2, v[k]); }
        mx2 = gm_wave_max(mx2);
        int pos = INT_MAX;                         // first row-major position holding the maximum: smallest e
#pragma unroll
        for (int k = 15; k >= 0; --k) if ((alive >> k & 1) && v[k] == mx2) pos = lane + 64 * k;
        pos = gm_wave_min(pos);
        if (pos == INT_MAX) break;                 // NaN scores: the reference would raise here
        const int r = pos / m, c = pos - r * m;
        if (lane == 0) { m0[r] = c; m1[c] = r; }
#pragma unroll
        for (int k = 0; k < 16; ++k) if (ri[k] == r || ci[k] == c) alive &= ~(1u << k);
    }
}

// n * m <= 1024 on FOUR waves (round 5): four entries per lane instead of sixteen.  The one-wave kernel above executes ~1000 instructions per assignment
// (sixteen IEEE divisions behind per-lane alive masks: 71 branches) = 1.9 us each, 62 us for 32 x 32 at the end of every encode + match + register step.
// Here: branch-free selects, four divisions per lane, and TWO workgroup reductions per assignment instead of three -- the maximum of the divided matrix is
// the divided maximum (a correctly rounded division by one positive denominator is monotone), so only the matrix maximum and the first position that
// holds the divided maximum are reduced; a non-positive or non-finite denominator (all scores <= -1e-5) takes the three-reduction form.  The matches are
// collected in LDS and written once.  Same arithmetic on every entry, same tie rule: identical matches (tests/test_hip_parity.py, test_hip_surface.py).
__global__ __launch_bounds__(256) void greedy_match_quad_kernel(const float* __restrict__ S, int n, int m, long long* __restrict__ m0,
                                                                long long* __restrict__ m1) {
    LS_LATENCY_CRITICAL();
    __shared__ float lmx[4], lmx2[4];
    __shared__ int lpos[4];
    __shared__ int lm[2048];     // matches of the rows | of the columns (n, m <= 1024)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = n * m;
    float v[4];
    int ri[4], ci[4];
    bool alive[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = tid + 256 * k;
        const bool in = e < total;
        v[k] = in ? S[e] : 0.f;
        ri[k] = in ? e / m : -1;
        ci[k] = in ? e - (e / m) * m : -1;
        alive[k] = in;
    }
    for (int i = tid; i < 2048; i += 256) lm[i] = -1;
    __syncthreads();
    const int iters = n < m ? n : m;
    for (int it = 0; it < iters; ++it) {
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k) mx = fmaxf(mx, alive[k] ? v[k] : -INFINITY);
        mx = gm_wave_max(mx);
        if (lane == 0) lmx[wave] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(lmx[0], lmx[1]), fmaxf(lmx[2], lmx[3]));
        const float denom = mx + 1e-5f;            // S /= (max + 1e-5)   (matcher_new.py:123)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = alive[k] ? v[k] / denom : v[k];
        float mx2;
        if (denom > 0.f && denom < INFINITY) mx2 = mx / denom;     // (workgroup-uniform)
        else {
            mx2 = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4; ++k) mx2 = fmaxf(mx2, alive[k] ? v[k] : -INFINITY);
            mx2 = gm_wave_max(mx2);
            if (lane == 0) lmx2[wave] = mx2;
            __syncthreads();
            mx2 = fmaxf(fmaxf(lmx2[0], lmx2[1]), fmaxf(lmx2[2], lmx2[3]));
        }
        int pos = INT_MAX;                         // first row-major position holding the maximum: smallest e
#pragma unroll
        for (int k = 3; k >= 0; --k) pos = (alive[k] && v[k] == mx2) ? tid + 256 * k : pos;
        pos = gm_wave_min(pos);
        if (lane == 0) lpos[wave] = pos;
        __syncthreads();
        pos = min(min(lpos[0], lpos[1]), min(lpos[2], lpos[3]));
        if (pos == INT_MAX) break;                 // NaN scores: the reference would raise here
        const int r = pos / m, c = pos - r * m;
        if (tid == 0) { lm[r] = c; lm[1024 + c] = r; }
#pragma unroll
        for (int k = 0; k < 4; ++k) alive[k] = alive[k] && ri[k] != r && ci[k] != c;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) m0[i] = lm[i];
    for (int j = tid; j < m; j += 256) m1[j] = lm[1024 + j];
}

// ---------------------------------------------------------------------------------------------- Kabsch
// one wave per problem; problem p pairs cloud x1[i1(p)] with x2[i2(p)]:
//   pair_mode 0: i1 = i2 = p (batched Kabsch);  pair_mode 1: p = i*m + j -> (i, j) (residual matrix)
__global__ __launch_bounds__(256) void kabsch_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                     const float* __restrict__ weights, int nprob, int n, int pair_m,
                                                     int raw_weights, float eps, float* __restrict__ Rout, float* __restrict__ tout,
                                                     float* __restrict__ res, float* __restrict__ res_mean,
                                                     int32_t* __restrict__ flags, const float* __restrict__ off1, const float* __restrict__ off2,
                                                     const long long* __restrict__ sel1, const long long* __restrict__ sel2) {
    LS_LATENCY_CRITICAL();
    // off1 / off2 (nullable) [*,3]: the point sets are x + off (More_Solver's pseudo-points z_so3 + t, more_solver.py:114-116, formed
    // here instead of by a separate element-wise launch); sel1 / sel2 (nullable) [nprob] int64: problem p reads set sel[p] (a negative
    // entry -- an unmatched row of matches0 -- reads set 0, as matches0.clamp(min=0) does)
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= nprob) return;
    int i1 = pair_m > 0 ? p / pair_m : p, i2 = pair_m > 0 ? p % pair_m : p;
    if (sel1) i1 = (int)max(sel1[p], 0ll);
    if (sel2) i2 = (int)max(sel2[p], 0ll);
    const float* a0 = x1 + (size_t)i1 * n * 3;
    const float* b0 = x2 + (size_t)i2 * n * 3;
    const float oa[3] = {off1 ? off1[i1 * 3] : 0.f, off1 ? off1[i1 * 3 + 1] : 0.f, off1 ? off1[i1 * 3 + 2] : 0.f};
    const float ob[3] = {off2 ? off2[i2 * 3] : 0.f, off2 ? off2[i2 * 3 + 1] : 0.f, off2 ? off2[i2 * 3 + 2] : 0.f};
    const bool shifted = off1 || off2;
    // a(i, x) / b(i, x): the point coordinates (x + off rounded to fp32 first, exactly what the separate add produced)
    auto A_ = [&](int i, int x) { const float v = a0[i * 3 + x]; return shifted ? __fadd_rn(v, oa[x]) : v; };
    auto B_ = [&](int i, int x) { const float v = b0[i * 3 + x]; return shifted ? __fadd_rn(v, ob[x]) : v; };
    const float* w = weights ? weights + (size_t)p * n : nullptr;
    // weights / (sum + eps)   (pose_estimation.py:52-54); raw_weights: the caller already applied :52-66 (normalisation,
    // best_k selection, w_threshold zeroing WITHOUT renormalising) and the weights are used as they are
    // Up to 256 points per problem (the codes' 256 pseudo-points): every point of the lane is fetched ONCE, up front, into registers -- the three passes below
    // (sums, covariance, residuals) were rolled loops of loads, i.e. one L2 round trip per 64 points and pass (round 5: 15 -> ~8 us at the end of every
    // encode + match + register step).  Same per-lane order (i = lane, lane + 64, ...), same arithmetic: identical results.  More points: the loops.
    const bool small = n <= 256;
    float pa[4][3], pb[4][3], pw[4];
    if (small) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = lane + 64 * u, ii = i < n ? i : 0;
#pragma unroll
            for (int x = 0; x < 3; ++x) { pa[u][x] = A_(ii, x); pb[u][x] = B_(ii, x); }
            pw[u] = w ? w[ii] : 1.0f;
        }
    }
#define LS_KB_POINTS(...)                                                                                                      \
    if (small) {                                                                                                               \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                                                        \
            const int i = lane + 64 * u;                                                                                       \
            if (i < n) { const float ax = pa[u][0], ay = pa[u][1], az = pa[u][2], bx = pb[u][0], by = pb[u][1], bz = pb[u][2], wraw = pw[u]; __VA_ARGS__ }  \
        }                                                                                                                      \
    } else {                                                                                                                   \
        for (int i = lane; i < n; i += 64) {                                                                                   \
            const float ax = A_(i, 0), ay = A_(i, 1), az = A_(i, 2), bx = B_(i, 0), by = B_(i, 1), bz = B_(i, 2), wraw = w ? w[i] : 1.0f; __VA_ARGS__ \
        }                                                                                                                      \
    }
    float sw = 0.f;
    LS_KB_POINTS({ (void)ax; (void)ay; (void)az; (void)bx; (void)by; (void)bz; sw += wraw; })
    sw = (raw_weights && w) ? 1.0f : wave_sum(sw) + eps;
    // weighted means, divided by (sum of normalised weights + eps)  (:68-69)
    float m1[3] = {0, 0, 0}, m2[3] = {0, 0, 0}, swn = 0.f;
    LS_KB_POINTS({
        const float wi = wraw / sw;
        swn += wi;
        m1[0] += wi * ax; m2[0] += wi * bx; m1[1] += wi * ay; m2[1] += wi * by; m1[2] += wi * az; m2[2] += wi * bz;
    })
    swn = wave_sum(swn) + eps;
#pragma unroll
    for (int x = 0; x < 3; ++x) { m1[x] = wave_sum(m1[x]) / swn; m2[x] = wave_sum(m2[x]) / swn; }
    // covariance H = sum_i w_i (x1_i - mu1)(x2_i - mu2)^T   (:71-77)
    float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    LS_KB_POINTS({
        const float wi = wraw / sw;
        const float c1[3] = {ax - m1[0], ay - m1[1], az - m1[2]}, c2[3] = {bx - m2[0], by - m2[1], bz - m2[2]};
        _Pragma("unroll") for (int r = 0; r < 3; ++r)
            _Pragma("unroll") for (int c = 0; c < 3; ++c) H[r * 3 + c] += wi * c1[r] * c2[c];
    })
    double Hd[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) Hd[e] = (double)wave_sum(H[e]);
    float R[9];
    const int code = kabsch_rotation(Hd, R);
    float t[3];
    if (code == LS_KABSCH_NONFINITE) {  // the reference's SVD-failure branch (:79-88): identity rotation, zero translation
        t[0] = t[1] = t[2] = 0.f;
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) t[r] = m2[r] - (R[r * 3] * m1[0] + R[r * 3 + 1] * m1[1] + R[r * 3 + 2] * m1[2]);
    }
    if (lane == 0) {
        if (Rout) for (int e = 0; e < 9; ++e) Rout[(size_t)p * 9 + e] = R[e];
        if (tout) for (int e = 0; e < 3; ++e) tout[(size_t)p * 3 + e] = t[e];
        if (flags) flags[p] = code;
    }
    // residuals |R x1 + t - x2|   (:105-121)
    float rs = 0.f;
    LS_KB_POINTS({
        (void)wraw;
        const float bb[3] = {bx, by, bz};
        float e2 = 0.f;
        _Pragma("unroll") for (int r = 0; r < 3; ++r) {
            const float d = R[r * 3] * ax + R[r * 3 + 1] * ay + R[r * 3 + 2] * az + t[r] - bb[r];
            e2 += d * d;
        }
        const float e = sqrtf(e2);
        if (res) res[(size_t)p * n + i] = e;
        rs += e;
    })
#undef LS_KB_POINTS
    if (res_mean) { rs = wave_sum(rs); if (lane == 0) res_mean[p] = rs / (float)n; }
}

int cosine_scores_launch(const float* m0, const float* m1, int n, int m, int D, float* inv_norm_ws, float* S, hipStream_t st) {
    if ((long long)n * m <= 4096) {   // latency-bound: one launch, every entry's wave recomputes its two row norms (bit-identical to the two-launch form)
        hipLaunchKernelGGL(cosine_scores_kernel, dim3(cdiv((long long)n * m, 4)), dim3(256), 0, st, m0, m1, n, m, D, inv_norm_ws, S, 2);
        LS_LAUNCH_CHECK();
        return LS_OK;
    }
    hipLaunchKernelGGL(cosine_scores_kernel, dim3(cdiv(n + m, 4)), dim3(256), 0, st, m0, m1, n, m, D, inv_norm_ws, S, 0);
    hipLaunchKernelGGL(cosine_scores_kernel, dim3(cdiv((long long)n * m, 4)), dim3(256), 0, st, m0, m1, n, m, D, inv_norm_ws, S, 1);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int greedy_match_launch(float* S, int n, int m, long long* m0, long long* m1, hipStream_t st) {
    if ((long long)n * m <= 1024) {
        static const bool one_wave = dev_knob("LS_GREEDY_ONE_WAVE", 0) != 0;   // dev A/B: the round-3 kernel
        if (one_wave || n * m <= 64) hipLaunchKernelGGL(greedy_match_wave_kernel, dim3(1), dim3(64), 0, st, S, n, m, m0, m1);
        else hipLaunchKernelGGL(greedy_match_quad_kernel, dim3(1), dim3(256), 0, st, S, n, m, m0, m1);
        LS_LAUNCH_CHECK();
        return LS_OK;
    }
    LS_REQUIRE((size_t)(n + m) * sizeof(int) <= 48 * 1024, "greedy_match: n+m=%d too large", n + m);
    hipLaunchKernelGGL(greedy_match_kernel, dim3(1), dim3(1024), (size_t)(n + m) * sizeof(int), st, S, n, m, m0, m1);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
// ---------------------------------------------------------------------------------------------- mutual-NN and Sinkhorn assignment
// nn_matcher            /root/reference/lib_more/matcher_new.py:85-107   (find_nn without thresholds, two mutual checks)
// sinkhorn_matcher      matcher_new.py:20-71                             (SuperGlue's log-space optimal transport with a dustbin row / column, 100
//                                                                         iterations, then mutual arg-maxes of the inner block and the exp(score) > threshold test)
// Round 5 ran these as ATen kernels: the Sinkhorn loop alone was 200 DEPENDENT logsumexp launches (+ adds), 2 - 4 ms for a 32 x 32 problem.  Here the
// whole matcher is ONE launch of one workgroup: the coupling matrix lives in LDS ((n+1) x (m+1), row stride odd: the column passes are conflict-free),
// eight lanes share a row (column), one barrier per half-iteration.  Ties: the FIRST maximum (lowest index), as torch.max / topk return on the reference's
// devices for the sizes in question.  logsumexp as ATen computes it: max, log(sum exp(x - max)) + max (an all -inf row keeps max = 0).
// mode 0 = mutual nearest neighbours on `S`; mode 1 = Sinkhorn on S / div.
__global__ __launch_bounds__(1024) void assign_kernel(const float* __restrict__ S, int n, int m, int mode, float div, float alpha, int iters, float thr,
                                                      long long* __restrict__ m0, long long* __restrict__ m1) {
    LS_LATENCY_CRITICAL();
    extern __shared__ float az[];
    const int tid = threadIdx.x, g = tid & 7, grp = tid >> 3;
    const int R = mode ? n + 1 : n, C = mode ? m + 1 : m, ld = C | 1;
    float* Z = az;                     // [R][ld]
    float* u = Z + (size_t)R * ld;     // [R]
    float* v = u + R;                  // [C]
    float* mx0 = v + C;                // [n]   row maxima of the inner block
    int* i0 = (int*)(mx0 + n);         // [n]
    int* i1 = i0 + n;                  // [m]
    int* ok0 = i1 + m;                 // [n]
    for (int e = tid; e < R * C; e += 1024) {
        const int i = e / C, j = e - i * C;
        Z[i * ld + j] = (i < n && j < m) ? (mode ? S[(size_t)i * m + j] / div : S[(size_t)i * m + j]) : alpha;
    }
    for (int i = tid; i < R; i += 1024) u[i] = 0.f;
    for (int j = tid; j < C; j += 1024) v[j] = 0.f;
    __syncthreads();
    float norm = 0.f;
    if (mode) {
        norm = -logf((float)n + (float)m);
        const float mu_bin = logf((float)m) + norm, nu_bin = logf((float)n) + norm;
        for (int it = 0; it < iters; ++it) {
            // u = log_mu - logsumexp_j(Z + v)
            for (int r = grp; r < R; r += 128) {
                const float* zr = Z + r * ld;
                float mx = -INFINITY;
                for (int j = g; j < C; j += 8) mx = fmaxf(mx, zr[j] + v[j]);
                mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
                const float mf = (mx == INFINITY || mx == -INFINITY) ? 0.f : mx;
                float sm = 0.f;
                for (int j = g; j < C; j += 8) sm += expf(zr[j] + v[j] - mf);
                sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
                if (g == 0) u[r] = (r < n ? norm : mu_bin) - (logf(sm) + mf);
            }
            __syncthreads();
            // v = log_nu - logsumexp_i(Z + u)
            for (int c = grp; c < C; c += 128) {
                float mx = -INFINITY;
                for (int i = g; i < R; i += 8) mx = fmaxf(mx, Z[i * ld + c] + u[i]);
                mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64)); mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
                const float mf = (mx == INFINITY || mx == -INFINITY) ? 0.f : mx;
                float sm = 0.f;
                for (int i = g; i < R; i += 8) sm += expf(Z[i * ld + c] + u[i] - mf);
                sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
                if (g == 0) v[c] = (c < m ? norm : nu_bin) - (logf(sm) + mf);
            }
            __syncthreads();
        }
    }
    // arg-maxes of the inner block of  Z + u + v - norm  (mode 0: of S), first maximum on ties (a NaN entry is never selected)
    for (int r = grp; r < n; r += 128) {
        float best = -INFINITY; int bj = 0x7fffffff;
        for (int j = g; j < m; j += 8) {
            const float x = mode ? ((Z[r * ld + j] + u[r]) + v[j]) - norm : Z[r * ld + j];
            if (x > best || (x == best && j < bj)) { best = x; bj = j; }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oj = __shfl_xor(bj, o, 64);
            if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
        }
        if (g == 0) { i0[r] = bj == 0x7fffffff ? 0 : bj; mx0[r] = best; }
    }
    for (int c = grp; c < m; c += 128) {
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int i = g; i < n; i += 8) {
            const float x = mode ? ((Z[i * ld + c] + u[i]) + v[c]) - norm : Z[i * ld + c];
            if (x > best || (x == best && i < bi)) { best = x; bi = i; }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (g == 0) i1[c] = bi == 0x7fffffff ? 0 : bi;
    }
    __syncthreads();
    // mode 0 (matcher_new.py:93-94): matches0 = mutual_check(i0, i1); matches1 = mutual_check(i1, matches0)
    // mode 1 (:58-66): valid0 = mutual0 & exp(max0) > thr; valid1 = mutual1 & valid0[i1]
    for (int i = tid; i < n; i += 1024) {
        const bool mutual = i1[i0[i]] == i;
        const bool ok = mode ? (mutual && expf(mx0[i]) > thr) : mutual;
        ok0[i] = ok;
        m0[i] = ok ? i0[i] : -1;
    }
    __syncthreads();
    for (int j = tid; j < m; j += 1024) {
        const int i = i1[j];
        const bool ok = mode ? (i0[i] == j && ok0[i]) : (ok0[i] && i0[i] == j);
        m1[j] = ok ? i : -1;
    }
}

size_t assign_lds_bytes(int n, int m, int mode) {
    const size_t R = mode ? n + 1 : n, C = mode ? m + 1 : m, ld = C | 1;
    return (R * ld + R + C + (size_t)n + 2 * (size_t)n + (size_t)m) * 4;
}
int assign_launch(const float* S, int n, int m, int mode, float div, float alpha, int iters, float thr, long long* m0, long long* m1, hipStream_t st) {
    const size_t lds = assign_lds_bytes(n, m, mode);
    LS_REQUIRE(lds <= 150 * 1024, "%s: a %d x %d problem needs %zu bytes of LDS (at most 153600: about 190 x 190)", mode ? "sinkhorn_match" : "nn_match", n, m, lds);
    if (lds > 64 * 1024) LS_HIP_CHECK(hipFuncSetAttribute((const void*)assign_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(assign_kernel, dim3(1), dim3(1024), lds, st, S, n, m, mode, div, alpha, iters, thr, m0, m1);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int kabsch_launch(const float* x1, const float* x2, const float* w, int nprob, int n, int pair_m, int raw_weights, float* R, float* t,
                  float* res, float* res_mean, int32_t* flags, hipStream_t st, const float* off1, const float* off2, const long long* sel1,
                  const long long* sel2) {
    hipLaunchKernelGGL(kabsch_kernel, dim3(cdiv(nprob, 4)), dim3(256), 0, st, x1, x2, w, nprob, n, pair_m, raw_weights, 1e-7f, R, t, res,
                       res_mean, flags, off1, off2, sel1, sel2);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

}  // namespace ls
