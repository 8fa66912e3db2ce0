// icp.hip -- point-to-point ICP refinement, one 1024-thread workgroup per registration problem, the whole
// iteration loop on device (no per-iteration host sync; the reference syncs every iteration to test convergence).
//
// Replaces pytorch3d.ops.iterative_closest_point(pc1, pc2, init_transform=SimilarityTransform(R^T, t, 1)) as called
// at /root/reference/lib_more/more_solver.py:182-187 (pytorch3d 0.7.4 is un-vendored: semantics restated in
// oracle/more.py::iterative_closest_point -- PARITY UNPINNED).  Row-vector convention Xt = X R + T.
// Per iteration: 1-NN of every Xt point in Y (canonical squared distance, first minimum), rigid alignment of the
// ORIGINAL X onto the matched points (means, 3x3 covariance / n, SVD, det fix), rmse, relative-improvement stop.
// Each problem stops on its own criterion (a batched call == independent batch-1 reference calls).
#include "ls_common.h"
#include "svd3.h"

namespace ls {

constexpr int ICP_PT = 4;  // source points per thread (n <= 4096)

template <bool FMA>
__device__ __forceinline__ float d3(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    float d = 0.f;
    if constexpr (FMA) {
        d = __builtin_fmaf(dx, dx, d); d = __builtin_fmaf(dy, dy, d); d = __builtin_fmaf(dz, dz, d);
    } else {
        float p = dx * dx; d = d + p; p = dy * dy; d = d + p; p = dz * dz; d = d + p;
    }
    return d;
}

__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k];
    return s;
}

template <bool FMA>
__global__ __launch_bounds__(1024) void icp_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                   const float* __restrict__ R0, const float* __restrict__ T0, int n, int m,
                                                   int max_iter, float thr, float* __restrict__ Rout, float* __restrict__ Tout,
                                                   float* __restrict__ rmse_out, int32_t* __restrict__ iters_out) {
    extern __shared__ __attribute__((aligned(16))) float ly[];  // Y [m][3]
    __shared__ float red[16];
    __shared__ float sR[9], sT[3];
    __shared__ int s_stop;
    const int p = blockIdx.x, tid = threadIdx.x;
    const float* Xp = X + (size_t)p * n * 3;
    const float* Yp = Y + (size_t)p * m * 3;
    for (int t = tid; t < m * 3; t += 1024) ly[t] = Yp[t];
    if (tid < 9) sR[tid] = R0[(size_t)p * 9 + tid];
    if (tid < 3) sT[tid] = T0[(size_t)p * 3 + tid];
    if (tid == 0) s_stop = 0;
    __syncthreads();
    float prev = -1.f, rmse = 0.f;
    int it = 0;
    const float invn = 1.0f / (float)n;
    // mean of X is iteration independent
    float mx[3] = {0, 0, 0};
    for (int i = tid; i < n; i += 1024) { mx[0] += Xp[i * 3]; mx[1] += Xp[i * 3 + 1]; mx[2] += Xp[i * 3 + 2]; }
#pragma unroll
    for (int a = 0; a < 3; ++a) mx[a] = block_sum_1024(mx[a], red) * invn;

    for (it = 0; it < max_iter; ++it) {
        float R[9], T[3];
#pragma unroll
        for (int e = 0; e < 9; ++e) R[e] = sR[e];
#pragma unroll
        for (int e = 0; e < 3; ++e) T[e] = sT[e];
        // pass 1: nearest neighbours of Xt in Y (kept in registers), mean of the matched points
        int match[ICP_PT];
        float sy[3] = {0, 0, 0};
#pragma unroll
        for (int u = 0; u < ICP_PT; ++u) {
            const int i = tid + u * 1024;
            match[u] = 0;
            if (i < n) {
                const float x0 = Xp[i * 3], x1 = Xp[i * 3 + 1], x2 = Xp[i * 3 + 2];
                const float q0 = x0 * R[0] + x1 * R[3] + x2 * R[6] + T[0];
                const float q1 = x0 * R[1] + x1 * R[4] + x2 * R[7] + T[1];
                const float q2 = x0 * R[2] + x1 * R[5] + x2 * R[8] + T[2];
                float best = INFINITY;
                int bj = 0;
                for (int j = 0; j < m; ++j) {
                    const float d = d3<FMA>(q0, q1, q2, ly[j * 3], ly[j * 3 + 1], ly[j * 3 + 2]);
                    if (d < best) { best = d; bj = j; }
                }
                match[u] = bj;
                sy[0] += ly[bj * 3]; sy[1] += ly[bj * 3 + 1]; sy[2] += ly[bj * 3 + 2];
            }
        }
        float my[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) my[a] = block_sum_1024(sy[a], red) * invn;
        // pass 2: cov = Xc^T Yc / n
        float sxy[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < ICP_PT; ++u) {
            const int i = tid + u * 1024;
            if (i < n) {
                const int bj = match[u];
                const float c0 = Xp[i * 3] - mx[0], c1 = Xp[i * 3 + 1] - mx[1], c2 = Xp[i * 3 + 2] - mx[2];
                const float y0 = ly[bj * 3] - my[0], y1 = ly[bj * 3 + 1] - my[1], y2 = ly[bj * 3 + 2] - my[2];
                sxy[0] += c0 * y0; sxy[1] += c0 * y1; sxy[2] += c0 * y2;
                sxy[3] += c1 * y0; sxy[4] += c1 * y1; sxy[5] += c1 * y2;
                sxy[6] += c2 * y0; sxy[7] += c2 * y1; sxy[8] += c2 * y2;
            }
        }
        double H[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) H[e] = (double)(block_sum_1024(sxy[e], red) * invn);
        float Rk[9];
        const bool ok = kabsch_rotation(H, Rk) == 0;   // rank-deficient covariance: keep the previous rotation
        float Rn[9], Tn[3];
        if (ok) {
            // pytorch3d row convention: R = U E V^T = (V E U^T)^T
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Rn[r * 3 + c] = Rk[c * 3 + r];
        } else {
#pragma unroll
            for (int e = 0; e < 9; ++e) Rn[e] = R[e];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) Tn[c] = my[c] - (mx[0] * Rn[c] + mx[1] * Rn[3 + c] + mx[2] * Rn[6 + c]);
        // pass 3: rmse of the updated transform against the same matches
        float se = 0.f;
#pragma unroll
        for (int u = 0; u < ICP_PT; ++u) {
            const int i = tid + u * 1024;
            if (i < n) {
                const int bj = match[u];
                const float x0 = Xp[i * 3], x1 = Xp[i * 3 + 1], x2 = Xp[i * 3 + 2];
                const float e0 = x0 * Rn[0] + x1 * Rn[3] + x2 * Rn[6] + Tn[0] - ly[bj * 3];
                const float e1 = x0 * Rn[1] + x1 * Rn[4] + x2 * Rn[7] + Tn[1] - ly[bj * 3 + 1];
                const float e2 = x0 * Rn[2] + x1 * Rn[5] + x2 * Rn[8] + Tn[2] - ly[bj * 3 + 2];
                se += e0 * e0 + e1 * e1 + e2 * e2;
            }
        }
        rmse = sqrtf(block_sum_1024(se, red) * invn);
        __syncthreads();
        if (tid == 0) {
            for (int e = 0; e < 9; ++e) sR[e] = Rn[e];
            for (int e = 0; e < 3; ++e) sT[e] = Tn[e];
            int stop = 0;
            if (prev >= 0.f) { const float rel = (prev - rmse) / prev; if (rel <= thr) stop = 1; }
            s_stop = stop;
        }
        __syncthreads();
        prev = rmse;
        if (s_stop) { ++it; break; }
    }
    if (tid == 0) {
        for (int e = 0; e < 9; ++e) Rout[(size_t)p * 9 + e] = sR[e];
        for (int e = 0; e < 3; ++e) Tout[(size_t)p * 3 + e] = sT[e];
        if (rmse_out) rmse_out[p] = rmse;
        if (iters_out) iters_out[p] = it;
    }
}

size_t icp_workspace_bytes(int, int) { return 256; }

int icp_run(const float* X, const float* Y, const float* R0, const float* T0, int b, int n, int m, int max_iter, float thr,
            unsigned flags, float* R, float* T, float* rmse, int32_t* iters_out, void*, size_t, hipStream_t st) {
    LS_REQUIRE(b > 0 && n > 0 && m > 0 && max_iter > 0, "icp: empty problem");
    LS_REQUIRE(n <= 1024 * ICP_PT, "icp: source cloud too large (n=%d, max %d)", n, 1024 * ICP_PT);
    LS_REQUIRE((size_t)m * 12 <= 120 * 1024, "icp: target cloud too large for LDS (m=%d, max 10240)", m);
    const size_t smem = (size_t)m * 3 * sizeof(float);
    if (flags & LS_FLAG_CONTRACT_FMA)
        hipLaunchKernelGGL(icp_kernel<true>, dim3(b), dim3(1024), smem, st, X, Y, R0, T0, n, m, max_iter, thr, R, T, rmse, iters_out);
    else
        hipLaunchKernelGGL(icp_kernel<false>, dim3(b), dim3(1024), smem, st, X, Y, R0, T0, n, m, max_iter, thr, R, T, rmse, iters_out);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

}  // namespace ls
