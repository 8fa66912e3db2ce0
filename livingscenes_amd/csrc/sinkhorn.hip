// sinkhorn.hip -- the log-domain "softmin" of entropic optimal transport between two small point clouds, and its gradient:
//     out_i  = -eps * log sum_j exp( h_j - C(x_i, y_j) / eps ),      C(x, y) = |x - y|^2 / 2
//     grad_i = sum_j softmax_j( h_j - C(x_i, y_j) / eps ) (x_i - y_j)                     (= d out_i / d x_i)
// This is the one primitive of the debiased Sinkhorn divergence the reference evaluates with
//     geomloss.SamplesLoss(loss='sinkhorn', p=2)                      /root/reference/lib_more/more_solver.py:146,158
// between the 1024 transformed source points and the 1024 target points inside the optimisation-based registration loop
// (registration half).  geomloss is neither vendored nor installed: the epsilon-scaling loop that calls this
// primitive (livingscenes_amd/sinkhorn.py) restates its published algorithm from memory -- PARITY UNPINNED.
//
// One wave per row i: the M columns are strided over the 64 lanes with an online (max, sum) pair per lane, combined by a wave
// reduction; the gradient is accumulated un-normalised against the running max and rescaled whenever the max moves.
#include "ls_common.h"

namespace ls {

__global__ __launch_bounds__(256) void softmin_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ h,
                                                      int N, int M, float eps, float* __restrict__ out, float* __restrict__ grad) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;
    const float xi = x[i * 3], yi = x[i * 3 + 1], zi = x[i * 3 + 2];
    const float inv = 1.0f / eps;
    float mx = -INFINITY, sum = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
    for (int j = lane; j < M; j += 64) {
        const float dx = xi - y[j * 3], dy = yi - y[j * 3 + 1], dz = zi - y[j * 3 + 2];
        const float v = h[j] - 0.5f * (dx * dx + dy * dy + dz * dz) * inv;
        if (v > mx) {
            const float sc = __expf(mx - v);   // exp(-inf) = 0 on the first hit
            sum *= sc; gx *= sc; gy *= sc; gz *= sc;
            mx = v;
        }
        const float e = __expf(v - mx);
        sum += e; gx += e * dx; gy += e * dy; gz += e * dz;
    }
    // combine the lanes: rescale everything to the wave maximum
    const float wmx = wave_max(mx);
    const float sc = mx == -INFINITY ? 0.f : __expf(mx - wmx);
    sum = wave_sum(sum * sc);
    if (grad) { gx = wave_sum(gx * sc); gy = wave_sum(gy * sc); gz = wave_sum(gz * sc); }
    if (lane == 0) {
        out[i] = -eps * (wmx + __logf(sum));
        if (grad) { grad[i * 3] = gx / sum; grad[i * 3 + 1] = gy / sum; grad[i * 3 + 2] = gz / sum; }
    }
}

}  // namespace ls

using namespace ls;

extern "C" int ls_sinkhorn_softmin_f32(const float* x, const float* y, const float* h, int N, int M, float eps, float* out, float* grad_x,
                                       void* stream) {
    LS_REQUIRE(x && y && h && out, "sinkhorn_softmin: null argument");
    LS_REQUIRE(N > 0 && M > 0 && eps > 0.f, "sinkhorn_softmin: empty problem or non-positive epsilon");
    hipLaunchKernelGGL(softmin_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, x, y, h, N, M, eps, out, grad_x);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
