// optim.hip -- device side of the optimisation-based registration, BATCHED over pairs:
// More_Solver._solve_pairwise_registration(optim=True), /root/reference/lib_more/more_solver.py:118-189 -- what eval_3rscan.py:381
// actually runs.  The reference advances ONE pair per call through 400 dependent Adam steps, each a decoder forward + backward and
// a Sinkhorn divergence with several host round trips; here P pairs advance in lock-step and the only host work per step is the
// launch sequence:
//   se3_transform_kernel      query = g . src                                                      (:149 pose.transform(src_pc))
//   [ls_sdf_decode_train / ls_sdf_backward on P x N queries: sdf.hip / gemm.hip]                    (:152-153 decoder, :160 backward)
//   smooth_l1_kernel          per-pair SmoothL1(sdf, 0) (mean) and its gradient                     (:152-156 loss_fn = SmoothL1Loss)
//   softmin_batched_kernel    the log-domain softmins of the debiased Sinkhorn divergence           (:146,158 geomloss SamplesLoss)
//   se3_adam_step_kernel      tangent gradient, Adam moments, g <- exp(-step) g, best-loss snapshot, geodesic early stop,
//                             and the NEXT step's transformed source cloud                          (:160-173)
// torchlie (LieTensor SE3 parameter), geomloss and roma are neither vendored nor installed: the retraction and the Sinkhorn loop are
// this build's documented definitions -- PARITY UNPINNED for them; the decoder gradients are pinned (sdf.hip).
// Every reduction runs in a fixed order (no atomics): a pair's trajectory does not depend on which other pairs share the launch.
#include "ls_common.h"
#include <algorithm>

namespace ls {

__device__ __forceinline__ float block_sum_256_opt(float v, float* red) {   // fixed-order block reduction, result in every thread
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// query[p,i] = R_p src[p,i] + t_p      g [P,3,4] row-major (R | t)
__global__ __launch_bounds__(256) void se3_transform_kernel(const float* __restrict__ g, const float* __restrict__ src, int N, float* __restrict__ q) {
    const int p = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float* G = g + p * 12;
    const float* s = src + ((size_t)p * N + i) * 3;
    float* o = q + ((size_t)p * N + i) * 3;
    const float x = s[0], y = s[1], z = s[2];
    o[0] = G[0] * x + G[1] * y + G[2] * z + G[3];
    o[1] = G[4] * x + G[5] * y + G[6] * z + G[7];
    o[2] = G[8] * x + G[9] * y + G[10] * z + G[11];
}

// loss[p] (+)= mean_i smooth_l1(sdf[p,i]) (beta = 1: 0.5 x^2 for |x| < 1, |x| - 0.5 otherwise; torch.nn.SmoothL1Loss defaults);
// grad[p,i] = d loss[p] / d sdf[p,i]
__global__ __launch_bounds__(256) void smooth_l1_kernel(const float* __restrict__ sdf, int N, int accumulate, float* __restrict__ loss,
                                                        float* __restrict__ grad) {
    __shared__ float red[4];
    const int p = blockIdx.x;
    const float inv = 1.0f / (float)N;
    float acc = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float x = sdf[(size_t)p * N + i], ax = fabsf(x);
        acc += ax < 1.0f ? 0.5f * x * x : ax - 0.5f;
        grad[(size_t)p * N + i] = (ax < 1.0f ? x : (x > 0.f ? 1.0f : -1.0f)) * inv;
    }
    acc = block_sum_256_opt(acc, red);
    if (threadIdx.x == 0) loss[p] = (accumulate ? loss[p] : 0.f) + acc * inv;
}

// loss[p] = mean_i sdf[p,i]^2 (torch.nn.MSELoss of (sdf, 0), more_solver.py:204,213), grad[p,i] = 2 sdf[p,i] / N; optionally the
// per-instance best-loss bookkeeping of More_Solver._optimize_code (:219-221: if loss < min_loss: min_loss = loss, a snapshot exists)
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ sdf, int N, float* __restrict__ loss, float* __restrict__ grad,
                                                  float* __restrict__ min_loss, int32_t* __restrict__ improved) {
    __shared__ float red[4];
    const int p = blockIdx.x;
    const float inv = 1.0f / (float)N;
    float acc = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float x = sdf[(size_t)p * N + i];
        acc = __builtin_fmaf(x, x, acc);
        grad[(size_t)p * N + i] = 2.0f * x * inv;
    }
    acc = block_sum_256_opt(acc, red);
    if (threadIdx.x == 0) {
        const float l = acc * inv;
        loss[p] = l;
        if (min_loss && l < min_loss[p]) { min_loss[p] = l; if (improved) improved[p] = 1; }
    }
}

// torch.optim.Adam (no weight decay, no amsgrad) on up to four parameter tensors in one launch, each with its own learning rate
// (more_solver.py:199-203: z_inv 1e-5, t 1e-4, z_so3 5e-4), in torch's operation order:
//   m <- m + (g - m)(1 - b1);  v <- b2 v + (1 - b2) g g;  p <- p - (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// Every scalar arrives ROUNDED ONCE from the double the host computed it in -- torch.optim.Adam forms 1 - beta, 1 - beta^t, lr / (1 - beta1^t)
// and sqrt(1 - beta2^t) from Python floats (doubles) and hands each to its fp32 kernel as one cast: with beta2 = 0.999, 1.0f - 0.999f is
// 1.3e-5 away from float(1 - 0.999), and 1 - powf(beta2, t) carries the cancellation of an fp32 power.
struct AdamSet { ls_adam_group g[4]; float step_size[4]; };   // step_size = float(lr / (1 - beta1^t))
__global__ __launch_bounds__(256) void adam_multi_kernel(AdamSet s, float omb1, float b2, float omb2, float eps, float bc2_sqrt) {
    const ls_adam_group& q = s.g[blockIdx.y];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= q.n) return;
    const float g = q.grad[i];
    const float m = __builtin_fmaf(g - q.m[i], omb1, q.m[i]);
    const float v = __builtin_fmaf(omb2 * g, g, b2 * q.v[i]);
    q.m[i] = m; q.v[i] = v;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    q.param[i] = __builtin_fmaf(-s.step_size[blockIdx.y], m / denom, q.param[i]);
}

// Batched log-domain softmin of entropic OT with cost |x - y|^2 / 2 (sinkhorn.hip's primitive with a pair index):
//   v_j    = logw + pot_y[p,j] / eps_p - |x_i - y_j|^2 / (2 eps_p)
//   out[i] = -eps_p log sum_j exp(v_j)          (averaged with prev[i] when `average`: the symmetric Sinkhorn update)
//   grad[i] = sum_j softmax_j(v) (x_i - y_j)    (optional)
// eps_p <= 0 marks a pair whose epsilon schedule has ended (shorter schedule than the batch maximum): out = prev, untouched.
// One wave per RB rows of one pair (a y point and its potential are loaded once for RB x rows: the kernel is bound by the
// L1 stream of the M y points per row, 16000 launches per 64-pair registration), online (max, sum) per lane and row, combined
// by wave reductions.
// Arithmetic per (row, j): three subtractions, |d|^2 (mul + 2 fma), ONE fma that forms the exponent already in base 2
// (v' = hj' + c |d|^2 with hj' = (logw + pot / eps) log2 e and c = -log2 e / (2 eps) folded per j / per pair), a compare, a subtraction,
// v_exp_f32, an add: 11 VALU slots + the quarter-rate exponential.  RB rows per wave share the y point and hj' (8 without the
// gradient accumulators, 4 with them).
template <bool GRAD, int RB, int CHK = 4>   // the gradient accumulators are needed by 2 of the ~40 softmins of a divergence only
__device__ __forceinline__ void softmin_rows(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ pot_y, float logw,
                                             const float* __restrict__ eps_p, const float* __restrict__ prev, int average, int N, int M,
                                             float* __restrict__ out, float* __restrict__ grad) {
    const int p = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int i0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RB;
    if (i0 >= N) return;
    const float eps = eps_p[p];
    if (!(eps > 0.f)) {
        // schedule ended: the potential passes through; without a previous potential the defined value is 0 (never left unwritten:
        // callers allocate `out` uninitialised)
        if (lane < RB && i0 + lane < N) out[(size_t)p * N + i0 + lane] = prev ? prev[(size_t)p * N + i0 + lane] : 0.f;
        if constexpr (GRAD) {
            if (lane < RB && i0 + lane < N) { float* gp = grad + ((size_t)p * N + i0 + lane) * 3; gp[0] = 0.f; gp[1] = 0.f; gp[2] = 0.f; }
        }
        return;
    }
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
    const float* yp = y + (size_t)p * M * 3;
    const float* hp = pot_y ? pot_y + (size_t)p * M : nullptr;
    const float inv2 = LOG2E / eps, c2 = -0.5f * inv2, logw2 = logw * LOG2E;
    float xi[RB], yi[RB], zi[RB], mx[RB], sum[RB], gx[GRAD ? RB : 1], gy[GRAD ? RB : 1], gz[GRAD ? RB : 1];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const float* xp = x + ((size_t)p * N + min(i0 + r, N - 1)) * 3;   // rows past N: clamped, computed, never stored
        xi[r] = xp[0]; yi[r] = xp[1]; zi[r] = xp[2];
        mx[r] = -INFINITY; sum[r] = 0.f;
        if constexpr (GRAD) gx[r] = gy[r] = gz[r] = 0.f;
    }
    if constexpr (!GRAD) {
        // chunks of eight y points per lane: the running maximum moves at most once per chunk and row (with a per-element update some
        // lane of the wave sees a new maximum in nearly every one of the M / 64 iterations, i.e. the rescaling path always runs)
        constexpr int CH = CHK;
        for (int j0 = lane; j0 < M; j0 += 64 * CH) {
            float yx[CH], yy[CH], yz[CH], hj[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int j = j0 + 64 * u, jc = min(j, M - 1);
                yx[u] = yp[jc * 3]; yy[u] = yp[jc * 3 + 1]; yz[u] = yp[jc * 3 + 2];
                hj[u] = j < M ? (hp ? fmaf(hp[jc], inv2, logw2) : logw2) : -INFINITY;   // past M: exponent -inf, contributes exactly 0
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                float v[CH], cm = -INFINITY;
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const float dx = xi[r] - yx[u], dy = yi[r] - yy[u], dz = zi[r] - yz[u];
                    v[u] = fmaf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)), c2, hj[u]);
                    cm = fmaxf(cm, v[u]);
                }
                if (cm > mx[r]) { sum[r] *= __builtin_amdgcn_exp2f(mx[r] - cm); mx[r] = cm; }   // 2^(-inf) = 0 on the first hit
#pragma unroll
                for (int u = 0; u < CH; ++u) sum[r] += __builtin_amdgcn_exp2f(v[u] - mx[r]);
            }
        }
    } else
    for (int j = lane; j < M; j += 64) {
        const float yx = yp[j * 3], yy = yp[j * 3 + 1], yz = yp[j * 3 + 2];
        const float hj = hp ? fmaf(hp[j], inv2, logw2) : logw2;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float dx = xi[r] - yx, dy = yi[r] - yy, dz = zi[r] - yz;
            const float v = fmaf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)), c2, hj);
            if (v > mx[r]) {
                const float sc = __builtin_amdgcn_exp2f(mx[r] - v);   // 2^(-inf) = 0 on the first hit
                sum[r] *= sc;
                gx[r] *= sc; gy[r] *= sc; gz[r] *= sc;
                mx[r] = v;
            }
            const float e = __builtin_amdgcn_exp2f(v - mx[r]);
            sum[r] += e;
            gx[r] += e * dx; gy[r] += e * dy; gz[r] += e * dz;
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const float wmx = wave_max(mx[r]);
        const float sc = mx[r] == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mx[r] - wmx);
        const float s = wave_sum(sum[r] * sc);
        float ax = 0.f, ay = 0.f, az = 0.f;
        if constexpr (GRAD) { ax = wave_sum(gx[r] * sc); ay = wave_sum(gy[r] * sc); az = wave_sum(gz[r] * sc); }
        if (lane == 0 && i0 + r < N) {
            const size_t row = (size_t)p * N + i0 + r;
            float o = -eps * LN2 * (wmx + __log2f(s));
            if (average) o = 0.5f * (prev[row] + o);
            out[row] = o;
            if constexpr (GRAD) { grad[row * 3] = ax / s; grad[row * 3 + 1] = ay / s; grad[row * 3 + 2] = az / s; }
        }
    }
}

template <bool GRAD, int RB, int CHK = 4>
__global__ __launch_bounds__(256) void softmin_batched_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ pot_y, float logw, const float* __restrict__ eps_p,
                                                              const float* __restrict__ prev, int average, int N, int M,
                                                              float* __restrict__ out, float* __restrict__ grad) {
    softmin_rows<GRAD, RB, CHK>(x, y, pot_y, logw, eps_p, prev, average, N, M, out, grad);
}
// Up to four INDEPENDENT softmins in one launch (blockIdx.z): the four potentials of one symmetric Sinkhorn iteration
// (f_ba, g_ab, f_aa, g_bb each read only the previous iteration's potentials).  At P = 64 a single softmin is a 29 us kernel behind
// ~30 us of Python + ctypes per call: the 400-step registration loop was bound by the host, not by the device.
struct SoftminSet { ls_softmin_problem q[4]; };
__global__ __launch_bounds__(256) void softmin_multi_kernel(SoftminSet s, const float* __restrict__ eps_p, int average) {
    const ls_softmin_problem& q = s.q[blockIdx.z];
    softmin_rows<false, 8, 4>(q.x, q.y, q.pot_y, q.logw, eps_p, q.prev, average, q.N, q.M, q.out, nullptr);
}

struct AdamCfg { float lr, b1, b2, omb1, omb2, eps, bc1, bc2, stop_angle; };   // omb = float(1 - beta), bc = float(1 - beta^(step + 1)): each formed in double on the host

// One workgroup per pair: d loss / d (v, omega) in the LEFT tangent space from the point gradients G = d loss / d query
//   grad = (sum_i G_i, sum_i query_i x G_i),
// Adam on the 6-vector, g <- exp(-step) g, snapshot of the pose after the step when this step's loss is the best so far
// (more_solver.py:166-168), geodesic angle to the initial rotation > stop_angle -> the pair stops (:172-173).  A stopped pair is
// frozen (the reference breaks out of its loop).  Finally query <- g_new . src for the next step.
__global__ __launch_bounds__(256) void se3_adam_step_kernel(const float* __restrict__ src, const float* __restrict__ G, const float* __restrict__ loss,
                                                            AdamCfg c, int N, float* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2,
                                                            float* __restrict__ min_loss, float* __restrict__ best_g, const float* __restrict__ init_R,
                                                            int32_t* __restrict__ active, float* __restrict__ query) {
    __shared__ float red[4];
    __shared__ float gs[12];
    const int p = blockIdx.x;
    if (!active[p]) return;                 // uniform per workgroup
    const float* q = query + (size_t)p * N * 3;
    const float* Gp = G + (size_t)p * N * 3;
    float acc[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < N; i += 256) {
        const float qx = q[i * 3], qy = q[i * 3 + 1], qz = q[i * 3 + 2];
        const float ax = Gp[i * 3], ay = Gp[i * 3 + 1], az = Gp[i * 3 + 2];
        acc[0] += ax; acc[1] += ay; acc[2] += az;
        acc[3] += qy * az - qz * ay; acc[4] += qz * ax - qx * az; acc[5] += qx * ay - qy * ax;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = block_sum_256_opt(acc[k], red);
    if (threadIdx.x == 0) {
        float st[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const float a = c.b1 * m1[p * 6 + k] + c.omb1 * acc[k];
            const float b = c.b2 * m2[p * 6 + k] + c.omb2 * acc[k] * acc[k];
            m1[p * 6 + k] = a; m2[p * 6 + k] = b;
            st[k] = -(c.lr * (a / c.bc1) / (sqrtf(b / c.bc2) + c.eps));
        }
        // exp of the twist (v, w) = st: Rodrigues, translation through the left Jacobian V
        const float vx = st[0], vy = st[1], vz = st[2], wx = st[3], wy = st[4], wz = st[5];
        const float th = sqrtf(wx * wx + wy * wy + wz * wz);
        float A, Bc, Cc;
        if (th < 1e-6f) { A = 1.f; Bc = 0.f; Cc = 0.f; }   // R = I + K, V = I + K / 2 (first order)
        else { A = sinf(th) / th; Bc = (1.f - cosf(th)) / (th * th); Cc = (th - sinf(th)) / (th * th * th); }
        const float K[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
        float K2[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) K2[r * 3 + cc] = K[r * 3] * K[cc] + K[r * 3 + 1] * K[3 + cc] + K[r * 3 + 2] * K[6 + cc];
        float E[9], V[9];
        const float vB = th < 1e-6f ? 0.5f : Bc;
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            const float id = (e % 4 == 0) ? 1.f : 0.f;
            E[e] = id + A * K[e] + Bc * K2[e];
            V[e] = id + vB * K[e] + Cc * K2[e];
        }
        const float tv[3] = {V[0] * vx + V[1] * vy + V[2] * vz, V[3] * vx + V[4] * vy + V[5] * vz, V[6] * vx + V[7] * vy + V[8] * vz};
        float gn[12];
        const float* go = g + p * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                float v = E[r * 3] * go[cc] + E[r * 3 + 1] * go[4 + cc] + E[r * 3 + 2] * go[8 + cc];
                if (cc == 3) v += tv[r];
                gn[r * 4 + cc] = v;
            }
        }
#pragma unroll
        for (int e = 0; e < 12; ++e) { g[p * 12 + e] = gn[e]; gs[e] = gn[e]; }
        if (loss[p] < min_loss[p]) {
            min_loss[p] = loss[p];
#pragma unroll
            for (int e = 0; e < 12; ++e) best_g[p * 12 + e] = gn[e];
        }
        const float* R0 = init_R + p * 9;
        float tr = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) tr += gn[r * 4] * R0[r * 3] + gn[r * 4 + 1] * R0[r * 3 + 1] + gn[r * 4 + 2] * R0[r * 3 + 2];
        const float cosang = fminf(1.f, fmaxf(-1.f, (tr - 1.f) * 0.5f));
        if (acosf(cosang) > c.stop_angle) active[p] = 0;   // radians against the configured number, as the reference does
    }
    __syncthreads();
    float* qo = query + (size_t)p * N * 3;
    const float* s = src + (size_t)p * N * 3;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float x = s[i * 3], y = s[i * 3 + 1], z = s[i * 3 + 2];
        qo[i * 3] = gs[0] * x + gs[1] * y + gs[2] * z + gs[3];
        qo[i * 3 + 1] = gs[4] * x + gs[5] * y + gs[6] * z + gs[7];
        qo[i * 3 + 2] = gs[8] * x + gs[9] * y + gs[10] * z + gs[11];
    }
}

}  // namespace ls

using namespace ls;

extern "C" {

int ls_se3_transform_f32(const float* g, const float* src, int P, int N, float* query, void* stream) {
    LS_REQUIRE(g && src && query && P > 0 && N > 0 && P <= 65535, "se3_transform: null argument or bad sizes (P=%d N=%d)", P, N);
    hipLaunchKernelGGL(se3_transform_kernel, dim3(cdiv(N, 256), P), dim3(256), 0, (hipStream_t)stream, g, src, N, query);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

int ls_smooth_l1_f32(const float* sdf, int P, int N, int accumulate, float* loss, float* grad_sdf, void* stream) {
    LS_REQUIRE(sdf && loss && grad_sdf && P > 0 && N > 0, "smooth_l1: null argument or empty problem");
    hipLaunchKernelGGL(smooth_l1_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, sdf, N, accumulate, loss, grad_sdf);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

int ls_mse_f32(const float* sdf, int P, int N, float* loss, float* grad_sdf, float* min_loss, int32_t* improved, void* stream) {
    LS_REQUIRE(sdf && loss && grad_sdf && P > 0 && N > 0, "mse: null argument or empty problem");
    hipLaunchKernelGGL(mse_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, sdf, N, loss, grad_sdf, min_loss, improved);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

int ls_adam_step_f32(const ls_adam_group* groups, int count, double beta1, double beta2, double adam_eps, int step, void* stream) {
    LS_REQUIRE(groups && count >= 1 && count <= 4 && step >= 0, "adam_step: bad arguments (count=%d step=%d)", count, step);
    AdamSet s;
    long long nmax = 0;
    for (int i = 0; i < count; ++i) {
        LS_REQUIRE(groups[i].param && groups[i].grad && groups[i].m && groups[i].v && groups[i].n > 0, "adam_step: group %d: null argument or empty tensor", i);
        s.g[i] = groups[i];
        nmax = std::max(nmax, groups[i].n);
    }
    for (int i = count; i < 4; ++i) s.g[i] = s.g[0];
    const double bc1 = 1.0 - pow(beta1, (double)(step + 1)), bc2 = 1.0 - pow(beta2, (double)(step + 1));   // as torch: Python floats
    for (int i = 0; i < 4; ++i) s.step_size[i] = (float)(s.g[i].lr / bc1);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(cdiv(nmax, 256), count), dim3(256), 0, (hipStream_t)stream, s, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)adam_eps, (float)sqrt(bc2));
    LS_LAUNCH_CHECK();
    return LS_OK;
}

int ls_sinkhorn_softmin_batched_f32(const float* x, const float* y, const float* pot_y, float logw, const float* eps, const float* prev,
                                    int average, int P, int N, int M, float* out, float* grad_x, void* stream) {
    LS_REQUIRE(x && y && eps && out, "sinkhorn_softmin_batched: null argument");
    LS_REQUIRE(P > 0 && N > 0 && M > 0 && P <= 65535, "sinkhorn_softmin_batched: bad sizes (P=%d N=%d M=%d)", P, N, M);
    LS_REQUIRE(!average || prev, "sinkhorn_softmin_batched: average needs prev");
    LS_REQUIRE(prev != out || !prev, "sinkhorn_softmin_batched: out must not alias prev (the symmetric update reads old potentials)");
    if (grad_x)
        hipLaunchKernelGGL((softmin_batched_kernel<true, 4>), dim3(cdiv(N, 4 * 4), P), dim3(256), 0, (hipStream_t)stream, x, y, pot_y, logw, eps, prev,
                           average, N, M, out, grad_x);
    else
        hipLaunchKernelGGL((softmin_batched_kernel<false, 8>), dim3(cdiv(N, 4 * 8), P), dim3(256), 0, (hipStream_t)stream, x, y, pot_y, logw, eps, prev,
                           average, N, M, out, grad_x);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

int ls_sinkhorn_softmin_multi_f32(const ls_softmin_problem* problems, int count, const float* eps, int average, int P, void* stream) {
    LS_REQUIRE(problems && eps && count >= 1 && count <= 4 && P > 0 && P <= 65535, "sinkhorn_softmin_multi: bad arguments (count=%d P=%d)", count, P);
    SoftminSet s;
    int nmax = 0;
    for (int i = 0; i < count; ++i) {
        const ls_softmin_problem& q = problems[i];
        LS_REQUIRE(q.x && q.y && q.out && q.N > 0 && q.M > 0, "sinkhorn_softmin_multi: problem %d: null argument or empty cloud", i);
        LS_REQUIRE(!average || q.prev, "sinkhorn_softmin_multi: problem %d: average needs prev", i);
        LS_REQUIRE(q.prev != q.out || !q.prev, "sinkhorn_softmin_multi: problem %d: out must not alias prev", i);
        s.q[i] = q;
        nmax = std::max(nmax, q.N);
    }
    for (int i = count; i < 4; ++i) s.q[i] = s.q[0];
    hipLaunchKernelGGL(softmin_multi_kernel, dim3(cdiv(nmax, 4 * 8), P, count), dim3(256), 0, (hipStream_t)stream, s, eps, average);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

int ls_se3_adam_step_f32(const float* src, const float* grad_query, const float* loss, int P, int N, double lr, double beta1, double beta2,
                         double adam_eps, int step, float stop_angle, float* g, float* m1, float* m2, float* min_loss, float* best_g,
                         const float* init_R, int32_t* active, float* query, void* stream) {
    LS_REQUIRE(src && grad_query && loss && g && m1 && m2 && min_loss && best_g && init_R && active && query, "se3_adam_step: null argument");
    LS_REQUIRE(P > 0 && N > 0 && step >= 0, "se3_adam_step: bad sizes");
    AdamCfg c{(float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)adam_eps,
              (float)(1.0 - pow(beta1, (double)(step + 1))), (float)(1.0 - pow(beta2, (double)(step + 1))), stop_angle};
    hipLaunchKernelGGL(se3_adam_step_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, src, grad_query, loss, c, N, g, m1, m2, min_loss, best_g,
                       init_R, active, query);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

}  // extern "C"
