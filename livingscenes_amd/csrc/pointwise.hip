// pointwise.hip -- the small HBM-bound kernels around the edge-conv: encode prologue, per-instance mean,
// point-wise VN activation (residual global conv), and the encoder tail (conv_c pooling + the four heads).
#include "ls_common.h"

namespace ls {

// ---------------------------------------------------------------------------------------------- prologue
// Shape_Prior.encode, /root/reference/model_utils.py:171-177: centroid = mean_n x; x -= centroid;
// scale_0 = mean(top-5 of the N*N entries of cdist(x,x)); x /= scale_0.  The symmetric matrix holds every
// unordered pair twice, so top-5 = (d1,d1,d2,d2,d3) with d1>=d2>=d3 the three largest pair distances.
// One workgroup per instance; cloud staged in LDS; each thread keeps a private top-3 of squared distances.
// (branch-free, five instructions: the nested-branch form diverges per lane inside the pair loops -- the same three values for finite input.  A NaN
//  distance is NOT ignored (fminf(a, NaN) = a duplicates the current maximum into b): clouds with NaN / Inf points are unsupported, as in the reference, whose
//  topk over a cdist matrix with NaN entries returns NaN and turns every code into NaN -- model_utils.py:175-177)
__device__ __forceinline__ void top3_insert(float v, float& a, float& b, float& c) {
    const float m = fminf(a, v);
    a = fmaxf(a, v);
    const float m2 = fminf(b, m);
    b = fmaxf(b, m);
    c = fmaxf(c, m2);
}

// centroid of instance b, cloud staged (un-centred) into sp[3][N]
__device__ __forceinline__ void stage_and_centroid(const float* __restrict__ xb, int N, float* sp, float* red, float c[3]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s[3] = {0.f, 0.f, 0.f};
#pragma unroll 4   // (twelve loads in flight at N = 1024: rolled, the cold cloud costs one HBM round trip per 256 points)
    for (int n = tid; n < N; n += 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = xb[(size_t)a * N + n]; sp[a * N + n] = v; s[a] += v; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { s[a] = wave_sum(s[a]); if (lane == 0) red[a * 4 + wave] = s[a]; }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = (red[a * 4] + red[a * 4 + 1] + red[a * 4 + 2] + red[a * 4 + 3]) / (float)N;
    __syncthreads();
}

// block-wide top-3 of per-thread (t0 >= t1 >= t2); result broadcast to every thread
__device__ __forceinline__ void block_top3(float& t0, float& t1, float& t2, float* red12) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float u0 = __shfl_xor(t0, o, 64), u1 = __shfl_xor(t1, o, 64), u2 = __shfl_xor(t2, o, 64);
        top3_insert(u0, t0, t1, t2); top3_insert(u1, t0, t1, t2); top3_insert(u2, t0, t1, t2);
    }
    __syncthreads();
    if (lane == 0) { red12[wave * 3] = t0; red12[wave * 3 + 1] = t1; red12[wave * 3 + 2] = t2; }
    __syncthreads();
    t0 = t1 = t2 = -1.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) top3_insert(red12[i], t0, t1, t2);
}

// One workgroup per instance.  The three largest pair distances only involve points near the hull, so the N^2/2 pair scan is
// pruned EXACTLY: with a = the point farthest from the centroid and LB = the third largest squared distance from a (three
// real pairs, so the overall third largest is >= LB), a pair (i, j) can reach the top three only if
// |p_i| + |p_j| >= sqrt(LB), hence only if both |p_i| and |p_j| are >= sqrt(LB) - max|p|  (triangle inequality about the
// centroid).  Those "outer" points are compacted and scanned pairwise with the same fp32 formula as the full scan, so the
// three values (and scale_0) are bit-identical to it; a cloud whose points all lie on a sphere degenerates to the full scan.
__global__ __launch_bounds__(256) void prologue_kernel(const float* __restrict__ x, int N, float* __restrict__ pts_out,
                                                       float* __restrict__ centroid_out, float* __restrict__ scale0_out, int dense) {
    LS_LATENCY_CRITICAL();
    extern __shared__ __attribute__((aligned(16))) float sp[];  // [3][N] centred cloud, then [N] outer-point indices, then their coordinates [3][N]
    __shared__ float red[12];
    __shared__ float redv[4];
    __shared__ int redi[4];
    __shared__ int nouter;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* outer = reinterpret_cast<int*>(sp + 3 * N);
    float c[3];
    stage_and_centroid(x + (size_t)b * 3 * N, N, sp, red, c);
    // centre; squared norms; a = arg-max norm (first index on ties)
    float best = -1.f;
    int besti = 0;
    for (int n = tid; n < N; n += 256) {
        float r2 = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = sp[a * N + n] - c[a]; sp[a * N + n] = v; r2 += v * v; }
        if (r2 > best) { best = r2; besti = n; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (lane == 0) { redv[wave] = best; redi[wave] = besti; }
    if (tid == 0) nouter = 0;
    __syncthreads();
    best = redv[0]; besti = redi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (redv[w] > best || (redv[w] == best && redi[w] < besti)) { best = redv[w]; besti = redi[w]; }
    const float rmax = sqrtf(best);
    // LB = third largest squared distance from a
    float t0 = -1.f, t1 = -1.f, t2 = -1.f;
    {
        const float px = sp[besti], py = sp[N + besti], pz = sp[2 * N + besti];
        for (int j = tid; j < N; j += 256) {
            if (j == besti) continue;
            const float dx = px - sp[j], dy = py - sp[N + j], dz = pz - sp[2 * N + j];
            top3_insert(dx * dx + dy * dy + dz * dz, t0, t1, t2);
        }
    }
    block_top3(t0, t1, t2, red);
    // outer points: |p| >= sqrt(LB) (1 - 1e-4) - max|p|   (t2 < 0 when N < 4: keep everything)
    const float cut = t2 > 0.f ? sqrtf(t2) * (1.0f - 1e-4f) - rmax : -1.f;
    for (int n = tid; n < N; n += 256) {
        const float r = sqrtf(sp[n] * sp[n] + sp[N + n] * sp[N + n] + sp[2 * N + n] * sp[2 * N + n]);
        if (r >= cut) outer[atomicAdd(&nouter, 1)] = n;
    }
    __syncthreads();
    const int M = nouter;
    // the outer points' coordinates, dense ([3][M] behind the index list): the pair loop then reads its operands directly -- one LDS round trip per pair
    // instead of two dependent ones (index, then coordinates) -- and its reads are independent, so the unrolled loop keeps several pairs in flight.  (Round 5:
    // the scan of the slowest instance of a batch -- a few hundred outer points -- was 33 of the kernel's 46 us.)
    t0 = t1 = t2 = -1.f;
    if (dense) {   // (kernel-uniform: the launch reserved the extra 3 N floats)
        float* ox = reinterpret_cast<float*>(outer + N);
        float* oy = ox + N;
        float* oz = oy + N;
        for (int i = tid; i < M; i += 256) { const int pi = outer[i]; ox[i] = sp[pi]; oy[i] = sp[N + pi]; oz[i] = sp[2 * N + pi]; }
        __syncthreads();
        // all pairs among the outer points: thread -> rows i = tid/32 + 8k of the pair matrix, columns strided by 32
        for (int i = tid >> 5; i < M; i += 8) {
            const float px = ox[i], py = oy[i], pz = oz[i];
#pragma unroll 4
            for (int j = i + 1 + (tid & 31); j < M; j += 32) {
                const float dx = px - ox[j], dy = py - oy[j], dz = pz - oz[j];
                top3_insert(dx * dx + dy * dy + dz * dz, t0, t1, t2);
            }
        }
    } else {       // large clouds (the dense copy would not fit the default LDS): through the index list
        for (int i = tid >> 5; i < M; i += 8) {
            const int pi = outer[i];
            const float px = sp[pi], py = sp[N + pi], pz = sp[2 * N + pi];
            for (int j = i + 1 + (tid & 31); j < M; j += 32) {
                const int pj = outer[j];
                const float dx = px - sp[pj], dy = py - sp[N + pj], dz = pz - sp[2 * N + pj];
                top3_insert(dx * dx + dy * dy + dz * dz, t0, t1, t2);
            }
        }
    }
    block_top3(t0, t1, t2, red);
    const float d1 = sqrtf(fmaxf(t0, 0.f)), d2 = sqrtf(fmaxf(t1, 0.f)), d3 = sqrtf(fmaxf(t2, 0.f));
    const float sc = ((((d1 + d1) + d2) + d2) + d3) / 5.0f;
    if (tid == 0) {
        scale0_out[b] = sc;
        centroid_out[b * 3 + 0] = c[0]; centroid_out[b * 3 + 1] = c[1]; centroid_out[b * 3 + 2] = c[2];
    }
    float* po = pts_out + (size_t)b * N * 3;
    for (int t = tid; t < N * 3; t += 256) { const int n = t / 3, a = t % 3; po[t] = sp[a * N + n] / sc; }
}

// [B,3,N] -> [B,N,3] without normalisation (pre_normalised path)
__global__ void transpose_cloud_kernel(const float* __restrict__ x, int N, float* __restrict__ pts_out, int total) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int b = t / (N * 3), r = t % (N * 3), n = r / 3, a = r % 3;
    pts_out[t] = x[((size_t)b * 3 + a) * N + n];
}

// ---------------------------------------------------------------------------------------------- mean over points
// dst_f.mean(-1) of vec_dgcnn_atten.py:223: in [B,N,3,C] -> out [B,3,C]; sequential over n (deterministic)
// block = 16 float4-column groups x 16 row slices; fixed-order LDS combine (bit-reproducible, no atomics)
__global__ __launch_bounds__(256) void mean_points_kernel(const float* __restrict__ f, int N, int row, float* __restrict__ out) {
    __shared__ float4 part[16][16];
    const int b = blockIdx.y, cg = threadIdx.x & 15, rs = threadIdx.x >> 4;
    const int col = (blockIdx.x * 16 + cg) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < row) {
        const float* p = f + (size_t)b * N * row + col;
#pragma unroll 8   // eight independent loads in flight; the sum keeps its order
        for (int n = rs; n < N; n += 16) {
            const float4 v = *reinterpret_cast<const float4*>(p + (size_t)n * row);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    part[rs][cg] = s;
    __syncthreads();
    if (rs == 0 && col < row) {
        float4 t = part[0][cg];
#pragma unroll
        for (int r = 1; r < 16; ++r) { const float4 v = part[r][cg]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        const float inv = 1.0f / (float)N;
        *reinterpret_cast<float4*>(out + (size_t)b * row + col) = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    }
}

// ---------------------------------------------------------------------------------------------- mean over points + its VecLinear
// The per-instance half of the residual global conv (vec_dgcnn_atten.py:222-225: VecLNA_G(cat(f, mean_n f))): the part of the
// contraction that multiplies the MEAN feature is the same for every point of an instance,
//     G[b][x][n] = sum_k mean_n f[b][n][x][k] * W[n][k]        n in [col0, col0 + ncols)  (the W_b / Wd W_b rows of the folded matrix)
// Until round 3 this took three launches (mean_points_kernel, a 192-row fp32-MFMA GEMM split along K, its reduce) of 5 - 14 us each on
// the critical path of every layer >= 2; here a workgroup computes the instance's mean rows into LDS (same summation order as
// mean_points_kernel where the layer is narrow) and then its block of columns, one column per lane, k ascending.
// sum over aligned groups of 16 lanes, every lane receives it (quad_perm, quad_perm, row_half_mirror, row_mirror: fixed order)
template <int CTRL>
__device__ __forceinline__ float dpp_addf(float v) { return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ float sum16_dpp(float v) { return dpp_addf<0x140>(dpp_addf<0x141>(dpp_addf<0x4E>(dpp_addf<0xB1>(v)))); }
__global__ __launch_bounds__(256) void glob_mean_gemv_kernel(const float* __restrict__ f, int N, int C, const float* __restrict__ W, int col0,
                                                             int cols_per_block, int ncols, float* __restrict__ G, int ldg, float inv) {
    LS_LATENCY_CRITICAL();
    // f [B][N][3][C]: the features themselves (inv = 1 / N), or N rows of partial column sums per instance written by the kernel that produced the
    // features (edge_attn_fq_kernel: one row per workgroup; edge_ft_v_kernel: N = 1) with inv = 1 / (points per instance)
    extern __shared__ float lmean[];           // [3][C]
    __shared__ float4 part[16][16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = 3 * C;
    const int cg = tid & 15, rs = tid >> 4;
    if (row >= 512) {
        // wide rows, few points (layers 5, 6: 32 points x 768 / 1536 floats): a thread owns float4 columns and walks all N rows (eight
        // loads in flight); no LDS combine, no barrier per 64-column chunk (24 chunks x 2 barriers of pure latency at layer 6)
        for (int c4 = tid; c4 < row / 4; c4 += 256) {
            const float* p = f + (size_t)b * N * row + c4 * 4;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
            for (int n = 0; n < N; ++n) {
                const float4 v = *reinterpret_cast<const float4*>(p + (size_t)n * row);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            *reinterpret_cast<float4*>(&lmean[c4 * 4]) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
        }
        __syncthreads();
    } else
    for (int cb = 0; cb < row; cb += 64) {     // 64 columns of the [N, 3C] matrix at a time: 16 float4 groups x 16 row slices
        const int col = cb + cg * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < row) {
            const float* p = f + (size_t)b * N * row + col;
#pragma unroll 8
            for (int n = rs; n < N; n += 16) {
                const float4 v = *reinterpret_cast<const float4*>(p + (size_t)n * row);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        part[rs][cg] = s;
        __syncthreads();
        if (rs == 0 && col < row) {
            float4 t = part[0][cg];
#pragma unroll
            for (int r = 1; r < 16; ++r) { const float4 v = part[r][cg]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
            *reinterpret_cast<float4*>(&lmean[col]) = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
        }
        __syncthreads();
    }
    // columns: SIXTEEN lanes own a column and walk its weight row together (lane l takes k = 4 l, 4 l + 64, ...: every load instruction reads whole
    // contiguous row segments), three partial dot products per lane, summed over the sixteen lanes on the DPP network.  (Until round 5 a lane owned a
    // column and read ITS row 16 bytes at a time: each of those loads is its own L2 transaction -- 26 us at layer 6 for 2 MB of weights.)
    const int j0 = blockIdx.y * cols_per_block, j1 = min(ncols, j0 + cols_per_block);
    const int kl = lane & 15, jr = wave * 4 + (lane >> 4);
    for (int jb = j0; jb < j1; jb += 32) {          // two columns per lane group and trip: more loads in flight
        float a[2][3];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = jb + 16 * u + jr;
            const float* wr = W + (size_t)(col0 + min(j, j1 - 1)) * C;
            a[u][0] = a[u][1] = a[u][2] = 0.f;
#pragma unroll 8
            for (int k = 4 * kl; k < C; k += 64) {
                const float4 w = *reinterpret_cast<const float4*>(wr + k);
                const float4 m0 = *reinterpret_cast<const float4*>(&lmean[k]), m1 = *reinterpret_cast<const float4*>(&lmean[C + k]),
                             m2 = *reinterpret_cast<const float4*>(&lmean[2 * C + k]);
                a[u][0] = __builtin_fmaf(m0.x, w.x, a[u][0]); a[u][0] = __builtin_fmaf(m0.y, w.y, a[u][0]); a[u][0] = __builtin_fmaf(m0.z, w.z, a[u][0]); a[u][0] = __builtin_fmaf(m0.w, w.w, a[u][0]);
                a[u][1] = __builtin_fmaf(m1.x, w.x, a[u][1]); a[u][1] = __builtin_fmaf(m1.y, w.y, a[u][1]); a[u][1] = __builtin_fmaf(m1.z, w.z, a[u][1]); a[u][1] = __builtin_fmaf(m1.w, w.w, a[u][1]);
                a[u][2] = __builtin_fmaf(m2.x, w.x, a[u][2]); a[u][2] = __builtin_fmaf(m2.y, w.y, a[u][2]); a[u][2] = __builtin_fmaf(m2.z, w.z, a[u][2]); a[u][2] = __builtin_fmaf(m2.w, w.w, a[u][2]);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = jb + 16 * u + jr;
#pragma unroll
            for (int x = 0; x < 3; ++x) a[u][x] = sum16_dpp(a[u][x]);
            if (kl == 0 && j < j1) {
                float* gp = G + (size_t)b * 3 * ldg + col0 + j;
                gp[0] = a[u][0]; gp[ldg] = a[u][1]; gp[2 * ldg] = a[u][2];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- point-wise VN activation
// global_conv VecLNA of vec_dgcnn_atten.py:222-225 after the GEMMs:
//   y  = T[p][x][0:C]   + G[b][x][2C:3C]      (W_a f[n]      + W_b mean_n f)
//   kd = T[p][x][C:2C]  + G[b][x][3C:4C]      (Wd W_a f[n]   + Wd W_b mean_n f)
// T [B*N*3, ldt] (cols lin|dir), G [B*3, ldg] (cols ..|..|lin_g|dir_g) or NULL; out [B,N,3,C]
__global__ __launch_bounds__(256) void vn_act_rows_kernel(const float* __restrict__ T, int ldt, const float* __restrict__ G,
                                                          int ldg, int N, int C, float oms, float* __restrict__ out,
                                                          long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % C);
    const long long p = t / C;
    const int b = (int)(p / N);
    const float* Tp = T + (size_t)p * 3 * ldt;
    float y[3], k[3];
#pragma unroll
    for (int x = 0; x < 3; ++x) {
        y[x] = Tp[x * ldt + c];
        k[x] = Tp[x * ldt + C + c];
        if (G) {
            y[x] += G[((size_t)b * 3 + x) * ldg + 2 * C + c];
            k[x] += G[((size_t)b * 3 + x) * ldg + 3 * C + c];
        }
    }
    vn_act(y[0], y[1], y[2], k[0], k[1], k[2], oms);
    float* op = out + (size_t)p * 3 * C + c;
    op[0] = y[0]; op[C] = y[1]; op[2 * C] = y[2];
}

// ---------------------------------------------------------------------------------------------- encoder tail
// vec_dgcnn_atten.py:231-250 + Shape_Prior.encode epilogue (model_utils.py:182-195), one workgroup per instance.
//   Tc [B, NP, 3, ldc]: conv_c.lin output in cols [0,Cd), the shared lin_dir output in col Cd
//   X = mean_n act_shared(...)               (:231-232; shared_nonlinearity -> one direction per point)
//   z_so3 = cevn(X); scale = mean_c |X_c| * SF; z_inv = <cevn(fc_inv X), z_so3>          (:234-238)
//   center = VecResBlock(X) * SF   (vec_layers.py:631-651: act2(shortcut X + lin1 VecLNA_fc0 X))  (:246-250)
//   t = center + centroid ; s = scale_0 * scale
struct TailW {
    const float* inv_t;    // [Cd][Cd]  fc_inv^T
    const float* fc0_t;    // [Cd][2h]  {fc0.lin ; fc0.dir*fc0.lin}^T
    const float* misc;     // lin1 [h] | shortcut [Cd] | act2 dir [1]
};

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// R[x][o] = sum_c Wt[c][o] X[x][c]  (x < 3, o < n_out; Wt row stride ldw, X [3][Cd] in LDS) for the whole workgroup: the c range is split
// over the four waves (a wave's lane l owns outputs l, l + 64, ...: coalesced weight rows, up to 16 independent fma chains of length
// Cd / 4 instead of one of length Cd per thread -- the one-chain form was a load latency per term: 70 us for the whole tail), partial sums
// through LDS (part [4][3][n_out]), combined in wave order.  Ends with a barrier: R is readable by every thread.
__device__ __forceinline__ void tail_gemv3(const float* __restrict__ Wt, int ldw, int Cd, int n_out, const float* X, float* part, float* R) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cq = (Cd + 3) / 4, c0 = wave * cq, c1 = min(Cd, c0 + cq);
    // four CONSECUTIVE outputs per lane per pass = one 16-byte weight load per c (round 5: the four-dword form issued 4x the load instructions for the same
    // bytes; every output still accumulates over its wave's c range in ascending order, so the sums are unchanged bit for bit)
    const bool v4 = (n_out % 4 == 0) && (ldw % 4 == 0) && ((reinterpret_cast<uintptr_t>(Wt) & 15) == 0);
    for (int ob = 0; ob < n_out; ob += 64 * 4) {
        float a[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j][0] = a[j][1] = a[j][2] = 0.f;
        if (v4) {
            const int o = ob + lane * 4;
            const bool in = o < n_out;
#pragma unroll 16  // sixteen 16-byte weight loads in flight: four L2 round trips per wave for a 64-row c range (a batch of loads is one round trip)
            for (int c = c0; c < c1; ++c) {
                const float x0 = X[c], x1 = X[Cd + c], x2 = X[2 * Cd + c];
                const float4 wv = in ? *reinterpret_cast<const float4*>(Wt + (size_t)c * ldw + o) : make_float4(0.f, 0.f, 0.f, 0.f);
                a[0][0] += wv.x * x0; a[0][1] += wv.x * x1; a[0][2] += wv.x * x2;
                a[1][0] += wv.y * x0; a[1][1] += wv.y * x1; a[1][2] += wv.y * x2;
                a[2][0] += wv.z * x0; a[2][1] += wv.z * x1; a[2][2] += wv.z * x2;
                a[3][0] += wv.w * x0; a[3][1] += wv.w * x1; a[3][2] += wv.w * x2;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (o + j < n_out) { part[(wave * 3 + 0) * n_out + o + j] = a[j][0]; part[(wave * 3 + 1) * n_out + o + j] = a[j][1]; part[(wave * 3 + 2) * n_out + o + j] = a[j][2]; }
            continue;
        }
#pragma unroll 4   // sixteen weight loads in flight
        for (int c = c0; c < c1; ++c) {
            const float x0 = X[c], x1 = X[Cd + c], x2 = X[2 * Cd + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = ob + j * 64 + lane;
                const float wv = o < n_out ? Wt[(size_t)c * ldw + o] : 0.f;
                a[j][0] += wv * x0; a[j][1] += wv * x1; a[j][2] += wv * x2;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = ob + j * 64 + lane;
            if (o < n_out) { part[(wave * 3 + 0) * n_out + o] = a[j][0]; part[(wave * 3 + 1) * n_out + o] = a[j][1]; part[(wave * 3 + 2) * n_out + o] = a[j][2]; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * n_out; i += 256) {
        const int x = i / n_out, o = i - x * n_out;
        R[i] = (part[(0 * 3 + x) * n_out + o] + part[(1 * 3 + x) * n_out + o]) + (part[(2 * 3 + x) * n_out + o] + part[(3 * 3 + x) * n_out + o]);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void tail_kernel(const float* __restrict__ Tc, int ldc, int NP, int Cd, TailW w, float oms,
                                                   float scale_factor, int center_pred, int center_scale,
                                                   const float* __restrict__ centroid, const float* __restrict__ scale0,
                                                   float* __restrict__ z_so3, float* __restrict__ z_inv,
                                                   float* __restrict__ s_out, float* __restrict__ t_out) {
    LS_LATENCY_CRITICAL();
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* X = sm;                 // [3][Cd]
    float* kdir = X + 3 * Cd;      // [NP][3] normalised shared directions
    float* H = kdir + 3 * NP;      // [3][h]
    float* part = H + 3 * (Cd / 2);  // [4][3][Cd] partial sums of tail_gemv3
    float* R = part + 12 * Cd;       // [3][Cd] its result
    __shared__ float red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int h = Cd / 2;
    const float* Tb = Tc + (size_t)b * NP * 3 * ldc;
    for (int n = tid; n < NP; n += 256) {
        const float k0 = Tb[((size_t)n * 3 + 0) * ldc + Cd], k1 = Tb[((size_t)n * 3 + 1) * ldc + Cd], k2 = Tb[((size_t)n * 3 + 2) * ldc + Cd];
        const float inv = 1.0f / fmaxf(sqrtf(k0 * k0 + k1 * k1 + k2 * k2), 1e-12f);
        kdir[n * 3 + 0] = k0 * inv; kdir[n * 3 + 1] = k1 * inv; kdir[n * 3 + 2] = k2 * inv;
    }
    __syncthreads();
    float ss = 0.f, sn = 0.f;
    for (int c = tid; c < Cd; c += 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll 16  // 48 loads in flight (un-unrolled: one L2 round trip per point, ~30 us of the kernel's 70; 8: 24 in flight)
        for (int n = 0; n < NP; ++n) {
            const float* r = Tb + (size_t)n * 3 * ldc + c;
            float y0 = r[0], y1 = r[ldc], y2 = r[2 * ldc];
            const float k0 = kdir[n * 3], k1 = kdir[n * 3 + 1], k2 = kdir[n * 3 + 2];
            const float p = y0 * k0 + y1 * k1 + y2 * k2;
            const float f = oms * fminf(p, 0.f);
            a0 += y0 - f * k0; a1 += y1 - f * k1; a2 += y2 - f * k2;
        }
        a0 /= (float)NP; a1 /= (float)NP; a2 /= (float)NP;
        X[c] = a0; X[Cd + c] = a1; X[2 * Cd + c] = a2;
        const float q = a0 * a0 + a1 * a1 + a2 * a2;
        ss += q; sn += sqrtf(q);
    }
    const float fro = sqrtf(block_sum_256(ss, red));
    const float nsum = block_sum_256(sn, red);
    const float invf = 1.0f / fmaxf(fro, 1e-12f);
    __syncthreads();
    // fc_inv
    tail_gemv3(w.inv_t, Cd, Cd, Cd, X, part, R);
    float yi[4][3];
    float ssi = 0.f;
    int cnt = 0;
    for (int o = tid; o < Cd; o += 256, ++cnt) {
        const float a0 = R[o], a1 = R[Cd + o], a2 = R[2 * Cd + o];
        if (cnt < 4) { yi[cnt][0] = a0; yi[cnt][1] = a1; yi[cnt][2] = a2; }
        ssi += a0 * a0 + a1 * a1 + a2 * a2;
    }
    const float invfi = 1.0f / fmaxf(sqrtf(block_sum_256(ssi, red)), 1e-12f);
    cnt = 0;
    for (int o = tid; o < Cd; o += 256, ++cnt) {
        const float z0 = X[o] * invf, z1 = X[Cd + o] * invf, z2 = X[2 * Cd + o] * invf;
        float* zp = z_so3 + ((size_t)b * Cd + o) * 3;
        zp[0] = z0; zp[1] = z1; zp[2] = z2;
        if (cnt < 4) z_inv[(size_t)b * Cd + o] = (yi[cnt][0] * z0 + yi[cnt][1] * z1 + yi[cnt][2] * z2) * invfi;
    }
    const float sc0 = scale0 ? scale0[b] : 1.0f;
    if (tid == 0) s_out[b] = sc0 * (nsum / (float)Cd * scale_factor);
    // fc_center
    float ctr[3] = {0.f, 0.f, 0.f};
    if (center_pred) {
        tail_gemv3(w.fc0_t, 2 * h, Cd, 2 * h, X, part, R);    // columns [0, h): fc0.lin, [h, 2h): fc0.dir * fc0.lin
        for (int o = tid; o < h; o += 256) {
            float y0 = R[o], y1 = R[2 * h + o], y2 = R[4 * h + o];
            const float k0 = R[h + o], k1 = R[2 * h + h + o], k2 = R[4 * h + h + o];
            vn_act(y0, y1, y2, k0, k1, k2, oms);
            H[o] = y0; H[h + o] = y1; H[2 * h + o] = y2;
        }
        __syncthreads();
        float v[3] = {0.f, 0.f, 0.f};
        for (int c = tid; c < Cd; c += 256) {
            const float wsv = w.misc[h + c];
            v[0] += wsv * X[c]; v[1] += wsv * X[Cd + c]; v[2] += wsv * X[2 * Cd + c];
        }
        for (int o = tid; o < h; o += 256) {
            const float wl = w.misc[o];
            v[0] += wl * H[o]; v[1] += wl * H[h + o]; v[2] += wl * H[2 * h + o];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) v[a] = block_sum_256(v[a], red);
        const float wd2 = w.misc[h + Cd];
        float k0 = wd2 * v[0], k1 = wd2 * v[1], k2 = wd2 * v[2];
        vn_act(v[0], v[1], v[2], k0, k1, k2, oms);
        const float sf = center_scale ? scale_factor : 1.0f;
        ctr[0] = v[0] * sf; ctr[1] = v[1] * sf; ctr[2] = v[2] * sf;
    }
    if (tid == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) t_out[b * 3 + a] = ctr[a] + (centroid ? centroid[b * 3 + a] : 0.f);
    }
}

// packed codes [z_so3 B*c*3 | z_inv B*c | s B | t B*3] (fixed addresses written by a captured encode graph) -> the caller's tensors
__global__ __launch_bounds__(256) void scatter_codes_kernel(const float* __restrict__ packed, int n_so3, int n_inv, int n_s, int n_t,
                                                            float* __restrict__ z_so3, float* __restrict__ z_inv, float* __restrict__ s,
                                                            float* __restrict__ t) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_so3) z_so3[i] = packed[i];
    else if (i < n_so3 + n_inv) z_inv[i - n_so3] = packed[i];
    else if (i < n_so3 + n_inv + n_s) s[i - n_so3 - n_inv] = packed[i];
    else if (i < n_so3 + n_inv + n_s + n_t) t[i - n_so3 - n_inv - n_s] = packed[i];
}
int scatter_codes_launch(const float* packed, int B, int c, float* z_so3, float* z_inv, float* s, float* t, hipStream_t st) {
    const int total = B * (4 * c + 4);
    hipLaunchKernelGGL(scatter_codes_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, packed, B * c * 3, B * c, B, B * 3, z_so3, z_inv, s, t);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

size_t prologue_scratch_floats(int B) { (void)B; return 0; }
int prologue_launch(const float* x, int B, int N, float* pts, float* centroid, float* scale0, float* /*unused*/, hipStream_t st) {
    LS_REQUIRE(N >= 3 && N <= 8192, "prologue: N=%d out of range (3..8192)", N);
    const int dense = (size_t)7 * N * sizeof(float) <= 64 * 1024;   // cloud [3][N], outer indices [N] (, their coordinates [3][N])
    hipLaunchKernelGGL(prologue_kernel, dim3(B), dim3(256), (size_t)(dense ? 7 : 4) * N * sizeof(float), st, x, N, pts, centroid, scale0, dense);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int transpose_cloud_launch(const float* x, int B, int N, float* pts, hipStream_t st) {
    const int total = B * N * 3;
    hipLaunchKernelGGL(transpose_cloud_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, x, N, pts, total);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int mean_points_launch(const float* f, int B, int N, int C, float* out, hipStream_t st) {
    const int row = 3 * C;
    LS_REQUIRE(row % 4 == 0, "mean_points: 3*C must be a multiple of 4");
    hipLaunchKernelGGL(mean_points_kernel, dim3(cdiv(row, 64), B), dim3(256), 0, st, f, N, row, out);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
// G[b][x][col0 + j] = <mean_n f[b][n][x][:], W[col0 + j][:]>, j < ncols   (f [B,N,3,C], W [*, C], G [B*3, ldg])
int glob_mean_gemv_launch(const float* f, int B, int N, int C, const float* W, int col0, int ncols, float* G, int ldg, hipStream_t st, int npoints) {
    // npoints > 0: f holds N rows of partial column sums over npoints points per instance (see the kernel)
    LS_REQUIRE(C % 4 == 0 && (size_t)3 * C * sizeof(float) <= 48 * 1024, "glob_mean_gemv: C=%d unsupported", C);
    int nblk = 1;
    // column blocks per instance: with the mean taken from a few rows of partial sums (npoints > 0) the kernel is two dependent round trips + its column
    // trips of 32 columns -- one trip per workgroup then (up to 2 048 small workgroups: 18 -> ~8 us at layer 6); reading the whole message (npoints == 0)
    // every extra block re-reads it: ~two workgroups per CU, at least 64 columns each
    const int wg_cap = npoints > 0 ? 2048 : 512, min_cols = npoints > 0 ? 32 : 64;
    while (B * nblk < wg_cap && ncols / (nblk * 2) >= min_cols) nblk *= 2;
    const int cpb = cdiv(ncols, nblk);
    hipLaunchKernelGGL(glob_mean_gemv_kernel, dim3(B, cdiv(ncols, cpb)), dim3(256), (size_t)3 * C * sizeof(float), st, f, N, C, W, col0, cpb, ncols, G, ldg,
                       1.0f / (float)(npoints > 0 ? npoints : N));
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int vn_act_rows_launch(const float* T, int ldt, const float* G, int ldg, int B, int N, int C, float neg_slope, float* out,
                       hipStream_t st) {
    const long long total = (long long)B * N * C;
    hipLaunchKernelGGL(vn_act_rows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, T, ldt, G, ldg, N, C, 1.0f - neg_slope, out, total);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int tail_launch(const float* Tc, int ldc, int B, int NP, int Cd, const float* inv_t, const float* fc0_t, const float* misc,
                float neg_slope, float scale_factor, int center_pred, int center_scale, const float* centroid,
                const float* scale0, float* z_so3, float* z_inv, float* s_out, float* t_out, hipStream_t st) {
    LS_REQUIRE(Cd <= 1024 && Cd % 2 == 0, "tail: c_dim=%d unsupported (even, <= 1024)", Cd);
    TailW w{inv_t, fc0_t, misc};
    const size_t smem = (size_t)(3 * Cd + 3 * NP + 3 * (Cd / 2) + 15 * Cd) * sizeof(float);   // X | kdir | H | gemv partials + result
    if (smem > 64 * 1024) LS_HIP_CHECK(hipFuncSetAttribute((const void*)tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));   // c_dim > ~800
    hipLaunchKernelGGL(tail_kernel, dim3(B), dim3(256), smem, st, Tc, ldc, NP, Cd, w, 1.0f - neg_slope, scale_factor,
                       center_pred, center_scale, centroid, scale0, z_so3, z_inv, s_out, t_out);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

}  // namespace ls
