// sdf.hip -- the code-dependent parts of the SDF query path (the 768-wide linears run in gemm.hip).
//
// Replaces FieldWrapper.forward, decoder_type "inner_deepsdf" (/root/reference/model_utils.py:236-251) and the
// first / skip-connected / last layers of DeepSDF_Decoder.forward
// (/root/reference/lib_shape_prior/core/lib/implicit_func/deepsdf_decoder.py:98-121).
//
// The reference builds u = [z_inv(256) | <q, z_so3_c>_c (256) | |q| (1)] per query and feeds the 513-vector to
// lin0 (and again to the skip layer).  Every one of those terms is linear in (q, |q|) once the instance code is
// fixed:   W u + b = (Wb z_so3) q + w_len |q| + (b + Wa z_inv)
// so per INSTANCE we fold a [out,3] matrix, an [out] vector and an [out] bias (sdf_prep_kernel) and the
// per-query work of those layers is a rank-4 affine map (sdf_affine_kernel, HBM-bound) instead of a K=513 GEMM;
// the 513-wide input tensor is never materialised.
#include "ls_common.h"

namespace ls {

// per (instance, code-fed layer): A [out][4] = {Wb z_so3 (3 cols), w_len}, beff [out] = b + Wa z_inv
//   inv_t [L][out] = Wa^T, so3_t [L][out] = Wb^T, wlen [out], bias [out]; z_so3 [B,L,3]; z_inv [B,L]
__global__ __launch_bounds__(256) void sdf_prep_kernel(const float* __restrict__ inv_t, const float* __restrict__ so3_t,
                                                       const float* __restrict__ wlen, const float* __restrict__ bias,
                                                       const float* __restrict__ z_so3, const float* __restrict__ z_inv,
                                                       int L, int out_dim, float* __restrict__ A, float* __restrict__ beff) {
    const int b = blockIdx.y, o = blockIdx.x * 256 + threadIdx.x;
    if (o >= out_dim) return;
    const float* zs = z_so3 + (size_t)b * L * 3;
    const float* zi = z_inv + (size_t)b * L;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, bb = bias[o];
#pragma unroll 8   // sixteen weight loads in flight per thread (un-unrolled, a single-instance fold took 115 us: one L2 round trip per step)
    for (int c = 0; c < L; ++c) {
        const float ws = so3_t[(size_t)c * out_dim + o];
        a0 += ws * zs[c * 3]; a1 += ws * zs[c * 3 + 1]; a2 += ws * zs[c * 3 + 2];
        bb += inv_t[(size_t)c * out_dim + o] * zi[c];
    }
    float* Ap = A + ((size_t)b * out_dim + o) * 4;
    Ap[0] = a0; Ap[1] = a1; Ap[2] = a2; Ap[3] = wlen[o];
    beff[(size_t)b * out_dim + o] = bb;
}

// h[row][o] = relu( (accumulate ? h[row][o] : 0) + A[b][o][0:3] . q + A[b][o][3] |q| + beff[b][o] ),
// q = (query - t[b]) / s[b]   (model_utils.py:236, :238-239).  Workgroup = (column block of 256, instance, slab of 64 rows); a lane owns
// FOUR consecutive columns (one 16-byte store per row), wave w the rows w, w + 4, ... of the slab.
// The normalised query (three IEEE divisions and a square root) is the same for every column: lane l computes it for row l of the slab and
// the row loop reads it back with v_readlane.  (Round 2 recomputed it in all 256 threads for every row and stored one float per thread:
// ~55 instructions per stored float, 581 us per 262 144-row chunk at width 768; once per row: 367 us; four columns per lane: 245 us.)  Same formulas per element: results unchanged.
struct AffCols {   // one lane's four columns of the folded layer
    float4 a[4];
    float4 bb;
};
__device__ __forceinline__ AffCols aff_load(const float* __restrict__ A, const float* __restrict__ beff, int b, int out_dim, int ow) {
    AffCols c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c.a[i] = *reinterpret_cast<const float4*>(A + ((size_t)b * out_dim + ow + i) * 4);
    c.bb = *reinterpret_cast<const float4*>(beff + (size_t)b * out_dim + ow);
    return c;
}
// one row: the lane's four outputs (stored), max|.| over the lane's 16-lane group = 64 columns (returned in every lane of the group)
__device__ __forceinline__ float aff_row(const AffCols& c, float qx, float qy, float qz, float len, bool on, int accumulate, float* __restrict__ hp) {
    float4 v;
    v.x = c.a[0].x * qx + c.a[0].y * qy + c.a[0].z * qz + c.a[0].w * len + c.bb.x;
    v.y = c.a[1].x * qx + c.a[1].y * qy + c.a[1].z * qz + c.a[1].w * len + c.bb.y;
    v.z = c.a[2].x * qx + c.a[2].y * qy + c.a[2].z * qz + c.a[2].w * len + c.bb.z;
    v.w = c.a[3].x * qx + c.a[3].y * qy + c.a[3].z * qz + c.a[3].w * len + c.bb.w;
    if (on) {
        if (accumulate) { const float4 p = *reinterpret_cast<const float4*>(hp); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        *reinterpret_cast<float4*>(hp) = v;
    } else v = make_float4(0.f, 0.f, 0.f, 0.f);
    float m = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
    m = dpp_fmax_rm<0xB1, 0xF>(m);    // quad_perm [1,0,3,2]
    m = dpp_fmax_rm<0x4E, 0xF>(m);    // quad_perm [2,3,0,1]
    m = dpp_fmax_rm<0x141, 0xF>(m);   // row_half_mirror
    m = dpp_fmax_rm<0x140, 0xF>(m);   // row_mirror
    return m;
}
__global__ __launch_bounds__(256) void sdf_affine_kernel(const float* __restrict__ query, const float* __restrict__ s,
                                                         const float* __restrict__ t, const float* __restrict__ A,
                                                         const float* __restrict__ beff, int M, int out_dim, int ldh,
                                                         int accumulate, int rows_per_block, float* __restrict__ h,
                                                         float* __restrict__ rowmax) {
    // rowmax (nullable) [rows][4 * gridDim.x]: max|h[row, 64-column group]| -- the operand range of the GEMM that reads h (gemm.hip, GemmAux)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rm_parts = 4 * gridDim.x, rm_part = 4 * blockIdx.x + (lane >> 4);
    const int b = blockIdx.y, o = blockIdx.x * 256 + lane * 4;
    const int r0 = blockIdx.z * rows_per_block;
    const bool on = o < out_dim;      // (out_dim % 4 == 0: a lane's four columns are all inside or all outside)
    const AffCols c = aff_load(A, beff, b, out_dim, on ? o : 0);
    const float sc = s[b], tx = t[b * 3], ty = t[b * 3 + 1], tz = t[b * 3 + 2];
    const int r1 = min(M, r0 + rows_per_block);
    for (int rb = r0; rb < r1; rb += 64) {
        const int rl = min(rb + lane, r1 - 1);
        const float* qp = query + ((size_t)b * M + rl) * 3;
        const float qxl = (qp[0] - tx) / sc, qyl = (qp[1] - ty) / sc, qzl = (qp[2] - tz) / sc;
        const float lenl = sqrtf(qxl * qxl + qyl * qyl + qzl * qzl);
        const int nr = min(64, r1 - rb);
        for (int j = wave; j < nr; j += 4) {
            const int r = rb + j;
            const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qxl), j));
            const float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qyl), j));
            const float qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qzl), j));
            const float len = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lenl), j));
            const float m = aff_row(c, qx, qy, qz, len, on, accumulate, h + ((size_t)b * M + r) * ldh + o);
            if (rowmax && (lane & 15) == 15) rowmax[((size_t)b * M + r) * rm_parts + rm_part] = m;
        }
    }
}

// Ragged form: rows of SEVERAL instances packed back to back, row r belongs to instance row_inst[r] (batched MISE rounds, where
// every instance contributes a different number of query points).  Same arithmetic as sdf_affine_kernel.
__global__ __launch_bounds__(256) void sdf_affine_rows_kernel(const float* __restrict__ query, const int32_t* __restrict__ row_inst,
                                                              const float* __restrict__ s, const float* __restrict__ t,
                                                              const float* __restrict__ A, const float* __restrict__ beff, long long R,
                                                              int out_dim, int ldh, int accumulate, int rows_per_block,
                                                              float* __restrict__ h, float* __restrict__ rowmax) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rm_parts = 4 * gridDim.x, rm_part = 4 * blockIdx.x + (lane >> 4);
    const int o = blockIdx.x * 256 + lane * 4;
    const long long r0 = (long long)blockIdx.y * rows_per_block;
    const bool on = o < out_dim;
    const int ow = on ? o : 0;
    const long long r1 = min(R, r0 + rows_per_block);
    int bprev = -1;
    AffCols c;
    for (long long rb = r0; rb < r1; rb += 64) {   // lane l normalises row rb + l, the row loop reads it back (see sdf_affine_kernel)
        const long long rl = min(rb + lane, r1 - 1);
        const int bl = row_inst[rl];
        const float sc = s[bl], tx = t[bl * 3], ty = t[bl * 3 + 1], tz = t[bl * 3 + 2];
        const float* qp = query + (size_t)rl * 3;
        const float qxl = (qp[0] - tx) / sc, qyl = (qp[1] - ty) / sc, qzl = (qp[2] - tz) / sc;
        const float lenl = sqrtf(qxl * qxl + qyl * qyl + qzl * qzl);
        const int nr = (int)min((long long)64, r1 - rb);
        for (int j = wave; j < nr; j += 4) {
            const long long r = rb + j;
            const int b = __builtin_amdgcn_readlane(bl, j);
            if (b != bprev) {   // rows of an instance are contiguous: reloaded a handful of times per block
                c = aff_load(A, beff, b, out_dim, ow);
                bprev = b;
            }
            const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qxl), j));
            const float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qyl), j));
            const float qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qzl), j));
            const float len = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lenl), j));
            const float m = aff_row(c, qx, qy, qz, len, on, accumulate, h + (size_t)r * ldh + o);
            if (rowmax && (lane & 15) == 15) rowmax[(size_t)r * rm_parts + rm_part] = m;
        }
    }
}

// last layer + tanh (deepsdf_decoder.py:104-121): sdf[row] = tanh(<h[row], w> + bias); one wave per row
__global__ __launch_bounds__(256) void sdf_out_kernel(const float* __restrict__ h, int ldh, int width, const float* __restrict__ w,
                                                      const float* __restrict__ bias, long long rows, float* __restrict__ sdf) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* hp = h + (size_t)row * ldh;
    float acc = 0.f;
    for (int c = lane * 4; c < width; c += 256) {
        const float4 hv = *reinterpret_cast<const float4*>(hp + c);
        const float4 wv = *reinterpret_cast<const float4*>(w + c);
        acc += hv.x * wv.x + hv.y * wv.y + hv.z * wv.z + hv.w * wv.w;
    }
    acc = wave_sum(acc);
    if (lane == 0) sdf[row] = tanhf(acc + bias[0]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the same path w.r.t. the instance code and the query points (the part the reference gets from
// autograd in More_Solver._optimize_code, /root/reference/lib_more/more_solver.py:191-228: loss.backward() through
// FieldWrapper.forward + DeepSDF_Decoder.forward).  The 768x768 layers are GEMMs against the transposed weights (gemm.hip);
// what is left is element-wise or a reduction:
//   sdf = tanh(<h7, w8> + b8)            dz7 = g (1 - sdf^2) w8 . [h7 > 0]
//   h_l = relu(z_l)                      dz_l = dh_l . [h_l > 0]
//   z_l = A (q,|q|) + beff (+ W h)       dA = sum_m dz (q,|q|)^T,  dbeff = sum_m dz,  d(q,|q|) = A^T dz      (l = 0, latent_in)
//   A = [Wb z_so3 | w_len], beff = b + Wa z_inv      dz_so3 = Wb^T dA[:, :3],  dz_inv = Wa^T dbeff
//   q = (query - t) / s                  dq = d(q)[:3] + d|q| q/|q|,  dquery = dq / s,  dt = -sum_m dq / s,  ds = -sum_m <dq, q> / s

// dz[row][c] = g[row] (1 - sdf[row]^2) w[c] [h[row][c] > 0]
__global__ __launch_bounds__(256) void sdf_out_bwd_kernel(const float* __restrict__ g, const float* __restrict__ sdf,
                                                          const float* __restrict__ w, const float* __restrict__ h, int ldh, int width,
                                                          long long rows, float* __restrict__ dz, float* __restrict__ rowmax,
                                                          const float* __restrict__ wmax) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // one float4 of a row
    const int w4 = width / 4;
    if (i >= rows * w4) return;
    const long long row = i / w4;
    const int c = (int)(i % w4) * 4;
    const float sv = sdf[row], gz = g[row] * (1.0f - sv * sv);
    if (rowmax && c == 0) rowmax[row] = fabsf(gz) * wmax[0];   // an upper bound of max|dz[row, :]| (GemmAux: any bound serves)
    const float4 hv = *reinterpret_cast<const float4*>(h + (size_t)row * ldh + c);
    const float4 wv = *reinterpret_cast<const float4*>(w + c);
    float4 o;
    o.x = hv.x > 0.f ? gz * wv.x : 0.f; o.y = hv.y > 0.f ? gz * wv.y : 0.f;
    o.z = hv.z > 0.f ? gz * wv.z : 0.f; o.w = hv.w > 0.f ? gz * wv.w : 0.f;
    *reinterpret_cast<float4*>(dz + (size_t)row * ldh + c) = o;
}
// dh[row][c] <- dh[row][c] [h[row][c] > 0]   (c < cols, both with row stride ld)
__global__ __launch_bounds__(256) void relu_mask_kernel(float* __restrict__ dh, const float* __restrict__ h, long long rows, int cols,
                                                        int ld) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c4n = cols / 4;
    if (i >= rows * c4n) return;
    const size_t o = (size_t)(i / c4n) * ld + (size_t)(i % c4n) * 4;
    float4 d = *reinterpret_cast<float4*>(dh + o);
    const float4 hv = *reinterpret_cast<const float4*>(h + o);
    d.x = hv.x > 0.f ? d.x : 0.f; d.y = hv.y > 0.f ? d.y : 0.f; d.z = hv.z > 0.f ? d.z : 0.f; d.w = hv.w > 0.f ? d.w : 0.f;
    *reinterpret_cast<float4*>(dh + o) = d;
}
// code-fed layer, reductions over the queries of an instance: dA[b][o][0..3] = sum_m dz[b,m,o] (q_m, |q_m|), dbeff[b][o] = sum_m dz
// (fixed summation order: bit-reproducible)
__global__ __launch_bounds__(256) void sdf_affine_bwd_cols_kernel(const float* __restrict__ query, const float* __restrict__ s,
                                                                  const float* __restrict__ t, const float* __restrict__ dz, int M,
                                                                  int out_dim, int ldh, float* __restrict__ dA, float* __restrict__ dbeff) {
    // workgroup = 64 output channels x 4 row slices (slice w takes rows w, w+4, ...); slices combined in fixed order through LDS
    __shared__ float red[4][64][5];
    const int b = blockIdx.y, lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + lane;
    const bool on = o < out_dim;
    const float sc = s[b], tx = t[b * 3], ty = t[b * 3 + 1], tz = t[b * 3 + 2];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, bb = 0.f;
    for (int r = slice; r < M; r += 4) {
        const float* qp = query + ((size_t)b * M + r) * 3;
        const float qx = (qp[0] - tx) / sc, qy = (qp[1] - ty) / sc, qz = (qp[2] - tz) / sc;
        const float len = sqrtf(qx * qx + qy * qy + qz * qz);
        const float d = on ? dz[((size_t)b * M + r) * ldh + o] : 0.f;
        a0 += d * qx; a1 += d * qy; a2 += d * qz; a3 += d * len; bb += d;
    }
    red[slice][lane][0] = a0; red[slice][lane][1] = a1; red[slice][lane][2] = a2; red[slice][lane][3] = a3; red[slice][lane][4] = bb;
    __syncthreads();
    if (slice == 0 && on) {
        float v[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = ((red[0][lane][k] + red[1][lane][k]) + red[2][lane][k]) + red[3][lane][k];
        float* Ap = dA + ((size_t)b * out_dim + o) * 4;
        Ap[0] = v[0]; Ap[1] = v[1]; Ap[2] = v[2]; Ap[3] = v[3];
        dbeff[(size_t)b * out_dim + o] = v[4];
    }
}
// code-fed layer, reduction over the output channels of a query: dQ[row][0..3] (+)= sum_o dz[row][o] A[b][o][0..3]; one wave per row
__global__ __launch_bounds__(256) void sdf_affine_bwd_rows_kernel(const float* __restrict__ dz, const float* __restrict__ A, int M,
                                                                  int out_dim, int ldh, int accumulate, long long rows,
                                                                  float* __restrict__ dQ) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = (int)(row / M);
    const float* Ab = A + (size_t)b * out_dim * 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int o = lane; o < out_dim; o += 64) {
        const float d = dz[(size_t)row * ldh + o];
        const float4 av = *reinterpret_cast<const float4*>(Ab + (size_t)o * 4);
        a0 += d * av.x; a1 += d * av.y; a2 += d * av.z; a3 += d * av.w;
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
    if (lane == 0) {
        float* p = dQ + (size_t)row * 4;
        if (accumulate) { a0 += p[0]; a1 += p[1]; a2 += p[2]; a3 += p[3]; }
        p[0] = a0; p[1] = a1; p[2] = a2; p[3] = a3;
    }
}
// dz_so3[b][c][x] = sum over the code-fed layers of sum_o so3_t[c][o] dA[b][o][x]; dz_inv[b][c] likewise with inv_t / dbeff
__global__ __launch_bounds__(256) void sdf_code_grad_kernel(const float* __restrict__ so3_t0, const float* __restrict__ inv_t0,
                                                            const float* __restrict__ dA0, const float* __restrict__ db0,
                                                            const float* __restrict__ so3_t1, const float* __restrict__ inv_t1,
                                                            const float* __restrict__ dA1, const float* __restrict__ db1, int L,
                                                            int out_dim, float* __restrict__ g_so3, float* __restrict__ g_inv) {
    // one wave per (instance, latent channel)
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (c >= L) return;
    float gx = 0.f, gy = 0.f, gz = 0.f, gi = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        const float* so3_t = pass ? so3_t1 : so3_t0;
        const float* inv_t = pass ? inv_t1 : inv_t0;
        const float* dA = pass ? dA1 : dA0;
        const float* db = pass ? db1 : db0;
        if (!so3_t) continue;
        for (int o = lane; o < out_dim; o += 64) {
            const float ws = so3_t[(size_t)c * out_dim + o], wi = inv_t[(size_t)c * out_dim + o];
            const float4 a = *reinterpret_cast<const float4*>(dA + ((size_t)b * out_dim + o) * 4);
            gx += ws * a.x; gy += ws * a.y; gz += ws * a.z;
            gi += wi * db[(size_t)b * out_dim + o];
        }
    }
    gx = wave_sum(gx); gy = wave_sum(gy); gz = wave_sum(gz); gi = wave_sum(gi);
    if (lane == 0) {
        float* p = g_so3 + ((size_t)b * L + c) * 3;
        p[0] = gx; p[1] = gy; p[2] = gz;
        g_inv[(size_t)b * L + c] = gi;
    }
}
// dq = dQ[:3] + dQ[3] q/|q| (0 at q = 0, as torch's norm backward); dquery = dq/s; per instance dt = -sum dq / s, ds = -sum <dq,q> / s
__global__ __launch_bounds__(256) void sdf_query_grad_kernel(const float* __restrict__ query, const float* __restrict__ s,
                                                             const float* __restrict__ t, const float* __restrict__ dQ, int M,
                                                             float* __restrict__ g_query, float* __restrict__ g_t, float* __restrict__ g_s) {
    __shared__ float red[4][4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float sc = s[b], tx = t[b * 3], ty = t[b * 3 + 1], tz = t[b * 3 + 2];
    float sx = 0.f, sy = 0.f, sz = 0.f, ss = 0.f;
    for (int r = tid; r < M; r += 256) {
        const size_t row = (size_t)b * M + r;
        const float* qp = query + row * 3;
        const float qx = (qp[0] - tx) / sc, qy = (qp[1] - ty) / sc, qz = (qp[2] - tz) / sc;
        const float len = sqrtf(qx * qx + qy * qy + qz * qz);
        const float4 d = *reinterpret_cast<const float4*>(dQ + row * 4);
        const float il = len > 0.f ? d.w / len : 0.f;
        const float dx = d.x + il * qx, dy = d.y + il * qy, dzv = d.z + il * qz;
        if (g_query) { g_query[row * 3] = dx / sc; g_query[row * 3 + 1] = dy / sc; g_query[row * 3 + 2] = dzv / sc; }
        sx += dx; sy += dy; sz += dzv; ss += dx * qx + dy * qy + dzv * qz;
    }
    sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz); ss = wave_sum(ss);
    if (lane == 0) { red[wave][0] = sx; red[wave][1] = sy; red[wave][2] = sz; red[wave][3] = ss; }
    __syncthreads();
    if (tid < 4) {
        const float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        if (tid < 3) g_t[b * 3 + tid] = -v / sc;
        else g_s[b] = -v / sc;
    }
}
// Wt[k][o] = W[o][k]
__global__ void transpose_kernel(const float* __restrict__ W, int rows, int cols, float* __restrict__ Wt) {
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (by + i < rows && bx + tx < cols) tile[i][tx] = W[(size_t)(by + i) * cols + bx + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (bx + i < cols && by + tx < rows) Wt[(size_t)(bx + i) * rows + by + tx] = tile[tx][i];
}

int sdf_out_bwd_launch(const float* g, const float* sdf, const float* w, const float* h, int ldh, int width, long long rows, float* dz,
                       hipStream_t st, float* rowmax, const float* wmax) {
    if (!wmax) rowmax = nullptr;
    hipLaunchKernelGGL(sdf_out_bwd_kernel, dim3(cdiv(rows * (width / 4), 256)), dim3(256), 0, st, g, sdf, w, h, ldh, width, rows, dz, rowmax, wmax);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int relu_mask_launch(float* dh, const float* h, long long rows, int cols, int ld, hipStream_t st) {
    LS_REQUIRE(cols % 4 == 0 && ld % 4 == 0, "relu_mask: cols/ld must be multiples of 4");
    hipLaunchKernelGGL(relu_mask_kernel, dim3(cdiv(rows * (cols / 4), 256)), dim3(256), 0, st, dh, h, rows, cols, ld);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int sdf_affine_bwd_launch(const float* query, const float* s, const float* t, const float* dz, const float* A, int B, int M, int out_dim,
                          int ldh, int accumulate, float* dA, float* dbeff, float* dQ, bool need_code, hipStream_t st) {
    // the reductions over the queries feed only the CODE gradient (a pose refinement with a fixed code does not need them)
    if (need_code)
        hipLaunchKernelGGL(sdf_affine_bwd_cols_kernel, dim3(cdiv(out_dim, 64), B), dim3(256), 0, st, query, s, t, dz, M, out_dim, ldh, dA, dbeff);
    hipLaunchKernelGGL(sdf_affine_bwd_rows_kernel, dim3(cdiv((long long)B * M, 4)), dim3(256), 0, st, dz, A, M, out_dim, ldh, accumulate,
                       (long long)B * M, dQ);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int sdf_code_grad_launch(const float* so3_t0, const float* inv_t0, const float* dA0, const float* db0, const float* so3_t1,
                         const float* inv_t1, const float* dA1, const float* db1, int B, int L, int out_dim, float* g_so3, float* g_inv,
                         hipStream_t st) {
    hipLaunchKernelGGL(sdf_code_grad_kernel, dim3(cdiv(L, 4), B), dim3(256), 0, st, so3_t0, inv_t0, dA0, db0, so3_t1, inv_t1, dA1, db1, L,
                       out_dim, g_so3, g_inv);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int sdf_query_grad_launch(const float* query, const float* s, const float* t, const float* dQ, int B, int M, float* g_query, float* g_t,
                          float* g_s, hipStream_t st) {
    hipLaunchKernelGGL(sdf_query_grad_kernel, dim3(B), dim3(256), 0, st, query, s, t, dQ, M, g_query, g_t, g_s);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int transpose_launch(const float* W, int rows, int cols, float* Wt, hipStream_t st) {
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, st, W, rows, cols, Wt);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

int sdf_prep_launch(const float* inv_t, const float* so3_t, const float* wlen, const float* bias, const float* z_so3,
                    const float* z_inv, int B, int L, int out_dim, float* A, float* beff, hipStream_t st) {
    hipLaunchKernelGGL(sdf_prep_kernel, dim3(cdiv(out_dim, 256), B), dim3(256), 0, st, inv_t, so3_t, wlen, bias, z_so3, z_inv, L,
                       out_dim, A, beff);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int sdf_affine_launch(const float* query, const float* s, const float* t, const float* A, const float* beff, int B, int M,
                      int out_dim, int ldh, int accumulate, float* h, hipStream_t st, float* rowmax) {
    const int rpb = 64;
    LS_REQUIRE(out_dim % 4 == 0 && ldh % 4 == 0, "sdf_affine: width %d / row stride %d must be multiples of 4", out_dim, ldh);
    // gridDim.y / .z are limited to 65535: say so instead of a generic launch failure (direct C-ABI callers; ops.sdf_decode chunks)
    LS_REQUIRE(B <= 65535 && cdiv(M, rpb) <= 65535, "sdf_decode: B=%d or M=%d too large for one call (B <= 65535, M <= %d): split the queries", B, M,
               65535 * rpb);
    hipLaunchKernelGGL(sdf_affine_kernel, dim3(cdiv(out_dim, 256), B, cdiv(M, rpb)), dim3(256), 0, st, query, s, t, A, beff, M,
                       out_dim, ldh, accumulate, rpb, h, rowmax);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int sdf_affine_rowmax_parts(int out_dim) { return 4 * cdiv(out_dim, 256); }
int sdf_affine_rows_launch(const float* query, const int32_t* row_inst, const float* s, const float* t, const float* A, const float* beff,
                           long long R, int out_dim, int ldh, int accumulate, float* h, hipStream_t st, float* rowmax) {
    const int rpb = 64;
    LS_REQUIRE(out_dim % 4 == 0 && ldh % 4 == 0, "sdf_affine: width %d / row stride %d must be multiples of 4", out_dim, ldh);
    LS_REQUIRE(cdiv(R, rpb) <= 65535, "sdf_decode_rows: R=%lld rows too many for one call (<= %d): split the rows", R, 65535 * rpb);
    hipLaunchKernelGGL(sdf_affine_rows_kernel, dim3(cdiv(out_dim, 256), (unsigned)cdiv(R, rpb)), dim3(256), 0, st, query, row_inst, s, t, A,
                       beff, R, out_dim, ldh, accumulate, rpb, h, rowmax);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int sdf_out_launch(const float* h, int ldh, int width, const float* w, const float* bias, long long rows, float* sdf,
                   hipStream_t st) {
    LS_REQUIRE(width % 4 == 0 && ldh % 4 == 0, "sdf_out: width/ldh must be multiples of 4");
    hipLaunchKernelGGL(sdf_out_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, h, ldh, width, w, bias, rows, sdf);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

}  // namespace ls
