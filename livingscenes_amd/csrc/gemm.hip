// gemm.hip -- out[M,N] = act(A[M,K] * W[N,K]^T + bias), exact fp32 on the CDNA4 matrix cores.
//
// This is the ONLY MFMA-shaped work on the path: the VecLinear channel contraction
// (/root/reference/lib_shape_prior/core/lib/vec_sim3/vec_layers.py:121-136, F.linear at :134) applied to
// x-major feature rows [B*N*3, C_in], and the DeepSDF linears
// (/root/reference/lib_shape_prior/core/lib/implicit_func/deepsdf_decoder.py:98-121).
//
// v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, bitwise an fmaf chain; 157 TFLOP/s peak = fp32 vector peak,
// but it leaves the VALU free and needs one VGPR per operand).  Workgroup tile 128x128, BK=16, 4 waves as
// 2(M) x 2(N), each wave 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).  Both operands are K-contiguous
// ("A * W^T"), staged through LDS with rows padded to 20 floats: a 16-lane ds_read_b128 group then touches
// 16 distinct 16-byte bank slots (20*i mod 64 is a permutation of the multiples of 4).  The k index is
// permuted inside a BK block (lanes 0-31 take k = 8h+s, lanes 32-63 take k = 8h+4+s at MFMA step (h,s)) so
// that every lane fetches its four k-values with ONE ds_read_b128; A and B use the same permutation, so the
// product is unchanged.  Next-tile global loads are issued before the MFMA block (register double buffer).
#include "ls_common.h"

namespace ls {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GM = 128, GN = 128, GK = 16, GLD = 20;

// Optional row gather (down-sampled encoder layers): output row m = (b*gNd + n)*3 + x reads A row
// (b*gNs + a_rows[b*gNd + n])*3 + x, i.e. the GEMM runs only on the FPS-selected points of each instance.
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                       int ldw, const float* __restrict__ bias, float* __restrict__ out,
                                                       int ldc, int M, int N, int K, int relu, int ntiles_n,
                                                       const int32_t* __restrict__ a_rows, int gNd, int gNs) {
    constexpr int STG = 32 * 68;  // epilogue staging: 32 rows x (64 + 4) floats per wave
    __shared__ __attribute__((aligned(16))) float smem[(4 * STG > (GM + GN) * GLD) ? 4 * STG : (GM + GN) * GLD];
    float* As = smem;
    float* Bs = smem + GM * GLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    // consecutive logical ids walk the N tiles of one M tile: they share the A panel in one XCD's L2
    const int tm = logical / ntiles_n, tn = logical % ntiles_n;
    const int m0 = tm * GM, n0 = tn * GN;
    const int wm = wave >> 1, wn = wave & 1;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // staging map: 128 rows x 4 float4 per operand tile = 512 float4, two per thread
    const int sr0 = tid >> 2, sk = (tid & 3) * 4;  // rows sr0 and sr0+64
    float4 ra[2], rb[2];
    long long arow[2];  // source row of A for this thread's two staged rows (-1 = out of range)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int gm = m0 + sr0 + h * 64;
        long long r = gm < M ? gm : -1;
        if (a_rows && gm < M) {
            const int pt = gm / 3, x = gm - pt * 3;
            const int bb = pt / gNd;
            r = ((long long)bb * gNs + a_rows[pt]) * 3 + x;
        }
        arow[h] = r;
    }
    auto gload = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = sr0 + h * 64;
            const int gn = n0 + r, gk = k0 + sk;
            ra[h] = (arow[h] >= 0 && gk < K) ? *reinterpret_cast<const float4*>(A + (size_t)arow[h] * lda + gk) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[h] = (gn < N && gk < K) ? *reinterpret_cast<const float4*>(W + (size_t)gn * ldw + gk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = sr0 + h * 64;
            *reinterpret_cast<float4*>(&As[r * GLD + sk]) = ra[h];
            *reinterpret_cast<float4*>(&Bs[r * GLD + sk]) = rb[h];
        }
    };

    gload(0);
    for (int k0 = 0; k0 < K; k0 += GK) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (k0 + GK < K) gload(k0 + GK);  // in flight under the MFMA block
        const int lr = lane & 31, lk = (lane >> 5) * 4;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const float4*>(&As[(wm * 64 + i * 32 + lr) * GLD + h * 8 + lk]);
                b[i] = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + i * 32 + lr) * GLD + h * 8 + lk]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }

    // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Storing straight from
    // that layout costs 64 global_store_dword per lane (store-issue-bound for the K = 32/64 table GEMMs, whose
    // epilogue outweighs their main loop).  Instead each wave transposes its 64x64 sub-tile through LDS in two 32-row
    // halves (the operand buffers are free now) and writes full 256-byte row segments with 16 dwordx4 stores per lane.
    __syncthreads();  // all waves are done with As / Bs
    float* stg = smem + wave * STG;  // 8.7 KB per wave
    const int col_l = lane & 31, rowh = (lane >> 5) * 4;
    const bool vec_ok = (ldc % 4 == 0) && (((uintptr_t)out & 15) == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wn * 64 + j * 32 + col_l;
            const float bv = (bias && gn < N) ? bias[gn] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] + bv;
                if (relu) v = fmaxf(v, 0.0f);
                stg[((r & 3) + 8 * (r >> 2) + rowh) * 68 + j * 32 + col_l] = v;
            }
        }
        // wave-local hand-off through LDS: same wave writes and reads, LDS ops of a wave complete in order
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = u * 64 + lane;           // 32 rows x 16 float4
            const int rr = idx >> 4, c4 = (idx & 15) * 4;
            const int gm = m0 + wm * 64 + i * 32 + rr;
            const int gn = n0 + wn * 64 + c4;
            if (gm < M && gn < N) {
                const float4 v = *reinterpret_cast<const float4*>(&stg[rr * 68 + c4]);
                float* op = out + (size_t)gm * ldc + gn;
                if (vec_ok && gn + 3 < N) {
                    *reinterpret_cast<float4*>(op) = v;
                } else {
                    op[0] = v.x;
                    if (gn + 1 < N) op[1] = v.y;
                    if (gn + 2 < N) op[2] = v.z;
                    if (gn + 3 < N) op[3] = v.w;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

int gemm_dispatch_gather(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M, int N,
                         int K, int relu, const int32_t* a_rows, int gNd, int gNs, hipStream_t st);
int gemm_dispatch(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M, int N,
                  int K, int relu, hipStream_t st) {
    return gemm_dispatch_gather(A, lda, W, ldw, bias, out, ldc, M, N, K, relu, nullptr, 0, 0, st);
}
int gemm_dispatch_gather(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M, int N,
                         int K, int relu, const int32_t* a_rows, int gNd, int gNs, hipStream_t st) {
    LS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem (M=%d N=%d K=%d)", M, N, K);
    LS_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0, "gemm: K, lda, ldw must be multiples of 4 (K=%d lda=%d ldw=%d)", K, lda, ldw);
    LS_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "gemm: A and W must be 16-byte aligned");
    const int tm = cdiv(M, GM), tn = cdiv(N, GN);
    hipLaunchKernelGGL(gemm_f32_kernel, dim3(tm * tn), dim3(256), 0, st, A, lda, W, ldw, bias, out, ldc, M, N, K, relu, tn, a_rows,
                       gNd, gNs);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

}  // namespace ls
