// knn_mfma.hip -- the same bit-exact 16-NN as knn.hip, with the idle matrix cores used as an EXACT-SAFE FILTER.
//
// Replaces pytorch3d.ops.knn_points as called at
//   /root/reference/lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:139-141        (C >= 32 feature layers)
//
// knn.hip spends ~150 M packed VALU instructions per layer-1 launch on the canonical (sub, mul, add) distance of EVERY
// pair, although after the first candidate tile only ~16/(64 t) of the pairs of tile t can still enter a top-16 list.
// Here, for every 64x64 tile after a workgroup's first:
//   1. S = q . s on the matrix cores (v_mfma_f32_32x32x2_f32, K = 3C), giving d^ = |q|^2 + |s|^2 - 2S;
//   2. a pair is DROPPED only if  d^ - eps > kth(q)  (kth = the query's current exact 16-th distance), where
//      eps = 6 (D+4) 2^-24 (|q|^2 + |s|^2) bounds |d^ - d_true| + |d_canonical - d_true| with 50 % slack
//      (gamma_{D+3} (|q|+|s|)^2 each, (|q|+|s|)^2 <= 2 (|q|^2+|s|^2)); fp32 accumulation of non-negative terms is
//      monotone, so a dropped pair provably has canonical distance > kth and could never have been inserted;
//   3. the surviving pairs (a few hundred of 4096) are compacted into an LDS list and get the CANONICAL distance
//      (same accq<> chain as knn.hip) one pair per thread; everything else is +inf in the distance tile;
//   4. the unchanged row-parallel selection (knn_common.h) merges the tile.
// The top-K lists only ever hold canonical distances, so the result is bit-identical to knn.hip / the oracle by
// construction; the filter only decides what is worth computing.  The first tile of a workgroup (empty lists: every
// pair passes) and any tile whose survivor list would overflow take the dense VALU path of knn.hip.
#include "knn_common.h"

namespace ls {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KM_CC = 32;
constexpr int KM_ROW = 3 * KM_CC + 4;
constexpr int KM_CAP = 2048;            // survivor-list capacity (pairs per tile)
constexpr int KM_PER = KM_CAP / 256;    // pairs per thread in the exact phase

// squared norms of feature rows: one wave per point
__global__ __launch_bounds__(256) void row_norms_kernel(const float* __restrict__ f, int row_f, long long npts,
                                                        float* __restrict__ norms) {
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= npts) return;
    const int lane = threadIdx.x & 63;
    const float* r = f + (size_t)p * row_f;
    float s = 0.f;
    for (int c = lane * 4; c < row_f; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(r + c);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s = wave_sum(s);
    if (lane == 0) norms[p] = s;
}

template <bool FMA>
__global__ __launch_bounds__(256, 2) void knn_mfma_kernel(const float* __restrict__ dstf, const float* __restrict__ srcf,
                                                          const int32_t* __restrict__ dst_rows, const float* __restrict__ nrm_dst,
                                                          const float* __restrict__ nrm_src, int Nd, int dst_n, int Ns, int C, int K,
                                                          int32_t* __restrict__ idx_out, float* __restrict__ dist_out, int qtiles,
                                                          int splits, int tiles_per_split, u64* __restrict__ partial, float epsE) {
    constexpr int CC = KM_CC, ROW = KM_ROW;
    constexpr int LC_FLOATS = (KNN_TS * ROW > KNN_TQ * KNN_LD) ? KNN_TS * ROW : KNN_TQ * KNN_LD;
    __shared__ __attribute__((aligned(16))) float lq[KNN_TQ * ROW];
    __shared__ __attribute__((aligned(16))) float lc[LC_FLOATS];  // candidate chunk; later the 64x64 distance tile
    __shared__ unsigned short llist[KM_CAP];
    __shared__ float lnq[KNN_TQ], lkth[KNN_TQ];
    __shared__ int lqrow[KNN_TQ];
    __shared__ int lcount;
    float* ldist = lc;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int sp = logical % splits;
    const int b = (logical / splits) / qtiles, qt = (logical / splits) % qtiles;
    const int q0 = qt * KNN_TQ;
    const int s_begin = sp * tiles_per_split * KNN_TS;
    const int s_end = min(Ns, s_begin + tiles_per_split * KNN_TS);
    const size_t row_f = (size_t)3 * C;
    const float* dbase = dstf + (size_t)b * dst_n * row_f;
    const float* sbase = srcf + (size_t)b * Ns * row_f;
    const float* nsb = nrm_src + (size_t)b * Ns;

    if (tid < KNN_TQ) {
        const int q = q0 + tid;
        int r = -1;
        if (q < Nd) r = dst_rows ? dst_rows[(size_t)b * Nd + q] : q;
        lqrow[tid] = r;
        lnq[tid] = r >= 0 ? nrm_dst[(size_t)b * dst_n + r] : 0.f;
        lkth[tid] = r >= 0 ? INFINITY : -INFINITY;  // padding queries never pass the filter
    }
    if (tid == 0) lcount = 0;
    __syncthreads();

    const int tx = tid & 15, ty = tid >> 4;             // dense micro-tile: candidates tx+16j, queries ty*4+i
    const int wm = wave >> 1, wn = wave & 1;            // MFMA tile: queries wm*32.., candidates wn*32..
    const int l31 = lane & 31, lh = lane >> 5;

    u64 lk[4], rkey[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { lk[i] = ~0ull; rkey[i] = ~0ull; }

    const int nch = C / CC;
    const bool q_once = nch == 1;
    Stager<CC> sq, sc;
    sq.load(dbase, lqrow, 0, 0, row_f, C, 0, tid);
    sc.load(sbase, nullptr, s_begin, Ns, row_f, C, 0, tid);
    if (q_once) sq.store(lq, ROW, tid);

    for (int s0 = s_begin; s0 < s_end; s0 += KNN_TS) {
        const bool first = s0 == s_begin;
        float acc[4][4];
        f32x16 S;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.0f;

        auto dense_chunk = [&]() {
#pragma unroll 1
            for (int d4 = 0; d4 < 3 * CC; d4 += 4) {
                float4 qv[4], cv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    qv[i] = *reinterpret_cast<const float4*>(&lq[(ty * 4 + i) * ROW + d4]);
                    cv[i] = *reinterpret_cast<const float4*>(&lc[(tx + 16 * i) * ROW + d4]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float a = acc[i][j];
                        a = accq<FMA>(a, qv[i].x, cv[j].x); a = accq<FMA>(a, qv[i].y, cv[j].y);
                        a = accq<FMA>(a, qv[i].z, cv[j].z); a = accq<FMA>(a, qv[i].w, cv[j].w);
                        acc[i][j] = a;
                    }
            }
        };
        // prefetch the stage that follows (pass, ch): pass 0 = A (MFMA or dense), pass 1 = B (exact re-walk, nch > 1 only)
        auto prefetch_after = [&](int pass, int ch, bool dense_tile) {
            int nc = ch + 1, ns0 = s0;
            if (nc == nch) {
                nc = 0;
                if (pass == 1 || dense_tile || q_once) ns0 = s0 + KNN_TS;  // otherwise pass B re-walks this tile
            }
            if (ns0 < s_end) {
                if (!q_once) sq.load(dbase, lqrow, 0, 0, row_f, C, nc * CC, tid);
                sc.load(sbase, nullptr, ns0, Ns, row_f, C, nc * CC, tid);
            }
        };

        // ---------------- pass A: dense distances (first tile) or S = q.s on the matrix cores
        for (int ch = 0; ch < nch; ++ch) {
            __syncthreads();
            if (!q_once) sq.store(lq, ROW, tid);
            sc.store(lc, ROW, tid);
            __syncthreads();
            prefetch_after(0, ch, first);
            if (first) {
                dense_chunk();
            } else {
#pragma unroll 2
                for (int d8 = 0; d8 < 3 * CC; d8 += 8) {
                    const float4 a = *reinterpret_cast<const float4*>(&lq[(wm * 32 + l31) * ROW + d8 + lh * 4]);
                    const float4 bb = *reinterpret_cast<const float4*>(&lc[(wn * 32 + l31) * ROW + d8 + lh * 4]);
                    S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bb.x, S, 0, 0, 0);
                    S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bb.y, S, 0, 0, 0);
                    S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bb.z, S, 0, 0, 0);
                    S = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bb.w, S, 0, 0, 0);
                }
            }
        }

        bool dense = first;
        float res[KM_PER];
        int total = 0;
        if (!first) {
            // ---------------- filter: compact the pairs that may still enter a list
            const int cc = wn * 32 + l31;
            const bool cvalid = (s0 + cc) < Ns;
            const float nsv = cvalid ? nsb[s0 + cc] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float nn = lnq[qr] + nsv;
                const float dh = nn - 2.0f * S[r];
                const bool pass = cvalid & !((dh - epsE * nn) > lkth[qr]);
                const u64 m = __ballot(pass);
                if (m) {
                    int base = 0;
                    if (lane == (int)__builtin_ctzll(m)) base = atomicAdd(&lcount, (int)__builtin_popcountll(m));
                    base = __shfl(base, (int)__builtin_ctzll(m), 64);
                    const int slot = base + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                    if (pass && slot < KM_CAP) llist[slot] = (unsigned short)((qr << 8) | cc);
                }
            }
            __syncthreads();
            total = lcount;
            dense = total > KM_CAP;  // survivor list overflow (adversarial data): fall back to the dense path
            // ---------------- pass B: canonical distances of the survivors (or of everything on overflow)
#pragma unroll
            for (int u = 0; u < KM_PER; ++u) res[u] = 0.0f;
            for (int ch = 0; ch < nch; ++ch) {
                if (!q_once) {  // multi-chunk layers re-walk the chunks of this tile
                    __syncthreads();
                    sq.store(lq, ROW, tid);
                    sc.store(lc, ROW, tid);
                    __syncthreads();
                    prefetch_after(1, ch, false);
                }
                if (dense) {
                    dense_chunk();
                } else {
                    const int rounds = (total + 255) >> 8;
#pragma unroll 1
                    for (int d4 = 0; d4 < 3 * CC; d4 += 4) {
#pragma unroll
                        for (int u = 0; u < KM_PER; ++u) {
                            if (u < rounds) {
                                const int i = tid + u * 256;
                                const unsigned pr = i < total ? llist[i] : 0u;
                                const float4 qv = *reinterpret_cast<const float4*>(&lq[(pr >> 8) * ROW + d4]);
                                const float4 cv = *reinterpret_cast<const float4*>(&lc[(pr & 255u) * ROW + d4]);
                                float a = res[u];
                                a = accq<FMA>(a, qv.x, cv.x); a = accq<FMA>(a, qv.y, cv.y);
                                a = accq<FMA>(a, qv.z, cv.z); a = accq<FMA>(a, qv.w, cv.w);
                                res[u] = a;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();  // everyone is done reading lc before it becomes the distance tile
        if (dense) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float4*>(&ldist[(ty * 4 + i) * KNN_LD + tx * 4]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
        } else {
            for (int t = tid; t < KNN_TQ * KNN_LD; t += 256) ldist[t] = INFINITY;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < KM_PER; ++u) {
                const int i = tid + u * 256;
                if (i < total) {
                    const unsigned pr = llist[i];
                    const unsigned qr = pr >> 8, cc = pr & 255u;
                    ldist[qr * KNN_LD + (cc & 15u) * 4 + (cc >> 4)] = res[u];
                }
            }
        }
        if (tid == 0) lcount = 0;
        __syncthreads();
        select_tile<false>(ldist, lk, rkey, s0, Ns, K, wave, lane);
        // refresh the filter thresholds: the row's K-th distance (+inf while the list is not full)
        if ((lane & 15) == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qr = wave * 16 + g * 4 + (lane >> 4);
                const unsigned hi = (unsigned)(rkey[g] >> 32);
                if (lqrow[qr] >= 0) lkth[qr] = hi == 0xFFFFFFFFu ? INFINITY : __uint_as_float(hi);
            }
        }
    }
    write_lists(lk, b, q0, Nd, K, wave, lane, splits, sp, partial, idx_out, dist_out);
}

int row_norms_launch(const float* f, int row_f, long long npts, float* norms, hipStream_t st) {
    LS_REQUIRE(row_f % 4 == 0, "row_norms: row length must be a multiple of 4");
    hipLaunchKernelGGL(row_norms_kernel, dim3(cdiv(npts, 4)), dim3(256), 0, st, f, row_f, npts, norms);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

int knn_mfma_launch(const float* dst, const float* src, const int32_t* dst_rows, const float* nrm_dst, const float* nrm_src, int B,
                    int Nd, int dst_n, int Ns, int C, int K, bool fma, int32_t* idx_out, float* dist_out, int splits, int tps,
                    u64* partial, hipStream_t st) {
    const int qtiles = cdiv(Nd, KNN_TQ);
    const float epsE = 6.0f * (float)(3 * C + 4) * 5.9604645e-8f;
    dim3 grid(B * qtiles * splits), block(256);
    if (fma)
        hipLaunchKernelGGL(knn_mfma_kernel<true>, grid, block, 0, st, dst, src, dst_rows, nrm_dst, nrm_src, Nd, dst_n, Ns, C, K, idx_out,
                           dist_out, qtiles, splits, tps, partial, epsE);
    else
        hipLaunchKernelGGL(knn_mfma_kernel<false>, grid, block, 0, st, dst, src, dst_rows, nrm_dst, nrm_src, Nd, dst_n, Ns, C, K, idx_out,
                           dist_out, qtiles, splits, tps, partial, epsE);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

}  // namespace ls
