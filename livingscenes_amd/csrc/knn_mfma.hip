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

// ---------------------------------------------------------------------------------------------------------------------
// v2 (C == 32 layers): lists may be SEEDED (previous layer's graph, knn_common.h), so the admission thresholds are
// near-final from the first tile and only ~2-3 % of the pairs survive the MFMA filter.  Survivors are appended to a compact
// list (exact phase: one pair per thread) and to per-query buckets of at most KM_BKT keys per tile, which feed the
// row-parallel insertion directly -- no dense distance tile is built.  A tile whose buckets / list would overflow (always
// the first tile of an un-seeded workgroup) falls back to the dense VALU path.
constexpr int KM_BKT = 16;      // survivors per query per tile kept in the bucket
constexpr int KM_LIST = 1024;   // survivors per tile in the compact list

template <bool FMA>
__global__ __launch_bounds__(256, 2) void knn_mfma_kernel(const float* __restrict__ dstf, const float* __restrict__ srcf,
                                                          const int32_t* __restrict__ dst_rows, const float* __restrict__ nrm_dst,
                                                          const float* __restrict__ nrm_src, int Nd, int dst_n, int Ns, int C, int K,
                                                          int32_t* __restrict__ idx_out, float* __restrict__ dist_out, int qtiles,
                                                          int splits, int tiles_per_split, u64* __restrict__ partial, float epsE,
                                                          const int32_t* __restrict__ seed_idx, int seed_n, int seed_by_row) {
    constexpr int CC = KM_CC, ROW = KM_ROW;
    constexpr int LC_FLOATS = (KNN_TS * ROW > KNN_TQ * KNN_LD) ? KNN_TS * ROW : KNN_TQ * KNN_LD;
    __shared__ __attribute__((aligned(16))) float lq[KNN_TQ * ROW];
    __shared__ __attribute__((aligned(16))) float lc[LC_FLOATS];   // candidate tile (row layout); dense fallback: distance tile
    __shared__ __attribute__((aligned(16))) u64 lbkey[KNN_TQ * KM_BKT];  // per-query survivor keys of the current tile
    __shared__ unsigned short llist[KM_LIST];                     // compact survivor list: q << 10 | slot << 6 | c
    __shared__ float lnq[KNN_TQ], lkth[KNN_TQ];
    __shared__ int lqrow[KNN_TQ], lcnt[KNN_TQ];
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
        lcnt[tid] = 0;
    }
    if (tid == 0) lcount = 0;
    __syncthreads();

    const int tx = tid & 15, ty = tid >> 4;             // dense fallback micro-tile: candidates tx+16j, queries ty*4+i
    const int wm = wave >> 1, wn = wave & 1;            // MFMA tile: queries wm*32.., candidates wn*32..
    const int l31 = lane & 31, lh = lane >> 5;

    u64 lk[4], rkey[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { lk[i] = ~0ull; rkey[i] = ~0ull; }

    Stager<CC> sq, sc;
    sq.load(dbase, lqrow, 0, 0, row_f, C, 0, tid);
    sc.load(sbase, nullptr, s_begin, Ns, row_f, C, 0, tid);
    sq.store(lq, ROW, tid);  // C == 32: the query tile is loop invariant

    auto refresh_kth = [&]() {  // the row's K-th canonical distance (+inf while the list is not full)
        if ((lane & 15) == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qr = wave * 16 + g * 4 + (lane >> 4);
                const unsigned hi = (unsigned)(rkey[g] >> 32);
                if (lqrow[qr] >= 0) lkth[qr] = hi == 0xFFFFFFFFu ? INFINITY : __uint_as_float(hi);
            }
        }
    };

    const bool seeded = seed_idx != nullptr && sp == 0;
    if (seeded) {
        u64* lseed = reinterpret_cast<u64*>(lc);
        compute_seed_keys<FMA>(lseed, seed_idx, seed_n, seed_by_row != 0, dbase, sbase, lqrow, b, q0, Ns, C, tid);
        __syncthreads();
        seed_lists(lseed, lk, rkey, K, wave, lane);
        refresh_kth();
    }

    // per-lane constants of the filter: the 16 accumulator rows of this lane and their query norms
    float nqv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) nqv[r] = lnq[wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];

    for (int s0 = s_begin; s0 < s_end; s0 += KNN_TS) {
        __syncthreads();               // previous tile fully consumed (lc, buckets, thresholds written)
        sc.store(lc, ROW, tid);
        __syncthreads();
        if (s0 + KNN_TS < s_end) sc.load(sbase, nullptr, s0 + KNN_TS, Ns, row_f, C, 0, tid);  // next tile in flight

        const bool force_dense = !seeded && s0 == s_begin;  // empty lists: every pair would pass the filter
        // ---------------- S = q . s on the matrix cores
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.0f;
        if (!force_dense) {
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
        // ---------------- filter: a pair survives unless d^ - eps > kth(q)
        if (!force_dense) {
            const int cc = wn * 32 + l31;
            const bool cvalid = (s0 + cc) < Ns;
            const float nsv = cvalid ? nsb[s0 + cc] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float nn = nqv[r] + nsv;
                const float dh = nn - 2.0f * S[r];
                const bool pass = cvalid & !((dh - epsE * nn) > lkth[qr]);
                const u64 m = __ballot(pass);
                if (m) {
                    int slot = KM_BKT, gpos = KM_LIST;
                    if (pass) slot = atomicAdd(&lcnt[qr], 1);
                    int base = 0;
                    const int first = (int)__builtin_ctzll(m);
                    if (lane == first) base = atomicAdd(&lcount, (int)__builtin_popcountll(m));
                    base = __shfl(base, first, 64);
                    if (pass) gpos = base + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                    if (pass && slot < KM_BKT && gpos < KM_LIST) llist[gpos] = (unsigned short)((qr << 10) | (slot << 6) | cc);
                }
            }
        }
        __syncthreads();
        const int total = lcount;
        bool overflow = force_dense | (total > KM_LIST);
        overflow = overflow | (__syncthreads_or((tid < KNN_TQ) && (lcnt[tid] > KM_BKT)) != 0);
        if (!overflow) {
            // ---------------- exact phase: canonical distance of each survivor, one pair per thread
            for (int i0 = 0; i0 < total; i0 += 256) {
                const int i = i0 + tid;
                if (i < total) {
                    const unsigned e = llist[i];
                    const unsigned qr = e >> 10, slot = (e >> 6) & 15u, cc = e & 63u;
                    const float* qp = &lq[qr * ROW];
                    const float* cp = &lc[cc * ROW];
                    float d = 0.0f;
#pragma unroll 4
                    for (int d4 = 0; d4 < 3 * CC; d4 += 4) {
                        const float4 qv = *reinterpret_cast<const float4*>(qp + d4);
                        const float4 cv = *reinterpret_cast<const float4*>(cp + d4);
                        d = accq<FMA>(d, qv.x, cv.x); d = accq<FMA>(d, qv.y, cv.y);
                        d = accq<FMA>(d, qv.z, cv.z); d = accq<FMA>(d, qv.w, cv.w);
                    }
                    lbkey[qr * KM_BKT + slot] = make_key(d, s0 + (int)cc, true);
                }
            }
            __syncthreads();
            // ---------------- insertion straight from the buckets (row r of group g <- query wave*16 + g*4 + r)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int qr = wave * 16 + g * 4 + (lane >> 4);
                const int cnt = lcnt[qr];
                const u64 k0 = (lane & 15) < cnt ? lbkey[qr * KM_BKT + (lane & 15)] : ~0ull;
                if (seeded) merge_keys<true>(k0, ~0ull, ~0ull, ~0ull, lk[g], rkey[g], K, lane);
                else merge_keys<false>(k0, ~0ull, ~0ull, ~0ull, lk[g], rkey[g], K, lane);
            }
        } else {
            // ---------------- dense fallback (first tile of an un-seeded workgroup, or adversarial data)
            float acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
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
            __syncthreads();  // everyone is done reading lc before it becomes the distance tile
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float4*>(&ldist[(ty * 4 + i) * KNN_LD + tx * 4]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            __syncthreads();
            if (seeded) select_tile<true>(ldist, lk, rkey, s0, Ns, K, wave, lane);
            else select_tile<false>(ldist, lk, rkey, s0, Ns, K, wave, lane);
        }
        refresh_kth();
        if (lane < 16) lcnt[wave * 16 + lane] = 0;  // each wave owns the buckets of its 16 queries (no cross-wave race)
        if (tid == 0) lcount = 0;
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
                    u64* partial, const int32_t* seed_idx, int seed_n, int seed_by_row, hipStream_t st) {
    LS_REQUIRE(C == KM_CC, "knn_mfma: only C == 32 layers are supported (C=%d)", C);
    const int qtiles = cdiv(Nd, KNN_TQ);
    const float epsE = 6.0f * (float)(3 * C + 4) * 5.9604645e-8f;
    dim3 grid(B * qtiles * splits), block(256);
    if (fma)
        hipLaunchKernelGGL(knn_mfma_kernel<true>, grid, block, 0, st, dst, src, dst_rows, nrm_dst, nrm_src, Nd, dst_n, Ns, C, K, idx_out,
                           dist_out, qtiles, splits, tps, partial, epsE, seed_idx, seed_n, seed_by_row);
    else
        hipLaunchKernelGGL(knn_mfma_kernel<false>, grid, block, 0, st, dst, src, dst_rows, nrm_dst, nrm_src, Nd, dst_n, Ns, C, K, idx_out,
                           dist_out, qtiles, splits, tps, partial, epsE, seed_idx, seed_n, seed_by_row);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

}  // namespace ls
