// knn.hip -- feature-space K-nearest-neighbour graph build (K <= 16), bit-exact w.r.t. oracle/ls_oracle.c.
//
// Replaces pytorch3d.ops.knn_points as called at
//   /root/reference/lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:139-141
//
// Data layout (HBM): features [B, N, 3, C] fp32 ("x-major rows").  Canonical distance
//   d(q,s) = sum_{c=0..C-1} sum_{x=0..2} (q[x][c]-s[x][c])^2   accumulated sequentially in that order
// (the reference flattens [B,C,3,N] -> [B,3C,N], so j = c*3+x), each term rounded per the FMA flag.
//
// Kernel shape (wave64, 256 threads = 4 waves per workgroup):
//   * a workgroup owns 64 queries of one instance and streams the instance's candidates in tiles of 64;
//   * both tiles are staged through LDS in channel chunks of CC (32 channels = 96 dims), rows padded to
//     100 floats so the 16-lane groups of ds_read_b128 hit 16 distinct 16-B bank slots (row stride/4 odd);
//   * each thread owns a 4x4 (query x candidate) register micro-tile: 24 ds_read_b128 feed 192 pair-dims;
//   * the 64x64 distance tile goes back to LDS and each wave merges 16 query rows into per-query sorted
//     top-K lists that live in the 16 lanes of a DPP row (4 queries per VGPR pair): every lane filters 4
//     candidates against its row's K-th entry under the lexicographic (dist, idx) order; survivors are
//     inserted one per row per step with two ballots + one DPP row_shr:1 -- four queries advance per wave
//     instruction, no LDS traffic besides the tile read, no divergence beyond a wave-uniform loop.
//   * blockIdx is remapped so that the tiles of one instance run on one XCD and share its L2.
// Roofline: VALU-bound (3 VALU ops per pair-dim without FMA, 2 with): algorithmic bytes per instance-layer
// are (Nd+Ns)*3C*4 + Nd*K*4, three orders of magnitude below the VALU time.
#include "knn_common.h"

namespace ls {

// CC = channels per LDS chunk (32, or 1 for raw xyz clouds where C == 1)
template <int CC, bool FMA>
__global__ __launch_bounds__(256, 3) void knn_kernel(const float* __restrict__ dstf, const float* __restrict__ srcf,
                                                  const int32_t* __restrict__ dst_rows, int Nd, int dst_n, int Ns, int C,
                                                  int K, int32_t* __restrict__ idx_out, float* __restrict__ dist_out,
                                                  int qtiles, int splits, int tiles_per_split, u64* __restrict__ partial,
                                                  const int32_t* __restrict__ seed_idx, int seed_n, int seed_by_row) {
    constexpr int ROW = (CC == 1) ? 4 : (3 * CC + 4);  // floats per staged row (ROW/4 odd: conflict-free ds_read_b128)
    constexpr int PR = 2 * 3 * CC + 4;                 // floats per candidate-PAIR row (PR/4 odd)
    constexpr int LC_FLOATS = (KNN_TS * ROW > KNN_TQ * KNN_LD) ? KNN_TS * ROW : KNN_TQ * KNN_LD;
    __shared__ __attribute__((aligned(16))) float lq[KNN_TQ * ROW];
    __shared__ __attribute__((aligned(16))) float lc[LC_FLOATS];   // candidate chunk; re-used as the 64x64 distance tile
    __shared__ int lqrow[KNN_TQ];
    float* ldist = lc;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int sp = logical % splits;                 // candidate range of this workgroup (split-S for under-filled grids)
    const int b = (logical / splits) / qtiles, qt = (logical / splits) % qtiles;
    const int q0 = qt * KNN_TQ;
    const int s_begin = sp * tiles_per_split * KNN_TS;
    const int s_end = min(Ns, s_begin + tiles_per_split * KNN_TS);
    const size_t row_f = (size_t)3 * C;
    const float* dbase = dstf + (size_t)b * dst_n * row_f;
    const float* sbase = srcf + (size_t)b * Ns * row_f;

    if (tid < KNN_TQ) {
        int q = q0 + tid;
        int r = -1;
        if (q < Nd) r = dst_rows ? dst_rows[(size_t)b * Nd + q] : q;
        lqrow[tid] = r;
    }
    __syncthreads();

    const int tx = tid & 15, ty = tid >> 4;  // candidates tx+16j, queries ty*4+i

    // per-wave top-K lists: group g, row r = lane>>4 -> query wave*16 + g*4 + r, entry e = lane&15 of its sorted key
    // list; rkey = the row's current K-th key replicated over the row (the admission threshold); ~0 = empty slot
    u64 lk[4], rkey[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { lk[i] = ~0ull; rkey[i] = ~0ull; }

    const int nchunks = (CC == 1) ? 1 : C / CC;
    const bool q_once = nchunks == 1;  // single chunk: the query tile is loop invariant, stage it once

    if constexpr (CC == 1) {
        for (int t = tid; t < KNN_TQ * 3; t += 256) {
            const int r = t / 3, x = t % 3;
            const int gr = lqrow[r];
            lq[r * ROW + x] = gr >= 0 ? dbase[(size_t)gr * 3 + x] : 0.0f;
        }
    }
    Stager<(CC == 1 ? 16 : CC)> sq, sc;  // register staging buffers (next chunk in flight under the current compute)
    if constexpr (CC != 1) {
        sq.load(dbase, lqrow, 0, 0, row_f, C, 0, tid);
        sc.load(sbase, nullptr, s_begin, Ns, row_f, C, 0, tid);
        if (q_once) { sq.store(lq, ROW, tid); }
    }
    // optional hints (e.g. the previous layer's neighbour list of the same point): start the lists from their exact
    // canonical distances so that the admission threshold is near-final from the first tile.  Only split 0 is seeded (the
    // merge kernel drops duplicates); the result does not depend on the hints.
    const bool seeded = (CC != 1) && seed_idx != nullptr && sp == 0;
    if constexpr (CC != 1) {
        if (seeded) {
            u64* lseed = reinterpret_cast<u64*>(lc);  // 8 KB, lc is not in use before the first tile is stored
            compute_seed_keys<FMA>(lseed, seed_idx, seed_n, seed_by_row != 0, dbase, sbase, lqrow, b, q0, Ns, C, tid);
            __syncthreads();
            seed_lists(lseed, lk, rkey, K, wave, lane);
        }
    }

    for (int s0 = s_begin; s0 < s_end; s0 += KNN_TS) {
        float acc[4][4];      // raw clouds (CC == 1)
        f32x2 acc2[4][2];     // feature layers: [query][candidate pair] x (even, odd candidate)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
#pragma unroll
            for (int j = 0; j < 2; ++j) acc2[i][j] = f32x2{0.0f, 0.0f};
        }

        if constexpr (CC == 1) {
            __syncthreads();  // previous selection finished with ldist (== lc)
            for (int t = tid; t < KNN_TS * 3; t += 256) {
                const int r = t / 3, x = t % 3;
                const int sidx = s0 + r;
                lc[r * ROW + x] = sidx < Ns ? sbase[(size_t)sidx * 3 + x] : 0.0f;
            }
            __syncthreads();
            float qv[4][3], cv[4][3];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    qv[i][x] = lq[(ty * 4 + i) * ROW + x];
                    cv[i][x] = lc[(tx + 16 * i) * ROW + x];
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int x = 0; x < 3; ++x) acc[i][j] = accq<FMA>(acc[i][j], qv[i][x], cv[j][x]);
            __syncthreads();  // everyone is done reading lc before it becomes the distance tile
        } else {
            for (int ch = 0; ch < nchunks; ++ch) {
                __syncthreads();  // previous chunk's compute / previous tile's selection done with lq, lc
                if (!q_once) sq.store(lq, ROW, tid);
                sc.store_pairs(lc, PR, tid);
                __syncthreads();
                // prefetch the next chunk (or the next tile's first chunk) into registers
                {
                    int nch = ch + 1, ns0 = s0;
                    if (nch == nchunks) { nch = 0; ns0 = s0 + KNN_TS; }
                    if (ns0 < s_end) {
                        if (!q_once) sq.load(dbase, lqrow, 0, 0, row_f, C, nch * CC, tid);
                        sc.load(sbase, nullptr, ns0, Ns, row_f, C, nch * CC, tid);
                    }
                }
#pragma unroll 1
                for (int d4 = 0; d4 < 3 * CC; d4 += 4) {
                    float4 qv[4];
                    f32x2 cp[2][4];  // [candidate pair][dim] = (c_{2pm}.d, c_{2pm+1}.d)
#pragma unroll
                    for (int i = 0; i < 4; ++i) qv[i] = *reinterpret_cast<const float4*>(&lq[(ty * 4 + i) * ROW + d4]);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const float4 lo = *reinterpret_cast<const float4*>(&lc[(tx + 16 * jj) * PR + d4 * 2]);
                        const float4 hi = *reinterpret_cast<const float4*>(&lc[(tx + 16 * jj) * PR + d4 * 2 + 4]);
                        cp[jj][0] = f32x2{lo.x, lo.y}; cp[jj][1] = f32x2{lo.z, lo.w};
                        cp[jj][2] = f32x2{hi.x, hi.y}; cp[jj][3] = f32x2{hi.z, hi.w};
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            f32x2 a = acc2[i][jj];  // canonical order j = c*3+x == LDS dim order
                            a = accq2<FMA>(a, qv[i].x, cp[jj][0]); a = accq2<FMA>(a, qv[i].y, cp[jj][1]);
                            a = accq2<FMA>(a, qv[i].z, cp[jj][2]); a = accq2<FMA>(a, qv[i].w, cp[jj][3]);
                            acc2[i][jj] = a;
                        }
                }
            }
            __syncthreads();  // everyone is done reading lc before it becomes the distance tile
        }
        // distance tile -> LDS, candidate c of a query row stored at slot (c & 15) * 4 + (c >> 4) (what a selection lane
        // reads with one ds_read_b128).  Feature layers own candidates c = 2*tx + 32*jj + h (pair-interleaved).
        if constexpr (CC == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float4*>(&ldist[(ty * 4 + i) * KNN_LD + tx * 4]) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int c0 = 2 * tx + 32 * jj;
                    ldist[(ty * 4 + i) * KNN_LD + (c0 & 15) * 4 + (c0 >> 4)] = acc2[i][jj].x;
                    ldist[(ty * 4 + i) * KNN_LD + ((c0 + 1) & 15) * 4 + ((c0 + 1) >> 4)] = acc2[i][jj].y;
                }
        }
        __syncthreads();

        if (seeded) select_tile<true>(ldist, lk, rkey, s0, Ns, K, wave, lane);
        else select_tile<false>(ldist, lk, rkey, s0, Ns, K, wave, lane);
    }
    write_lists(lk, b, q0, Nd, K, wave, lane, splits, sp, partial, idx_out, dist_out);
}

// merge the per-split sorted key lists of every query: one 16-lane row per query, same insertion step as above
__global__ __launch_bounds__(256) void knn_merge_kernel(const u64* __restrict__ partial, int total_q, int splits, int K,
                                                        int32_t* __restrict__ idx_out, float* __restrict__ dist_out) {
    const int lane = threadIdx.x & 63, e16 = lane & 15, rowbase = lane & 48;
    const int q = (blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool live = q < total_q;
    const u64* pq = partial + (size_t)(live ? q : 0) * splits * 16;
    u64 lkey = live ? pq[e16] : ~0ull;                       // split 0 is already sorted
    u64 kk = bperm64(rowbase + K - 1, lkey);
    for (int s = 1; s < splits; ++s) {
        u64 k0 = live ? pq[s * 16 + e16] : ~0ull;
        u64 m64 = __ballot(k0 < kk);
        while (m64) {
            const unsigned rb = (unsigned)(m64 >> rowbase) & 0xFFFFu;
            const bool rowhas = rb != 0;
            const int srclane = rowbase + __builtin_ctz(rb | 0x10000u);
            const u64 cand = bperm64(srclane, k0);
            k0 = (lane == srclane) ? ~0ull : k0;
            const u64 d64 = __ballot(rowhas & (lkey == cand));  // a seeded split may already hold this candidate
            const bool rowok = rowhas & (((unsigned)(d64 >> rowbase) & 0xFFFFu) == 0u);
            const u64 l64 = __ballot(rowok & (lkey < cand));
            const int pos = __builtin_popcount((unsigned)(l64 >> rowbase) & 0xFFFFu);
            const u64 up = dpp_row_shr1(lkey);
            const bool s1 = rowok & (e16 == pos), s2 = rowok & (e16 > pos);
            lkey = s1 ? cand : (s2 ? up : lkey);
            kk = bperm64(rowbase + K - 1, lkey);
            m64 = __ballot(k0 < kk);
        }
    }
    if (live && e16 < K) {
        const size_t o = (size_t)q * K + e16;
        const unsigned hi = (unsigned)(lkey >> 32), lo = (unsigned)lkey;
        idx_out[o] = hi == 0xFFFFFFFFu ? -1 : (int)lo;
        if (dist_out) dist_out[o] = hi == 0xFFFFFFFFu ? INFINITY : __uint_as_float(hi);
    }
}

// number of candidate splits for a launch: fill >= ~4 workgroups per CU when the (instance x query-tile) grid is small
static int knn_choose_splits(int B, int Nd, int Ns) {
    const int qtiles = cdiv(Nd, KNN_TQ), ctiles = cdiv(Ns, KNN_TS);
    const int blocks = B * qtiles;
    if (blocks >= 512 || ctiles < 2) return 1;  // >= 2 workgroups per CU: splitting would only add merge work (and un-seeded splits)
    // one workgroup per CU is the target: every split re-stages the 64 x 3C query tile and only split 0 is seeded, so at the
    // layer-4 shape (128 x 512, D = 192) 2 splits run in 0.14 ms where 8 splits (1024 workgroups) took 0.23 ms and 1 split 0.19 ms
    static const int target = dev_knob("LS_KNN_SPLIT_TARGET", 256);
    int sp = cdiv(target, blocks);
    if (sp > ctiles) sp = ctiles;
    if (sp > 16) sp = 16;
    return sp < 1 ? 1 : sp;
}
// scratch layout: [split partial lists (u64)] [squared norms of the src rows (B*Ns floats)] [of the dst rows (B*dst_n)]
static size_t knn_partial_bytes(int B, int Nd, int Ns) {
    const int sp = knn_choose_splits(B, Nd, Ns);
    return sp > 1 ? (size_t)B * Nd * sp * 16 * sizeof(u64) : 0;
}
// the fused MFMA-filtered kernel (knn_mfma.hip) takes the C == 32 / 64 calls whose candidates fit it, unless the caller forces the all-VALU kernel
int knn_sweep_max_ns();
static bool knn_uses_sweep(int C, bool /*seeded*/, int Ns, unsigned flags) {
    if (!(C == 32 || C == 64) || (flags & LS_FLAG_KNN_VALU_ONLY)) return false;
    return Ns >= 1 && ((Ns + 31) & ~31) <= knn_sweep_max_ns();
}
size_t knn_sweep_scratch_bytes(int B, int Nd, int dst_n, int Ns, int C);
int knn_sweep_launch(const float*, const float*, const int32_t*, int, int, int, int, int, int, bool, int32_t*, float*, const int32_t*, int,
                     int, void*, hipStream_t);
int knn_xyz_launch(const float*, const float*, const int32_t*, int, int, int, int, int, bool, int32_t*, float*, hipStream_t);
int knn_small_launch(const float*, const float*, const int32_t*, int, int, int, int, int, int, bool, int32_t*, float*, hipStream_t);
size_t knn_scratch_bytes(int B, int Nd, int dst_n, int Ns, int C, bool seeded, unsigned flags) {
    if (knn_uses_sweep(C, seeded, Ns, flags)) return knn_sweep_scratch_bytes(B, Nd, dst_n, Ns, C);
    return knn_partial_bytes(B, Nd, Ns);
}

bool knn_would_sweep(int C, int Ns, unsigned flags) { return knn_uses_sweep(C, true, Ns, flags); }

template <int CC, bool FMA>
static int launch_knn(const float* dst, const float* src, const int32_t* dst_rows, int B, int Nd, int dst_n, int Ns,
                      int C, int K, int32_t* idx_out, float* dist_out, void* scratch, const int32_t* seed_idx, int seed_n,
                      int seed_by_row, hipStream_t st) {
    const int qtiles = cdiv(Nd, KNN_TQ), ctiles = cdiv(Ns, KNN_TS);
    int splits = scratch ? knn_choose_splits(B, Nd, Ns) : 1;
    const int tps = cdiv(ctiles, splits);
    splits = cdiv(ctiles, tps);
    dim3 grid(B * qtiles * splits), block(256);
    hipLaunchKernelGGL((knn_kernel<CC, FMA>), grid, block, 0, st, dst, src, dst_rows, Nd, dst_n, Ns, C, K, idx_out,
                       dist_out, qtiles, splits, tps, (u64*)scratch, seed_idx, seed_n, seed_by_row);
    LS_LAUNCH_CHECK();
    if (splits > 1) {
        const int total_q = B * Nd;
        hipLaunchKernelGGL(knn_merge_kernel, dim3(cdiv(total_q, 16)), dim3(256), 0, st, (const u64*)scratch, total_q, splits, K,
                           idx_out, dist_out);
        LS_LAUNCH_CHECK();
    }
    return LS_OK;
}

int knn_dispatch(const float* dst, const float* src, const int32_t* dst_rows, int B, int Nd, int dst_n, int Ns, int C,
                 int K, unsigned flags, int32_t* idx_out, float* dist_out, void* scratch, const int32_t* seed_idx, int seed_n,
                 int seed_by_row, hipStream_t st) {
    LS_REQUIRE(B > 0 && Nd > 0 && Ns > 0 && dst_n > 0, "knn: empty problem (B=%d Nd=%d Ns=%d)", B, Nd, Ns);
    LS_REQUIRE(K >= 1 && K <= KNN_MAXK, "knn: K=%d unsupported (1..16)", K);
    LS_REQUIRE(C == 1 || C % 32 == 0, "knn: C=%d must be 1 or a multiple of 32", C);
    const bool fma = (flags & LS_FLAG_CONTRACT_FMA) != 0;
    // C == 32 / 64 rows, at most 1024 candidates, more than a handful of queries: f16 image + the fused kernel (knn_mfma.hip)
    if (scratch && knn_uses_sweep(C, seed_idx != nullptr, Ns, flags) && (seed_idx != nullptr || Nd > 32))
        return knn_sweep_launch(dst, src, dst_rows, B, Nd, dst_n, Ns, C, K, fma, idx_out, dist_out, seed_idx, seed_n, seed_by_row, scratch, st);
    if (C == 1) {
        // raw clouds: wave-per-query kernel (knn_xyz.hip)
        return knn_xyz_launch(dst, src, dst_rows, B, Nd, dst_n, Ns, K, fma, idx_out, dist_out, st);
    }
    // few queries per instance (encoder layers 5, 6): one pair per thread on whole rows (knn_xyz.hip) instead of 64 x 64 tiles
    if (Nd <= 32) return knn_small_launch(dst, src, dst_rows, B, Nd, dst_n, Ns, C, K, fma, idx_out, dist_out, st);
    return fma ? launch_knn<32, true>(dst, src, dst_rows, B, Nd, dst_n, Ns, C, K, idx_out, dist_out, scratch, seed_idx, seed_n, seed_by_row, st)
               : launch_knn<32, false>(dst, src, dst_rows, B, Nd, dst_n, Ns, C, K, idx_out, dist_out, scratch, seed_idx, seed_n, seed_by_row, st);
}

}  // namespace ls
