// knn_mfma.hip -- the same bit-exact 16-NN as knn.hip for the SEEDED C == 32 / 64 encoder layers (1-4), with the distance sweep
// moved onto the matrix cores as an EXACT-SAFE FILTER.
//
// Replaces pytorch3d.ops.knn_points as called at
//   /root/reference/lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:139-141        (layers whose input has 32 or 64 channels)
//
// knn.hip spends ~250 M VALU instructions per layer-1 launch on the canonical (sub, mul, add) distance of EVERY pair, although
// once a query's list holds its previous-layer neighbours only ~25 of 1024 candidates can still enter it.  Here:
//   1. S = q . s on the matrix cores, giving d^ = |q|^2 + |s|^2 - 2S.  Default: f16 operands on centred, row-scaled rows
//      (v_mfma_f32_32x32x16_f16, knn_sweep_f16_kernel, error bound in the comment above it; bf16 until round 3); fp32 operands
//      (v_mfma_f32_32x32x2_f32, knn_sweep_kernel) for Ns > 2048 or LS_KNN_SWEEP_FP32=1;
//   2. a pair is DROPPED only if  d^ - eps > kth(q)  (kth = the query's current exact K-th distance), where for the fp32 sweep
//      eps = 6 (D+4) 2^-24 (|q|^2+|s|^2) bounds |d^ - d_true| + |d_canonical - d_true| with 50 % slack
//      (gamma_{D+3} (|q|+|s|)^2 each, (|q|+|s|)^2 <= 2 (|q|^2+|s|^2)); fp32 accumulation of non-negative terms is
//      monotone, so a dropped pair provably has canonical distance > kth and could never have been inserted;
//   3. the survivors get the CANONICAL distance (same fp32 chain as knn.hip / the oracle) and go through the same
//      key merge (knn_common.h).
// The top-K lists only ever hold canonical distances, so the result is bit-identical to knn.hip / the oracle by
// construction; the filter only decides what is worth computing.
#include "knn_common.h"

namespace ls {

typedef float f32x16 __attribute__((ext_vector_type(16)));


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
// Three launches per seeded C == 32 layer (all on the caller's stream):
//   1. knn_seed_kernel    exact canonical keys of each query's hints (the previous layer's list of the same point), sorted
//                         and de-duplicated: seedkeys[B*Nd][16]; the K-th key is the query's admission threshold kth.
//   2. knn_sweep_kernel   S = q . s for ALL pairs on the matrix cores; a candidate that passes  d^ - eps <= kth  and is not a
//                         hint is appended to the query's survivor list (u16 indices in global memory, ~8 of 1024 on real
//                         features).  No canonical arithmetic, no sorting, no lists: 100 VGPRs, 21 KB LDS, four workgroups
//                         per CU, the whole layer-1 grid (1024 workgroups) resident in one round.
//   3. knn_finish_kernel  canonical keys of the survivors, merged into the seeded list (knn_common.h) -> idx / dist.
// The sweep is matrix-core bound, 1. and 3. are L2-gather-latency bound (tiny VALU load, high occupancy): as separate kernels
// they overlap with whatever the other in-flight stream runs, instead of serialising inside every workgroup (a fused version
// measured 83 k + 154 k + 50 k cycles per workgroup for the three phases).
// Exactness: thresholds are static (kth of the hints); a dropped pair has canonical distance > kth >= the final K-th
// distance; everything kept is decided by canonical keys.  A query whose hints give no finite threshold (fewer than K
// distinct valid hints) or more than KS_CAP survivors is finished by brute force over all candidates: slow, still exact.
constexpr int KS_CAP = 256;   // survivor slots per query (u16 indices in the workspace)
constexpr int KS_LD = 36;     // chunk row stride (floats): 32 dims + 4, 9 x 16 B -> conflict-free ds_read_b128

// Canonical distance of 16 (query row, candidate row) pairs per wave-instruction stream.  A lane that walks its own 384-byte
// rows makes every load touch 64 different cache lines for 16 useful bytes each (the vector L1 handles one line per cycle:
// measured 136 k cycles per workgroup for the 1024 seed pairs).  Here the four lanes of a quad share one pair: lane i loads
// channels 8i..8i+7 of the x, y and z segments (six 16-byte loads, 16 distinct lines per wave load, every line fully used)
// -- in canonical order (j = c*3 + x) exactly the run j = 24i .. 24i+23.  The fp32 chain  d = (d + t_j)  is serial by
// definition, so it runs as four stages: every lane adds its 24 terms to the running value it holds, then the quad shifts
// the value one lane up (DPP quad_perm); after stage s lane s holds the exact prefix over runs 0..s.  96 dependent adds per
// 16 pairs instead of 96 per 64 pairs -- 2x the VALU work of the lane-per-pair form, 8x fewer L1 line accesses.
// Result valid in lanes with (lane & 3) == 3.
// Rows wider than 32 channels (CC = 64) take CC/32 rounds of the same four stages; between rounds the running value rotates
// from lane 3 back to lane 0 of the quad (quad_perm [3,0,1,2]).
template <int CC>
struct QuadRow {   // one lane's share of one 32-channel round of a feature row: channels 8i..8i+7 of the x, y, z segments (i = lane & 3)
    float4 v[6];
    __device__ __forceinline__ void load(const float* __restrict__ row, int lane, int round = 0) {
        const int off = round * 32 + (lane & 3) * 8;
#pragma unroll
        for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int h = 0; h < 2; ++h) v[x * 2 + h] = *reinterpret_cast<const float4*>(row + x * CC + off + 4 * h);
    }
};
// one round: d := d + (the round's 96 terms in canonical order); LAST: no rotation after the fourth stage (result in lane 3)
template <int CC, bool FMA, bool LAST>
__device__ __forceinline__ float quad_round(float d, const QuadRow<CC>& q, const QuadRow<CC>& c) {
#pragma clang fp contract(off)
    // (Round 4: the differences and squares as packed fp32 on the natural register pairs of the 16-byte loads -- 24 instead of 48 instructions,
    //  same IEEE operations, bit-exact -- measured SLOWER: seed 31 -> 35, finish 34 -> 36, finish_select 47 -> 56 us.  Scalar it stays.)
    float t[24];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const float4 a = q.v[x * 2 + h], b = c.v[x * 2 + h];
            const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
            t[(h * 4 + 0) * 3 + x] = FMA ? d0 : d0 * d0;
            t[(h * 4 + 1) * 3 + x] = FMA ? d1 : d1 * d1;
            t[(h * 4 + 2) * 3 + x] = FMA ? d2 : d2 * d2;
            t[(h * 4 + 3) * 3 + x] = FMA ? d3 : d3 * d3;
        }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int j = 0; j < 24; ++j) d = FMA ? __builtin_fmaf(t[j], t[j], d) : d + t[j];
        if (s < 3)        // lane i <- lane i-1 within the quad (quad_perm [0,0,1,2])
            d = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x90, 0xF, 0xF, false));
        else if (!LAST)   // next round starts in lane 0 with lane 3's prefix (quad_perm [3,0,1,2])
            d = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x93, 0xF, 0xF, false));
    }
    return d;
}
// canonical distance of the quad's (query row, candidate row) pair; q0 = the query's round-0 share (held by the caller)
template <int CC, bool FMA>
__device__ __forceinline__ float quad_pair_distance(const QuadRow<CC>& q0, const float* __restrict__ qrow, const float* __restrict__ crow,
                                                    int lane) {
    QuadRow<CC> c;
    c.load(crow, lane, 0);
    if constexpr (CC == 32) {
        return quad_round<CC, FMA, true>(0.0f, q0, c);
    } else {
        float d = quad_round<CC, FMA, false>(0.0f, q0, c);
#pragma unroll
        for (int r = 1; r < CC / 32; ++r) {
            QuadRow<CC> q;
            q.load(qrow, lane, r);
            c.load(crow, lane, r);
            d = (r == CC / 32 - 1) ? quad_round<CC, FMA, true>(d, q, c) : quad_round<CC, FMA, false>(d, q, c);
        }
        return d;
    }
}

int row_norms_launch(const float* f, int row_f, long long npts, float* norms, hipStream_t st);

// ---- 1. seeds.  One wave per four queries (row r of the wave = query 4*wave_id + r), four quad-steps of four hints.
// AUTO (round 3): the hints are not read from seed_idx but SELECTED here from the class winners of knn_sweep_winners_kernel (win_val /
// win_idx [query][W], W = 32 or 64): the 16 lanes of a query row take W / 16 winners each as (descending value, index) keys and
// merge_keys keeps the row's 16 best -- the same 16, in the same order, as the separate knn_autohint_select_kernel launch produced.
template <int CC, bool FMA, bool AUTO = false>
__global__ __launch_bounds__(256, CC == 32 ? 6 : 4) void knn_seed_kernel(const float* __restrict__ dstf, const float* __restrict__ srcf,
                                                          const int32_t* __restrict__ dst_rows, int Nd, int dst_n, int Ns, int K,
                                                          const int32_t* __restrict__ seed_idx, int seed_n, int seed_by_row,
                                                          u64* __restrict__ seedkeys, int groups_per_inst, int total_groups,
                                                          const float* __restrict__ win_val = nullptr, const int32_t* __restrict__ win_idx = nullptr,
                                                          int W = 0) {
    constexpr int RF = 3 * CC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware order: consecutive logical workgroups (= the queries of one instance) share an XCD, so the instance's
    // 393 KB candidate table is gathered out of that XCD's L2 instead of the fabric (round-robin placement: every XCD
    // streams all 64 tables, measured 194 MB of L2 fills per launch for 25 MB of tables)
    const int wg = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;   // global wave id
    if (wg >= total_groups) return;                           // (wave-uniform; no workgroup barrier in this kernel)
    const int b = wg / groups_per_inst, q = (wg % groups_per_inst) * 4 + (lane >> 4);
    const float* dbase = dstf + (size_t)b * dst_n * RF;
    const float* sbase = srcf + (size_t)b * Ns * RF;
    const bool live = q < Nd;
    const int r = live ? (dst_rows ? dst_rows[(size_t)b * Nd + q] : q) : -1;
    const int quad = (lane >> 2) & 3;
    const bool qlast = (lane & 3) == 3;
    int sidx[4];
    if constexpr (AUTO) {
        // the wave's four queries one after the other: W <= 64 winners, one per lane, through the 64-lane sorting network (the
        // insertion-based merge_keys took ~40 ballot rounds for this: +17 us on the layer-3 seed launch)
        int my_hint = -1;                                                // hint number (lane & 15) of this lane's row
        const int q0 = (wg % groups_per_inst) * 4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const size_t qg = (size_t)b * Nd + min(q0 + g, Nd - 1);
            u64 k = ~0ull;
            if (lane < W) {
                const int idx = win_idx[qg * W + lane];
                if (idx >= 0) {
                    unsigned u = __float_as_uint(win_val[qg * W + lane]);
                    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;       // monotone in the float value
                    k = ((u64)(~u) << 32) | (unsigned)idx;             // ascending key order = descending value
                }
            }
            LS_SORT64(cx64, k, lane)
            const int hsel = (k == ~0ull) ? -1 : (int)(unsigned)k;
            const int got = __shfl(hsel, lane & 15, 64);
            my_hint = ((lane >> 4) == g) ? got : my_hint;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int si = __shfl(my_hint, (lane & 48) + u * 4 + quad, 64);
            sidx[u] = (r >= 0 && si >= 0 && si < Ns) ? si : -1;
        }
    } else {
        const int32_t* hp = seed_idx + ((size_t)b * seed_n + (seed_by_row ? max(r, 0) : min(q, seed_n - 1))) * 16;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int si = hp[u * 4 + quad];
            sidx[u] = (r >= 0 && si >= 0 && si < Ns) ? si : -1;
        }
    }
    const float* qrow = dbase + (size_t)max(r, 0) * RF;
    QuadRow<CC> qv;
    qv.load(qrow, lane);
    u64 ks[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        ks[u] = make_key(quad_pair_distance<CC, FMA>(qv, qrow, sbase + (size_t)max(sidx[u], 0) * RF, lane), sidx[u], (sidx[u] >= 0) & qlast);
    // The row's 16 keys -- key (step u, quad g) sits in lane 4 g + 3 -- as ONE key per lane (lane 4 g + u takes step u's), sorted by the 16-lane
    // network; repeated hints (equal keys: same index, same canonical distance) are adjacent afterwards, all but the first become "none" and the
    // row is sorted once more (rare: wave-uniform branch).  Until round 4 the list was built by merge_keys' insertion loop: one ballot round
    // of ~40 instructions per key that entered, ~7 rounds per wave = a quarter of this kernel's instructions; the result is the same list.
    u64 lk = ~0ull;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const u64 bq = dpp_quad_bcast3(ks[u]);
        lk = (lane & 3) == u ? bq : lk;
    }
    LS_SORT16(cx64, lk, lane)
    {
        const u64 prev = dpp_row_shr1(lk);
        const bool dup = (lane & 15) > 0 && prev == lk && lk != ~0ull;
        if (__any(dup)) {
            lk = dup ? ~0ull : lk;
            LS_SORT16(cx64, lk, lane)
        }
    }
    if (live) seedkeys[((size_t)b * Nd + q) * 16 + (lane & 15)] = lk;
}

// ---- 2. sweep.  64 queries x all candidates per workgroup; wave (wm, wn) owns the 32x32 block of S for queries wm*32..,
// candidates wn*32.. of each 64-candidate tile; the query fragments stay in registers, candidates stream through LDS in
// 64-row x 32-dim chunks (double buffered, one barrier per chunk, three chunks = one tile).
template <int CC>
__global__ __launch_bounds__(256, CC == 32 ? 4 : 2) void knn_sweep_kernel(const float* __restrict__ dstf, const float* __restrict__ srcf,
                                                           const int32_t* __restrict__ dst_rows, const float* __restrict__ nrm_dst,
                                                           const float* __restrict__ nrm_src, int Nd, int dst_n, int Ns, int K,
                                                           int qtiles, float epsE, const u64* __restrict__ seedkeys,
                                                           int32_t* __restrict__ surv_cnt, unsigned short* __restrict__ surv) {
    constexpr int RF = 3 * CC, NCH = RF / 32;   // a tile = NCH chunks of 32 consecutive floats of every candidate row
    __shared__ __attribute__((aligned(16))) float lc[2][KNN_TS * KS_LD];              // 18 KB: two candidate chunks
    __shared__ __attribute__((aligned(16))) unsigned short lseedidx[KNN_TQ * 16];     // 2 KB: the hints (0xFFFF = none)
    __shared__ float lnq[KNN_TQ], lkth[KNN_TQ];
    __shared__ int lqrow[KNN_TQ], lcnt[KNN_TQ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int b = logical / qtiles, qt = logical % qtiles;
    const int q0 = qt * KNN_TQ;
    const float* dbase = dstf + (size_t)b * dst_n * RF;
    const float* sbase = srcf + (size_t)b * Ns * RF;
    const float* nsb = nrm_src + (size_t)b * Ns;

    if (tid < KNN_TQ) {
        const int q = q0 + tid;
        int r = -1;
        if (q < Nd) r = dst_rows ? dst_rows[(size_t)b * Nd + q] : q;
        lqrow[tid] = r;
        lnq[tid] = r >= 0 ? nrm_dst[(size_t)b * dst_n + r] : 0.f;
        float kth = -INFINITY;                                  // padding queries never pass the filter
        if (r >= 0) {
            const unsigned hi = (unsigned)(seedkeys[((size_t)b * Nd + q) * 16 + (K - 1)] >> 32);
            kth = hi == 0xFFFFFFFFu ? INFINITY : __uint_as_float(hi);   // fewer than K distinct hints: everything passes
        }
        lkth[tid] = kth;
        lcnt[tid] = 0;
    }
    for (int i = tid; i < KNN_TQ * 16; i += 256) {
        const int q = q0 + (i >> 4);
        unsigned short h = 0xFFFFu;
        if (q < Nd) {
            const u64 k = seedkeys[((size_t)b * Nd + q) * 16 + (i & 15)];
            if ((unsigned)(k >> 32) != 0xFFFFFFFFu) h = (unsigned short)(unsigned)k;
        }
        lseedidx[i] = h;
    }
    __syncthreads();

    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // loop-invariant A fragments: query row wm*32 + l31, dims k*32 + j*8 + lh*4 .. +3 (padding queries: row 0, kth = -inf)
    float4 a[NCH * 4];
    {
        const int r = lqrow[wm * 32 + l31];
        const float* qp = dbase + (size_t)(r >= 0 ? r : 0) * RF + lh * 4;
#pragma unroll
        for (int i = 0; i < NCH * 4; ++i) a[i] = *reinterpret_cast<const float4*>(qp + i * 8);
    }

    // chunk staging: chunk g = (tile g/NCH, 32-float slice g%NCH of the row); thread -> rows sr and sr+32, float4 column sc4
    const int sr = tid >> 3, sc4 = (tid & 7) * 4;
    const int ntiles = (Ns + KNN_TS - 1) / KNN_TS;
    float4 st0, st1;
    auto gload = [&](int t, int k) {
        const int r0 = t * KNN_TS + sr, r1 = r0 + 32;
        // rows past Ns are clamped, not zeroed: their S column is garbage but `cvalid` keeps it out of the filter, and a
        // select on the loaded value would make the wave wait for the load right here instead of one chunk later
        st0 = *reinterpret_cast<const float4*>(sbase + (size_t)min(r0, Ns - 1) * RF + k * 32 + sc4);
        st1 = *reinterpret_cast<const float4*>(sbase + (size_t)min(r1, Ns - 1) * RF + k * 32 + sc4);
    };
    auto lstore = [&](int buf) {
        *reinterpret_cast<float4*>(&lc[buf][sr * KS_LD + sc4]) = st0;
        *reinterpret_cast<float4*>(&lc[buf][(sr + 32) * KS_LD + sc4]) = st1;
    };

    gload(0, 0);
    lstore(0);
    gload(0, 1);
    __syncthreads();

    const int cc = wn * 32 + l31;
    unsigned short* sv = surv + ((size_t)b * Nd + q0) * KS_CAP;
#pragma unroll 1
    for (int t = 0; t < ntiles; ++t) {
        // candidate norm of this lane's column, consumed two chunks later (clamped: columns past Ns are masked by cvalid).
        // Loaded and used inside the same iteration: a load result carried over the back edge makes the compiler drain
        // every load in flight (s_waitcnt vmcnt(0)) at the loop head, including the chunk prefetch.
        float nsv = nsb[min(t * KNN_TS + cc, Ns - 1)];
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.0f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int buf = (t * NCH + k) & 1;
            const float* bp = &lc[buf][(wn * 32 + l31) * KS_LD + lh * 4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 bb = *reinterpret_cast<const float4*>(bp + j * 8);
                const float4 av = a[k * 4 + j];
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bb.x, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bb.y, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bb.z, S, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bb.w, S, 0, 0, 0);
            }
            if (k == NCH - 1) {
                // filter epilogue: candidate t*64 + cc against the 16 queries of this lane's accumulator rows
                const int cglob = t * KNN_TS + cc;
                const bool cvalid = cglob < Ns;
                asm volatile("" : "+v"(nsv));   // keep the norm's first use (and its s_waitcnt) down here, two chunks after the load
                unsigned mask = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float nn = lnq[qr] + nsv;
                    const float dh = nn - 2.0f * S[r];
                    const bool pass = cvalid & !((dh - epsE * nn) > lkth[qr]);
                    mask |= pass ? (1u << r) : 0u;
                }
                while (mask) {
                    const int r = __builtin_ctz(mask);
                    mask &= mask - 1;
                    const int qr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    // a hint's exact key is in the seeded list already: do not record it again
                    const uint4 h0 = *reinterpret_cast<const uint4*>(&lseedidx[qr * 16]);
                    const uint4 h1 = *reinterpret_cast<const uint4*>(&lseedidx[qr * 16 + 8]);
                    const unsigned cg = (unsigned)cglob;
                    auto has = [&](unsigned w) { return ((w & 0xFFFFu) == cg) | ((w >> 16) == cg); };
                    const bool hinted = has(h0.x) | has(h0.y) | has(h0.z) | has(h0.w) | has(h1.x) | has(h1.y) | has(h1.z) | has(h1.w);
                    if (!hinted) {
                        const int pos = atomicAdd(&lcnt[qr], 1);
                        if (pos < KS_CAP) sv[(size_t)qr * KS_CAP + pos] = (unsigned short)cglob;
                    }
                }
            }
            // unconditional (clamped) stores / loads: with branches around them the compiler cannot count the loads in
            // flight and falls back to s_waitcnt vmcnt(0) in front of the MFMA block
            lstore(buf ^ 1);
            {
                const int t2 = k + 2 < NCH ? t : t + 1;     // chunk g + 2
                gload(min(t2, ntiles - 1), (k + 2) % NCH);
            }
            __syncthreads();
        }
    }
    if (tid < KNN_TQ && q0 + tid < Nd) surv_cnt[(size_t)b * Nd + q0 + tid] = lcnt[tid];   // > KS_CAP: finish by brute force
}

// ---- 3. finish.  Same wave layout as the seed kernel; the seeded list is reloaded, survivors get canonical keys in rounds of
// 16 per query (four quad-steps), a query flagged as overflowed scans every candidate the same way.
// (Round 4, built and not kept: the four rows' survivors POOLED into one list of (row, survivor) pairs for the wave's 16 quads -- ceil(sum / 16)
//  steps instead of max_r ceil(cnt_r / 4): 13.2 M -> 10.8 M VALU wave-instructions per launch, but every pair then gathers its query share too
//  and the keys pass through an LDS table before they are merged: 36.5 -> 38.5 us alone (44 us with two steps unrolled) -- the kernel waits on
//  its row gathers, not on issue slots.)
template <int CC, bool FMA>
__global__ __launch_bounds__(256, 4) void knn_finish_kernel(const float* __restrict__ dstf, const float* __restrict__ srcf,
                                                            const int32_t* __restrict__ dst_rows, int Nd, int dst_n, int Ns, int K,
                                                            const u64* __restrict__ seedkeys, const int32_t* __restrict__ surv_cnt,
                                                            const unsigned short* __restrict__ surv, int32_t* __restrict__ idx_out,
                                                            float* __restrict__ dist_out, int groups_per_inst, int total_groups) {
    constexpr int RF = 3 * CC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;   // XCD-aware order, as in the seed kernel
    if (wg >= total_groups) return;
    const int b = wg / groups_per_inst, q = (wg % groups_per_inst) * 4 + (lane >> 4);
    const float* dbase = dstf + (size_t)b * dst_n * RF;
    const float* sbase = srcf + (size_t)b * Ns * RF;
    const bool live = q < Nd;
    const int r = live ? (dst_rows ? dst_rows[(size_t)b * Nd + q] : q) : -1;
    const int quad = (lane >> 2) & 3, e16 = lane & 15;
    const bool qlast = (lane & 3) == 3;
    const size_t qg = (size_t)b * Nd + (live ? q : 0);
    // the seeded list, truncated to K entries (entry e in lane e of the row)
    u64 lk = (live && e16 < K) ? seedkeys[qg * 16 + e16] : ~0ull;
    u64 rkey = bperm64((lane & 48) + K - 1, lk);
    int cnt = live ? surv_cnt[qg] : 0;
    const bool brute = cnt > KS_CAP;
    if (brute) cnt = Ns;
    const unsigned short* sp = surv + qg * KS_CAP;
    const float* qrow = dbase + (size_t)max(r, 0) * RF;
    QuadRow<CC> qv;
    qv.load(qrow, lane);
    for (int base = 0; __any(base < cnt); base += 16) {
        u64 ks[4] = {~0ull, ~0ull, ~0ull, ~0ull};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = base + u * 4 + quad;
            const bool v = j < cnt;
            if (__any(v)) {   // wave-uniform: skip the steps no row of the wave needs
                const int c = v ? (brute ? j : (int)sp[j]) : 0;
                ks[u] = make_key(quad_pair_distance<CC, FMA>(qv, qrow, sbase + (size_t)c * RF, lane), c, v & qlast);
            }
        }
        key_cx(ks[0], ks[1]); key_cx(ks[2], ks[3]); key_cx(ks[0], ks[2]); key_cx(ks[1], ks[3]); key_cx(ks[1], ks[2]);
        merge_keys<true>(ks[0], ks[1], ks[2], ks[3], lk, rkey, K, lane);
    }
    if (live && e16 < K) {
        const size_t o = qg * K + e16;
        const unsigned hi = (unsigned)(lk >> 32), lo = (unsigned)lk;
        idx_out[o] = hi == 0xFFFFFFFFu ? -1 : (int)lo;
        if (dist_out) dist_out[o] = hi == 0xFFFFFFFFu ? INFINITY : __uint_as_float(hi);
    }
}

// ---- 3'. finish, one WAVE per query (rows wider than 32 channels).  The row-per-query form above serialises a query's
// survivors four at a time and a wave waits for its slowest row (layer 4, D = 192, ~38 survivors: 104 us for 8 k queries).
// Here the 16 quads of a wave take 16 survivors per step, three steps are independent instruction streams, and the 48 new keys
// are sorted together with the seeded list by one 64-lane network (knn_common.h).
template <int CC, bool FMA>
__global__ __launch_bounds__(256, 3) void knn_finish_wave_kernel(const float* __restrict__ dstf, const float* __restrict__ srcf,
                                                                 const int32_t* __restrict__ dst_rows, int Nd, int dst_n, int Ns, int K,
                                                                 const u64* __restrict__ seedkeys, const int32_t* __restrict__ surv_cnt,
                                                                 const unsigned short* __restrict__ surv, int32_t* __restrict__ idx_out,
                                                                 float* __restrict__ dist_out, int total_q) {
    constexpr int RF = 3 * CC;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qg = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;   // consecutive queries (one instance) share an XCD
    if (qg >= total_q) return;
    const int b = qg / Nd, q = qg % Nd;
    const float* sbase = srcf + (size_t)b * Ns * RF;
    const int r = dst_rows ? dst_rows[qg] : q;
    const float* qrow = dstf + ((size_t)b * dst_n + r) * RF;
    const int quad = lane >> 2;
    const bool qlast = (lane & 3) == 3;
    int cnt = surv_cnt[qg];
    const bool brute = cnt > KS_CAP;    // overflowed survivor list: scan every candidate, starting from an empty list
    if (brute) cnt = Ns;
    u64 best = (!brute && lane < K) ? seedkeys[(size_t)qg * 16 + lane] : ~0ull;   // lanes 0..15: the list so far
    const unsigned short* sp = surv + (size_t)qg * KS_CAP;
    QuadRow<CC> qv;
    qv.load(qrow, lane);
    for (int base = 0; base < cnt; base += 48) {
        u64 ks[3] = {~0ull, ~0ull, ~0ull};
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (base + u * 16 < cnt) {   // wave-uniform
                const int j = base + u * 16 + quad;
                const bool v = j < cnt;
                const int c = v ? (brute ? j : (int)sp[j]) : 0;
                ks[u] = make_key(quad_pair_distance<CC, FMA>(qv, qrow, sbase + (size_t)c * RF, lane), c, v & qlast);
            }
        }
        // lane 16 + 16 u + g <- the key of quad g in step u (held by lane 4 g + 3)
        const int src = ((lane & 15) << 2) + 3;
        const u64 n0 = bperm64(src, ks[0]), n1 = bperm64(src, ks[1]), n2 = bperm64(src, ks[2]);
        u64 k = lane < 16 ? best : (lane < 32 ? n0 : (lane < 48 ? n1 : n2));
        LS_SORT64(cx64, k, lane)
        best = k;
    }
    if (lane < K) {
        const size_t o = (size_t)qg * K + lane;
        const unsigned hi = (unsigned)(best >> 32), lo = (unsigned)best;
        idx_out[o] = hi == 0xFFFFFFFFu ? -1 : (int)lo;
        if (dist_out) dist_out[o] = hi == 0xFFFFFFFFu ? INFINITY : __uint_as_float(hi);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 2'. f16 sweep.  The filter only has to be SAFE, not accurate, so S = q . s does not need fp32 operands: with the rows
// centred on the instance centre (distances are translation invariant), scaled by an exact power of two PER ROW (largest element
// -> [2^14, 2^15): the f16 window follows the row as in gemm.hip; the inverse scales multiply S back) and rounded to f16 (unit
// roundoff u = 2^-11),
//     |S~ - S| <= (2u + u^2) |q'||s'| <= (2^-11 + 2^-23) (|q'|^2 + |s'|^2)      (fp32 accumulation adds D 2^-23 of the same scale)
// so  d^ = |q'|^2 + |s'|^2 - 2 S~  is within  eps_b (|q'|^2 + |s'|^2),  eps_b = 1.02 * 2^-10 + 6 (D+4) 2^-24 + 2^-20,  of the canonical
// distance (second term: fp32 norms, the canonical chain's own rounding, the centring subtraction; third: elements more than 24
// binades below their row's maximum go subnormal).  A pair is dropped only if  d^ - eps_b (..) > kth.  Rounds 1-2 used bf16
// (u = 2^-8, no row scale needed): the margin was 8x wider -- on the encoder's features kth / (|q'|^2+|s'|^2) is 0.04 .. 0.3, so bf16
// admitted 5 .. 20 % more survivors than an fp32 sweep, f16 admits < 3 % more -- at the same matrix-core rate
// (v_mfma_f32_32x32x16_f16).  Without centring a common offset would drown the distances in eps_b.
//
// Layout: knn_prep_f16_kernel writes the centred rows "fragment-major": [instance][32-row tile][k-step of 16 dims][lane
// (k-half h, row j)][8 f16], i.e. exactly the 1 KB a wave's B operand load wants -> one fully coalesced 16-byte-per-lane load
// per MFMA, no LDS staging, no workgroup barriers; the waves are independent (32 queries x a candidate range each) and any
// (queries x splits) grid fills the chip.  Per-query hints live in a wave-private LDS bitmap (a pass that is a hint is not
// recorded), survivors are appended through global counters.
typedef _Float16 f16x8k __attribute__((ext_vector_type(8)));
typedef _Float16 kh2_t __attribute__((ext_vector_type(2)));
typedef float kf2_t __attribute__((ext_vector_type(2)));
constexpr int KB_MAXNS = 2048;   // bitmap words per query = Ns / 32 <= 64
constexpr int KB_CAPW = 64;      // survivor slots per (wave, query) in LDS


// centre of an instance: the mean of its first min(N, 64) rows (any centre is valid -- distances are translation invariant
// and the error bound is relative to the centred norms; the encoder's rows are in FPS order, so a prefix is a spread-out
// sample).  grid (B, ceil(D/64)), 1024 threads = 64 dims x 16 row groups
__global__ __launch_bounds__(1024) void knn_mean_rows_kernel(const float* __restrict__ f, int N, int D, float* __restrict__ mu) {
    __shared__ float part[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int b = blockIdx.x, d = blockIdx.y * 64 + tx;
    const int n = min(N, 64);
    float s = 0.f;
    if (d < D)
        for (int r = ty; r < n; r += 16) s += f[((size_t)b * N + r) * D + d];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && d < D) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i][tx];
        mu[(size_t)b * D + d] = t / (float)n;
    }
}

// one wave per 32-row tile: lane (h, j) walks row j in 16-dim steps (dims kk*16 + h*8 .. +7: two float4 in, one 16-byte store
// into the step's 1 KB fragment block), accumulating the centred row's squared norm on the way
// The centre of the instance is computed HERE (round 3: the separate knn_mean_rows_kernel launch is gone from the encoder's path): every
// wave sums the first min(Nc, 16) rows of the centre source `fc` (the candidate set: queries and candidates of a call share one centre)
// into its own LDS slot -- lane l takes dims l, l + 64, l + 128, coalesced row reads out of L2.  Any centre is valid (header comment of
// knn_mean_rows_kernel); 16 FPS-ordered rows instead of 64 widen the filter's margin by ~5 %.
constexpr int KNN_CENTRE_ROWS = 16;
__global__ __launch_bounds__(256) void knn_prep_f16_kernel(const float* __restrict__ f, const float* __restrict__ fc, int Nc, int N, int Npad, int D,
                                                            int tiles_total, unsigned short* __restrict__ out, float* __restrict__ norms,
                                                            float* __restrict__ iscale, int32_t* __restrict__ zero_buf, long long zero_n) {
    __shared__ __attribute__((aligned(16))) float lmu[4][192];
    // the sweep's per-query survivor counters are cleared here (a separate memset launch cost 5 - 7 us per layer)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < zero_n; i += (long long)gridDim.x * 256) zero_buf[i] = 0;
    const int tg = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tg >= tiles_total) return;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int tpi = Npad >> 5, b = tg / tpi, tile = tg % tpi, r = tile * 32 + j;
    const int KK = D >> 4;
    const bool live = r < N;
    {
        const int nc = min(Nc, KNN_CENTRE_ROWS);
        const float* cp = fc + (size_t)b * Nc * D;
        float* mw = lmu[threadIdx.x >> 6];
        for (int d = lane; d < D; d += 64) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four independent chains: the 16 loads of a dim are in flight together
            for (int rr = 0; rr + 3 < nc; rr += 4) {
                s0 += cp[(size_t)rr * D + d]; s1 += cp[(size_t)(rr + 1) * D + d]; s2 += cp[(size_t)(rr + 2) * D + d]; s3 += cp[(size_t)(rr + 3) * D + d];
            }
            for (int rr = nc & ~3; rr < nc; ++rr) s0 += cp[(size_t)rr * D + d];
            mw[d] = ((s0 + s1) + (s2 + s3)) / (float)nc;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the slot is wave-private, LDS ops of a wave complete in order
        __builtin_amdgcn_wave_barrier();
    }
    const float* rp = f + ((size_t)b * N + (live ? r : 0)) * D + h * 8;
    const float* mp = lmu[threadIdx.x >> 6] + h * 8;
    unsigned short* op = out + (((size_t)b * tpi + tile) * KK * 64 + lane) * 8;
    // pass 1: the centred row's largest magnitude -> its exact power-of-two scale (largest element -> [2^14, 2^15): the f16 window follows
    // every row, as in gemm.hip; elements below 2^-24 of the row maximum go subnormal: absolute error 2^-39 of it, inside the bound's slack)
    float amax = 0.f;
#pragma unroll 3
    for (int kk = 0; kk < KK; ++kk) {
        const float4 x0 = *reinterpret_cast<const float4*>(rp + kk * 16), x1 = *reinterpret_cast<const float4*>(rp + kk * 16 + 4);
        const float4 m0 = *reinterpret_cast<const float4*>(mp + kk * 16), m1 = *reinterpret_cast<const float4*>(mp + kk * 16 + 4);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(x0.x - m0.x), fabsf(x0.y - m0.y))), fmaxf(fabsf(x0.z - m0.z), fabsf(x0.w - m0.w)));
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(x1.x - m1.x), fabsf(x1.y - m1.y))), fmaxf(fabsf(x1.z - m1.z), fabsf(x1.w - m1.w)));
    }
    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
    unsigned be = (__float_as_uint(amax) >> 23) & 0xffu;
    be = be < 15u ? 15u : be;
    const float sc = live ? __uint_as_float((268u - be) << 23) : 0.f;      // (Inf / NaN rows: the image row is non-finite, every pair with it survives)
    float s = 0.f;
#pragma unroll 3
    for (int kk = 0; kk < KK; ++kk) {
        const float4 x0 = *reinterpret_cast<const float4*>(rp + kk * 16), x1 = *reinterpret_cast<const float4*>(rp + kk * 16 + 4);
        const float4 m0 = *reinterpret_cast<const float4*>(mp + kk * 16), m1 = *reinterpret_cast<const float4*>(mp + kk * 16 + 4);
        float c[8] = {x0.x - m0.x, x0.y - m0.y, x0.z - m0.z, x0.w - m0.w, x1.x - m1.x, x1.y - m1.y, x1.z - m1.z, x1.w - m1.w};
        uint4 w;
        unsigned pk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float c0 = live ? c[2 * i] : 0.f, c1 = live ? c[2 * i + 1] : 0.f;
            s += c0 * c0 + c1 * c1;
            const kh2_t hv = __builtin_convertvector(kf2_t{c0 * sc, c1 * sc}, kh2_t);     // round to nearest even
            pk[i] = __builtin_bit_cast(unsigned, hv);
        }
        w.x = pk[0]; w.y = pk[1]; w.z = pk[2]; w.w = pk[3];
        *reinterpret_cast<uint4*>(op + (size_t)kk * 512) = w;
    }
    s += __shfl_xor(s, 32, 64);
    if (h == 0 && live) { norms[(size_t)b * N + r] = s; iscale[(size_t)b * N + r] = __uint_as_float((be - 14u) << 23); }
}

// Round 4: the same image, one WORKGROUP per 32-row tile.  The one-wave form walks a row twice (maximum, then scale + convert) in KK dependent
// steps of two loads each and leaves 2 waves per SIMD on the chip at the encoder's shapes (2 048 tiles): 17.5 us per layer for 25 MB in and
// 12.5 MB out, i.e. latency, not bandwidth.  Here the WPT waves of a workgroup take KK / WPT k-steps each, every load of a wave is issued
// up front and the row stays in registers; the row maximum and the squared norm are combined through LDS (two barriers).  Same centre (the
// mean of the first min(Nc, 16) rows in the same summation order), same scale, same f16 values; the norm is summed in a different order
// (per wave, then over the waves ascending), which the filter's margin covers like any other fp32 rounding of the norms.
template <int KK, int WPT>
__global__ __launch_bounds__(64 * WPT) void knn_prep_f16_tile_kernel(const float* __restrict__ f, const float* __restrict__ fc, int Nc, int N, int Npad,
                                                                    unsigned short* __restrict__ out, float* __restrict__ norms, float* __restrict__ iscale,
                                                                    int32_t* __restrict__ zero_buf, long long zero_n) {
    constexpr int D = KK * 16, KPW = KK / WPT, DW = KPW * 16;     // dims of one wave
    static_assert(KK % WPT == 0 && DW <= 64, "a wave's dims fit one lane each for the centre");
    __shared__ __attribute__((aligned(16))) float lmu[WPT][64];
    __shared__ float lamax[WPT][32], lsum[WPT][32];
    for (long long i = (long long)blockIdx.x * (64 * WPT) + threadIdx.x; i < zero_n; i += (long long)gridDim.x * (64 * WPT)) zero_buf[i] = 0;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), j = lane & 31, h = lane >> 5;
    const int tpi = Npad >> 5, b = blockIdx.x / tpi, tile = blockIdx.x % tpi, r = tile * 32 + j;
    const bool live = r < N;
    // the row's share of this wave: dims w * DW + kk * 16 + h * 8 .. + 7, all loads in flight before the centre arrives
    const float* rp = f + ((size_t)b * N + (live ? r : 0)) * D + w * DW + h * 8;
    float4 x[KPW][2];
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) { x[kk][0] = *reinterpret_cast<const float4*>(rp + kk * 16); x[kk][1] = *reinterpret_cast<const float4*>(rp + kk * 16 + 4); }
    {   // centre of this wave's dims (lane l: dim w * DW + l), summation order of knn_prep_f16_kernel
        const int nc = min(Nc, KNN_CENTRE_ROWS);
        const float* cp = fc + (size_t)b * Nc * D + w * DW + lane;
        if (lane < DW) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (int rr = 0; rr + 3 < nc; rr += 4) {
                s0 += cp[(size_t)rr * D]; s1 += cp[(size_t)(rr + 1) * D]; s2 += cp[(size_t)(rr + 2) * D]; s3 += cp[(size_t)(rr + 3) * D];
            }
            for (int rr = nc & ~3; rr < nc; ++rr) s0 += cp[(size_t)rr * D];
            lmu[w][lane] = ((s0 + s1) + (s2 + s3)) / (float)nc;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the slot is wave-private
        __builtin_amdgcn_wave_barrier();
    }
    float c[KPW][8];
    float amax = 0.f;
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) {
        const float4 m0 = *reinterpret_cast<const float4*>(&lmu[w][kk * 16 + h * 8]), m1 = *reinterpret_cast<const float4*>(&lmu[w][kk * 16 + h * 8 + 4]);
        const float4 x0 = x[kk][0], x1 = x[kk][1];
        const float t[8] = {x0.x - m0.x, x0.y - m0.y, x0.z - m0.z, x0.w - m0.w, x1.x - m1.x, x1.y - m1.y, x1.z - m1.z, x1.w - m1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) { c[kk][i] = live ? t[i] : 0.f; amax = fmaxf(amax, fabsf(t[i])); }
    }
    amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
    if (h == 0) lamax[w][j] = amax;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < WPT; ++u) amax = fmaxf(amax, lamax[u][j]);
    unsigned be = (__float_as_uint(amax) >> 23) & 0xffu;
    be = be < 15u ? 15u : be;
    const float sc = live ? __uint_as_float((268u - be) << 23) : 0.f;
    unsigned short* op = out + ((((size_t)b * tpi + tile) * KK + w * KPW) * 64 + lane) * 8;
    float s = 0.f;
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) {
        unsigned pk[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float c0 = c[kk][2 * i], c1 = c[kk][2 * i + 1];
            s += c0 * c0 + c1 * c1;
            const kh2_t hv = __builtin_convertvector(kf2_t{c0 * sc, c1 * sc}, kh2_t);     // round to nearest even
            pk[i] = __builtin_bit_cast(unsigned, hv);
        }
        uint4 wv;
        wv.x = pk[0]; wv.y = pk[1]; wv.z = pk[2]; wv.w = pk[3];
        *reinterpret_cast<uint4*>(op + (size_t)kk * 512) = wv;
    }
    s += __shfl_xor(s, 32, 64);
    if (h == 0) lsum[w][j] = s;
    __syncthreads();
    if (w == 0 && h == 0 && live) {
        float t = lsum[0][j];
#pragma unroll
        for (int u = 1; u < WPT; ++u) t += lsum[u][j];
        norms[(size_t)b * N + r] = t;
        iscale[(size_t)b * N + r] = __uint_as_float((be - 14u) << 23);
    }
}
// the image of `f` ([B, N, D] rows, centre from fc): tile kernel for the encoder's row widths, the one-wave kernel otherwise / for A/B (LS_KNN_PREP_WAVE=1)
static int knn_prep_launch(const float* f, const float* fc, int Nc, int B, int N, int Npad, int D, unsigned short* out, float* norms, float* iscale,
                           int32_t* zero_buf, long long zero_n, hipStream_t st) {
    static const bool wave_form = getenv("LS_KNN_PREP_WAVE") && atoi(getenv("LS_KNN_PREP_WAVE")) != 0;
    const int tiles = B * (Npad / 32);
    if (D == 96 && !wave_form)
        hipLaunchKernelGGL((knn_prep_f16_tile_kernel<6, 3>), dim3(tiles), dim3(192), 0, st, f, fc, Nc, N, Npad, out, norms, iscale, zero_buf, zero_n);
    else if (D == 192 && !wave_form)
        hipLaunchKernelGGL((knn_prep_f16_tile_kernel<12, 4>), dim3(tiles), dim3(256), 0, st, f, fc, Nc, N, Npad, out, norms, iscale, zero_buf, zero_n);
    else
        hipLaunchKernelGGL(knn_prep_f16_kernel, dim3(cdiv(tiles, 4)), dim3(256), 0, st, f, fc, Nc, N, Npad, D, tiles, out, norms, iscale, zero_buf, zero_n);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

// QG (round 3, A/B only -- LS_KNN_SWEEP_QG=2): a wave sweeps QG groups of 32 queries against every candidate fragment it loads, which
// halves the L2 -> CU stream of the candidate image (layer 1: 2 048 waves x 196 KB = 403 MB per launch with one group).  Measured slower
// (see knn_sweep_launch_t): the kernel is latency-, not stream-bound.  One group is the default.
template <int D, int QG>
__global__ __launch_bounds__(256) void knn_sweep_f16_kernel(const unsigned short* __restrict__ dq, const unsigned short* __restrict__ sq,
                                                             const int32_t* __restrict__ dst_rows, const float* __restrict__ nrm_dst,
                                                             const float* __restrict__ nrm_src, const float* __restrict__ isc_dst,
                                                             const float* __restrict__ isc_src, int Nd, int dst_n, int dst_npad, int Ns,
                                                             int ns_pad, int K, int qgroups, int nsplit, int total_waves, float epsB,
                                                             const u64* __restrict__ seedkeys, int32_t* __restrict__ surv_cnt,
                                                             unsigned short* __restrict__ surv) {
    constexpr int KK = D / 16;
    constexpr int CAPW = KB_CAPW / QG;   // survivor slots per (wave, query) in LDS
    constexpr int NQW = 32 * QG;         // queries per wave
    // dynamic LDS, per wave: hint bitmap NQW x (ns_pad / 32) words | survivor counters [NQW] | survivor lists [NQW][CAPW] u16
    extern __shared__ __attribute__((aligned(16))) unsigned lds_dyn[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;   // consecutive waves (one instance) share an XCD's L2
    if (wg >= total_waves) return;                                // (no workgroup barrier in this kernel)
    const int sp = wg % nsplit, g = (wg / nsplit) % qgroups, b = wg / (nsplit * qgroups);
    const int q0 = g * NQW, l31 = lane & 31, lh = lane >> 5;
    const unsigned short* dqb = dq + (size_t)b * dst_npad * D;
    const unsigned short* sqb = sq + (size_t)b * ns_pad * D;
    const float* nsb = nrm_src + (size_t)b * Ns;
    const float om = 1.0f - epsB;
    const int words = ns_pad >> 5;
    const int per_wave = NQW * words + NQW + NQW * CAPW / 2;   // 32-bit words
    unsigned* bits = lds_dyn + (size_t)wave * per_wave;
    int* lcnt = reinterpret_cast<int*>(bits + NQW * words);
    unsigned short* llist = reinterpret_cast<unsigned short*>(lcnt + NQW);

    // A fragments: query row q0 + 32 u + l31 (padding queries: row 0; their threshold drops everything)
    f16x8k a[QG][KK];
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        const int qi = q0 + 32 * u + l31;
        const int r = qi < Nd ? (dst_rows ? dst_rows[(size_t)b * Nd + qi] : qi) : 0;
        const unsigned short* ap = dqb + ((size_t)(r >> 5) * KK * 64 + lh * 32 + (r & 31)) * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) a[u][kk] = __builtin_bit_cast(f16x8k, *reinterpret_cast<const uint4*>(ap + (size_t)kk * 512));
    }
    // drop  <=>  d^ - eps nn > kth  <=>  S~ < ((1-eps)(nq + ns) - kth) / 2 = A[row] + Bc[candidate];  S~ = S / (s_q s_c) (the rows' scales)
    // (Round 4: branch-free, in two batches -- row indices, then everything that depends on them.  Written as `if (q < Nd) { load; load;
    //  load }` per row the compiler emitted sixteen blocks with an s_waitcnt vmcnt(0) inside each: ~2 dependent L2 round trips x 16 rows = 11 us of a
    //  44 us launch before the first tile.)
    float A[QG][16], IQ[QG][16];
    {
        int rowi[QG][16];
#pragma unroll
        for (int u = 0; u < QG; ++u)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int qc = min(q0 + 32 * u + (rr & 3) + 8 * (rr >> 2) + 4 * lh, Nd - 1);
                rowi[u][rr] = dst_rows ? dst_rows[(size_t)b * Nd + qc] : qc;
            }
        unsigned khi[QG][16];
        float nd[QG][16];
#pragma unroll
        for (int u = 0; u < QG; ++u)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int qc = min(q0 + 32 * u + (rr & 3) + 8 * (rr >> 2) + 4 * lh, Nd - 1);
                khi[u][rr] = reinterpret_cast<const unsigned*>(seedkeys + ((size_t)b * Nd + qc) * 16 + (K - 1))[1];   // high word = the K-th distance's bits
                nd[u][rr] = nrm_dst[(size_t)b * dst_n + rowi[u][rr]];
                IQ[u][rr] = isc_dst[(size_t)b * dst_n + rowi[u][rr]];
            }
#pragma unroll
        for (int u = 0; u < QG; ++u)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int q = q0 + 32 * u + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
                const float kth = khi[u][rr] == 0xFFFFFFFFu ? INFINITY : __uint_as_float(khi[u][rr]);   // fewer than K distinct hints: nothing is dropped
                A[u][rr] = q < Nd ? 0.5f * (om * nd[u][rr] - kth) : INFINITY;                             // padding query: 0 < inf, always dropped
                IQ[u][rr] = q < Nd ? IQ[u][rr] : 0.f;
            }
        // The test  S iq ic < A + Bc  with the row / column scales iq, ic exact powers of two, rearranged (round 4) so that a tile costs one
        // multiply, one fma, one compare and one add-with-carry per pair instead of seven instructions:  S < (A / iq) (1 / ic) + (Bc / ic) (1 / iq).
        // Scaling by powers of two commutes with the one rounding of the sum (A + Bc), so every decision is the one the old form took.
#pragma unroll
        for (int u = 0; u < QG; ++u)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const float inv = IQ[u][rr] > 0.f ? __uint_as_float(0x7F000000u - __float_as_uint(IQ[u][rr])) : 0.f;   // 1 / 2^e exactly (0 marks a padding query, whose A is +inf)
                A[u][rr] = IQ[u][rr] > 0.f ? A[u][rr] * inv : INFINITY;
                IQ[u][rr] = inv;
            }
    }
    // hint bitmap of the wave's queries.  The keys are loaded BEFORE the LDS clear (the wave barriers are scheduling fences: the
    // loads would otherwise be issued only after the clear, one more exposed memory round trip per wave)
    uint4 hk[QG][4];
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        const int qh = min(q0 + 32 * u + l31, Nd - 1);
        const uint4* kp = reinterpret_cast<const uint4*>(seedkeys + ((size_t)b * Nd + qh) * 16 + lh * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) hk[u][e] = kp[e];
    }
    for (int i = lane; i < NQW * words + NQW; i += 64) bits[i] = 0u;   // bitmap and counters
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < QG; ++u)
        if (q0 + 32 * u + l31 < Nd) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (hk[u][e].y != 0xFFFFFFFFu) atomicOr(&bits[(32 * u + l31) * words + (hk[u][e].x >> 5)], 1u << (hk[u][e].x & 31));
                if (hk[u][e].w != 0xFFFFFFFFu) atomicOr(&bits[(32 * u + l31) * words + (hk[u][e].z >> 5)], 1u << (hk[u][e].z & 31));
            }
        }
    __builtin_amdgcn_wave_barrier();

    const int ntiles = ns_pad >> 5, tps = (ntiles + nsplit - 1) / nsplit;
    const int t0 = sp * tps, t1 = min(ntiles, t0 + tps);
    // Round 4: the candidate fragments of tile t + 1 are REQUESTED before tile t's MFMAs start (two fragment sets, the loop unrolled by two so
    // that no register moves are needed; clamped unconditional loads).  With `load 6 fragments, 6 MFMAs, filter` per tile every wave exposed one L2
    // round trip (~1 500 cycles) per tile against ~190 matrix-pipe + ~400 VALU cycles of work; +24 VGPRs keep the kernel at three waves per SIMD.
    // Where the layer-1 launch's 42 us go (timing variants, 12 steps in flight): without the survivor loop 35, without filter arithmetic and loop 28,
    // without the MFMAs 32 -- i.e. MFMA ~10, filter ~7, survivor bookkeeping ~7, and ~18 that are the fragment stream (403 MB through the L1 path
    // = 11.7 us at 64 B/clk/CU), the per-wave set-up and the flush: no single lever is left in this kernel.
    // (Late round 4: forced to four waves per SIMD -- __launch_bounds__(256, 4): 136 -> 128 VGPRs with ten spilled, two scratch accesses per tile --
    //  the layer-1 launch went from 43 to 64 us; three waves it stays.)
    struct TileIn { f16x8k bf[KK]; float Bc, ic; };
    auto load_tile = [&](int t, TileIn& ti) {
        const int tc = min(t, t1 - 1);
        const unsigned short* bp = sqb + ((size_t)tc * KK * 64 + lane) * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) ti.bf[kk] = __builtin_bit_cast(f16x8k, *reinterpret_cast<const uint4*>(bp + (size_t)kk * 512));
        const int cg = tc * 32 + l31;
        const float ic = isc_src[(size_t)b * Ns + min(cg, Ns - 1)], iic = __uint_as_float(0x7F000000u - __float_as_uint(ic));   // 1 / 2^e exactly
        ti.Bc = 0.5f * om * nsb[min(cg, Ns - 1)] * iic;
        ti.ic = iic;
    };
    auto do_tile = [&](int t, const TileIn& ti) {
        const int cg = t * 32 + l31;
        const float Bc = ti.Bc, ic = ti.ic;
#pragma unroll
        for (int u = 0; u < QG; ++u) {
            f32x16 S;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) S = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][kk], ti.bf[kk], S, 0, 0, 0);
            unsigned mask = 0;   // bit r: row r of this lane's column survives.  Compare + add-with-carry (mask = 2 mask + !(S < rhs)), spelled in
                                 // asm: hipcc builds the same mask from v_cndmask / v_or / v_lshl, three to four instructions per pair.
            // The right-hand sides first (they do not depend on S), then a fence that owns S: the hazard recogniser does not look into inline
            // asm, so nothing would keep the compares the 18 wait states behind the last MFMA that a VALU read of its result needs -- found the
            // hard way: with the results in VGPRs (build.py: -amdgpu-mfma-vgpr-form) the compares read them early and k-NN lists came out wrong.
            float rhs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rhs[r] = __builtin_fmaf(A[u][r], ic, Bc * IQ[u][r]);
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(S));
#pragma unroll
            for (int r = 15; r >= 0; --r)
                asm("v_cmp_nlt_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(S[r]), "v"(rhs[r]) : "vcc");
            if (cg >= Ns) mask = 0;
            while (mask) {
                const int r = __builtin_ctz(mask);
                mask &= mask - 1;
                const int qr = 32 * u + (r & 3) + 8 * (r >> 2) + 4 * lh, q = q0 + qr;
                if (q < Nd && !((bits[qr * words + (cg >> 5)] >> (cg & 31)) & 1u)) {   // a hint's key is in the seeded list already
                    // wave-private list first (an LDS atomic returns in ~100 cycles; a returning global atomic per survivor inside
                    // this per-lane serial loop costs a memory round trip each); a full list spills to the global path directly
                    const int lp = atomicAdd(&lcnt[qr], 1);
                    if (lp < CAPW) {
                        llist[qr * CAPW + lp] = (unsigned short)cg;
                    } else {
                        const size_t qg = (size_t)b * Nd + q;
                        const int pos = atomicAdd(&surv_cnt[qg], 1);
                        if (pos < KS_CAP) surv[qg * KS_CAP + pos] = (unsigned short)cg;
                    }
                }
            }
        }
    };
    {
        TileIn ta, tb;
        if (t0 < t1) load_tile(t0, ta);
        for (int t = t0; t < t1; t += 2) {
            load_tile(t + 1, tb);
            __builtin_amdgcn_sched_barrier(0);
            do_tile(t, ta);
            load_tile(t + 2, ta);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < t1) do_tile(t + 1, tb);
        }
    }
    // flush: one global atomic per query reserves the wave's slots
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < QG; ++u)
        if (lh == 0 && q0 + 32 * u + l31 < Nd) {
            const int qr = 32 * u + l31;
            const int n = min(lcnt[qr], CAPW);
            if (n > 0) {
                const size_t qg = (size_t)b * Nd + q0 + qr;
                const int base = atomicAdd(&surv_cnt[qg], n);
                for (int i = 0; i < n && base + i < KS_CAP; ++i) surv[qg * KS_CAP + base + i] = llist[qr * CAPW + i];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// 0'. hints for UN-SEEDED calls ("auto hints").  Without a previous layer's graph the thresholds come from the data itself: one
// more f16 sweep in which every lane keeps, for each of its 16 query rows, the best candidate it has seen -- lane l31 of a wave
// sees the candidates of residue class l31 (mod 32) of its tile range, so a query gets 32 x nsplit class winners -- and
// knn_autohint_select_kernel keeps the 16 best of them by approximate distance.  A true neighbour is missed only if a better
// one shares its class (~2 of 16 with 64 classes), so the K-th exact distance among these hints is close to the final one and
// the seeded pipeline above runs unchanged.  Hints never influence the result.
template <int D, int QG>
__global__ __launch_bounds__(256) void knn_sweep_winners_kernel(const unsigned short* __restrict__ dq, const unsigned short* __restrict__ sq,
                                                                const int32_t* __restrict__ dst_rows, const float* __restrict__ nrm_src,
                                                                const float* __restrict__ isc_dst, const float* __restrict__ isc_src, int dst_n,
                                                                int Nd, int dst_npad, int Ns, int ns_pad, int qgroups, int nsplit,
                                                                int total_waves, float* __restrict__ win_val, int32_t* __restrict__ win_idx) {
    constexpr int KK = D / 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    if (wg >= total_waves) return;
    const int sp = wg % nsplit, g = (wg / nsplit) % qgroups, b = wg / (nsplit * qgroups);
    const int q0 = g * 32 * QG, l31 = lane & 31, lh = lane >> 5;
    const unsigned short* dqb = dq + (size_t)b * dst_npad * D;
    const unsigned short* sqb = sq + (size_t)b * ns_pad * D;
    const float* nsb = nrm_src + (size_t)b * Ns;
    f16x8k a[QG][KK];
#pragma unroll
    for (int u = 0; u < QG; ++u) {
        const int qi = q0 + 32 * u + l31;
        const int r = qi < Nd ? (dst_rows ? dst_rows[(size_t)b * Nd + qi] : qi) : 0;
        const unsigned short* ap = dqb + ((size_t)(r >> 5) * KK * 64 + lh * 32 + (r & 31)) * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) a[u][kk] = __builtin_bit_cast(f16x8k, *reinterpret_cast<const uint4*>(ap + (size_t)kk * 512));
    }
    float best[QG][16], IQ[QG][16];
    int bt[QG][16];
#pragma unroll
    for (int u = 0; u < QG; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            best[u][r] = -INFINITY; bt[u][r] = -1;
            const int q = q0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * lh;
            IQ[u][r] = q < Nd ? isc_dst[(size_t)b * dst_n + (dst_rows ? dst_rows[(size_t)b * Nd + q] : q)] : 0.f;
        }
    const int ntiles = ns_pad >> 5, tps = (ntiles + nsplit - 1) / nsplit;
    const int t0 = sp * tps, t1 = min(ntiles, t0 + tps);
#pragma unroll 2
    for (int t = t0; t < t1; ++t) {
        const unsigned short* bp = sqb + ((size_t)t * KK * 64 + lane) * 8;
        f16x8k bf[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) bf[kk] = __builtin_bit_cast(f16x8k, *reinterpret_cast<const uint4*>(bp + (size_t)kk * 512));
        const int cg = t * 32 + l31;
        const float hb = cg < Ns ? 0.5f * nsb[min(cg, Ns - 1)] : INFINITY;   // padding columns can never win
        const float ic = isc_src[(size_t)b * Ns + min(cg, Ns - 1)];
#pragma unroll
        for (int u = 0; u < QG; ++u) {
            f32x16 S;
#pragma unroll
            for (int r = 0; r < 16; ++r) S[r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) S = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u][kk], bf[kk], S, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = S[r] * (IQ[u][r] * ic) - hb;                  // S~ - |s|^2 / 2: largest = nearest (|q|^2 is per query)
                const bool better = v > best[u][r];
                best[u][r] = better ? v : best[u][r];
                bt[u][r] = better ? t : bt[u][r];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < QG; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = q0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (q < Nd) {
                const size_t o = (((size_t)b * Nd + q) * nsplit + sp) * 32 + l31;
                win_val[o] = best[u][r];
                win_idx[o] = bt[u][r] >= 0 ? bt[u][r] * 32 + l31 : -1;
            }
        }
}
// one wave per query: the 16 best of its W = 32 * nsplit <= 64 class winners -> hints[q][16]
__global__ __launch_bounds__(256) void knn_autohint_select_kernel(const float* __restrict__ win_val, const int32_t* __restrict__ win_idx, int W,
                                                                  int total_q, int32_t* __restrict__ hints) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= total_q) return;
    u64 k = ~0ull;
    if (lane < W) {
        const int idx = win_idx[(size_t)q * W + lane];
        if (idx >= 0) {
            unsigned u = __float_as_uint(win_val[(size_t)q * W + lane]);
            u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;       // monotone in the float value
            k = ((u64)(~u) << 32) | (unsigned)idx;             // ascending key order = descending value
        }
    }
    LS_SORT64(cx64, k, lane)
    if (lane < 16) hints[(size_t)q * 16 + lane] = (k == ~0ull) ? -1 : (int)(unsigned)k;
}

// ---------------------------------------------------------------------------------------------------------------------
// ONE-SWEEP path for un-seeded calls with at most KO_MAXNS candidates (round 3; the encoder's layers 3 and 4).  The auto-hint path above
// sweeps the pair matrix twice (class winners, then the filter against the exact K-th distance of the 16 hints) with an exact "seed"
// phase in between: five launches, 16 + ~40 exact distances per query.  Here ONE sweep stores the approximate cosine of every pair
// (16-bit fixed point: |q'||s'| cos~ = S~ to 2^-16 |q'||s'|, far inside the f16 bound) and ONE wave per query then
//   1. rebuilds d^ = |q'|^2 + |s'|^2 - 2 |q'||s'| cos~ with its two-sided bound  lo = d^ - eps nn <= d_canonical <= d^ + eps nn = hi
//      (nn = |q'|^2 + |s'|^2, eps = eps_b of the f16 sweep + 2^-14 for the fixed point and this kernel's own roundings),
//   2. takes T = the K-th smallest of the 64 per-lane minima of hi -- 64 disjoint candidate groups contribute one candidate each, so at
//      least K candidates have a canonical distance <= T: T bounds the K-th canonical distance from above,
//   3. keeps every candidate with lo <= T (a candidate of the true top K has d_canonical <= K-th <= T, hence lo <= T),
//   4. computes the canonical distance of the survivors (same quad chains, same key network as knn_finish_wave_kernel).
// Three launches (image, sweep, finish), no hints, no thresholds from exact distances; the lists only ever hold canonical keys, so the
// result is bit-identical by construction.  Non-finite rows: T is not finite -> everything survives -> brute force, still exact.
constexpr int KO_MAXNS = 512;
#ifndef LS_KO_US
#define LS_KO_US 1
#endif
#ifndef LS_KO_WPS
#define LS_KO_WPS 4
#endif
// quad steps (16 survivors each) per iteration of the finish / waves per SIMD it is compiled for.  Measured at layer 3 (32 768 queries,
// ~20 survivors each, 12 steps in flight): 3 steps at 3 waves 82 us, 2 at 3: 82, 2 at 4 (8 spills): 83, 1 at 4 (110 VGPRs): 73.  The
// front end (bounds, T, compaction) is 25 us of that; the rest is the canonical chains: VALU-issue-bound at ~70 % of the issue rate
constexpr int KO_US = LS_KO_US;
constexpr int KO_WPS = LS_KO_WPS;
template <int D>
__global__ __launch_bounds__(256) void knn_sweep_store_kernel(const unsigned short* __restrict__ dq, const unsigned short* __restrict__ sq,
                                                              const int32_t* __restrict__ dst_rows, const float* __restrict__ nrm_dst,
                                                              const float* __restrict__ nrm_src, const float* __restrict__ isc_dst,
                                                              const float* __restrict__ isc_src, int Nd, int dst_n, int dst_npad, int Ns, int ns_pad,
                                                              int qgroups, int nsplit, int total_waves, short* __restrict__ cosq) {
    constexpr int KK = D / 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;
    if (wg >= total_waves) return;
    const int sp = wg % nsplit, g = (wg / nsplit) % qgroups, b = wg / (nsplit * qgroups);
    const int q0 = g * 32, l31 = lane & 31, lh = lane >> 5;
    const unsigned short* dqb = dq + (size_t)b * dst_npad * D;
    const unsigned short* sqb = sq + (size_t)b * ns_pad * D;
    const float* nsb = nrm_src + (size_t)b * Ns;
    f16x8k a[KK];
    {
        const int qi = q0 + l31;
        const int r = qi < Nd ? (dst_rows ? dst_rows[(size_t)b * Nd + qi] : qi) : 0;
        const unsigned short* ap = dqb + ((size_t)(r >> 5) * KK * 64 + lh * 32 + (r & 31)) * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) a[kk] = __builtin_bit_cast(f16x8k, *reinterpret_cast<const uint4*>(ap + (size_t)kk * 512));
    }
    float rq[16];          // 32767 / |q'| of the lane's 16 query rows (0: padding query or a row at the centre)
    short* orow[16];       // where the row's cosines go (null: padding query)
    {   // branch-free, in two batches (row indices, then the norms and scales): see knn_sweep_f16_kernel
        int rowi[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int qc = min(q0 + (rr & 3) + 8 * (rr >> 2) + 4 * lh, Nd - 1);
            rowi[rr] = dst_rows ? dst_rows[(size_t)b * Nd + qc] : qc;
        }
        float nq[16], is[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) { nq[rr] = nrm_dst[(size_t)b * dst_n + rowi[rr]]; is[rr] = isc_dst[(size_t)b * dst_n + rowi[rr]]; }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int q = q0 + (rr & 3) + 8 * (rr >> 2) + 4 * lh;
            rq[rr] = (q < Nd && nq[rr] > 0.f) ? 32767.0f * __builtin_amdgcn_rsqf(nq[rr]) * is[rr] : 0.f;   // (the row's image scale folded in)
            orow[rr] = q < Nd ? cosq + ((size_t)b * Nd + q) * ns_pad : nullptr;
        }
    }
    const int ntiles = ns_pad >> 5, tps = (ntiles + nsplit - 1) / nsplit;
    const int t0 = sp * tps, t1 = min(ntiles, t0 + tps);
#pragma unroll 2
    for (int t = t0; t < t1; ++t) {
        const unsigned short* bp = sqb + ((size_t)t * KK * 64 + lane) * 8;
        f16x8k bf[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) bf[kk] = __builtin_bit_cast(f16x8k, *reinterpret_cast<const uint4*>(bp + (size_t)kk * 512));
        const int cg = t * 32 + l31;
        const float ns = nsb[min(cg, Ns - 1)];
        const float rs = ns > 0.f ? __builtin_amdgcn_rsqf(ns) * isc_src[(size_t)b * Ns + min(cg, Ns - 1)] : 0.f;
        f32x16 S;
#pragma unroll
        for (int r = 0; r < 16; ++r) S[r] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) S = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], bf[kk], S, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // |S| <= |q'||s'| for the exact product, so the clamp can only move S~ towards it
            const float c = fminf(fmaxf(S[r] * rq[r] * rs, -32767.0f), 32767.0f);
            if (orow[r]) orow[r][cg] = (short)__float2int_rn(c);
        }
    }
}

template <int CC, bool FMA>
__global__ __launch_bounds__(256, KO_WPS) void knn_finish_select_kernel(const float* __restrict__ dstf, const float* __restrict__ srcf,
                                                                   const int32_t* __restrict__ dst_rows, const float* __restrict__ nrm_dst,
                                                                   const float* __restrict__ nrm_src, int Nd, int dst_n, int Ns, int ns_pad, int K,
                                                                   float epsS, const short* __restrict__ cosq, int32_t* __restrict__ idx_out,
                                                                   float* __restrict__ dist_out, int total_q, int32_t* __restrict__ surv_cnt) {
    constexpr int RF = 3 * CC;
    constexpr int NV = KO_MAXNS / 64;
    __shared__ unsigned short llist[4][KO_MAXNS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int qg = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;   // consecutive queries (one instance) share an XCD
    if (qg >= total_q) return;
    const int b = qg / Nd, q = qg % Nd;
    const float* sbase = srcf + (size_t)b * Ns * RF;
    const int r = dst_rows ? dst_rows[qg] : q;
    const float* qrow = dstf + ((size_t)b * dst_n + r) * RF;
    const int quad = lane >> 2;
    const bool qlast = (lane & 3) == 3;
    QuadRow<CC> qv;
    qv.load(qrow, lane);

    // 1. two-sided bounds of every candidate's canonical distance (candidate j = 64 i + lane)
    // (v_sqrt_f32, 1 ulp, instead of the correctly rounded sqrtf -- ten instructions per candidate column in a VALU-bound kernel; its relative
    //  error 2^-23 on |q'||s'| <= nn / 2 is covered by the 2^-21 the launch adds to epsS)
    const float nq = nrm_dst[(size_t)b * dst_n + r], snq = __builtin_amdgcn_sqrtf(nq);
    const short* cp = cosq + (size_t)qg * ns_pad;
    const float* nsb = nrm_src + (size_t)b * Ns;
    float lo[NV];
    float himin = INFINITY;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = i * 64 + lane;
        const bool valid = j < Ns;
        const float c = (float)cp[min(j, ns_pad - 1)] * (1.0f / 32767.0f);
        const float ns = nsb[min(j, Ns - 1)];
        const float nn = nq + ns;
        const float dh = nn - 2.0f * c * (snq * __builtin_amdgcn_sqrtf(ns)), e = epsS * nn;
        lo[i] = valid ? dh - e : INFINITY;
        const float hi = (valid && dh == dh) ? fmaxf(dh + e, 0.0f) : INFINITY;   // (a NaN bound never wins the minimum; its candidate survives below)
        himin = fminf(himin, hi);
    }
    // 2. T = the K-th smallest of the 64 lane minima (non-negative floats order like their bit patterns)
    const float T = __uint_as_float(kth_smallest_upper_bound(__float_as_uint(himin), K < 64 ? K : 64));   // (knn_common.h: a valid bound, <= 2^-8 above the K-th minimum)
    // 3. survivors -> the wave's LDS list
    unsigned short* sp = llist[wave];
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = i * 64 + lane;
        const bool keep = j < Ns && !(lo[i] > T);              // (NaN bounds survive)
        const unsigned long long m = __ballot(keep);
        const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (keep) sp[pos] = (unsigned short)j;
        cnt += __popcll(m);
    }
    if (lane == 0) surv_cnt[qg] = -cnt - 1;   // how many candidates get a canonical distance: the exact-phase statistic (knn_sweep_stats_launch).  Stored as
                                              // -(count + 1): a RAW count of up to KO_MAXNS, not a sweep-path list length (whose values above KS_CAP mean "scanned all Ns")
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the list is wave-private, LDS ops of a wave complete in order
    __builtin_amdgcn_wave_barrier();
    // 4. canonical keys of the survivors, 16 KO_US per step, sorted with the list so far by the 64-lane network
    // (Measured and not kept: two passes -- first the candidates with hi <= T, which give an exact K-th distance, then the rest against
    // that one-sided bound: fewer exact distances but a second dependent round of gathers per query; 86 -> 97 us at layer 3 with the
    // bf16 image.  With the f16 image the survivors fit one 48-wide step anyway.)
    u64 best = ~0ull;
    for (int base = 0; base < cnt; base += 16 * KO_US) {
        u64 ks[3] = {~0ull, ~0ull, ~0ull};
#pragma unroll
        for (int u = 0; u < KO_US; ++u) {
            if (base + u * 16 < cnt) {   // wave-uniform
                const int j = base + u * 16 + quad;
                const bool v = j < cnt;
                const int c = v ? (int)sp[j] : 0;
                ks[u] = make_key(quad_pair_distance<CC, FMA>(qv, qrow, sbase + (size_t)c * RF, lane), c, v & qlast);
            }
        }
        const int src = ((lane & 15) << 2) + 3;
        const u64 n0 = bperm64(src, ks[0]), n1 = bperm64(src, ks[1]), n2 = KO_US > 2 ? bperm64(src, ks[2]) : ~0ull;
        u64 k = lane < 16 ? best : (lane < 32 ? n0 : (lane < 48 ? n1 : n2));
        if constexpr (KO_US == 1) { LS_SORT32(cx64, k, lane) }   // 16 + 16 keys: the 32-lane network (15 of the 21 exchanges) does it
        else { LS_SORT64(cx64, k, lane) }
        best = k;
    }
    if (lane < K) {
        const size_t o = (size_t)qg * K + lane;
        const unsigned hi = (unsigned)(best >> 32), lw = (unsigned)best;
        idx_out[o] = hi == 0xFFFFFFFFu ? -1 : (int)lw;
        if (dist_out) dist_out[o] = hi == 0xFFFFFFFFu ? INFINITY : __uint_as_float(hi);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// FUSED build (round 5): image -> ONE kernel.  The seeded path above is five launches (image, seed, sweep, finish) that hand seed keys
// (u64 [B Nd][16]), survivor lists (u16 [B Nd][256]) and counters to each other through HBM -- 150 MB moved per layer-1 call for 54 MB of
// compulsory traffic, three launch boundaries, and 16 hint distances per query whose only use is a threshold.  The one-sweep path needs no
// hints but stores a cosine per pair (128 MB at Ns = 1024).  Here a workgroup of eight waves owns 32 queries of one instance and does
// everything on chip:
//   1. sweep      wave w takes the candidate tiles w, w + 8, ...: S = q'.s' on the f16 matrix cores into REGISTERS (TPW x 16 per lane), then in
//                 place  lo = d^ - eps nn  with  d^ = |q'|^2 + |s'|^2 - 2 S iq ic  (lo <= d_canonical <= hi = d^ + eps nn, the bound of the f16
//                 sweep: eps_b above, + 2^-21 for the roundings of d^ / lo / hi themselves); every lane keeps the minimum of hi per query row;
//   2. threshold  the minima of the 8 x 32 (wave, column class) groups of a row are folded into 64 LDS slots by atomic-min (disjoint candidate
//                 groups stay disjoint); T = an upper bound of the K-th smallest slot (knn_common.h) -- at least K candidates have a canonical
//                 distance <= T, so T bounds the K-th canonical distance from above;
//   3. filter     a candidate survives iff !(lo > T) (a candidate of the true top K has d_canonical <= K-th <= T, hence lo <= T; NaN bounds
//                 survive): appended to the query's LDS list (~18 of 1024);
//   4. exact      the workgroup's survivors as ONE flat list of (query, candidate) pairs, 16 per wave step on the quad chains
//                 (quad_pair_distance: the canonical fp32 chain), query rows out of LDS -- every step is full, whereas one wave per query leaves
//                 its second step of 16 nearly empty;
//   5. select     per query the K smallest (distance, index) keys by the 32- / 64-lane network, two queries per wave pass when both fit 32.
// A query with more than KF_LCAP survivors (duplicates, non-finite rows) is finished by brute force over all candidates: slow, still exact.
// The lists only ever hold canonical keys and a dropped candidate provably cannot enter them: bit-identical to the oracle by construction.
// Hints are not used at all (they never influenced results).
constexpr int KF_QT = 32;        // queries per workgroup (one 32-row MFMA operand)
constexpr int KF_WAVES = 8;
constexpr int KF_LCAP = 64;      // survivors per query that go through the flat exact phase (one 64-lane sorting pass)
constexpr int KF_MAXNS = KF_WAVES * 4 * 32;   // at most four tiles per wave

template <int CC>
struct KfLds {
    float qrows[KF_QT][3 * CC];          // the workgroup's query rows (fp32, x-major as in global memory)
    float2 qn[KF_QT];                    // {|q'|^2, -2 iq} per query row (padding queries: {0, 0})
    unsigned hm[KF_QT][64];              // group minima of hi (bit patterns of non-negative floats)
    float T[KF_QT];
    int cnt[KF_QT];
    int base[KF_QT + 1];
    unsigned short list[KF_QT][KF_LCAP];
    unsigned plist[KF_QT * KF_LCAP];     // flat pair list: candidate | row << 16 | slot << 21
    u64 keys[KF_QT][KF_LCAP];
};

// brute force of one query by one wave: every candidate's canonical key, 48 per pass, merged with the best 16 so far
template <int CC, bool FMA>
__device__ __noinline__ u64 kf_brute_row(const float* __restrict__ qrow, const float* __restrict__ sbase, int Ns, int lane) {
    constexpr int RF = 3 * CC;
    const int quad = lane >> 2;
    const bool qlast = (lane & 3) == 3;
    QuadRow<CC> qv;
    qv.load(qrow, lane);
    u64 best = ~0ull;
    for (int base = 0; base < Ns; base += 48) {
        u64 ks[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = base + u * 16 + quad;
            const bool v = c < Ns;
            ks[u] = make_key(quad_pair_distance<CC, FMA>(qv, qrow, sbase + (size_t)(v ? c : 0) * RF, lane), c, v & qlast);
        }
        const int src = ((lane & 15) << 2) + 3;
        const u64 n0 = bperm64(src, ks[0]), n1 = bperm64(src, ks[1]), n2 = bperm64(src, ks[2]);
        u64 k = lane < 16 ? best : (lane < 32 ? n0 : (lane < 48 ? n1 : n2));
        LS_SORT64(cx64, k, lane)
        best = k;
    }
    return best;     // lanes 0 .. 15: the 16 smallest keys, ascending
}

// Compiled for four waves per SIMD = two workgroups per CU (116 - 128 registers, no spills).  Measured (round 5): one workgroup per CU by LDS padding
// 151 / 81 us at layers 1 / 2 against 110 / 71 us.
#ifndef LS_KF_WPE
#define LS_KF_WPE(TPW) 4
#endif
template <int CC, bool FMA, int TPW>
__global__ __launch_bounds__(64 * KF_WAVES, LS_KF_WPE(TPW)) void knn_fused_kernel(const float* __restrict__ dstf, const float* __restrict__ srcf, const int32_t* __restrict__ dst_rows,
                                                                  const unsigned short* __restrict__ dq, const unsigned short* __restrict__ sq,
                                                                  const float* __restrict__ nrm_dst, const float* __restrict__ nrm_src,
                                                                  const float* __restrict__ isc_dst, const float* __restrict__ isc_src, int Nd, int dst_n,
                                                                  int dst_npad, int Ns, int ns_pad, int K, int qtiles, float eps,
                                                                  int32_t* __restrict__ idx_out, float* __restrict__ dist_out, int32_t* __restrict__ surv_cnt) {
    constexpr int D = 3 * CC, KK = D / 16, RF = 3 * CC;
    __shared__ __attribute__((aligned(16))) KfLds<CC> L;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);          // the query tiles of one instance share an XCD's L2 (its image and rows)
    const int b = logical / qtiles, q0 = (logical - b * qtiles) * KF_QT;
    const unsigned short* dqb = dq + (size_t)b * dst_npad * D;
    const unsigned short* sqb = sq + (size_t)b * ns_pad * D;
    const float* sbase = srcf + (size_t)b * Ns * RF;
    const float* dbase = dstf + (size_t)b * dst_n * RF;
    const int ntiles = ns_pad >> 5;

    // ---- 0. the query rows (fp32) into LDS, their norms / scales, the LDS state
    for (int i = tid; i < KF_QT * (RF / 4); i += 64 * KF_WAVES) {
        const int qr = i / (RF / 4), c4 = i - qr * (RF / 4);
        const int q = min(q0 + qr, Nd - 1);
        const int r = dst_rows ? dst_rows[(size_t)b * Nd + q] : q;
        *reinterpret_cast<float4*>(&L.qrows[qr][c4 * 4]) = *reinterpret_cast<const float4*>(dbase + (size_t)r * RF + c4 * 4);
    }
    if (tid < KF_QT) {
        const int q = q0 + tid;
        const int r = q < Nd ? (dst_rows ? dst_rows[(size_t)b * Nd + q] : q) : 0;
        const float nq = nrm_dst[(size_t)b * dst_n + r], iq = isc_dst[(size_t)b * dst_n + r];
        L.qn[tid] = q < Nd ? make_float2(nq, -2.0f * iq) : make_float2(0.f, 0.f);
        L.cnt[tid] = 0;
    }
    for (int i = tid; i < KF_QT * 64; i += 64 * KF_WAVES) (&L.hm[0][0])[i] = 0x7F800000u;

    // ---- 1. sweep: this wave's tiles, S in registers
    f16x8k a[KK];
    {
        const int qi = q0 + l31;
        const int r = qi < Nd ? (dst_rows ? dst_rows[(size_t)b * Nd + qi] : qi) : 0;
        const unsigned short* ap = dqb + ((size_t)(r >> 5) * KK * 64 + lh * 32 + (r & 31)) * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) a[kk] = __builtin_bit_cast(f16x8k, *reinterpret_cast<const uint4*>(ap + (size_t)kk * 512));
    }
    f32x16 S[TPW];
    float cns[TPW], cic[TPW];          // the tile's column: |s'|^2 and its inverse image scale (this lane's candidate)
    {
        f16x8k bf[2][KK];
        auto load_b = [&](int t, f16x8k (&dstb)[KK]) {
            const int tg = min(wave + KF_WAVES * t, ntiles - 1);
            const unsigned short* bp = sqb + ((size_t)tg * KK * 64 + lane) * 8;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) dstb[kk] = __builtin_bit_cast(f16x8k, *reinterpret_cast<const uint4*>(bp + (size_t)kk * 512));
        };
        load_b(0, bf[0]);
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            if (t + 1 < TPW) load_b(t + 1, bf[(t + 1) & 1]);
            const int cg = min((wave + KF_WAVES * t) * 32 + l31, Ns - 1);
            cns[t] = nrm_src[(size_t)b * Ns + cg];
            cic[t] = isc_src[(size_t)b * Ns + cg];
#pragma unroll
            for (int r = 0; r < 16; ++r) S[t][r] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) S[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], bf[t & 1][kk], S[t], 0, 0, 0);
        }
    }
    __syncthreads();     // qn, hm, cnt are initialised
    // bounds in place: S := lo; per row the minimum of hi over this lane's columns, folded straight into the row's LDS slot (step 2).  Rows outside,
    // tiles inside: a row's {|q'|^2, -2 iq} is one broadcast LDS read, nothing but S stays live across rows.  (fminf ignores a NaN operand: a NaN
    // bound never wins the minimum; its candidate survives the filter below.)  Padding columns (candidate >= Ns) and padding tiles contribute nothing.
    bool colv[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) colv[t] = (wave + KF_WAVES * t) < ntiles && (wave + KF_WAVES * t) * 32 + l31 < Ns;
    {
        const int slot = (wave & 1) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float2 qn = L.qn[qr];
            float hmin = INFINITY;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const float nn = qn.x + cns[t], w = qn.y * cic[t];
                const float dh = __builtin_fmaf(S[t][r], w, nn), e = eps * nn;
                hmin = colv[t] ? fminf(hmin, dh + e) : hmin;
                float lo = dh - e;
                asm volatile("" : "+v"(lo));      // (materialised HERE: left free, the optimiser sinks the subtraction into the filter and keeps dh AND nn alive instead -- two registers per pair)
                S[t][r] = lo;
            }
            // ---- 2. threshold: (non-negative floats order like their bit patterns; +inf = "no candidate")
            const float h = fmaxf(hmin, 0.0f);
            if (h < INFINITY) atomicMin(&L.hm[qr][slot], __float_as_uint(h));
#ifndef LS_KF_NOSB
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (four rows' LDS reads in flight at a time: hoisted all sixteen, they cost 32 registers beside S)
#endif
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < KF_QT / KF_WAVES; ++u) {
        const int qr = wave * (KF_QT / KF_WAVES) + u;
        const unsigned tb = kth_smallest_upper_bound(L.hm[qr][lane], K < 64 ? K : 64);
        if (lane == 0) L.T[qr] = __uint_as_float(tb);
    }
    __syncthreads();
    // ---- 3. filter
    {
        unsigned mask[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) mask[t] = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float Tr = L.T[(r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
            for (int t = 0; t < TPW; ++t) mask[t] |= (S[t][r] > Tr) ? 0u : (1u << r);
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int cg = (wave + KF_WAVES * t) * 32 + l31;
            unsigned m = colv[t] ? mask[t] : 0u;
            while (m) {
                const int r = __builtin_ctz(m);
                m &= m - 1;
                const int qr = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (q0 + qr < Nd) {
                    const int pos = atomicAdd(&L.cnt[qr], 1);
                    if (pos < KF_LCAP) L.list[qr][pos] = (unsigned short)cg;
                }
            }
        }
    }
    __syncthreads();
    // ---- 4. the flat pair list (queries that overflowed their list are left to the brute-force pass of step 5)
    if (wave == 0) {
        const int c = (lane < KF_QT && L.cnt[lane] <= KF_LCAP) ? L.cnt[lane] : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < KF_QT; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            incl += lane >= o ? v : 0;
        }
        if (lane < KF_QT) L.base[lane] = incl - c;
        if (lane == KF_QT - 1) L.base[KF_QT] = incl;
        if (lane < KF_QT && q0 + lane < Nd && surv_cnt) surv_cnt[(size_t)b * Nd + q0 + lane] = -(L.cnt[lane] > KF_LCAP ? Ns : L.cnt[lane]) - 1;   // statistic (knn_stats_kernel)
    }
    __syncthreads();
    for (int i = tid; i < KF_QT * KF_LCAP; i += 64 * KF_WAVES) {
        const int qr = i / KF_LCAP, j = i - qr * KF_LCAP;
        const int c = L.cnt[qr];
        if (c <= KF_LCAP && j < c) L.plist[L.base[qr] + j] = (unsigned)L.list[qr][j] | ((unsigned)qr << 16) | ((unsigned)j << 21);
    }
    __syncthreads();
    {
        // Two pairs in flight per quad: the candidate row of step s + 1 is requested before the chain of step s runs (a step is ~2 us of L2
        // gather latency in front of ~600 cycles of dependent adds; a workgroup has ~5 steps).
        const int P = L.base[KF_QT];
        const int Q = wave * 16 + (lane >> 2);
        const bool qlast = (lane & 3) == 3;
        struct Pair { unsigned ent; bool v; QuadRow<CC> c; };
        auto fetch = [&](int p0, Pair& pr) {
            const int p = p0 + Q;
            pr.v = p < P;
            pr.ent = L.plist[pr.v ? p : 0];
            pr.c.load(sbase + (size_t)(pr.ent & 0xFFFFu) * RF, lane, 0);
        };
        auto finish = [&](const Pair& pr) {
            const int cand = pr.ent & 0xFFFFu, qr = (pr.ent >> 16) & 31, slot = pr.ent >> 21;
            const float* qrow = &L.qrows[qr][0];
            QuadRow<CC> qv;
            qv.load(qrow, lane);
            float d;
            if constexpr (CC == 32) {
                d = quad_round<CC, FMA, true>(0.0f, qv, pr.c);
            } else {
                const float* crow = sbase + (size_t)cand * RF;
                d = quad_round<CC, FMA, false>(0.0f, qv, pr.c);
#pragma unroll
                for (int r = 1; r < CC / 32; ++r) {
                    QuadRow<CC> q, c;
                    q.load(qrow, lane, r);
                    c.load(crow, lane, r);
                    d = (r == CC / 32 - 1) ? quad_round<CC, FMA, true>(d, q, c) : quad_round<CC, FMA, false>(d, q, c);
                }
            }
            if (pr.v && qlast) L.keys[qr][slot] = make_key(d, cand, true);
        };
#ifdef LS_KF_NOPIPE      // dev A/B: one pair per quad at a time
        for (int p0 = 0; p0 < P; p0 += 16 * KF_WAVES) { Pair pa; fetch(p0, pa); finish(pa); }
        if (false) {
#else
        if (P > 0) {
#endif
            Pair pa, pb;
            fetch(0, pa);
            for (int p0 = 0; p0 < P; p0 += 2 * 16 * KF_WAVES) {
                fetch(p0 + 16 * KF_WAVES, pb);
                __builtin_amdgcn_sched_barrier(0);
                finish(pa);
                if (p0 + 16 * KF_WAVES < P) {       // (uniform)
                    fetch(p0 + 2 * 16 * KF_WAVES, pa);
                    __builtin_amdgcn_sched_barrier(0);
                    finish(pb);
                }
            }
        }
    }
    __syncthreads();
    // ---- 5. select: this wave's four queries, two per pass when both lists fit 32 lanes
    auto emit = [&](int qr, u64 k, int e) {
        const int q = q0 + qr;
        if (q < Nd && e < K) {
            const size_t o = ((size_t)b * Nd + q) * K + e;
            const unsigned hi = (unsigned)(k >> 32), lw = (unsigned)k;
            idx_out[o] = hi == 0xFFFFFFFFu ? -1 : (int)lw;
            if (dist_out) dist_out[o] = hi == 0xFFFFFFFFu ? INFINITY : __uint_as_float(hi);
        }
    };
#pragma unroll
    for (int u = 0; u < KF_QT / KF_WAVES; u += 2) {
        const int qa = wave * (KF_QT / KF_WAVES) + u, qb = qa + 1;
        const int ca = L.cnt[qa], cb = L.cnt[qb];
        if (ca <= 32 && cb <= 32) {
            const int qr = lh ? qb : qa, c = lh ? cb : ca;
            u64 k = l31 < c ? L.keys[qr][l31] : ~0ull;
            LS_SORT32(cx64, k, lane)
            emit(qr, k, l31);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int qr = h ? qb : qa, c = h ? cb : ca;
                u64 k;
                if (c <= KF_LCAP) {
                    k = lane < c ? L.keys[qr][lane] : ~0ull;
                    LS_SORT64(cx64, k, lane)
                } else {
                    k = q0 + qr < Nd ? kf_brute_row<CC, FMA>(&L.qrows[qr][0], sbase, Ns, lane) : ~0ull;
                }
                emit(qr, k, lane);
            }
        }
    }
}

static inline size_t pad32(size_t n) { return (n + 31) & ~(size_t)31; }
static bool knn_sweep_f16_enabled() {
    static const bool off = getenv("LS_KNN_SWEEP_FP32") && atoi(getenv("LS_KNN_SWEEP_FP32")) != 0;   // A/B: fp32 sweep kernel
    return !off;
}
size_t knn_sweep_scratch_bytes(int B, int Nd, int dst_n, int Ns, int C) {
    const size_t nq = (size_t)B * Nd;
    const size_t D = (size_t)3 * C;
    return ((size_t)B * Ns + (size_t)B * dst_n) * sizeof(float) + 256      // row norms
           + nq * 16 * sizeof(u64) + nq * sizeof(int32_t) + nq * KS_CAP * sizeof(unsigned short) + 256
           + (size_t)B * D * sizeof(float) + 256                            // instance means
           + ((size_t)B * pad32(Ns) + (size_t)B * pad32(dst_n)) * D * sizeof(unsigned short) + 512    // f16 images (src, dst)
           + nq * 64 * (sizeof(float) + sizeof(int32_t)) + nq * 16 * sizeof(int32_t) + 512            // auto hints: class winners, hints
           + (pad32(Ns) <= (size_t)KO_MAXNS ? nq * pad32(Ns) * sizeof(short) + 256 : 0)                // one-sweep path: the pair cosines
           + ((size_t)B * Ns + (size_t)B * dst_n) * sizeof(float) + 256;                               // inverse row scales of the f16 images
}

// byte offset of the per-query survivor counters inside a sweep-path scratch area: row norms | inverse row scales | seed keys | COUNTERS | ...
// (ONE definition for knn_sweep_launch_t, which lays the area out, and knn_sweep_stats_launch, which reads the counters back)
static size_t knn_sweep_surv_cnt_offset(int B, int Ns, int dst_n, size_t nq) {
    size_t off = (size_t)B * Ns * sizeof(float) + (size_t)B * dst_n * sizeof(float);     // row norms
    off = (off + 255) & ~(size_t)255;
    off += (size_t)B * Ns * sizeof(float) + (size_t)B * dst_n * sizeof(float);            // inverse row scales
    off = (off + 255) & ~(size_t)255;
    return off + nq * 16 * sizeof(u64);                                                   // seed keys
}

template <int CC>
static int knn_sweep_launch_t(const float* dst, const float* src, const int32_t* dst_rows, int B, int Nd, int dst_n, int Ns, int K, bool fma,
                              int32_t* idx_out, float* dist_out, const int32_t* seed_idx, int seed_n, int seed_by_row, void* scratch,
                              hipStream_t st) {
    const int C = CC;
    constexpr int D = 3 * CC;
    const size_t nq = (size_t)B * Nd;
    const bool f16img = knn_sweep_f16_enabled() && Ns <= KB_MAXNS;
    char* sc = (char*)scratch;
    float* nsrc = (float*)sc;
    float* ndst = nsrc;
    size_t off = (size_t)B * Ns * sizeof(float);
    if (dst != src) ndst = (float*)(sc + off);
    off += (size_t)B * dst_n * sizeof(float);
    off = (off + 255) & ~(size_t)255;
    float* isrc = (float*)(sc + off);          // inverse row scales of the f16 images (knn_prep_f16_kernel)
    float* idst = isrc;
    off += (size_t)B * Ns * sizeof(float);
    if (dst != src) idst = (float*)(sc + off);
    off += (size_t)B * dst_n * sizeof(float);
    off = (off + 255) & ~(size_t)255;
    u64* seedkeys = (u64*)(sc + off);
    off += nq * 16 * sizeof(u64);
    if (off != knn_sweep_surv_cnt_offset(B, Ns, dst_n, nq)) { set_error("knn sweep: scratch layout and knn_sweep_surv_cnt_offset disagree"); return LS_ERR_INVALID; }
    int32_t* surv_cnt = (int32_t*)(sc + off);
    off += nq * sizeof(int32_t);
    unsigned short* surv = (unsigned short*)(sc + off);
    off += nq * KS_CAP * sizeof(unsigned short);
    off = (off + 255) & ~(size_t)255;
    float* mu = (float*)(sc + off);
    off += (size_t)B * D * sizeof(float);
    off = (off + 255) & ~(size_t)255;
    unsigned short* sq = (unsigned short*)(sc + off);
    const int ns_pad = (int)pad32(Ns), dst_npad = (int)pad32(dst_n);
    unsigned short* dq = sq;
    if (dst != src) dq = sq + (size_t)B * ns_pad * D;
    int rc;
    // query groups per sweep wave (knn_sweep_f16_kernel): LS_KNN_SWEEP_QG=2 halves the candidate stream per query (A/B; measured SLOWER in
    // round 3 -- k-NN build 138 / 102 / 148 / 92 us at layers 1 - 4 with one group, 161 / 133 / 152 / 97 us with two: the sweep is bound by
    // the latency of its fragment loads at 2 - 3 waves per SIMD, not by the L2 stream -- so one group stays the default)
    static const int qg_env = getenv("LS_KNN_SWEEP_QG") ? atoi(getenv("LS_KNN_SWEEP_QG")) : 0;
    const int qg = (qg_env == 2 && Nd >= 64) ? 2 : 1;
    float* win_val = nullptr;      // auto hints: class winners of the first sweep (selected inside the seed kernel)
    int32_t* win_idx = nullptr;
    int win_w = 0;
    if (f16img) {
        static_assert(D <= 192, "knn_prep_f16_kernel keeps the centre in a 192-float LDS slot per wave");
        (void)mu;
        rc = knn_prep_launch(src, src, Ns, B, Ns, ns_pad, D, sq, nsrc, isrc, surv_cnt, (long long)nq, st);
        if (rc != LS_OK) return rc;
        if (dst != src) {   // same centre for both sets: the candidates' first rows
            rc = knn_prep_launch(dst, src, Ns, B, dst_n, dst_npad, D, dq, ndst, idst, (int32_t*)nullptr, 0LL, st);
            if (rc != LS_OK) return rc;
        }
        static const bool fused_on = !(getenv("LS_KNN_FUSED") && atoi(getenv("LS_KNN_FUSED")) == 0);   // A/B: the multi-launch paths below
        if (fused_on && ns_pad <= KF_MAXNS && K <= 16 && Ns >= 1) {   // one kernel per (instance, 32 queries): see knn_fused_kernel   // one kernel per (instance, 32 queries): see knn_fused_kernel
            const int qtiles = cdiv(Nd, KF_QT), tpw = cdiv(ns_pad / 32, KF_WAVES);
            const float epsF = 1.02f * 0.0009765625f + 6.0f * (float)(D + 4) * 5.9604645e-8f + 9.5367431640625e-7f + 4.76837158203125e-7f;   // eps_b of the f16 sweep + 2^-21 (d^, lo, hi)
#define LS_KF(T) do { if (fma) hipLaunchKernelGGL((knn_fused_kernel<CC, true, T>), dim3(B * qtiles), dim3(64 * KF_WAVES), 0, st, dst, src, dst_rows, dq, sq, ndst, nsrc, idst, isrc, Nd, dst_n, dst_npad, Ns, ns_pad, K, qtiles, epsF, idx_out, dist_out, surv_cnt); \
                   else hipLaunchKernelGGL((knn_fused_kernel<CC, false, T>), dim3(B * qtiles), dim3(64 * KF_WAVES), 0, st, dst, src, dst_rows, dq, sq, ndst, nsrc, idst, isrc, Nd, dst_n, dst_npad, Ns, ns_pad, K, qtiles, epsF, idx_out, dist_out, surv_cnt); } while (0)
            if (tpw <= 1) LS_KF(1); else if (tpw == 2) LS_KF(2); else LS_KF(4);
#undef LS_KF
            LS_LAUNCH_CHECK();
            return LS_OK;
        }
        static const bool one_sweep_on = !(getenv("LS_KNN_ONE_SWEEP") && atoi(getenv("LS_KNN_ONE_SWEEP")) == 0);   // A/B: the two-sweep auto-hint path
        if (!seed_idx && one_sweep_on && ns_pad <= KO_MAXNS && K <= 16) {   // un-seeded, few candidates: one sweep + one finish (see above)
            unsigned short* img_end = dq + (size_t)B * dst_npad * D;
            char* wend = (char*)((((uintptr_t)img_end + 255) & ~(uintptr_t)255) + nq * 64 * (sizeof(float) + sizeof(int32_t)) + nq * 16 * sizeof(int32_t) + 256);
            short* cosq = (short*)(((uintptr_t)wend + 255) & ~(uintptr_t)255);
            const int qgroups = cdiv(Nd, 32);
            int nsplit = 1;
            while ((long long)B * qgroups * nsplit < 4096 && nsplit * 2 <= ns_pad / 32 && nsplit < 16) nsplit *= 2;
            const int total_waves = B * qgroups * nsplit;
            hipLaunchKernelGGL((knn_sweep_store_kernel<D>), dim3(cdiv(total_waves, 4)), dim3(256), 0, st, dq, sq, dst_rows, ndst, nsrc, idst, isrc, Nd, dst_n, dst_npad,
                               Ns, ns_pad, qgroups, nsplit, total_waves, cosq);
            LS_LAUNCH_CHECK();
            const float epsS = 1.02f * 0.0009765625f + 6.0f * (float)(D + 4) * 5.9604645e-8f + 6.103515625e-5f + 4.76837158203125e-7f;   // f16 image + 2^-14 (fixed point, this kernel) + 2^-21 (its 1-ulp square roots)
            const int wblocks = cdiv((long long)nq, 4);
            if (fma)
                hipLaunchKernelGGL((knn_finish_select_kernel<CC, true>), dim3(wblocks), dim3(256), 0, st, dst, src, dst_rows, ndst, nsrc, Nd, dst_n, Ns, ns_pad,
                                   K, epsS, cosq, idx_out, dist_out, (int)nq, surv_cnt);
            else
                hipLaunchKernelGGL((knn_finish_select_kernel<CC, false>), dim3(wblocks), dim3(256), 0, st, dst, src, dst_rows, ndst, nsrc, Nd, dst_n, Ns, ns_pad,
                                   K, epsS, cosq, idx_out, dist_out, (int)nq, surv_cnt);
            LS_LAUNCH_CHECK();
            return LS_OK;
        }
        if (!seed_idx) {   // un-seeded call: hints from a first sweep (class winners; the 16 best are picked inside the seed kernel)
            unsigned short* img_end = dq + (size_t)B * dst_npad * D;
            win_val = (float*)(((uintptr_t)img_end + 255) & ~(uintptr_t)255);
            win_idx = (int32_t*)(win_val + nq * 64);
            const int nsplit = ((long long)B * cdiv(Nd, 32) < 4096 && ns_pad / 32 >= 16) ? 2 : 1;   // W = 32 nsplit <= 64 winners per query
            win_w = 32 * nsplit;
            if (qg == 2) {
                const int qgroups = cdiv(Nd, 64), total_waves = B * qgroups * nsplit;
                hipLaunchKernelGGL((knn_sweep_winners_kernel<D, 2>), dim3(cdiv(total_waves, 4)), dim3(256), 0, st, dq, sq, dst_rows, nsrc, idst, isrc, dst_n, Nd, dst_npad,
                                   Ns, ns_pad, qgroups, nsplit, total_waves, win_val, win_idx);
            } else {
                const int qgroups = cdiv(Nd, 32), total_waves = B * qgroups * nsplit;
                hipLaunchKernelGGL((knn_sweep_winners_kernel<D, 1>), dim3(cdiv(total_waves, 4)), dim3(256), 0, st, dq, sq, dst_rows, nsrc, idst, isrc, dst_n, Nd, dst_npad,
                                   Ns, ns_pad, qgroups, nsplit, total_waves, win_val, win_idx);
            }
            LS_LAUNCH_CHECK();
            static const bool sel_launch = getenv("LS_KNN_SELECT_LAUNCH") && atoi(getenv("LS_KNN_SELECT_LAUNCH")) != 0;   // A/B: the separate select launch
            if (sel_launch) {
                int32_t* hints = win_idx + nq * 64;
                hipLaunchKernelGGL(knn_autohint_select_kernel, dim3(cdiv((long long)nq, 4)), dim3(256), 0, st, win_val, win_idx, win_w, (int)nq, hints);
                LS_LAUNCH_CHECK();
                seed_idx = hints;
                win_w = 0;
            }
            seed_n = Nd;
            seed_by_row = 0;
        }
    } else {
        LS_REQUIRE(seed_idx != nullptr, "knn_sweep: the fp32 sweep needs seed lists");
        rc = row_norms_launch(src, 3 * C, (long long)B * Ns, nsrc, st);
        if (rc != LS_OK) return rc;
        if (dst != src) {
            rc = row_norms_launch(dst, 3 * C, (long long)B * dst_n, ndst, st);
            if (rc != LS_OK) return rc;
        }
    }

    const int groups = cdiv(Nd, 4);                 // one wave per four queries
    const int gblocks = cdiv((long long)B * groups, 4);
    const int qtiles = cdiv(Nd, KNN_TQ);
    const float epsE = 6.0f * (float)(3 * C + 4) * 5.9604645e-8f;
    if (win_w) {
        if (fma)
            hipLaunchKernelGGL((knn_seed_kernel<CC, true, true>), dim3(gblocks), dim3(256), 0, st, dst, src, dst_rows, Nd, dst_n, Ns, K, seed_idx, seed_n,
                               seed_by_row, seedkeys, groups, B * groups, win_val, win_idx, win_w);
        else
            hipLaunchKernelGGL((knn_seed_kernel<CC, false, true>), dim3(gblocks), dim3(256), 0, st, dst, src, dst_rows, Nd, dst_n, Ns, K, seed_idx, seed_n,
                               seed_by_row, seedkeys, groups, B * groups, win_val, win_idx, win_w);
    } else if (fma)
        hipLaunchKernelGGL((knn_seed_kernel<CC, true>), dim3(gblocks), dim3(256), 0, st, dst, src, dst_rows, Nd, dst_n, Ns, K, seed_idx, seed_n,
                           seed_by_row, seedkeys, groups, B * groups, (const float*)nullptr, (const int32_t*)nullptr, 0);
    else
        hipLaunchKernelGGL((knn_seed_kernel<CC, false>), dim3(gblocks), dim3(256), 0, st, dst, src, dst_rows, Nd, dst_n, Ns, K, seed_idx, seed_n,
                           seed_by_row, seedkeys, groups, B * groups, (const float*)nullptr, (const int32_t*)nullptr, 0);
    LS_LAUNCH_CHECK();
    if (f16img) {
        const int nqw = 32 * qg;                    // queries per wave
        const int qgroups = cdiv(Nd, nqw);
        int nsplit = 1;   // >= ~4 waves per SIMD over the chip (4096 waves) when the query grid alone is smaller
        static const int wave_target = getenv("LS_KNN_SWEEP_WAVES") ? atoi(getenv("LS_KNN_SWEEP_WAVES")) : 4096;   // A/B
        while ((long long)B * qgroups * nsplit < wave_target && nsplit * 8 <= ns_pad / 32 && nsplit < 16) nsplit *= 2;   // >= 4 tiles per wave
        const int total_waves = B * qgroups * nsplit;
        const float epsB = 1.02f * 0.0009765625f + epsE + 9.5367431640625e-7f;   // f16 image: 2 u = 2^-10 (+ 2^-20: subnormal tails of a row)
        const size_t lds = (size_t)4 * (nqw * (ns_pad / 32) + nqw + nqw * (KB_CAPW / qg) / 2) * sizeof(unsigned);   // <= 52 KB (QG = 2: <= 50 KB)
        if (qg == 2)
            hipLaunchKernelGGL((knn_sweep_f16_kernel<D, 2>), dim3(cdiv(total_waves, 4)), dim3(256), lds, st, dq, sq, dst_rows, ndst, nsrc, idst, isrc, Nd, dst_n,
                               dst_npad, Ns, ns_pad, K, qgroups, nsplit, total_waves, epsB, seedkeys, surv_cnt, surv);
        else
            hipLaunchKernelGGL((knn_sweep_f16_kernel<D, 1>), dim3(cdiv(total_waves, 4)), dim3(256), lds, st, dq, sq, dst_rows, ndst, nsrc, idst, isrc, Nd, dst_n,
                               dst_npad, Ns, ns_pad, K, qgroups, nsplit, total_waves, epsB, seedkeys, surv_cnt, surv);
    } else {
        hipLaunchKernelGGL(knn_sweep_kernel<CC>, dim3(B * qtiles), dim3(256), 0, st, dst, src, dst_rows, ndst, nsrc, Nd, dst_n, Ns, K, qtiles,
                           epsE, seedkeys, surv_cnt, surv);
    }
    LS_LAUNCH_CHECK();
    static const bool finish_wave32 = getenv("LS_KNN_FINISH_WAVE32") && atoi(getenv("LS_KNN_FINISH_WAVE32")) != 0;   // A/B
    if (CC == 32 && !finish_wave32) {
        if (fma)
            hipLaunchKernelGGL((knn_finish_kernel<CC, true>), dim3(gblocks), dim3(256), 0, st, dst, src, dst_rows, Nd, dst_n, Ns, K, seedkeys,
                               surv_cnt, surv, idx_out, dist_out, groups, B * groups);
        else
            hipLaunchKernelGGL((knn_finish_kernel<CC, false>), dim3(gblocks), dim3(256), 0, st, dst, src, dst_rows, Nd, dst_n, Ns, K, seedkeys,
                               surv_cnt, surv, idx_out, dist_out, groups, B * groups);
    } else {
        const int wblocks = cdiv((long long)nq, 4);
        if (fma)
            hipLaunchKernelGGL((knn_finish_wave_kernel<CC, true>), dim3(wblocks), dim3(256), 0, st, dst, src, dst_rows, Nd, dst_n, Ns, K,
                               seedkeys, surv_cnt, surv, idx_out, dist_out, (int)nq);
        else
            hipLaunchKernelGGL((knn_finish_wave_kernel<CC, false>), dim3(wblocks), dim3(256), 0, st, dst, src, dst_rows, Nd, dst_n, Ns, K,
                               seedkeys, surv_cnt, surv, idx_out, dist_out, (int)nq);
    }
    LS_LAUNCH_CHECK();
    return LS_OK;
}

int knn_sweep_launch(const float* dst, const float* src, const int32_t* dst_rows, int B, int Nd, int dst_n, int Ns, int C, int K,
                     bool fma, int32_t* idx_out, float* dist_out, const int32_t* seed_idx, int seed_n, int seed_by_row, void* scratch,
                     hipStream_t st) {
    LS_REQUIRE(C == 32 || C == 64, "knn_sweep: only C == 32 / 64 layers are supported (C=%d)", C);
    LS_REQUIRE(Ns <= 65535, "knn_sweep: Ns=%d exceeds the 16-bit survivor index", Ns);
    if (C == 32)
        return knn_sweep_launch_t<32>(dst, src, dst_rows, B, Nd, dst_n, Ns, K, fma, idx_out, dist_out, seed_idx, seed_n, seed_by_row, scratch, st);
    return knn_sweep_launch_t<64>(dst, src, dst_rows, B, Nd, dst_n, Ns, K, fma, idx_out, dist_out, seed_idx, seed_n, seed_by_row, scratch, st);
}

// Exact-phase statistics of the LAST sweep-path call that used `scratch` (bench.py's hardware-utilisation roofline; profiled passes only):
// out[0] += sum over the queries of the candidates that were given a canonical distance BEYOND the hints (survivor lists; a query that
// overflowed its list scanned all Ns), out[1] += the number of queries.  The counters sit where knn_sweep_launch_t put them.
__global__ __launch_bounds__(256) void knn_stats_kernel(const int32_t* __restrict__ surv_cnt, long long nq, int Ns, unsigned long long* __restrict__ out) {
    unsigned long long s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long long)gridDim.x * 256) {
        const int c = surv_cnt[i];
        s += (unsigned long long)(c < 0 ? -(c + 1) : (c > KS_CAP ? Ns : c));      // negative: the one-sweep path's raw count (knn_finish_select_kernel)
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[0], s);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&out[1], (unsigned long long)nq);
}
int knn_sweep_stats_launch(const void* scratch, int B, int Nd, int dst_n, int Ns, unsigned long long* out, hipStream_t st) {
    const size_t nq = (size_t)B * Nd;
    const size_t off = knn_sweep_surv_cnt_offset(B, Ns, dst_n, nq);
    hipLaunchKernelGGL(knn_stats_kernel, dim3((unsigned)std::min<size_t>(cdiv((long long)nq, 256), 256)), dim3(256), 0, st,
                       (const int32_t*)((const char*)scratch + off), (long long)nq, Ns, out);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

int row_norms_launch(const float* f, int row_f, long long npts, float* norms, hipStream_t st) {
    LS_REQUIRE(row_f % 4 == 0, "row_norms: row length must be a multiple of 4");
    hipLaunchKernelGGL(row_norms_kernel, dim3(cdiv(npts, 4)), dim3(256), 0, st, f, row_f, npts, norms);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

}  // namespace ls
