// knn_mfma.hip -- the same bit-exact 16-NN as knn.hip for the SEEDED C == 32 encoder layers, with the distance sweep moved
// onto the matrix cores as an EXACT-SAFE FILTER.
//
// Replaces pytorch3d.ops.knn_points as called at
//   /root/reference/lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:139-141        (layers whose input has 32 channels)
//
// knn.hip spends ~250 M VALU instructions per layer-1 launch on the canonical (sub, mul, add) distance of EVERY pair, although
// once a query's list holds its previous-layer neighbours only ~25 of 1024 candidates can still enter it.  Here:
//   1. S = q . s on the matrix cores (v_mfma_f32_32x32x2_f32, K = 3C), giving d^ = |q|^2 + |s|^2 - 2S;
//   2. a pair is DROPPED only if  d^ - eps > kth(q)  (kth = the query's current exact K-th distance), where
//      eps = 6 (D+4) 2^-24 (|q|^2 + |s|^2) bounds |d^ - d_true| + |d_canonical - d_true| with 50 % slack
//      (gamma_{D+3} (|q|+|s|)^2 each, (|q|+|s|)^2 <= 2 (|q|^2+|s|^2)); fp32 accumulation of non-negative terms is
//      monotone, so a dropped pair provably has canonical distance > kth and could never have been inserted;
//   3. the survivors get the CANONICAL distance (same fp32 chain as knn.hip / the oracle) and go through the same
//      row-parallel key merge (knn_common.h).
// The top-K lists only ever hold canonical distances, so the result is bit-identical to knn.hip / the oracle by
// construction; the filter only decides what is worth computing.
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
// "sweep" kernel (C == 32 layers, seeded, un-split): the candidate sweep is a pure MFMA GEMM with a filtering epilogue.
//   * lists are seeded from the previous layer's graph (knn_common.h), so each query's admission threshold kth (its exact
//     K-th canonical distance so far) is near-final before the sweep starts: ~25 of 1024 candidates pass;
//   * the query fragments of S = q . s stay in registers (48 VGPRs), candidates stream through LDS in 64-row x 32-dim
//     chunks (double buffered, one barrier per chunk), three chunks = one 64x64 tile of S;
//   * the epilogue of a tile appends the candidates that pass  d^ - eps <= kth  to a per-query survivor buffer (u16
//     indices, KS_CAP slots) -- no canonical arithmetic, no sorting inside the sweep;
//   * survivors get their CANONICAL distance (quad_pair_distance, rows re-read from L2) and are merged into the row-parallel
//     lists only when a buffer could overflow during the next tile (rare; tightens kth) and once at the end.
// Exactness: a dropped pair has canonical distance > kth >= final K-th distance; everything else is decided by canonical
// keys.  Worst case (all-equal points, repeated hints) every tile flushes: slow but still exact.
constexpr int KS_CAP = 128;   // survivor slots per query; a flush is forced once a buffer holds more than KS_CAP - 64
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
template <bool FMA>
__device__ __forceinline__ float quad_pair_distance(const float* __restrict__ qrow, const float* __restrict__ crow, int lane) {
#pragma clang fp contract(off)
    constexpr int C = KM_CC;
    const int off = (lane & 3) * 8;
    float4 qv[6], cv[6];
#pragma unroll
    for (int x = 0; x < 3; ++x)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            qv[x * 2 + h] = *reinterpret_cast<const float4*>(qrow + x * C + off + 4 * h);
            cv[x * 2 + h] = *reinterpret_cast<const float4*>(crow + x * C + off + 4 * h);
        }
    float t[24];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            const float4 a = qv[x * 2 + h], b = cv[x * 2 + h];
            const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
            t[(h * 4 + 0) * 3 + x] = FMA ? d0 : d0 * d0;
            t[(h * 4 + 1) * 3 + x] = FMA ? d1 : d1 * d1;
            t[(h * 4 + 2) * 3 + x] = FMA ? d2 : d2 * d2;
            t[(h * 4 + 3) * 3 + x] = FMA ? d3 : d3 * d3;
        }
    float d = 0.0f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int j = 0; j < 24; ++j) d = FMA ? __builtin_fmaf(t[j], t[j], d) : d + t[j];
        if (s < 3)   // lane i <- lane i-1 within the quad (quad_perm [0,0,1,2])
            d = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), 0x90, 0xF, 0xF, false));
    }
    return d;
}

template <bool FMA>
__global__ __launch_bounds__(256, 3) void knn_sweep_kernel(const float* __restrict__ dstf, const float* __restrict__ srcf,
                                                           const int32_t* __restrict__ dst_rows, const float* __restrict__ nrm_dst,
                                                           const float* __restrict__ nrm_src, int Nd, int dst_n, int Ns, int K,
                                                           int32_t* __restrict__ idx_out, float* __restrict__ dist_out, int qtiles,
                                                           float epsE, const int32_t* __restrict__ seed_idx, int seed_n, int seed_by_row) {
    constexpr int C = KM_CC, RF = 3 * KM_CC;
    __shared__ __attribute__((aligned(16))) float lc[2][KNN_TS * KS_LD];              // 18 KB: two candidate chunks
    __shared__ __attribute__((aligned(16))) unsigned short lbuf[KNN_TQ * KS_CAP];     // 16 KB: survivor indices
    __shared__ __attribute__((aligned(16))) unsigned short lseedidx[KNN_TQ * 16];     // 2 KB: the hints (0xFFFF = none)
    __shared__ float lnq[KNN_TQ], lkth[KNN_TQ];
    __shared__ int lqrow[KNN_TQ], lcnt[KNN_TQ];
    __shared__ int lflush;

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
        lkth[tid] = r >= 0 ? INFINITY : -INFINITY;  // padding queries never pass the filter
        lcnt[tid] = 0;
    }
    if (tid == 0) lflush = 0;
    __syncthreads();

    const int wm = wave >> 1, wn = wave & 1;            // MFMA tile: queries wm*32.., candidates wn*32..
    const int l31 = lane & 31, lh = lane >> 5;

    u64 lk[4], rkey[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { lk[i] = ~0ull; rkey[i] = ~0ull; }

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
    // Quad-cooperative exact phase (quad_pair_distance): in group g the 16-lane row r of the wave belongs to query
    // wave*16 + g*4 + r; its four quads take four candidates per step, four steps feed one merge_keys call (<= 16 keys a row).
    const int quad = (lane >> 2) & 3;              // quad index inside the 16-lane row
    const bool qlast = (lane & 3) == 3;            // the lane of a quad that ends up with the distance
    // canonical keys for the buffered survivors of this wave's 16 queries -> merged into the lists; buffers emptied
    auto flush_own = [&]() {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int qr = wave * 16 + g * 4 + (lane >> 4);
            const int cnt = lcnt[qr];
            const int r = lqrow[qr];
            const float* qp = dbase + (size_t)(r >= 0 ? r : 0) * RF;
            for (int base = 0; __any(base < cnt); base += 16) {
                u64 ks[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = base + u * 4 + quad;
                    const bool v = j < cnt;
                    ks[u] = ~0ull;
                    if (__any(v)) {   // wave-uniform: skip steps no row needs
                        const int c = v ? (int)lbuf[qr * KS_CAP + j] : 0;
                        const float d = quad_pair_distance<FMA>(qp, sbase + (size_t)c * RF, lane);
                        ks[u] = make_key(d, c, v & qlast);
                    }
                }
                key_cx(ks[0], ks[1]); key_cx(ks[2], ks[3]); key_cx(ks[0], ks[2]); key_cx(ks[1], ks[3]); key_cx(ks[1], ks[2]);
                merge_keys<true>(ks[0], ks[1], ks[2], ks[3], lk[g], rkey[g], K, lane);
            }
            if ((lane & 15) == 0) lcnt[qr] = 0;
        }
        refresh_kth();
    };

    {   // seed the lists with the exact keys of the hints; remember the hint indices (they are not buffered again)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int qr = wave * 16 + g * 4 + (lane >> 4);
            const int r = lqrow[qr];
            const float* qp = dbase + (size_t)(r >= 0 ? r : 0) * RF;
            const int32_t* hp = seed_idx + ((size_t)b * seed_n + (seed_by_row ? (r >= 0 ? r : 0) : min(q0 + qr, seed_n - 1))) * 16;
            u64 ks[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = u * 4 + quad;
                const int sidx = hp[e];
                const bool v = r >= 0 && sidx >= 0 && sidx < Ns;
                const int c = v ? sidx : 0;
                const float d = quad_pair_distance<FMA>(qp, sbase + (size_t)c * RF, lane);
                ks[u] = make_key(d, c, v & qlast);
                if (qlast) lseedidx[qr * 16 + e] = v ? (unsigned short)c : (unsigned short)0xFFFFu;
            }
            key_cx(ks[0], ks[1]); key_cx(ks[2], ks[3]); key_cx(ks[0], ks[2]); key_cx(ks[1], ks[3]); key_cx(ks[1], ks[2]);
            merge_keys<true>(ks[0], ks[1], ks[2], ks[3], lk[g], rkey[g], K, lane);
        }
        refresh_kth();
    }

    // loop-invariant A fragments: query row wm*32 + l31, dims k*32 + j*8 + lh*4 .. +3
    float4 a[12];
    float nqv[16];
    auto load_query_frags = [&]() {   // (re)loaded after a mid-sweep flush so that nothing has to live across it
        const int r = lqrow[wm * 32 + l31];
        const float* qp = dbase + (size_t)(r >= 0 ? r : 0) * RF + lh * 4;
#pragma unroll
        for (int i = 0; i < 12; ++i) a[i] = *reinterpret_cast<const float4*>(qp + i * 8);  // padding queries: row 0, kth = -inf
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) nqv[r2] = lnq[wm * 32 + (r2 & 3) + 8 * (r2 >> 2) + 4 * lh];
    };
    load_query_frags();

    // chunk staging: chunk g = (tile g/3, xyz component g%3); thread -> rows sr and sr+32, float4 column sc4
    const int sr = tid >> 3, sc4 = (tid & 7) * 4;
    const int ntiles = (Ns + KNN_TS - 1) / KNN_TS;
    float4 st0, st1;
    auto gload = [&](int t, int k) {
        const int r0 = t * KNN_TS + sr, r1 = r0 + 32;
        // rows past Ns are clamped, not zeroed: their S column is garbage but `cvalid` keeps it out of the filter, and a
        // select on the loaded value would make the wave wait for the load right here instead of one chunk later
        st0 = *reinterpret_cast<const float4*>(sbase + (size_t)min(r0, Ns - 1) * RF + k * C + sc4);
        st1 = *reinterpret_cast<const float4*>(sbase + (size_t)min(r1, Ns - 1) * RF + k * C + sc4);
    };
    auto lstore = [&](int buf) {
        *reinterpret_cast<float4*>(&lc[buf][sr * KS_LD + sc4]) = st0;
        *reinterpret_cast<float4*>(&lc[buf][(sr + 32) * KS_LD + sc4]) = st1;
    };

    __syncthreads();  // lkth and lseedidx visible to all waves
    gload(0, 0);
    lstore(0);
    gload(0, 1);
    __syncthreads();

    const int cc = wn * 32 + l31;
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
        for (int k = 0; k < 3; ++k) {
            const int g = t * 3 + k, buf = (t + k) & 1;
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
            if (k == 2) {
                // filter epilogue: candidate t*64 + cc against the 16 queries of this lane's accumulator rows
                const int cglob = t * KNN_TS + cc;
                const bool cvalid = cglob < Ns;
                asm volatile("" : "+v"(nsv));   // keep the norm's first use (and its s_waitcnt) down here, two chunks after the load
                unsigned mask = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const float nn = nqv[r] + nsv;
                    const float dh = nn - 2.0f * S[r];
                    const bool pass = cvalid & !((dh - epsE * nn) > lkth[qr]);
                    mask |= pass ? (1u << r) : 0u;
                }
                while (mask) {
                    const int r = __builtin_ctz(mask);
                    mask &= mask - 1;
                    const int qr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    // a hint's exact key is in the list already: do not buffer it again
                    const uint4 h0 = *reinterpret_cast<const uint4*>(&lseedidx[qr * 16]);
                    const uint4 h1 = *reinterpret_cast<const uint4*>(&lseedidx[qr * 16 + 8]);
                    const unsigned cg = (unsigned)cglob;
                    auto has = [&](unsigned w) { return ((w & 0xFFFFu) == cg) | ((w >> 16) == cg); };
                    const bool hinted = has(h0.x) | has(h0.y) | has(h0.z) | has(h0.w) | has(h1.x) | has(h1.y) | has(h1.z) | has(h1.w);
                    if (!hinted) {
                        const int pos = atomicAdd(&lcnt[qr], 1);
                        if (pos < KS_CAP) lbuf[qr * KS_CAP + pos] = (unsigned short)cglob;
                        if (pos >= KS_CAP - KNN_TS) lflush = 1;
                    }
                }
            }
            // unconditional (clamped) stores / loads: with branches around them the compiler cannot count the loads in
            // flight and falls back to s_waitcnt vmcnt(0) in front of the MFMA block
            lstore(buf ^ 1);
            {
                const int t2 = k == 0 ? t : t + 1;          // chunk g + 2
                gload(min(t2, ntiles - 1), (k + 2) % 3);
            }
            __syncthreads();
        }
        if (lflush) {   // block-uniform (read after the barrier): some buffer could overflow during the next tile
            flush_own();
            __syncthreads();
            if (tid == 0) lflush = 0;
            load_query_frags();
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): leave this rare branch with nothing in flight, so that the
                                                 // common path keeps exact load counts at the join
        }
    }
    flush_own();
    write_lists(lk, b, q0, Nd, K, wave, lane, 1, 0, nullptr, idx_out, dist_out);
}

int knn_sweep_launch(const float* dst, const float* src, const int32_t* dst_rows, const float* nrm_dst, const float* nrm_src, int B,
                     int Nd, int dst_n, int Ns, int C, int K, bool fma, int32_t* idx_out, float* dist_out, const int32_t* seed_idx,
                     int seed_n, int seed_by_row, hipStream_t st) {
    LS_REQUIRE(C == KM_CC, "knn_sweep: only C == 32 layers are supported (C=%d)", C);
    LS_REQUIRE(seed_idx != nullptr, "knn_sweep: needs seed lists");
    LS_REQUIRE(Ns <= 65535, "knn_sweep: Ns=%d exceeds the 16-bit survivor index", Ns);
    const int qtiles = cdiv(Nd, KNN_TQ);
    const float epsE = 6.0f * (float)(3 * C + 4) * 5.9604645e-8f;
    dim3 grid(B * qtiles), block(256);
    if (fma)
        hipLaunchKernelGGL(knn_sweep_kernel<true>, grid, block, 0, st, dst, src, dst_rows, nrm_dst, nrm_src, Nd, dst_n, Ns, K, idx_out,
                           dist_out, qtiles, epsE, seed_idx, seed_n, seed_by_row);
    else
        hipLaunchKernelGGL(knn_sweep_kernel<false>, grid, block, 0, st, dst, src, dst_rows, nrm_dst, nrm_src, Nd, dst_n, Ns, K, idx_out,
                           dist_out, qtiles, epsE, seed_idx, seed_n, seed_by_row);
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
