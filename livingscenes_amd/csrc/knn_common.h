// knn_common.h -- pieces shared by the two k-NN kernels (knn.hip: all-VALU; knn_mfma.hip: MFMA-filtered)
#pragma once
#include "ls_common.h"

namespace ls {

constexpr int KNN_TQ = 64;   // queries per workgroup
constexpr int KNN_TS = 64;   // candidates per tile
constexpr int KNN_MAXK = 16;
constexpr int KNN_LD = KNN_TS + 4;  // distance-tile row stride (floats), multiple of 4 for 16-byte rows

template <bool FMA>
__device__ __forceinline__ float accq(float d, float a, float b) {
#pragma clang fp contract(off)
    const float diff = a - b;
    if constexpr (FMA) {
        return __builtin_fmaf(diff, diff, d);
    } else {
        const float p = diff * diff;
        return d + p;
    }
}

typedef unsigned long long u64;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// two canonical accumulations at once (two candidates, same query value): maps onto v_pk_add / v_pk_mul (/ v_pk_fma)
template <bool FMA>
__device__ __forceinline__ f32x2 accq2(f32x2 d, float q, f32x2 c) {
#pragma clang fp contract(off)
    const f32x2 qq = {q, q};
    const f32x2 diff = qq - c;
    if constexpr (FMA) {
        return __builtin_elementwise_fma(diff, diff, d);
    } else {
        const f32x2 p = diff * diff;
        return d + p;
    }
}

// (dist >= 0, idx) -> sortable key; invalid candidates get the "empty" key ~0
__device__ __forceinline__ u64 make_key(float d, int idx, bool valid) {
    const u64 k = ((u64)__float_as_uint(d) << 32) | (unsigned)idx;
    return valid ? k : ~0ull;
}
__device__ __forceinline__ void key_cx(u64& a, u64& b) {  // compare-exchange: a <= b afterwards
    const bool sw = b < a;
    const u64 lo = sw ? b : a, hi = sw ? a : b;
    a = lo; b = hi;
}
__device__ __forceinline__ u64 bperm64(int srclane, u64 v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(srclane << 2, (int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(srclane << 2, (int)(unsigned)(v >> 32));
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 dpp_quad_bcast3(u64 v) {   // every lane of a quad receives the value of the quad's lane 3 (quad_perm [3,3,3,3])
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, 0xFF, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), 0xFF, 0xF, 0xF, true);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 dpp_row_shr1(u64 v) {  // lane e of a 16-lane row receives lane e-1's value (lane 0: 0)
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, 0x111, 0xF, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), 0x111, 0xF, 0xF, false);
    return ((u64)hi << 32) | lo;
}

// Stage one 64-row x CC-channel chunk: global rows are x-major ([x][c], c contiguous -> coalesced float4 loads); the LDS
// image is CANONICAL (dim j = c*3 + x contiguous), so that one ds_read_b128 yields four consecutive summation terms.
template <int CC>
struct Stager {
    // thread -> (row rr = tid>>2, quarter q = tid&3); its float4 number u covers x = u / U2, channels (u % U2)*16 + q*4..+3
    // (the four threads of a row read 64 contiguous bytes per load; LDS offsets are compile-time per u)
    static constexpr int U2 = CC / 16;                 // float4-quads per xyz component (2 for CC = 32)
    static constexpr int PER = 3 * U2;                 // float4 per thread (6)
    float4 r[PER];
    __device__ __forceinline__ void load(const float* __restrict__ base, const int* rowmap, int row0, int nrows, size_t row_f,
                                         int C, int c0, int tid) {
        const int rr = tid >> 2, q = tid & 3;
        const int gr = rowmap ? rowmap[rr] : ((row0 + rr) < nrows ? row0 + rr : -1);
        const float* p = base + (size_t)(gr >= 0 ? gr : 0) * row_f + c0 + q * 4;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const float4 v = *reinterpret_cast<const float4*>(p + (size_t)(u / U2) * C + (u % U2) * 16);
            r[u] = gr >= 0 ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // candidate-PAIR interleaved image: rows 2m and 2m+1 share one LDS row of stride PR, element (dim dd, parity h) at
    // dd*2 + h, so that one ds_read_b128 yields (c0.d, c1.d, c0.d+1, c1.d+1): natural operand pairs for v_pk_* math.
    __device__ __forceinline__ void store_pairs(float* lds, int PR, int tid) const {
        const int rr = tid >> 2, q = tid & 3;
        float* p0 = lds + (rr >> 1) * PR + (rr & 1) + 2 * (q * 12);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            float* p = p0 + 2 * ((u % U2) * 48 + (u / U2));
            p[0] = r[u].x; p[6] = r[u].y; p[12] = r[u].z; p[18] = r[u].w;
        }
    }
    __device__ __forceinline__ void store(float* lds, int ROW, int tid) const {
        const int rr = tid >> 2, q = tid & 3;
        float* p0 = lds + rr * ROW + q * 12;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            float* p = p0 + (u % U2) * 48 + (u / U2);
            p[0] = r[u].x; p[3] = r[u].y; p[6] = r[u].z; p[9] = r[u].w;
        }
    }
};


// One row-parallel merge: every lane offers up to four keys (k0 <= k1 <= k2 <= k3, ~0 = none); while any lane's smallest
// pending key beats its row's K-th key: ballot -> first proposing lane per row -> bpermute its key to the row -> ballot of
// "entry < candidate" gives the insert position -> DPP row_shr:1 shifts the tail.  Four queries (one per 16-lane row)
// advance per wave instruction; the only control flow is the wave-uniform loop exit.  DEDUP: a candidate whose key equals
// an existing entry (same index, same canonical distance: a seed met again by the scan) is dropped.
template <bool DEDUP>
__device__ __forceinline__ void merge_keys(u64 k0, u64 k1, u64 k2, u64 k3, u64& lkg, u64& kk, int K, int lane) {
    const int e16 = lane & 15, rowbase = lane & 48;
    u64 m64 = __ballot(k0 < kk);
    while (m64) {
        const unsigned rb = (unsigned)(m64 >> rowbase) & 0xFFFFu;
        bool rowhas = rb != 0;
        const int srclane = rowbase + __builtin_ctz(rb | 0x10000u);  // rowbase+16 (no lane of this row) if !rowhas
        const u64 cand = bperm64(srclane, k0);
        const bool is_src = lane == srclane;
        k0 = is_src ? k1 : k0; k1 = is_src ? k2 : k1; k2 = is_src ? k3 : k2; k3 = is_src ? ~0ull : k3;
        if constexpr (DEDUP) {
            const u64 d64 = __ballot(rowhas & (lkg == cand));
            rowhas = rowhas & (((unsigned)(d64 >> rowbase) & 0xFFFFu) == 0u);
        }
        const u64 l64 = __ballot(rowhas & (lkg < cand));
        const int pos = __builtin_popcount((unsigned)(l64 >> rowbase) & 0xFFFFu);
        const u64 up = dpp_row_shr1(lkg);
        const bool s1 = rowhas & (e16 == pos), s2 = rowhas & (e16 > pos);
        lkg = s1 ? cand : (s2 ? up : lkg);
        kk = bperm64(rowbase + K - 1, lkg);
        m64 = __ballot(k0 < kk);
    }
}

// Merge one 64x64 distance tile (LDS, slot layout (c & 15) * 4 + (c >> 4)) into the wave's top-K key lists.  (dist, idx)
// pairs are packed into one u64 key (non-negative float bits << 32 | idx) so the lexicographic order is a single unsigned
// compare.  The 16-lane row r of group g owns query wave*16 + g*4 + r; its sorted top-K keys live in the row's lanes
// (entry e in lane 16r+e).  Each lane sorts its 4 candidate keys once (c = e + 16j), then merge_keys.
template <bool DEDUP>
__device__ __forceinline__ void select_tile(const float* ldist, u64 (&lk)[4], u64 (&rkey)[4], int s0, int Ns, int K, int wave,
                                            int lane) {
    const int e16 = lane & 15;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int qrow = wave * 16 + g * 4 + (lane >> 4);
        const float4 dv = *reinterpret_cast<const float4*>(&ldist[qrow * KNN_LD + e16 * 4]);
        const int cbase = s0 + e16;
        u64 k0 = make_key(dv.x, cbase, cbase < Ns), k1 = make_key(dv.y, cbase + 16, cbase + 16 < Ns);
        u64 k2 = make_key(dv.z, cbase + 32, cbase + 32 < Ns), k3 = make_key(dv.w, cbase + 48, cbase + 48 < Ns);
        key_cx(k0, k1); key_cx(k2, k3); key_cx(k0, k2); key_cx(k1, k3); key_cx(k1, k2);
        merge_keys<DEDUP>(k0, k1, k2, k3, lk[g], rkey[g], K, lane);
    }
}

// Seed the wave's lists from lseed[64][16] (one key per (query, hint), ~0 = none); repeated hints are dropped.
__device__ __forceinline__ void seed_lists(const u64* lseed, u64 (&lk)[4], u64 (&rkey)[4], int K, int wave, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int qrow = wave * 16 + g * 4 + (lane >> 4);
        merge_keys<true>(lseed[qrow * 16 + (lane & 15)], ~0ull, ~0ull, ~0ull, lk[g], rkey[g], K, lane);
    }
}

// canonical distance between two feature rows in global memory (x-major [3][C]), same chain as the tile path
template <bool FMA>
__device__ __forceinline__ float row_distance(const float* __restrict__ a, const float* __restrict__ b, int C) {
    float d = 0.0f;
    for (int c4 = 0; c4 < C; c4 += 4) {
        const float4 ax = *reinterpret_cast<const float4*>(a + c4), ay = *reinterpret_cast<const float4*>(a + C + c4),
                     az = *reinterpret_cast<const float4*>(a + 2 * C + c4);
        const float4 bx = *reinterpret_cast<const float4*>(b + c4), by = *reinterpret_cast<const float4*>(b + C + c4),
                     bz = *reinterpret_cast<const float4*>(b + 2 * C + c4);
        d = accq<FMA>(d, ax.x, bx.x); d = accq<FMA>(d, ay.x, by.x); d = accq<FMA>(d, az.x, bz.x);
        d = accq<FMA>(d, ax.y, bx.y); d = accq<FMA>(d, ay.y, by.y); d = accq<FMA>(d, az.y, bz.y);
        d = accq<FMA>(d, ax.z, bx.z); d = accq<FMA>(d, ay.z, by.z); d = accq<FMA>(d, az.z, bz.z);
        d = accq<FMA>(d, ax.w, bx.w); d = accq<FMA>(d, ay.w, by.w); d = accq<FMA>(d, az.w, bz.w);
    }
    return d;
}

// Compute the seed keys of a 64-query workgroup into lseed[64][16]: hint e of query q is seed_idx[(b*seed_n + row)*16 + e]
// with row = the query's source row (seed_by_row, the previous layer's list of the same point) or its query index.
template <bool FMA>
__device__ __forceinline__ void compute_seed_keys(u64* lseed, const int32_t* __restrict__ seed_idx, int seed_n, bool seed_by_row,
                                                  const float* __restrict__ dbase, const float* __restrict__ sbase, const int* lqrow,
                                                  int b, int q0, int Ns, int C, int tid) {
    const size_t row_f = (size_t)3 * C;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int p = tid + u * 256;
        const int q = p >> 4, e = p & 15;
        const int r = lqrow[q];
        int sidx = -1;
        if (r >= 0) sidx = seed_idx[((size_t)b * seed_n + (seed_by_row ? r : q0 + q)) * 16 + e];
        u64 key = ~0ull;
        if (sidx >= 0 && sidx < Ns) key = make_key(row_distance<FMA>(dbase + (size_t)r * row_f, sbase + (size_t)sidx * row_f, C), sidx, true);
        lseed[q * 16 + e] = key;
    }
}

// write one wave's 16 sorted lists (final result, or a per-split partial list for the merge kernel)
__device__ __forceinline__ void write_lists(const u64 (&lk)[4], int b, int q0, int Nd, int K, int wave, int lane, int splits, int sp,
                                            u64* __restrict__ partial, int32_t* __restrict__ idx_out, float* __restrict__ dist_out) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int q = q0 + wave * 16 + g * 4 + (lane >> 4);
        const int e = lane & 15;
        if (q < Nd && splits > 1) {
            partial[(((size_t)b * Nd + q) * splits + sp) * 16 + e] = e < K ? lk[g] : ~0ull;
        } else if (q < Nd && e < K) {
            const size_t o = ((size_t)b * Nd + q) * K + e;
            const unsigned hi = (unsigned)(lk[g] >> 32), lo = (unsigned)lk[g];
            idx_out[o] = hi == 0xFFFFFFFFu ? -1 : (int)lo;
            if (dist_out) dist_out[o] = hi == 0xFFFFFFFFu ? INFINITY : __uint_as_float(hi);
        }
    }
}

// ---- 64-lane sorting network (one value per lane)
template <int CTRL>
__device__ __forceinline__ unsigned dpp_mov(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// value of lane (lane ^ M)
template <int M>
__device__ __forceinline__ unsigned lane_xor(unsigned v, int lane) {
    if constexpr (M == 1) return dpp_mov<0xB1>(v);          // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
    else if constexpr (M == 3) return dpp_mov<0x1B>(v);     // quad_perm [3,2,1,0]
    else if constexpr (M == 7) return dpp_mov<0x141>(v);    // row_half_mirror
    else if constexpr (M == 15) return dpp_mov<0x140>(v);   // row_mirror
    else if constexpr (M < 32) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (M << 10) | 0x1F);   // bit mode: lane ^ M
    else return (unsigned)__builtin_amdgcn_ds_bpermute((lane ^ M) << 2, (int)v);
}
template <int M>
__device__ __forceinline__ constexpr int pair_bit() { return (M & (M + 1)) == 0 ? (M + 1) / 2 : M; }   // the lower lane of a pair has this bit clear

template <int M>
__device__ __forceinline__ void cx32(unsigned& v, int lane) {
    const unsigned o = lane_xor<M>(v, lane);
    v = (lane & pair_bit<M>()) == 0 ? min(v, o) : max(v, o);
}
template <int M>
__device__ __forceinline__ void cx64(u64& v, int lane) {
    const u64 o = ((u64)lane_xor<M>((unsigned)(v >> 32), lane) << 32) | lane_xor<M>((unsigned)v, lane);
    const bool lt = o < v, takemin = (lane & pair_bit<M>()) == 0;
    v = (takemin == lt) ? o : v;
}
// ascending sort of one value per lane over the 64 lanes: bitonic network in the "flip" form (first step of every merge
// pairs lane i with lane i ^ (k-1), the rest are plain half-cleaners), so that every exchange is a lane-xor and 13 of the 21
// stages are DPP moves
// An upper bound of the K-th smallest of the wave's 64 values v (bit patterns of non-negative floats, +inf = 0x7F800000 for "none"): the
// smallest T whose low 15 bits are ones with at least K values <= T -- a radix select on bits 30 .. 15, one v_cmp per bit, the counting on
// the scalar unit (ballot, s_bcnt1) -- at most 2^-8 above the K-th smallest itself.  The admission thresholds of knn_xyz_kernel and
// knn_finish_select_kernel only have to be VALID (>= the K-th smallest) and tight; until round 4 they sorted the 64 values with the 21-stage
// network below and read lane K - 1: ~125 VALU instructions per query against 16 here, in kernels that are VALU-issue-bound.
__device__ __forceinline__ unsigned kth_smallest_upper_bound(unsigned v, int K) {
    unsigned ans = 0;
#pragma unroll
    for (int bit = 30; bit >= 15; --bit) {
        const unsigned t = ans | ((1u << bit) - 1u);
        const int cnt = __builtin_popcountll(__ballot(v <= t));
        ans |= cnt >= K ? 0u : (1u << bit);
    }
    return min(ans | 0x7FFFu, 0x7F800000u);
}

// the first ten exchanges of the 64-lane network: every aligned group of 16 lanes sorted ascending (DPP / ds_swizzle only)
#define LS_SORT16(CX, v, lane)                                                                                         \
    CX<1>(v, lane);                                                                                                    \
    CX<3>(v, lane); CX<1>(v, lane);                                                                                    \
    CX<7>(v, lane); CX<2>(v, lane); CX<1>(v, lane);                                                                    \
    CX<15>(v, lane); CX<4>(v, lane); CX<2>(v, lane); CX<1>(v, lane);

// the first fifteen: every aligned group of 32 lanes sorted ascending
#define LS_SORT32(CX, v, lane)                                                                                         \
    LS_SORT16(CX, v, lane)                                                                                             \
    CX<31>(v, lane); CX<8>(v, lane); CX<4>(v, lane); CX<2>(v, lane); CX<1>(v, lane);

#define LS_SORT64(CX, v, lane)                                                                                         \
    CX<1>(v, lane);                                                                                                    \
    CX<3>(v, lane); CX<1>(v, lane);                                                                                    \
    CX<7>(v, lane); CX<2>(v, lane); CX<1>(v, lane);                                                                    \
    CX<15>(v, lane); CX<4>(v, lane); CX<2>(v, lane); CX<1>(v, lane);                                                   \
    CX<31>(v, lane); CX<8>(v, lane); CX<4>(v, lane); CX<2>(v, lane); CX<1>(v, lane);                                   \
    CX<63>(v, lane); CX<16>(v, lane); CX<8>(v, lane); CX<4>(v, lane); CX<2>(v, lane); CX<1>(v, lane);


}  // namespace ls
