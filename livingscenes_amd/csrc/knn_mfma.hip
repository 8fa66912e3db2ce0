// knn_mfma.hip -- the same bit-exact 16-NN as knn.hip for the C == 32 / 64 encoder layers (1 - 4, up to 1024 candidates), with the distance sweep
// on the matrix cores as an EXACT-SAFE FILTER and everything else of a build in ONE kernel (knn_fused_kernel, round 5).
//
// Replaces pytorch3d.ops.knn_points as called at
//   /root/reference/lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:139-141        (layers whose input has 32 or 64 channels)
//
// knn.hip spends ~250 M VALU instructions per layer-1 launch on the canonical (sub, mul, add) distance of EVERY pair, although only ~18 of 1024
// candidates per query can be among the 16 nearest once approximate distances are known.  Here:
//   1. S = q . s on the f16 matrix cores over centred, row-scaled rows (the "image", knn_prep_f16_tile_kernel; bound eps_b below), giving
//      d^ = |q'|^2 + |s'|^2 - 2 S with  lo = d^ - eps nn <= d_canonical <= d^ + eps nn = hi,  nn = |q'|^2 + |s'|^2;
//   2. T = the K-th smallest of the minima of hi over disjoint candidate groups bounds the K-th canonical distance from above; a pair is DROPPED
//      only if lo > T -- fp32 accumulation of non-negative terms is monotone, so a dropped pair provably has a canonical distance above the K-th;
//   3. the survivors get the CANONICAL distance (same fp32 chain as knn.hip / the oracle) and the K smallest (distance, index) keys are the list.
// The lists only ever hold canonical distances, so the result is bit-identical to knn.hip / the oracle by construction; the filter only
// decides what is worth computing.  History: rounds 2 - 4 ran this as four to five launches per layer (image, seed distances of the previous
// layer's lists as thresholds, sweep, finish; or image, cosine store, select) that handed seed keys, survivor lists and pair cosines to each
// other through HBM (150 MB moved for 54 MB compulsory at layer 1); they are gone (docs/history.md).
#include "knn_common.h"

namespace ls {

typedef float f32x16 __attribute__((ext_vector_type(16)));


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
constexpr int KNN_CENTRE_ROWS = 16;
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
#ifndef LS_KF_PK
#define LS_KF_PK 1
#endif
constexpr int KF_QT = 32;        // queries per workgroup (one 32-row MFMA operand)
constexpr int KF_WAVES = 8;
constexpr int KF_LCAP = 64;      // survivors per query that go through the flat exact phase (one 64-lane sorting pass)
constexpr int KF_MAXNS = KF_WAVES * 4 * 32;   // at most four tiles per wave

template <int CC>
struct KfLds {
    float qrows[KF_QT][3 * CC];          // the workgroup's query rows (fp32, x-major as in global memory)
    float qnx[KF_QT], qny[KF_QT];        // |q'|^2 and -2 iq per query row (padding queries: 0, 0); separate arrays: rows qr, qr + 1 are a register pair for the packed bounds
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
#ifdef LS_KF_WPE_ALL            // dev A/B (scripts/dev/build_variants.py)
#define LS_KF_WPE(TPW, FMA) LS_KF_WPE_ALL
#else
#define LS_KF_WPE(TPW, FMA) ((FMA) ? 3 : 4)      // (the fused-multiply-add variants -- LS_FLAG_CONTRACT_FMA, a secondary mode -- spill at four)
#endif
template <int CC, bool FMA, int TPW>
__global__ __launch_bounds__(64 * KF_WAVES, LS_KF_WPE(TPW, FMA)) void knn_fused_kernel(const float* __restrict__ dstf, const float* __restrict__ srcf, const int32_t* __restrict__ dst_rows,
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
        L.qnx[tid] = q < Nd ? nq : 0.f;
        L.qny[tid] = q < Nd ? -2.0f * iq : 0.f;
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
#if LS_KF_PK
    {
        // Round 6: the bounds of TWO accumulator rows per instruction on packed fp32 (rows r, r + 1 of a 32 x 32 accumulator are an aligned register pair and
        // their query rows qr, qr + 1 adjacent LDS words): nn = |q'|^2 + |s'|^2, w = (-2 iq) ic, d^ = fma(S, w, nn), e = eps nn, hi = d^ + e, lo = d^ - e -- the
        // same IEEE operations per element as the scalar form (so the same survivors), 6 packed instead of 12 scalar instructions per pair, and a padding
        // column carries |s'|^2 = +inf instead of a select on the minimum: hi = +inf never wins, lo = NaN is masked by colv in the filter below.
        const int slot = (wave & 1) * 32 + l31;
        kf2_t cn2[TPW], ci2[TPW];
#pragma unroll
        for (int t = 0; t < TPW; ++t) { const float cn = colv[t] ? cns[t] : INFINITY; cn2[t] = kf2_t{cn, cn}; ci2[t] = kf2_t{cic[t], cic[t]}; }
        const kf2_t eps2 = {eps, eps};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const int qr = (r & 3) + 8 * (r >> 2) + 4 * lh;       // r even: rows qr, qr + 1
            const kf2_t qx = *reinterpret_cast<const kf2_t*>(&L.qnx[qr]), qy = *reinterpret_cast<const kf2_t*>(&L.qny[qr]);
            float hm0 = INFINITY, hm1 = INFINITY;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const kf2_t nn = qx + cn2[t], w = qy * ci2[t];
                const kf2_t sv = {S[t][r], S[t][r + 1]};
                const kf2_t dh = __builtin_elementwise_fma(sv, w, nn), e = eps2 * nn;
                const kf2_t hi = dh + e;
                kf2_t lo = dh - e;
                hm0 = fminf(hm0, hi.x); hm1 = fminf(hm1, hi.y);
                asm volatile("" : "+v"(lo));      // (materialised here, as in the scalar form: sunk into the filter it keeps d^ and nn alive)
                S[t][r] = lo.x; S[t][r + 1] = lo.y;
            }
            // ---- 2. threshold: (non-negative floats order like their bit patterns; +inf = "no candidate")
            const float h0 = fmaxf(hm0, 0.0f), h1 = fmaxf(hm1, 0.0f);
            if (h0 < INFINITY) atomicMin(&L.hm[qr][slot], __float_as_uint(h0));
            if (h1 < INFINITY) atomicMin(&L.hm[qr + 1][slot], __float_as_uint(h1));
#ifndef LS_KF_NOSB
            if ((r & 3) == 2) __builtin_amdgcn_sched_barrier(0);
#endif
        }
    }
#else
    {
        const int slot = (wave & 1) * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float2 qn = make_float2(L.qnx[qr], L.qny[qr]);
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
#endif
    __syncthreads();
#pragma unroll
    for (int u = 0; u < KF_QT / KF_WAVES; ++u) {
        const int qr = wave * (KF_QT / KF_WAVES) + u;
        const unsigned tb = kth_smallest_upper_bound(L.hm[qr][lane], K < 64 ? K : 64);
        if (lane == 0) L.T[qr] = __uint_as_float(tb);
    }
    __syncthreads();
#if defined(LS_KF_STOP) && LS_KF_STOP == 1      // dev timing variants (wrong results): the kernel up to the threshold
    if (L.T[0] != 12345.f) return;
#endif
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
#if defined(LS_KF_STOP) && LS_KF_STOP == 2      // ... up to the flat pair list
    if (L.T[0] != 12345.f) { if (tid == 0) idx_out[(size_t)b * Nd * K + q0] = L.plist[0]; return; }
#endif
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
#if defined(LS_KF_STOP) && LS_KF_STOP == 3      // ... up to the exact distances
    if (L.T[0] != 12345.f) { if (tid == 0) idx_out[(size_t)b * Nd * K + q0] = (int)L.keys[0][0]; return; }
#endif
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
// the f16 image of `f` ([B, N, D] rows, centre from the first rows of fc): D = 96 / 192 (C = 32 / 64)
static int knn_prep_launch(const float* f, const float* fc, int Nc, int B, int N, int Npad, int D, unsigned short* out, float* norms, float* iscale, hipStream_t st) {
    const int tiles = B * (Npad / 32);
    if (D == 96)
        hipLaunchKernelGGL((knn_prep_f16_tile_kernel<6, 3>), dim3(tiles), dim3(192), 0, st, f, fc, Nc, N, Npad, out, norms, iscale, (int32_t*)nullptr, 0LL);
    else
        hipLaunchKernelGGL((knn_prep_f16_tile_kernel<12, 4>), dim3(tiles), dim3(256), 0, st, f, fc, Nc, N, Npad, out, norms, iscale, (int32_t*)nullptr, 0LL);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
// largest candidate set the fused kernel takes (four 32-candidate tiles per wave); beyond it knn.hip's all-VALU kernel runs (knn_uses_sweep)
int knn_sweep_max_ns() { return KF_MAXNS; }

// scratch of one call: per-query exact-distance counters | row norms (src, dst) | inverse row scales (src, dst) | f16 images (src, dst)
struct KfScratch { float *nsrc, *ndst, *isrc, *idst; int32_t* cnt; unsigned short *sq, *dq; size_t bytes; };
static KfScratch kf_layout(void* scratch, int B, int Nd, int dst_n, int Ns, int D, bool self) {
    char* sc = (char*)scratch;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = sc + off; off = (off + bytes + 255) & ~(size_t)255; return p; };
    KfScratch k;
    k.cnt = (int32_t*)take((size_t)B * Nd * sizeof(int32_t));      // (first: knn_sweep_stats_launch finds it whatever the rest looks like)
    k.nsrc = (float*)take((size_t)B * Ns * sizeof(float));
    k.ndst = self ? k.nsrc : (float*)take((size_t)B * dst_n * sizeof(float));
    k.isrc = (float*)take((size_t)B * Ns * sizeof(float));
    k.idst = self ? k.isrc : (float*)take((size_t)B * dst_n * sizeof(float));
    k.sq = (unsigned short*)take((size_t)B * pad32(Ns) * D * sizeof(unsigned short));
    k.dq = self ? k.sq : (unsigned short*)take((size_t)B * pad32(dst_n) * D * sizeof(unsigned short));
    k.bytes = off;
    return k;
}
size_t knn_sweep_scratch_bytes(int B, int Nd, int dst_n, int Ns, int C) { return kf_layout(nullptr, B, Nd, dst_n, Ns, 3 * C, false).bytes; }

template <int CC>
static int knn_sweep_launch_t(const float* dst, const float* src, const int32_t* dst_rows, int B, int Nd, int dst_n, int Ns, int K, bool fma,
                              int32_t* idx_out, float* dist_out, void* scratch, hipStream_t st) {
    constexpr int D = 3 * CC;
    const KfScratch k = kf_layout(scratch, B, Nd, dst_n, Ns, D, dst == src);
    const int ns_pad = (int)pad32(Ns), dst_npad = (int)pad32(dst_n);
    int rc = knn_prep_launch(src, src, Ns, B, Ns, ns_pad, D, k.sq, k.nsrc, k.isrc, st);
    if (rc != LS_OK) return rc;
    if (dst != src) {   // same centre for both sets: the candidates' first rows
        rc = knn_prep_launch(dst, src, Ns, B, dst_n, dst_npad, D, k.dq, k.ndst, k.idst, st);
        if (rc != LS_OK) return rc;
    }
    const int qtiles = cdiv(Nd, KF_QT), tpw = cdiv(ns_pad / 32, KF_WAVES);
    const float epsF = 1.02f * 0.0009765625f + 6.0f * (float)(D + 4) * 5.9604645e-8f + 9.5367431640625e-7f + 4.76837158203125e-7f;   // eps_b of the f16 image + 2^-21 (d^, lo, hi)
#define LS_KF(T) do { if (fma) hipLaunchKernelGGL((knn_fused_kernel<CC, true, T>), dim3(B * qtiles), dim3(64 * KF_WAVES), 0, st, dst, src, dst_rows, k.dq, k.sq, k.ndst, k.nsrc, k.idst, k.isrc, Nd, dst_n, dst_npad, Ns, ns_pad, K, qtiles, epsF, idx_out, dist_out, k.cnt); \
                   else hipLaunchKernelGGL((knn_fused_kernel<CC, false, T>), dim3(B * qtiles), dim3(64 * KF_WAVES), 0, st, dst, src, dst_rows, k.dq, k.sq, k.ndst, k.nsrc, k.idst, k.isrc, Nd, dst_n, dst_npad, Ns, ns_pad, K, qtiles, epsF, idx_out, dist_out, k.cnt); } while (0)
    if (tpw <= 1) LS_KF(1); else if (tpw == 2) LS_KF(2); else LS_KF(4);
#undef LS_KF
    LS_LAUNCH_CHECK();
    return LS_OK;
}

// hints (seed_idx: the previous layer's lists) are accepted for the caller's sake and IGNORED: the fused kernel derives its thresholds from the
// sweep itself, and hints never influenced a result
int knn_sweep_launch(const float* dst, const float* src, const int32_t* dst_rows, int B, int Nd, int dst_n, int Ns, int C, int K,
                     bool fma, int32_t* idx_out, float* dist_out, const int32_t* /*seed_idx*/, int /*seed_n*/, int /*seed_by_row*/, void* scratch,
                     hipStream_t st) {
    LS_REQUIRE(C == 32 || C == 64, "knn_sweep: only C == 32 / 64 layers are supported (C=%d)", C);
    LS_REQUIRE((int)pad32(Ns) <= KF_MAXNS && K <= KNN_MAXK, "knn_sweep: Ns=%d K=%d beyond the fused kernel (Ns <= %d, K <= 16)", Ns, K, KF_MAXNS);
    if (C == 32) return knn_sweep_launch_t<32>(dst, src, dst_rows, B, Nd, dst_n, Ns, K, fma, idx_out, dist_out, scratch, st);
    return knn_sweep_launch_t<64>(dst, src, dst_rows, B, Nd, dst_n, Ns, K, fma, idx_out, dist_out, scratch, st);
}

// Exact-phase statistics of the LAST fused call that used `scratch` (bench.py's hardware-utilisation roofline; profiled passes only):
// out[0] += sum over the queries of the candidates that were given a canonical distance (stored by the kernel as -(count + 1); a query that
// overflowed its list scanned all Ns), out[1] += the number of queries.
__global__ __launch_bounds__(256) void knn_stats_kernel(const int32_t* __restrict__ surv_cnt, long long nq, unsigned long long* __restrict__ out) {
    unsigned long long s = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long long)gridDim.x * 256) {
        const int c = surv_cnt[i];
        s += (unsigned long long)(c < 0 ? -(c + 1) : 0);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&out[0], s);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&out[1], (unsigned long long)nq);
}
int knn_sweep_stats_launch(const void* scratch, int B, int Nd, int dst_n, int Ns, unsigned long long* out, hipStream_t st) {
    const size_t nq = (size_t)B * Nd;
    (void)dst_n; (void)Ns;
    hipLaunchKernelGGL(knn_stats_kernel, dim3((unsigned)std::min<size_t>(cdiv((long long)nq, 256), 256)), dim3(256), 0, st, (const int32_t*)scratch, (long long)nq, out);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

}  // namespace ls
