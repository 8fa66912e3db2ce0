// knn_xyz.hip -- bit-exact 16-NN on RAW xyz clouds (C == 1, D = 3): encoder layer 0 and every ls_knn_f32 call on points.
//
// Replaces pytorch3d.ops.knn_points as called at
//   /root/reference/lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:139-141   (layer 0: src_f = the normalised cloud, :184)
//
// With D = 3 the distance is 8 VALU operations per pair and the tiled kernel (knn.hip) spends ~85 % of its time in the
// incremental top-K insertion (an un-seeded query inserts ~16 (1 + ln(Ns/16)) = 83 candidates, one ballot round each) and in
// the four workgroup barriers per 64 x 64 tile.  Here a WAVE owns a query and a lane owns 16 candidates (1024 per chunk):
//   1. d[16] per lane in registers (same fp32 chain as knn.hip / the oracle: ((0 + dx^2) + dy^2) + dz^2, FMA flag honoured);
//   2. the K-th smallest of the 64 per-lane minima (round 4: a radix select on the float bits' top 16 bits, 2^-8 above it at most; before: a
//      21-stage 64-lane sorting network) is an upper bound of the K-th smallest distance -- 64 disjoint candidate groups contribute one candidate each -- and admits ~18 of 1024
//      candidates on average;
//   3. the admitted (dist, idx) keys are compacted through LDS (ballot + mbcnt) and sorted by one 64-lane network on the
//      u64 keys together with the list carried over from the previous chunk (more than 48 admitted keys -- exact ties, tiny
//      clouds -- take further rounds of the same network).
// No barriers, no tiles, no insertion loop; exact by construction (everything <= an upper bound of the K-th key is sorted by
// the full lexicographic key).  Encoder layer-0 shape (64 x 1024 x 1024): 164 -> 67 us (rocprofv3); ~750 wave instructions
// per query (distances 150, threshold network 125, compaction 270, key network 200) = VALU-issue bound.
#include "knn_common.h"

namespace ls {

constexpr int KX_CH = 1024;   // candidates per chunk (16 per lane)

template <bool FMA, bool SINGLE>
__global__ __launch_bounds__(256, 4) void knn_xyz_kernel(const float* __restrict__ dst, const float* __restrict__ src,
                                                      const int32_t* __restrict__ dst_rows, int Nd, int dst_n, int Ns, int K,
                                                      int32_t* __restrict__ idx_out, float* __restrict__ dist_out, int qpw,
                                                      int qblocks) {
    __shared__ u64 lsurv[4][KX_CH + 1];   // + one dump slot per wave for the lanes that do not pass (branch-free compaction)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int logical = xcd_remap(blockIdx.x, gridDim.x);   // the query blocks of one instance share an XCD (12 KB cloud in its L2)
    const int b = logical / qblocks, qb = logical % qblocks;
    const float* sb = src + (size_t)b * Ns * 3;
    const float* db = dst + (size_t)b * dst_n * 3;
    u64* ls = lsurv[wave];
    constexpr bool single = SINGLE;   // Ns <= KX_CH: the candidates are loaded once per wave

    // candidates as PAIRS (j, j + 8) -> packed fp32 math (v_pk_add / v_pk_mul: half the distance instructions; no MFMA in this
    // kernel, so the packed ops cost nothing extra).  Columns past Ns hold +inf: their distance is +inf and never passes.
    f32x2 cx[8], cy[8], cz[8];
    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ca = c0 + lane + 64 * j, cb = ca + 512;
            const int la = min(ca, Ns - 1), lb = min(cb, Ns - 1);
            const float ax = sb[(size_t)la * 3 + 0], ay = sb[(size_t)la * 3 + 1], az = sb[(size_t)la * 3 + 2];
            const float bx = sb[(size_t)lb * 3 + 0], by = sb[(size_t)lb * 3 + 1], bz = sb[(size_t)lb * 3 + 2];
            cx[j] = f32x2{ca < Ns ? ax : INFINITY, cb < Ns ? bx : INFINITY};
            cy[j] = f32x2{ca < Ns ? ay : INFINITY, cb < Ns ? by : INFINITY};
            cz[j] = f32x2{ca < Ns ? az : INFINITY, cb < Ns ? bz : INFINITY};
        }
    };
    if (single) load_chunk(0);

    const int qbase = (qb * 4 + wave) * qpw;
    for (int qi = 0; qi < qpw; ++qi) {
        const int q = qbase + qi;
        if (q >= Nd) break;                                   // wave-uniform
        const int r = __builtin_amdgcn_readfirstlane(dst_rows ? dst_rows[(size_t)b * Nd + q] : q);
        const float qx = db[(size_t)r * 3 + 0], qy = db[(size_t)r * 3 + 1], qz = db[(size_t)r * 3 + 2];
        u64 best = ~0ull;                                     // lanes 0..15: the sorted list so far
        for (int c0 = 0; c0 < Ns; c0 += KX_CH) {
            if (!single) load_chunk(c0);
            f32x2 d[8];
            unsigned mn = 0x7F800000u;                        // +inf
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f32x2 a = accq2<FMA>(f32x2{0.0f, 0.0f}, qx, cx[j]);
                a = accq2<FMA>(a, qy, cy[j]);
                a = accq2<FMA>(a, qz, cz[j]);
                d[j] = a;
                mn = min(mn, min(__float_as_uint(a.x), __float_as_uint(a.y)));
            }
            // admission threshold: K-th smallest lane minimum, tightened by the carried list's K-th key; capped at the largest
            // finite float so that the +inf padding columns never pass
            unsigned thr = kth_smallest_upper_bound(mn, K);   // (knn_common.h)
            thr = min(thr, (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(best >> 32), K - 1));
            thr = min(thr, 0x7F7FFFFFu);
            int n = 0;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                const int j = jj & 7;
                const float dv = jj < 8 ? d[j].x : d[j].y;
                const int c = c0 + lane + 64 * j + (jj < 8 ? 0 : 512);
                const bool pass = __float_as_uint(dv) <= thr;
                const u64 bal = __ballot(pass);
                const int pos = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
                ls[pass ? pos : KX_CH] = make_key(dv, c, true);   // non-passing lanes write the dump slot
                n += __builtin_popcountll(bal);
            }
            __builtin_amdgcn_wave_barrier();
            int pos = 0;
            if (c0 == 0) {   // nothing carried yet: the first 64 keys start at lane 0, and the usual ~18 of them need only the 32-lane network
                u64 k = lane < n ? ls[lane] : ~0ull;
                if (n <= 32) { LS_SORT32(cx64, k, lane) }   // (wave-uniform)
                else { LS_SORT64(cx64, k, lane) }
                best = k;
                pos = 64;
            }
            for (; pos < n; pos += 48) {
                u64 k = best;
                if (lane >= 16) {
                    const int i = pos + lane - 16;
                    k = i < n ? ls[i] : ~0ull;
                }
                LS_SORT64(cx64, k, lane)
                best = k;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (lane < K) {
            const size_t o = ((size_t)b * Nd + q) * K + lane;
            const unsigned hi = (unsigned)(best >> 32), lo = (unsigned)best;
            idx_out[o] = hi == 0xFFFFFFFFu ? -1 : (int)lo;
            if (dist_out) dist_out[o] = hi == 0xFFFFFFFFu ? INFINITY : __uint_as_float(hi);
        }
    }
}

int knn_xyz_launch(const float* dst, const float* src, const int32_t* dst_rows, int B, int Nd, int dst_n, int Ns, int K, bool fma,
                   int32_t* idx_out, float* dist_out, hipStream_t st) {
    // queries per wave: 16 amortises the candidate registers; fewer when the launch would not fill the 256 CUs
    int qpw = 16;
    while (qpw > 1 && (long long)B * cdiv(Nd, 4 * qpw) < 1024) qpw >>= 1;
    const int qblocks = cdiv(Nd, 4 * qpw);
    const bool single = Ns <= KX_CH;
#define LS_KX(F, S) hipLaunchKernelGGL((knn_xyz_kernel<F, S>), dim3(B * qblocks), dim3(256), 0, st, dst, src, dst_rows, Nd, dst_n, Ns, K, \
                                       idx_out, dist_out, qpw, qblocks)
    if (fma) { if (single) LS_KX(true, true); else LS_KX(true, false); }
    else { if (single) LS_KX(false, true); else LS_KX(false, false); }
#undef LS_KX
    LS_LAUNCH_CHECK();
    return LS_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Small feature-space problems (encoder layers 5 and 6: 32 queries x 128 / 32 candidates per instance, D = 384 / 768).
// The 64 x 64-tile kernel of knn.hip leaves 3/4 (layer 5) or 15/16 (layer 6) of its register micro-tiles on padding and walks
// the channels in 32-wide chunks with two workgroup barriers each: 65 + 77 us of mostly latency.  Here a workgroup owns 8
// queries of one instance and 32 candidates at a time, one (query, candidate) pair per thread, whole rows (up to 128
// channels per pass) staged x-major through LDS with 16-byte copies; a thread reads the x, y and z float4 of four channels
// and adds the twelve terms in canonical order (j = c*3 + x), so no LDS transpose is needed.  Selection: the 8 x 32 distance
// tile goes through LDS to two waves whose 16-lane rows merge it (knn_common.h).  Same arithmetic chain, same keys.
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
constexpr int KSM_Q = 8, KSM_S = 32, KSM_CCH = 128;
constexpr int KSM_ROW = 3 * KSM_CCH + 4;     // floats; (ROW / 4) odd -> the 16 lanes of a ds_read_b128 group hit distinct banks

template <bool FMA>
__global__ __launch_bounds__(256, 2) void knn_small_kernel(const float* __restrict__ dstf, const float* __restrict__ srcf,
                                                           const int32_t* __restrict__ dst_rows, int Nd, int dst_n, int Ns, int C, int K,
                                                           int32_t* __restrict__ idx_out, float* __restrict__ dist_out, int qblocks) {
    LS_LATENCY_CRITICAL();
    __shared__ __attribute__((aligned(16))) float lq[KSM_Q * KSM_ROW];
    __shared__ __attribute__((aligned(16))) float lc[KSM_S * KSM_ROW];
    __shared__ float ldist[KSM_Q][KSM_S + 1];
    __shared__ int lqrow[KSM_Q];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int b = logical / qblocks, q0 = (logical % qblocks) * KSM_Q;
    const size_t row_f = (size_t)3 * C;
    const float* dbase = dstf + (size_t)b * dst_n * row_f;
    const float* sbase = srcf + (size_t)b * Ns * row_f;
    if (tid < KSM_Q) {
        const int q = q0 + tid;
        lqrow[tid] = q < Nd ? (dst_rows ? dst_rows[(size_t)b * Nd + q] : q) : -1;
    }
    __syncthreads();
    const int qi = tid >> 5, ci = tid & 31;
    // rows of this wave's merge: waves 0 and 1, row r = lane >> 4 -> query wave * 4 + r
    u64 lk = ~0ull, rkey = ~0ull;
    const int nch = cdiv_dev(C, KSM_CCH);
    // candidate staging: thread -> row sr = tid >> 3, float4 columns (tid & 7) + 8 m of each x segment; the next pass's rows are
    // in flight (registers) under the current pass's arithmetic
    const int sr = tid >> 3, sc = (tid & 7) * 4;
    // (named scalars: hipcc keeps a prefetch ARRAY that is written in one iteration and read in the next in scratch memory)
#define KSM_PF_LIST(F) F(0, 0) F(0, 1) F(0, 2) F(0, 3) F(1, 0) F(1, 1) F(1, 2) F(1, 3) F(2, 0) F(2, 1) F(2, 2) F(2, 3)
#define KSM_PF_DECL(x, m) float4 pf_##x##_##m;
#define KSM_PF_LOAD(x, m) pf_##x##_##m = *reinterpret_cast<const float4*>(rp + (size_t)x * C + min(m * 32, cwp - 32));
#define KSM_PF_STORE(x, m) \
    if (sc + m * 32 < cw) *reinterpret_cast<float4*>(&lc[sr * KSM_ROW + x * KSM_CCH + sc + m * 32]) = pf_##x##_##m;
    KSM_PF_LIST(KSM_PF_DECL)
#define KSM_PREFETCH(S0, CH)                                                                                  \
    {                                                                                                         \
        const int c0p = (CH) * KSM_CCH, cwp = min(C - c0p, KSM_CCH);                                          \
        const float* rp = sbase + (size_t)min((S0) + sr, Ns - 1) * row_f + c0p + sc;                          \
        KSM_PF_LIST(KSM_PF_LOAD)   /* columns past this pass's width: a clamped re-read inside the row */     \
    }
    KSM_PREFETCH(0, 0)
    for (int s0 = 0; s0 < Ns; s0 += KSM_S) {
        float d = 0.0f;
        for (int ch = 0; ch < nch; ++ch) {
            const int c0 = ch * KSM_CCH, cw = min(C - c0, KSM_CCH);   // channels of this pass
            __syncthreads();   // previous pass / previous merge done with lq, lc, ldist
            if (nch > 1 || s0 == 0) {   // query rows: 32 threads per row
                const int gr = lqrow[qi];
                for (int x = 0; x < 3; ++x)
                    for (int c4 = ci * 4; c4 < cw; c4 += 128)
                        *reinterpret_cast<float4*>(&lq[qi * KSM_ROW + x * KSM_CCH + c4]) =
                            gr >= 0 ? *reinterpret_cast<const float4*>(dbase + (size_t)gr * row_f + (size_t)x * C + c0 + c4)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            KSM_PF_LIST(KSM_PF_STORE)
            __syncthreads();
            {
                int nch2 = ch + 1, ns0 = s0;
                if (nch2 == nch) { nch2 = 0; ns0 = s0 + KSM_S; }
                KSM_PREFETCH(min(ns0, Ns - 1), nch2)   // past the end: a harmless re-read of the last row
            }
            const float* qp = &lq[qi * KSM_ROW];
            const float* cp = &lc[ci * KSM_ROW];
#pragma unroll 2
            for (int c4 = 0; c4 < cw; c4 += 4) {
                const float4 ax = *reinterpret_cast<const float4*>(qp + c4), ay = *reinterpret_cast<const float4*>(qp + KSM_CCH + c4),
                             az = *reinterpret_cast<const float4*>(qp + 2 * KSM_CCH + c4);
                const float4 bx = *reinterpret_cast<const float4*>(cp + c4), by = *reinterpret_cast<const float4*>(cp + KSM_CCH + c4),
                             bz = *reinterpret_cast<const float4*>(cp + 2 * KSM_CCH + c4);
                d = accq<FMA>(d, ax.x, bx.x); d = accq<FMA>(d, ay.x, by.x); d = accq<FMA>(d, az.x, bz.x);
                d = accq<FMA>(d, ax.y, bx.y); d = accq<FMA>(d, ay.y, by.y); d = accq<FMA>(d, az.y, bz.y);
                d = accq<FMA>(d, ax.z, bx.z); d = accq<FMA>(d, ay.z, by.z); d = accq<FMA>(d, az.z, bz.z);
                d = accq<FMA>(d, ax.w, bx.w); d = accq<FMA>(d, ay.w, by.w); d = accq<FMA>(d, az.w, bz.w);
            }
        }
        ldist[qi][ci] = d;
        __syncthreads();
        if (wave < 2) {
            const int qr = wave * 4 + (lane >> 4), e = lane & 15;
            const int ca = s0 + e, cb = s0 + e + 16;
            const bool live = lqrow[qr] >= 0;
            u64 k0 = make_key(ldist[qr][e], ca, live && ca < Ns), k1 = make_key(ldist[qr][e + 16], cb, live && cb < Ns);
            key_cx(k0, k1);
            merge_keys<false>(k0, k1, ~0ull, ~0ull, lk, rkey, K, lane);
        }
    }
    if (wave < 2) {
        const int qr = wave * 4 + (lane >> 4), e = lane & 15, q = q0 + qr;
        if (q < Nd && e < K) {
            const size_t o = ((size_t)b * Nd + q) * K + e;
            const unsigned hi = (unsigned)(lk >> 32), lo = (unsigned)lk;
            idx_out[o] = hi == 0xFFFFFFFFu ? -1 : (int)lo;
            if (dist_out) dist_out[o] = hi == 0xFFFFFFFFu ? INFINITY : __uint_as_float(hi);
        }
    }
}

int knn_small_launch(const float* dst, const float* src, const int32_t* dst_rows, int B, int Nd, int dst_n, int Ns, int C, int K, bool fma,
                     int32_t* idx_out, float* dist_out, hipStream_t st) {
    LS_REQUIRE(C % 4 == 0, "knn_small: C=%d must be a multiple of 4", C);
    const int qblocks = cdiv(Nd, KSM_Q);
    if (fma)
        hipLaunchKernelGGL(knn_small_kernel<true>, dim3(B * qblocks), dim3(256), 0, st, dst, src, dst_rows, Nd, dst_n, Ns, C, K, idx_out,
                           dist_out, qblocks);
    else
        hipLaunchKernelGGL(knn_small_kernel<false>, dim3(B * qblocks), dim3(256), 0, st, dst, src, dst_rows, Nd, dst_n, Ns, C, K, idx_out,
                           dist_out, qblocks);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

}  // namespace ls
