// edge_staged.hip -- attention edge-conv (encoder layers 2 - 4) with LDS-STAGED NEIGHBOUR TILES (round 5).
//
// Same layer as edge.hip: edge_attn_fq_kernel (vec_dgcnn_atten.py:196-219: K / Q / V VecLNAs over cat(nbr - ctr, ctr),
// channel_equi_vec_normalize, head soft-max, weighted sum; folded per-point tables as in the edge.hip header), different data movement.
// edge_attn_fq_kernel gathers every neighbour's table row through the vector-memory path: 16 x 3 KB per destination point, 1.6 GB per
// launch at layers 2 / 3 for 100 - 200 MB of distinct table bytes, delivered at 13 TB/s -- 80 % of what a pure gather probe reaches
// (profiles/r4_final/gather_probe.txt): the kernel's bound is the L1 path, not HBM, not the VALU.  A source row is used by 4 - 16 edges of its
// instance, so the re-reads are served here by the LDS instead:
//
//   * a workgroup = 1024 threads = ONE CU, owns PTS destination points of one instance (G = Nd / PTS workgroups per instance, neighbours
//     on one XCD);
//   * the table is stored SLICE-MAJOR by the table GEMM (gemm.hip: GemmAux::slice_cols): per instance 2 Co / CS slices
//     [K slices | V slices], a slice = [source row = point * 3 + xyz][CS channels][lin, dir] floats = Ns * 3 * CS * 8 bytes <= 48 KB, contiguous;
//   * phase t streams slice t + 1 global -> LDS with global_load_lds_dwordx4 (no registers, 48 x 1 KB per slice, double-buffered) while every
//     lane works on slice t: lane = (point, channel cl of the slice, edge group eg), EPL = 16 / EG edges each, three ds_read_b64 per edge;
//   * K pass (all K slices): per-edge partial scores <k, q> and |k|^2 stay in registers, reduced over the channel lanes by DPP at the end
//     of every head / of the pass; soft-max over the 16 edges in registers; V pass (all V slices): weighted sum, written per phase;
//   * the destination side (Q_lin / Q_dir of the K and V branches, q) comes from the f16 matrix cores as in edge_attn_fq_kernel, per "Q block"
//     of 16 output columns: v_mfma_f32_16x16x32_f16, A = the wave's own 3 * PPW feature rows, split once into registers, B = pre-split weight
//     planes DMA-ed beside the slices, results into a wave-private LDS slab.
//
// Traffic per launch at layer 3 (B = 64): G = 4 x 100 MB streamed L2 -> LDS (each instance's table once per CU that works on it) instead
// of 1.6 GB gathered; HBM sees the table once.  Arithmetic per edge and channel is the fq kernel's; the ORDER of the head / norm sums differs
// (per-channel chains + lane trees instead of four-channel chains), so the two paths agree to ~1e-6 of the tensor maximum, not bit for bit
// (tests/test_hip_layers.py::test_staged_attention_*).
#include "ls_common.h"

namespace ls {

namespace {

constexpr int SK = 16;                 // neighbours per point
constexpr int ST_SLICE_MAX = 49152;    // bytes of one staged slice buffer
constexpr int ST_THREADS = 1024;

typedef _Float16 sh8_t __attribute__((ext_vector_type(8)));
typedef _Float16 sh2_t __attribute__((ext_vector_type(2)));
typedef float sf2_t __attribute__((ext_vector_type(2)));
typedef float sf4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st_split_pair(sf2_t v, unsigned& h, unsigned& l) {
    const sh2_t hv = __builtin_convertvector(v, sh2_t);
    const sh2_t lv = __builtin_convertvector(v - __builtin_convertvector(hv, sf2_t), sh2_t);
    h = __builtin_bit_cast(unsigned, hv);
    l = __builtin_bit_cast(unsigned, lv);
}
// eight consecutive-k fp32 values (already scaled) -> the (hi, lo) f16 operand fragments of this lane (gemm.hip: two-piece split)
__device__ __forceinline__ void st_split8(const float4& a, const float4& b, float sc, sh8_t& h, sh8_t& l) {
    uint4 hh, ll;
    st_split_pair(sf2_t{a.x * sc, a.y * sc}, hh.x, ll.x); st_split_pair(sf2_t{a.z * sc, a.w * sc}, hh.y, ll.y);
    st_split_pair(sf2_t{b.x * sc, b.y * sc}, hh.z, ll.z); st_split_pair(sf2_t{b.z * sc, b.w * sc}, hh.w, ll.w);
    h = __builtin_bit_cast(sh8_t, hh);
    l = __builtin_bit_cast(sh8_t, ll);
}
// the row's power of two: largest element -> [2^14, 2^15) (gemm.hip, "operand range of the f16 split"); e_inv = exponent of the inverse scale
__device__ __forceinline__ void st_pow2_scale(float amax, float& s, int& e_inv) {
    unsigned be = (__float_as_uint(amax) >> 23) & 0xffu;
    be = be < 15u ? 15u : be;
    s = __uint_as_float((268u - be) << 23);
    e_inv = (int)be - 14 - 127;
}
template <int CTRL>
__device__ __forceinline__ float st_dpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ float st_dpp_max(float v) {
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false)));
}
// all-reduce over the CS channel lanes of a (point, edge group): CS = 2 -> lane pairs, CS = 4 -> quads
template <int CS>
__device__ __forceinline__ float cs_sum(float v) {
    v = st_dpp_add<0xB1>(v);                          // quad_perm [1,0,3,2]
    if constexpr (CS == 4) v = st_dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    return v;
}
template <int CS>
__device__ __forceinline__ float cs_max(float v) {
    v = st_dpp_max<0xB1>(v);
    if constexpr (CS == 4) v = st_dpp_max<0x4E>(v);
    return v;
}
// all-reduce over the EG edge-group lanes of a point (lane stride CS inside an aligned group of CS * EG lanes)
template <int CS, int EG>
__device__ __forceinline__ float eg_sum(float v) {
#pragma unroll
    for (int o = CS; o < CS * EG; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int CS, int EG>
__device__ __forceinline__ float eg_max(float v) {
#pragma unroll
    for (int o = CS; o < CS * EG; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int CTRL>
__device__ __forceinline__ float st_dpp_mov(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
// Sums over the CS channel lanes, kept where the registers are scarce.  SCAT (CS = 4, 8 edges per lane): a transpose-reduce over the quad -- the total
// of edge e = 4 k + q ends up in slot k of quad lane q (18 instructions, against 16 for eight all-reduces that would keep eight registers per head);
// otherwise: plain all-reduces, slot = edge.
template <int CS, int EPL>
struct StRed {
    static constexpr bool SCAT = CS == 4 && EPL == 8;
    static constexpr int NS = SCAT ? 2 : EPL;
    static __device__ __forceinline__ void run(const float (&sp)[EPL], float (&o)[NS], int lane) {
        if constexpr (SCAT) {
            const bool b0 = lane & 1, b1 = lane & 2;
            float t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = (b0 ? sp[2 * j + 1] : sp[2 * j]) + st_dpp_mov<0xB1>(b0 ? sp[2 * j] : sp[2 * j + 1]);
#pragma unroll
            for (int k = 0; k < 2; ++k) o[k] = (b1 ? t[2 * k + 1] : t[2 * k]) + st_dpp_mov<0x4E>(b1 ? t[2 * k] : t[2 * k + 1]);
        } else {
#pragma unroll
            for (int i = 0; i < EPL; ++i) o[i] = cs_sum<CS>(sp[i]);
        }
    }
    // the value of this lane's edge i out of the slots (SCAT: slot i >> 2 of quad lane i & 3)
    template <int I>
    static __device__ __forceinline__ float get(const float (&o)[NS]) {
        if constexpr (SCAT) return st_dpp_mov<(I & 3) * 0x55>(o[I >> 2]);
        else return o[I];
    }
};
__device__ __forceinline__ float st_inv_fro(float ss) { return __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f)); }   // as edge.hip: inv_fro

// geometry shared by the kernel, the weight preparation and the host side
template <int CS, int EG, int CIN, int CO>
struct StGeo {
    static constexpr int LPP = CS * EG;             // lanes per destination point
    static constexpr int PPW = 64 / LPP;            // points per wave
    static constexpr int PTS = ST_THREADS / LPP;    // points per workgroup
    static constexpr int EPL = SK / EG;             // edges per lane
    static constexpr int NSL = CO / CS;             // slices per pass
    static constexpr int NH = CO / 16;              // heads
    static constexpr int PPH = 16 / CS;             // phases (slices) per head
    static constexpr int ROWB = 3 * CS * 8;         // bytes of one source point in a slice: [xyz][cl][lin, dir]
    static constexpr int KS = CIN / 32;             // k-steps of the 16x16x32 MFMA
    static constexpr int MT = (PPW * 3 + 15) / 16;  // 16-row tiles of a wave's destination rows
    static constexpr int PBK = 4 / CS;              // K phases served by one Q block of 16 columns (QK_lin, QK_dir, Qq_lin, Qq_dir per channel)
    static constexpr int PBV = 8 / CS;              // V phases per Q block (QV_lin, QV_dir per channel)
    static constexpr int NKB = NSL / PBK, NVB = NSL / PBV;
    static constexpr int QCH = KS * 2 + 1;          // 1 KB chunks of a Q block: [ks][hi, lo] fragment planes + one chunk with the 16 column exponents
    static constexpr int QBB = QCH * 1024;
    static constexpr int SLABF = MT * 16 * 20;      // floats of a wave's slab: [row][16 columns, stride 20]
    static constexpr int WPRIV = SLABF * 4 + 128;   // + the MT * 16 row exponents
    static constexpr int LDS_BYTES = 2 * ST_SLICE_MAX + 2 * QBB + 16 * WPRIV;
    static_assert(PPH % 2 == 0 && NSL % 2 == 0, "buffer parity = phase parity inside a head");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

}  // namespace

// ---------------------------------------------------------------------------------------------- weight preparation (once per model)
// P-side rows of the folded edge weights in SLICE order: Wp[n] = W[orig(n)], n = (pass * NSL + j) * 2 CS + cl * 2 + g (pass 0 = K, 1 = V;
// g 0 = lin, 1 = dir), orig = the [PV_lin | PV_dir | PK_lin | PK_dir] row of channel j CS + cl (edge.hip header).
__global__ __launch_bounds__(256) void edge_st_permute_w_kernel(const float* __restrict__ W, int Co, int Cin, int CS, float* __restrict__ Wp) {
    const int n = blockIdx.x, NSL = Co / CS;
    const int sidx = n / (2 * CS), c = n % (2 * CS), pass = sidx / NSL, j = sidx % NSL, cl = c >> 1, g = c & 1;
    const int orig = ((pass == 0 ? 2 : 0) + g) * Co + j * CS + cl;
    for (int k = threadIdx.x; k < Cin; k += 256) Wp[(size_t)n * Cin + k] = W[(size_t)orig * Cin + k];
}
// Q-side rows ([QV_lin | QV_dir | QK_lin | QK_dir | Qq_lin | Qq_dir] x Co, Cin) -> per Q block of 16 output columns the B fragments of
// v_mfma_f32_16x16x32_f16: [ks][hi, lo][lane] x 16 bytes, lane = column (l & 15) + 16 * (k / 8 & 3), each row scaled by its own power of two,
// then one 1 KB chunk whose first 16 ints are the columns' inverse-scale exponents.  K blocks first, then V blocks.
__global__ __launch_bounds__(64) void edge_st_presplit_q_kernel(const float* __restrict__ Wq, int Co, int Cin, int CS, char* __restrict__ planes) {
    const int NSL = Co / CS, PBK = 4 / CS, PBV = 8 / CS, NKB = NSL / PBK, KS = Cin / 32, QBB = (KS * 2 + 1) * 1024;
    const int blk = blockIdx.x, lane = threadIdx.x, c = lane & 15, kg = lane >> 4;
    int orig;
    if (blk < NKB) {
        const int pb = c / (4 * CS), rem = c % (4 * CS), cl = rem >> 2, g4 = rem & 3;
        orig = (2 + g4) * Co + (blk * PBK + pb) * CS + cl;
    } else {
        const int pb = c / (2 * CS), rem = c % (2 * CS), cl = rem >> 1, g2 = rem & 1;
        orig = g2 * Co + ((blk - NKB) * PBV + pb) * CS + cl;
    }
    const float* wr = Wq + (size_t)orig * Cin;
    float am = 0.f;
    for (int k = 0; k < Cin; ++k) am = fmaxf(am, fabsf(wr[k]));
    float sc; int ei;
    st_pow2_scale(am, sc, ei);
    char* out = planes + (size_t)blk * QBB;
    for (int ks = 0; ks < KS; ++ks) {
        const float4 a = *reinterpret_cast<const float4*>(wr + ks * 32 + kg * 8), b = *reinterpret_cast<const float4*>(wr + ks * 32 + kg * 8 + 4);
        sh8_t h, l;
        st_split8(a, b, sc, h, l);
        *reinterpret_cast<uint4*>(out + (ks * 2) * 1024 + lane * 16) = __builtin_bit_cast(uint4, h);
        *reinterpret_cast<uint4*>(out + (ks * 2 + 1) * 1024 + lane * 16) = __builtin_bit_cast(uint4, l);
    }
    int* ex = reinterpret_cast<int*>(out + KS * 2048);
    for (int u = lane; u < 256; u += 64) ex[u] = 0;
    __syncthreads();
    if (kg == 0) ex[c] = ei;
}

// ---------------------------------------------------------------------------------------------- the kernel
template <int CS, int EG, int CIN, int CO>
__global__ __launch_bounds__(ST_THREADS) void edge_attn_staged_kernel(const float* __restrict__ T, const float* __restrict__ cur, const char* __restrict__ Qp,
                                                                     const int32_t* __restrict__ knn, const int32_t* __restrict__ dst_rows, int Nd, int Ns,
                                                                     int G, float oms, float inv_sqrt_dk, float* __restrict__ out, float* __restrict__ rowmax) {
    using GEO = StGeo<CS, EG, CIN, CO>;
    constexpr int LPP = GEO::LPP, PPW = GEO::PPW, PTS = GEO::PTS, EPL = GEO::EPL, NSL = GEO::NSL, NH = GEO::NH, PPH = GEO::PPH, ROWB = GEO::ROWB,
                  KS = GEO::KS, MT = GEO::MT, PBK = GEO::PBK, PBV = GEO::PBV, NKB = GEO::NKB, QCH = GEO::QCH, QBB = GEO::QBB, SLABF = GEO::SLABF,
                  WPRIV = GEO::WPRIV;
    constexpr int OFF_Q = 2 * ST_SLICE_MAX, OFF_W = OFF_Q + 2 * QBB;
    __shared__ __attribute__((aligned(16))) char lds[GEO::LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane / LPP, li = lane % LPP, cl = li % CS, eg = li / CS;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);      // the G workgroups of an instance on one XCD: its slices are streamed out of one L2
    const int b = logical / G, g = logical - b * G;
    const int pw0 = g * PTS + wave * PPW;                      // first destination point (inside the instance) of this wave
    const int pid = b * Nd + pw0 + pl;
    const int slice_bytes = Ns * ROWB, nchunks = slice_bytes >> 10;
    const char* Tb = reinterpret_cast<const char*>(T) + (size_t)b * 2 * NSL * slice_bytes;
    float* slab = reinterpret_cast<float*>(lds + OFF_W + wave * WPRIV);
    int* rexp = reinterpret_cast<int*>(lds + OFF_W + wave * WPRIV + SLABF * 4);
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned lane16 = (unsigned)lane * 16u;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);

    // global -> LDS, 1 KB per instruction: lane l's 16 bytes land at M0 + 16 l.  Inline asm (as gemm.hip: WDIR): the compiler does not track these
    // loads, so every phase ends with an explicit s_waitcnt vmcnt(0) in front of its barrier.
    auto dma = [&](unsigned lds_addr, const char* src) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(lane16), "s"(src) : "memory", "m0");
#pragma clang diagnostic pop
    };
    // stage slice tn (0 .. 2 NSL - 1: K slices, then V slices) and, when phase tn opens a Q block, that block's weight planes
    auto stage = [&](int tn) {
        const char* src = Tb + (size_t)tn * slice_bytes;
        const unsigned dst = lds0 + (unsigned)(tn & 1) * ST_SLICE_MAX;
#ifdef LS_ST_NODMA        // dev timing variants (scripts/dev/build_variants.py): never defined in the release build
        if (tn > 0) return;
#endif
#ifdef LS_ST_ROT          // the G workgroups of an instance start their slice at different chunks
        const int rot = g * (nchunks / G);
        for (int c = wave_s; c < nchunks; c += 16) { int cc = c + rot; if (cc >= nchunks) cc -= nchunks; dma(dst + (unsigned)cc * 1024u, src + (size_t)cc * 1024); }
#else
        for (int c = wave_s; c < nchunks; c += 16) dma(dst + (unsigned)c * 1024u, src + (size_t)c * 1024);
#endif
        int blk = -1;
        if (tn < NSL) { if (tn % PBK == 0) blk = tn / PBK; }
        else if ((tn - NSL) % PBV == 0) blk = NKB + (tn - NSL) / PBV;
        if (blk >= 0 && wave_s < QCH) dma(lds0 + OFF_Q + (unsigned)(blk & 1) * QBB + (unsigned)wave_s * 1024u, Qp + (size_t)blk * QBB + (size_t)wave_s * 1024);
    };
    stage(0);

    // ---- this lane's edges: byte offset of the neighbour's row in a slice (+ this lane's channel)
    unsigned noff[EPL];
    {
        const int32_t* kp = knn + (size_t)pid * SK + eg * EPL;
        if constexpr (EPL >= 4) {
#pragma unroll
            for (int u = 0; u < EPL / 4; ++u) {
                const int4 v = reinterpret_cast<const int4*>(kp)[u];
                noff[4 * u] = (unsigned)v.x * ROWB + cl * 8; noff[4 * u + 1] = (unsigned)v.y * ROWB + cl * 8;
                noff[4 * u + 2] = (unsigned)v.z * ROWB + cl * 8; noff[4 * u + 3] = (unsigned)v.w * ROWB + cl * 8;
            }
        } else {
            const int2 v = *reinterpret_cast<const int2*>(kp);
            noff[0] = (unsigned)v.x * ROWB + cl * 8; noff[1] = (unsigned)v.y * ROWB + cl * 8;
        }
    }

    // ---- the wave's destination feature rows as (hi, lo) f16 A fragments, kept in registers for every Q block
    sh8_t ah[MT][KS], al[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int rr = mt * 16 + (lane & 15);
        const bool on = rr < PPW * 3;
        const int pp = on ? rr / 3 : 0, x = on ? rr - 3 * pp : 0;
        const int pr = b * Nd + pw0 + pp;
        const int sp = dst_rows ? dst_rows[pr] : pw0 + pp;
        const float* fr = cur + (((size_t)b * Ns + sp) * 3 + x) * CIN + (lane >> 4) * 8;
        float4 va[KS], vb[KS];
        float am = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            va[ks] = on ? *reinterpret_cast<const float4*>(fr + ks * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
            vb[ks] = on ? *reinterpret_cast<const float4*>(fr + ks * 32 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            am = fmaxf(am, fmaxf(fmaxf(fabsf(va[ks].x), fabsf(va[ks].y)), fmaxf(fabsf(va[ks].z), fabsf(va[ks].w))));
            am = fmaxf(am, fmaxf(fmaxf(fabsf(vb[ks].x), fabsf(vb[ks].y)), fmaxf(fabsf(vb[ks].z), fabsf(vb[ks].w))));
        }
        am = fmaxf(am, __shfl_xor(am, 16, 64));
        am = fmaxf(am, __shfl_xor(am, 32, 64));
        float sc; int ei;
        st_pow2_scale(am, sc, ei);
        if (lane < 16) rexp[rr] = ei;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) st_split8(va[ks], vb[ks], sc, ah[mt][ks], al[mt][ks]);
    }

    // ---- destination-side product of one Q block: slab[row][0 .. 15] = x_rows . Wq_block^T  (wave-private: no workgroup barrier)
    auto qblock = [&](int blk) {
        const char* bp = lds + OFF_Q + (blk & 1) * QBB;
        const int ew = reinterpret_cast<const int*>(bp + KS * 2048)[lane & 15];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            sf4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const sh8_t bh = __builtin_bit_cast(sh8_t, *reinterpret_cast<const uint4*>(bp + (ks * 2) * 1024 + lane * 16));
                const sh8_t bl = __builtin_bit_cast(sh8_t, *reinterpret_cast<const uint4*>(bp + (ks * 2 + 1) * 1024 + lane * 16));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[mt][ks], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt][ks], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[mt][ks], bl, acc, 0, 0, 0);
            }
            // C/D map of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + r
            const int row0 = mt * 16 + 4 * (lane >> 4);
            const int4 re = *reinterpret_cast<const int4*>(rexp + row0);
            float* sp_ = slab + row0 * 20 + (lane & 15);
            sp_[0] = __builtin_ldexpf(acc[0], re.x + ew); sp_[20] = __builtin_ldexpf(acc[1], re.y + ew);
            sp_[40] = __builtin_ldexpf(acc[2], re.z + ew); sp_[60] = __builtin_ldexpf(acc[3], re.w + ew);
        }
    };
    auto phase_end = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the next slice has landed (and its output stores have left)
        __syncthreads();
    };
    phase_end();   // slice 0 + Q block 0 are in the LDS, rexp is written

    // ================================================================================================ K pass
    using RED = StRed<CS, EPL>;
    constexpr int NS = RED::NS;
    float sc_[NH][NS];           // head scores of this lane's slots (StRed), later the soft-max weights
    float np[EPL];               // |k|^2 partial of this lane's channels
    float ssq = 0.f;             // |q|^2 partial
#pragma unroll
    for (int i = 0; i < EPL; ++i) np[i] = 0.f;
    const float* qrow = slab + (3 * pl) * 20;
    struct Row { float2 a0, a1, a2; };
    auto ldrow = [&](const char* sb, unsigned off) {
        Row r;
        const char* rp = sb + off;
        r.a0 = *reinterpret_cast<const float2*>(rp); r.a1 = *reinterpret_cast<const float2*>(rp + CS * 8); r.a2 = *reinterpret_cast<const float2*>(rp + 2 * CS * 8);
        return r;
    };
    auto kphase = [&](int t, const char* sb, float (&sp)[EPL]) {
        stage(t + 1);
        if (t % PBK == 0) qblock(t / PBK);
        const float* qr = qrow + ((t % PBK) * CS + cl) * 4;
        const float4 qx = *reinterpret_cast<const float4*>(qr), qy = *reinterpret_cast<const float4*>(qr + 20), qz = *reinterpret_cast<const float4*>(qr + 40);
        float q0 = qx.z, q1 = qy.z, q2 = qz.z;
        vn_act(q0, q1, q2, qx.w, qy.w, qz.w, oms);
        ssq = __builtin_fmaf(q2, q2, __builtin_fmaf(q1, q1, __builtin_fmaf(q0, q0, ssq)));
        // two rows of LDS reads in flight ahead of the arithmetic (the scheduler, left free, issues all EPL rows first: 6 EPL registers)
        Row pr[2];
        pr[0] = ldrow(sb, noff[0]); pr[1] = ldrow(sb, noff[1]);
#ifdef LS_ST_NOCOMPUTE
        asm volatile("" :: "v"(pr[0].a0.x), "v"(pr[1].a0.x));
#else
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const Row r = pr[i & 1];
            if (i + 2 < EPL) pr[i & 1] = ldrow(sb, noff[i + 2]);
            __builtin_amdgcn_sched_barrier(0);
            float y0 = r.a0.x + qx.x, y1 = r.a1.x + qy.x, y2 = r.a2.x + qz.x;
            const float d0 = r.a0.y + qx.y, d1 = r.a1.y + qy.y, d2 = r.a2.y + qz.y;
            vn_act(y0, y1, y2, d0, d1, d2, oms);
            sp[i] = __builtin_fmaf(y2, q2, __builtin_fmaf(y1, q1, __builtin_fmaf(y0, q0, sp[i])));
            np[i] = __builtin_fmaf(y2, y2, __builtin_fmaf(y1, y1, __builtin_fmaf(y0, y0, np[i])));
            // (the empty asm pins this edge's arithmetic HERE: left alone, the optimiser sinks it to the accumulators' next use -- behind the phase's
            //  barrier, with every row of the phase still live in registers)
            asm volatile("" : "+v"(sp[i]), "+v"(np[i]));
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        phase_end();
    };
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        float sp[EPL];
#pragma unroll
        for (int i = 0; i < EPL; ++i) sp[i] = 0.f;
        for (int ph = 0; ph < PPH; ph += 2) {
            kphase(h * PPH + ph, lds, sp);
            kphase(h * PPH + ph + 1, lds + ST_SLICE_MAX, sp);
        }
        RED::run(sp, sc_[h], lane);
    }

    // ================================================================================================ soft-max over the 16 neighbours per head
    {
        const float inv_q = st_inv_fro(cs_sum<CS>(ssq));
        float invk[NS];
        RED::run(np, invk, lane);
#pragma unroll
        for (int i = 0; i < NS; ++i) invk[i] = st_inv_fro(invk[i]) * inv_q * inv_sqrt_dk;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < NS; ++i) { sc_[h][i] *= invk[i]; mx = fmaxf(mx, sc_[h][i]); }
            if constexpr (RED::SCAT) mx = cs_max<CS>(mx);      // the 8 edges of this edge group live in the quad's slots
            mx = eg_max<CS, EG>(mx);
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < NS; ++i) { sc_[h][i] = expf(sc_[h][i] - mx); sum += sc_[h][i]; }
            if constexpr (RED::SCAT) sum = cs_sum<CS>(sum);
            sum = eg_sum<CS, EG>(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int i = 0; i < NS; ++i) sc_[h][i] *= inv;
        }
    }

    // ================================================================================================ V pass
    float rm0 = 0.f, rm1 = 0.f, rm2 = 0.f;
    float* orow = out + (size_t)pid * 3 * CO + cl;
    auto vedge = [&](const Row& r, float w, const float2& qx, const float2& qy, const float2& qz, float& a0_, float& a1_, float& a2_) {
        float y0 = r.a0.x + qx.x, y1 = r.a1.x + qy.x, y2 = r.a2.x + qz.x;
        const float d0 = r.a0.y + qx.y, d1 = r.a1.y + qy.y, d2 = r.a2.y + qz.y;
        vn_act(y0, y1, y2, d0, d1, d2, oms);
        a0_ = __builtin_fmaf(w, y0, a0_); a1_ = __builtin_fmaf(w, y1, a1_); a2_ = __builtin_fmaf(w, y2, a2_);
    };
    auto vphase = [&](int tv, const char* sb, const float (&w)[NS]) {      // tv = 0 .. NSL - 1
        if (tv + 1 < NSL) stage(NSL + tv + 1);
        if (tv % PBV == 0) qblock(NKB + tv / PBV);
        const float* qr = qrow + ((tv % PBV) * CS + cl) * 2;
        const float2 qx = *reinterpret_cast<const float2*>(qr), qy = *reinterpret_cast<const float2*>(qr + 20), qz = *reinterpret_cast<const float2*>(qr + 40);
        float a0_ = 0.f, a1_ = 0.f, a2_ = 0.f;
        Row pr[2];
        pr[0] = ldrow(sb, noff[0]); pr[1] = ldrow(sb, noff[1]);
#define LS_VEDGE(I)                                                                   \
        if constexpr (I < EPL) {                                                      \
            const Row r = pr[I & 1];                                                  \
            if constexpr (I + 2 < EPL) pr[I & 1] = ldrow(sb, noff[I + 2 < EPL ? I + 2 : 0]); \
            __builtin_amdgcn_sched_barrier(0);                                        \
            vedge(r, RED::template get<I>(w), qx, qy, qz, a0_, a1_, a2_);             \
            asm volatile("" : "+v"(a0_), "+v"(a1_), "+v"(a2_));                       \
            __builtin_amdgcn_sched_barrier(0);                                        \
        }
#ifndef LS_ST_NOCOMPUTE
        LS_VEDGE(0) LS_VEDGE(1) LS_VEDGE(2) LS_VEDGE(3) LS_VEDGE(4) LS_VEDGE(5) LS_VEDGE(6) LS_VEDGE(7)
#endif
#undef LS_VEDGE
        a0_ = eg_sum<CS, EG>(a0_); a1_ = eg_sum<CS, EG>(a1_); a2_ = eg_sum<CS, EG>(a2_);
        if (eg == 0) {
            float* op = orow + tv * CS;
            op[0] = a0_; op[CO] = a1_; op[2 * CO] = a2_;
        }
        rm0 = fmaxf(rm0, fabsf(a0_)); rm1 = fmaxf(rm1, fabsf(a1_)); rm2 = fmaxf(rm2, fabsf(a2_));
        phase_end();
    };
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        for (int ph = 0; ph < PPH; ph += 2) {
            vphase(h * PPH + ph, lds, sc_[h]);
            vphase(h * PPH + ph + 1, lds + ST_SLICE_MAX, sc_[h]);
        }
    }
    if (rowmax) {   // max|out[row, :]| for the GEMM that reads `out` (gemm.hip, GemmAux)
        rm0 = cs_max<CS>(rm0); rm1 = cs_max<CS>(rm1); rm2 = cs_max<CS>(rm2);
        if (li == 0) { float* rp = rowmax + (size_t)pid * 3; rp[0] = rm0; rp[1] = rm1; rp[2] = rm2; }
    }
}

// ---------------------------------------------------------------------------------------------- host side
// which instantiation serves a layer shape: 1 = <2,4,32,64> (released layer 2), 2 = <4,2,64,64> (layer 3), 3 = <4,8,64,128> (layer 4); 0 = none
int edge_st_variant(int Co, int Cin) { return (Co == 64 && Cin == 32) ? 1 : (Co == 64 && Cin == 64) ? 2 : (Co == 128 && Cin == 64) ? 3 : 0; }
int edge_st_cs(int variant) { return variant == 1 ? 2 : 4; }
int edge_st_pts(int variant) { return variant == 3 ? 32 : 128; }
size_t edge_st_q_bytes(int Co, int Cin) {
    const int v = edge_st_variant(Co, Cin);
    if (!v) return 0;
    const int CS = edge_st_cs(v), NSL = Co / CS, nblk = NSL / (4 / CS) + NSL / (8 / CS);
    return (size_t)nblk * ((Cin / 32) * 2 + 1) * 1024;
}
// can the staged kernel take this problem?  (slices of at most 48 KB in whole KB, whole workgroups of destination points, whole 32-row store
// groups of the table GEMM inside an instance)
bool edge_st_fits(int Co, int Cin, int Ns, int Nd) {
    const int v = edge_st_variant(Co, Cin);
    if (!v) return false;
    const int rowb = 3 * edge_st_cs(v) * 8;
    const long long sb = (long long)Ns * rowb;
    return sb <= ST_SLICE_MAX && sb % 1024 == 0 && Nd % edge_st_pts(v) == 0 && (Ns * 3) % 32 == 0 && Ns >= SK;
}
int edge_st_prepare_launch(const float* W, int Co, int Cin, float* Wp, void* qplanes, hipStream_t st) {
    const int v = edge_st_variant(Co, Cin);
    LS_REQUIRE(v, "edge_staged: unsupported shape (Co=%d Cin=%d)", Co, Cin);
    const int CS = edge_st_cs(v), NSL = Co / CS, nblk = NSL / (4 / CS) + NSL / (8 / CS);
    hipLaunchKernelGGL(edge_st_permute_w_kernel, dim3(4 * Co), dim3(256), 0, st, W, Co, Cin, CS, Wp);
    hipLaunchKernelGGL(edge_st_presplit_q_kernel, dim3(nblk), dim3(64), 0, st, W + (size_t)4 * Co * Cin, Co, Cin, CS, (char*)qplanes);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int edge_attn_staged_launch(const float* T, const float* cur, int Cin, const void* qplanes, const int32_t* knn, const int32_t* dst_rows, int B, int Nd,
                            int Ns, int Co, int head_c, float neg_slope, float* out, hipStream_t st, float* rowmax) {
    LS_REQUIRE(head_c == 16 && edge_st_fits(Co, Cin, Ns, Nd) && qplanes, "edge_attn_staged: unsupported problem (Co=%d Cin=%d Ns=%d Nd=%d)", Co, Cin, Ns, Nd);
    const int v = edge_st_variant(Co, Cin), G = Nd / edge_st_pts(v);
    const float isd = 1.0f / sqrtf(3.0f * head_c), oms = 1.0f - neg_slope;
#define LS_ST(CS, EG, CIN, CO) hipLaunchKernelGGL((edge_attn_staged_kernel<CS, EG, CIN, CO>), dim3(B * G), dim3(ST_THREADS), 0, st, T, cur, (const char*)qplanes, knn, dst_rows, Nd, Ns, G, oms, isd, out, rowmax)
    if (v == 1) LS_ST(2, 4, 32, 64);
    else if (v == 2) LS_ST(4, 2, 64, 64);
    else LS_ST(4, 8, 64, 128);
#undef LS_ST
    LS_LAUNCH_CHECK();
    return LS_OK;
}

}  // namespace ls
