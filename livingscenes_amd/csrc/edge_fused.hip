// edge_fused.hip -- attention layers with FEW points per instance (released encoder: layers 5 and 6, 32 destination points, 128 / 32 source
// points, 128 -> 256 and 256 -> 512 channels): the folded VN-Linear tables are never written to memory.
//
// Replaces, for those layers, the table GEMM (gemm.hip) + edge_attn_v4_kernel (edge.hip) pair behind
//   /root/reference/lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:196-219   (get_graph_feature, V / K / Q VecLNA, QK soft-max attention)
//   /root/reference/lib_shape_prior/core/lib/vec_sim3/vec_layers.py:24-31,121-136,241-268   (cevn, VecLinear, VecActivation)
//
// Round 3 measured those two layers at 270 us of a 1.3 ms step: the table of layer 6 is 6 144 rows x 5 120 columns = 126 MB written by a
// K = 256 GEMM (96 us: short main loop, 256 KB epilogue per tile, 20 % of the matrix peak) and read back once by the attention kernel
// (54 us); layer 5 the same with 138 MB.  An instance of these layers is 32 x 3 (or 128 x 3) feature rows: a workgroup that owns
// (instance, HG heads) can form its table slice -- [rows] x [lin 16 | dir 16] columns per head and column-group pair -- with the f16 matrix
// cores straight into LDS and consume it there.  What keeps one workgroup from doing the whole layer is channel_equi_vec_normalize: the K
// feature of an edge is normalised by its Frobenius norm over ALL channels (vec_layers.py:24-31), i.e. over all heads.  Hence three launches:
//   edge_ft_qk_kernel     q = VecLNA_Q(dst), k = VecLNA_K(edge): per head the raw scores sum_c <k_c, q_c> and the partial squared norms
//                         |k|^2_head, |q|^2_head -> small global arrays [instance][head][point][neighbour]
//   edge_ft_norms_kernel  the partial norms summed over the heads in ascending head order (deterministic), once per instance -> 1 / |k|_F, 1 / |q|_F
//   edge_ft_v_kernel      soft-max over the 16 neighbours, v = VecLNA_V(edge) from its own table slice, weighted sum -> out [B, Nd, 3, Co]
//                         (+ one row maximum per head group for the GEMM that reads `out`: GemmAux a_parts)
// Operands: both MFMA operands arrive FRAGMENT-MAJOR (one coalesced 1 KB load per 32 x 16 operand block): the weights are split once per model
// (edge_ft_presplit_w_kernel: tile = (head, column-group pair), rows [lin 16 | dir 16], per-row power-of-two scale), the feature rows once per
// layer call (edge_ft_prep_a_kernel: per-row power-of-two scale, (hi, lo) f16 planes) -- the same two-piece split, the same three products per
// 16 k in the same order (l_a h_w, h_a h_w, h_a l_w) into one fp32 accumulator, ascending k, and the same integer-exponent scale in the epilogue as
// gemm.hip (the row scales come from the exact row maxima; the table GEMM may be handed an upper bound by its producer, so the two paths agree
// to fp32 round-off, not bit for bit).  The attention arithmetic uses the helpers of edge.hip (vn_act, dot43, fma43: every multiply-add spelled
// as an fma); what differs from edge_attn_v4_kernel is the ORDER in which the squared norms are summed over the channels (per head by a DPP quad
// sum, then over the heads ascending, instead of one 64-lane tree).
// [The numbers of this paragraph are those of the FIRST form of the GEMM phases (ft_gemm_phase: one 32 x 32 item at a time, resident W tile); the
//  phases now run as ft_gemm_phase_g3 -- see there: layers 5 / 6 92 / 115 -> 80 / 100 us, 242 - 254 -> 148 - 174 VGPRs, bench 52.7k -> 54.3k.]
// Measured (MI355X, B = 64, one step in flight, us per launch): layer 5 image 11 + q/k 37 + norms 6 + v 42 = 96 (table path 83 + 37 = 120);
// layer 6 image 5 + q/k 59 + norms 6 + v 47 = 117 (96 + 54 = 150); whole bench 48.1k -> 50.0k object-instances/s.  What bounds the two big
// kernels (timing variants, 12 steps in flight, us: whole | attention loops only | GEMM phases only): layer 6 q/k 56 | 18 | 45, v 50 | 18 | 31;
// layer 5 q/k 35 | 12 | 29, v 43 | 19 | 27.  The GEMM phases run at 20 - 30 % of the matrix peak: every wave streams its own copy of the A tiles
// through the L1 (layer 6, k phase: 4 waves x (96 KB of A + 32 KB of W) per 576 MFMAs = 28 B/clk per wave against 64 B/clk per CU).  Sharing the A
// image through LDS would cut that 2.3x, but 96 KB of planes + 52 KB of slabs leave one workgroup per CU.  BUILT AND MEASURED instead (round 4, not
// kept): (i) layer 6 with every A batch (4 k-steps, 8 KB) staged once per workgroup in a double-buffered 16 KB LDS ring and each wave keeping one
// W tile: q/k 56 -> 59 us, v 50 -> 46 us; (ii) two interleaved accumulator chains per tile (even / odd k-steps): no change -- a dependent MFMA chain is
// not what holds the phases back.  SQ counters (scripts/dev/sq_counters.sh, layer-6 q/k): a wave is parked on s_waitcnt / barriers 43 % of its
// life, issue-stalled behind its own MFMAs 28 %, issuing 28 %; the matrix pipes are busy 20 % of the launch.  With 8 waves per CU (242 - 254 VGPRs, 52 - 69 KB
// of LDS) nothing overlaps a workgroup's GEMM phase with another's attention phase except by chance; the structural remedy -- a persistent
// workgroup per instance whose MFMA waves run one head group ahead of its VALU waves -- is the next step, not a tweak of this one.
#include "ls_common.h"

namespace ls {

constexpr int FK = 16;            // neighbours per point
constexpr int FND = 32;           // destination points per instance (the fused path's shape: layers 5 / 6 of the released schedule)
typedef _Float16 fh8_t __attribute__((ext_vector_type(8)));
typedef _Float16 fh2_t __attribute__((ext_vector_type(2)));
typedef float ff2_t __attribute__((ext_vector_type(2)));
typedef float ff16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void fsplit_pair(ff2_t v, unsigned& h, unsigned& l) {
    const fh2_t hv = __builtin_convertvector(v, fh2_t);
    const fh2_t lv = __builtin_convertvector(v - __builtin_convertvector(hv, ff2_t), fh2_t);
    h = __builtin_bit_cast(unsigned, hv);
    l = __builtin_bit_cast(unsigned, lv);
}
// exact powers of two: s * amax in [2^14, 2^15), e = exponent of 1 / s  (gemm.hip: pow2_scale / pow2_e)
__device__ __forceinline__ void fpow2(float amax, float& s, int& e_inv) {
    unsigned be = (__float_as_uint(amax) >> 23) & 0xffu;
    be = be < 15u ? 15u : be;
    s = __uint_as_float((268u - be) << 23);
    e_inv = (int)be - 14 - 127;
}
template <int CTRL>
__device__ __forceinline__ float fdpp_max(float v) {
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false)));
}
template <int CTRL>
__device__ __forceinline__ float fdpp_add(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float fquad_sum(float v) { return fdpp_add<0x4E>(fdpp_add<0xB1>(v)); }
__device__ __forceinline__ float fquad_max(float v) { return fdpp_max<0x4E>(fdpp_max<0xB1>(v)); }
__device__ __forceinline__ float finv_fro(float ss) { return __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f)); }

// ---------------------------------------------------------------------------------------------------------------- operand images
// Weight tiles.  W [10 Co][Cin] (column groups PV_lin PV_dir PK_lin PK_dir QV_lin QV_dir QK_lin QK_dir Qq_lin Qq_dir, edge.hip header).
// Tile T = head * 5 + p, p in {0 PV, 1 PK, 2 QV, 3 QK, 4 Qq}: its 32 rows are [lin: W row 2p Co + 16 head + 0..15 | dir: W row (2p+1) Co + 16 head + 0..15].
// planes [T][ks][hi, lo][64 lanes] x 16 B (lane = row + 32 (k / 8 & 1), eight consecutive k), wexp [T * 32] = exponent of the row's inverse scale.
__global__ __launch_bounds__(64) void edge_ft_presplit_w_kernel(const float* __restrict__ W, int Co, int Cin, int KS, uint4* __restrict__ planes,
                                                                int* __restrict__ wexp) {
    const int T = blockIdx.x / KS, ks = blockIdx.x % KS, lane = threadIdx.x;
    const int head = T / 5, p = T % 5, r = lane & 31;
    const float* wr = W + ((size_t)(2 * p + (r >> 4)) * Co + 16 * head + (r & 15)) * Cin;
    float am = 0.f;
    for (int c = 0; c < Cin; ++c) am = fmaxf(am, fabsf(wr[c]));
    float sc;
    int e;
    fpow2(am, sc, e);
    if (ks == 0 && lane < 32) wexp[T * 32 + r] = e;
    const int k = ks * 16 + 8 * (lane >> 5);
    uint4 h, l;
    fsplit_pair(ff2_t{wr[k] * sc, wr[k + 1] * sc}, h.x, l.x);
    fsplit_pair(ff2_t{wr[k + 2] * sc, wr[k + 3] * sc}, h.y, l.y);
    fsplit_pair(ff2_t{wr[k + 4] * sc, wr[k + 5] * sc}, h.z, l.z);
    fsplit_pair(ff2_t{wr[k + 6] * sc, wr[k + 7] * sc}, h.w, l.w);
    planes[((size_t)blockIdx.x * 2) * 64 + lane] = h;
    planes[((size_t)blockIdx.x * 2 + 1) * 64 + lane] = l;
}
size_t edge_ft_w_bytes(int Co, int Cin) { return (size_t)(Co / 16) * 5 * (Cin / 16) * 2 * 64 * sizeof(uint4) + (size_t)(Co / 16) * 5 * 32 * sizeof(int); }
int edge_ft_presplit_w_launch(const float* W, int Co, int Cin, void* planes, hipStream_t st) {
    const int tiles = (Co / 16) * 5, KS = Cin / 16;
    hipLaunchKernelGGL(edge_ft_presplit_w_kernel, dim3(tiles * KS), dim3(64), 0, st, W, Co, Cin, KS, (uint4*)planes,
                       (int*)((uint4*)planes + (size_t)tiles * KS * 128));
    LS_LAUNCH_CHECK();
    return LS_OK;
}

// Feature rows -> A planes.  Output row R (tile R / 32): source row = rows ? (b Ns + rows[b Nd + n]) 3 + x with (b, n, x) from R : R.
// One workgroup per 32-row tile, its four waves take KS / 4 k-steps each (every load up front, the row maximum combined through LDS).
// planes [tile][ks][hi, lo][64 lanes] x 16 B, aexp [tile * 32 + row] = exponent of the row's inverse power-of-two scale.
template <int KS>
__global__ __launch_bounds__(256) void edge_ft_prep_a_kernel(const float* __restrict__ cur, const int32_t* __restrict__ rows, int Ns, int Nd,
                                                             uint4* __restrict__ planes, int* __restrict__ aexp) {
    constexpr int CIN = KS * 16, KPW = KS / 4;
    __shared__ float lmax[4][32];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int R = blockIdx.x * 32 + j;
    size_t src = (size_t)R;
    if (rows) {
        const int x = R % 3, pn = R / 3, b = pn / Nd;
        src = ((size_t)b * Ns + rows[pn]) * 3 + x;
    }
    const float* rp = cur + src * CIN + (w * KPW) * 16 + h * 8;
    float4 v[KPW][2];
    float am = 0.f;
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) {
        v[kk][0] = *reinterpret_cast<const float4*>(rp + kk * 16);
        v[kk][1] = *reinterpret_cast<const float4*>(rp + kk * 16 + 4);
    }
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk)
#pragma unroll
        for (int u = 0; u < 2; ++u)
            am = fmaxf(am, fmaxf(fmaxf(fabsf(v[kk][u].x), fabsf(v[kk][u].y)), fmaxf(fabsf(v[kk][u].z), fabsf(v[kk][u].w))));
    am = fmaxf(am, __shfl_xor(am, 32, 64));
    if (h == 0) lmax[w][j] = am;
    __syncthreads();
    am = fmaxf(fmaxf(lmax[0][j], lmax[1][j]), fmaxf(lmax[2][j], lmax[3][j]));
    float sc;
    int e;
    fpow2(am, sc, e);
    if (w == 0 && h == 0) aexp[R] = e;
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) {
        uint4 hh, ll;
        fsplit_pair(ff2_t{v[kk][0].x * sc, v[kk][0].y * sc}, hh.x, ll.x);
        fsplit_pair(ff2_t{v[kk][0].z * sc, v[kk][0].w * sc}, hh.y, ll.y);
        fsplit_pair(ff2_t{v[kk][1].x * sc, v[kk][1].y * sc}, hh.z, ll.z);
        fsplit_pair(ff2_t{v[kk][1].z * sc, v[kk][1].w * sc}, hh.w, ll.w);
        uint4* op = planes + (((size_t)blockIdx.x * KS + w * KPW + kk) * 2) * 64 + lane;
        op[0] = hh;
        op[64] = ll;
    }
}

// ---------------------------------------------------------------------------------------------------------------- shared pieces
struct FT43 { float4 x, y, z; };
__device__ __forceinline__ FT43 ft_lds43(const float* p, int sld) {
    FT43 r;
    r.x = *reinterpret_cast<const float4*>(p);
    r.y = *reinterpret_cast<const float4*>(p + sld);
    r.z = *reinterpret_cast<const float4*>(p + 2 * sld);
    return r;
}
__device__ __forceinline__ FT43 ft_add43(const FT43& a, const FT43& b) {
    FT43 r;
    r.x = make_float4(a.x.x + b.x.x, a.x.y + b.x.y, a.x.z + b.x.z, a.x.w + b.x.w);
    r.y = make_float4(a.y.x + b.y.x, a.y.y + b.y.y, a.y.z + b.y.z, a.y.w + b.y.w);
    r.z = make_float4(a.z.x + b.z.x, a.z.y + b.z.y, a.z.z + b.z.z, a.z.w + b.z.w);
    return r;
}
__device__ __forceinline__ void ft_act43(FT43& y, const FT43& k, float oms) {
    vn_act(y.x.x, y.y.x, y.z.x, k.x.x, k.y.x, k.z.x, oms);
    vn_act(y.x.y, y.y.y, y.z.y, k.x.y, k.y.y, k.z.y, oms);
    vn_act(y.x.z, y.y.z, y.z.z, k.x.z, k.y.z, k.z.z, oms);
    vn_act(y.x.w, y.y.w, y.z.w, k.x.w, k.y.w, k.z.w, oms);
}
// (explicit fma chain, the order of edge.hip's dot43)
__device__ __forceinline__ float ft_dot43(const FT43& a, const FT43& b) {
    float s = a.x.x * b.x.x;
    s = __builtin_fmaf(a.y.x, b.y.x, s); s = __builtin_fmaf(a.z.x, b.z.x, s);
    s = __builtin_fmaf(a.x.y, b.x.y, s); s = __builtin_fmaf(a.y.y, b.y.y, s); s = __builtin_fmaf(a.z.y, b.z.y, s);
    s = __builtin_fmaf(a.x.z, b.x.z, s); s = __builtin_fmaf(a.y.z, b.y.z, s); s = __builtin_fmaf(a.z.z, b.z.z, s);
    s = __builtin_fmaf(a.x.w, b.x.w, s); s = __builtin_fmaf(a.y.w, b.y.w, s); s = __builtin_fmaf(a.z.w, b.z.w, s);
    return s;
}
__device__ __forceinline__ void ft_fma43(FT43& acc, float w, const FT43& y) {
    acc.x.x = __builtin_fmaf(w, y.x.x, acc.x.x); acc.x.y = __builtin_fmaf(w, y.x.y, acc.x.y); acc.x.z = __builtin_fmaf(w, y.x.z, acc.x.z); acc.x.w = __builtin_fmaf(w, y.x.w, acc.x.w);
    acc.y.x = __builtin_fmaf(w, y.y.x, acc.y.x); acc.y.y = __builtin_fmaf(w, y.y.y, acc.y.y); acc.y.z = __builtin_fmaf(w, y.y.z, acc.y.z); acc.y.w = __builtin_fmaf(w, y.y.w, acc.y.w);
    acc.z.x = __builtin_fmaf(w, y.z.x, acc.z.x); acc.z.y = __builtin_fmaf(w, y.z.y, acc.z.y); acc.z.z = __builtin_fmaf(w, y.z.z, acc.z.z); acc.z.w = __builtin_fmaf(w, y.z.w, acc.z.w);
}

// One "job" of a phase: slab[rows of the instance][HG heads x (lin 16 | dir 16)] = A rows . W tile^T for weight pair `p` of the workgroup's heads.
//   A: the instance's M-tiles mt0 .. mt0 + MT of an A-plane image; W: tile (head0 + hl) * 5 + p.
template <int KS>
struct FtJob { const uint4* a_planes; const int* a_exp; int mt0; int MT; int p; float* slab; int sld; };

// (First form, until late round 4 -- kept as a note, the code is gone: the flattened (head-local, M-tile) list of a phase cut into four contiguous pieces,
//  one 32 x 32 item at a time, the item's W tile resident in registers (KS x 8 VGPRs), its A fragments in batches of eight k-steps, double-buffered
//  whole tiles at K = 128.  Lessons that still hold: `load, load, 3 MFMA` per k-step makes hipcc wait vmcnt(0) before every MFMA group -- request a
//  batch, fence, then consume it; a run-time-trip-count loop of dependent global loads serialises.)
// The GEMM phase: THREE M-tiles per unit of work and nothing resident.  Why: the first form kept a W tile in registers (128 VGPRs at
// K = 256) -- 242 - 254 VGPRs per wave, i.e. the two workgroups of a CU own its whole register file while their waves sit through ~13 sequential L2
// round trips per q / k phase pair, and LS_SKIP=hi32 shows that these layers cost 0.165 ms of the 1.21 ms step in steady state: nothing can be
// resident beside them.  Here a unit = (job, head, three consecutive M-tiles = the 96 feature rows of 32 points): three independent accumulators,
// and per batch of two k-steps 12 A fragments + 4 W fragments (64 VGPRs) are requested, awaited and consumed by 18 MFMAs.  Same products in the same
// order per accumulator (ascending k: l_a h_w, h_a h_w, h_a l_w) => bit-identical slabs; the W tile is streamed once per unit, as before once per
// (job, head); ~120 VGPRs.  Every job's M-tile count is a multiple of three by construction (96 or 384 rows).
// (Order: the operand images are read-only no-alias memory and the loop stores nothing, so only data dependencies keep hipcc from hoisting all sixteen
//  batches' loads to the top -- the batch's lane offset passes through a volatile asm, and so do the accumulators after its MFMAs: edge.hip, pool kernel.)
#ifndef LS_FT_KB
#define LS_FT_KB 2
#endif
// one unit of work: MPU consecutive M-tiles (first: tile `tile0` of the job) x one weight tile -> the job's slab
template <int KS, int MPU>
__device__ __forceinline__ void ft_gemm_unit(const FtJob<KS>& jb, int tile0, int hl, int T, const uint4* __restrict__ wplanes, const int* __restrict__ wexp, int lane) {
    const char* ab = reinterpret_cast<const char*>(jb.a_planes + ((size_t)(jb.mt0 + tile0) * KS * 2) * 64);
    const char* wb = reinterpret_cast<const char*>(wplanes + ((size_t)T * KS * 2) * 64);
    const int we = wexp[T * 32 + (lane & 31)];
    ff16_t acc[MPU];
#pragma unroll
    for (int m = 0; m < MPU; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    unsigned voff = (unsigned)lane * 16u;
    constexpr int KB = LS_FT_KB;   // k-steps per batch
    struct Stage { fh8_t ah[MPU][KB], al[MPU][KB], bh[KB], bl[KB]; };
    auto load_stage = [&](Stage& sg, int k0) {
        asm volatile("" : "+v"(voff));
#pragma unroll
        for (int u = 0; u < KB; ++u) {
            sg.bh[u] = __builtin_bit_cast(fh8_t, *reinterpret_cast<const uint4*>(wb + (size_t)((k0 + u) * 2) * 1024 + voff));
            sg.bl[u] = __builtin_bit_cast(fh8_t, *reinterpret_cast<const uint4*>(wb + (size_t)((k0 + u) * 2 + 1) * 1024 + voff));
#pragma unroll
            for (int m = 0; m < MPU; ++m) {
                sg.ah[m][u] = __builtin_bit_cast(fh8_t, *reinterpret_cast<const uint4*>(ab + (size_t)((m * KS + k0 + u) * 2) * 1024 + voff));
                sg.al[m][u] = __builtin_bit_cast(fh8_t, *reinterpret_cast<const uint4*>(ab + (size_t)((m * KS + k0 + u) * 2 + 1) * 1024 + voff));
            }
        }
    };
    auto run_stage = [&](const Stage& sg) {
#pragma unroll
        for (int u = 0; u < KB; ++u)
#pragma unroll
            for (int m = 0; m < MPU; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sg.al[m][u], sg.bh[u], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sg.ah[m][u], sg.bh[u], acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sg.ah[m][u], sg.bl[u], acc[m], 0, 0, 0);
            }
#pragma unroll
        for (int m = 0; m < MPU; ++m) asm volatile("" : "+v"(acc[m])::"memory");
    };
    // (Two register stages -- batch b + 1 requested before batch b's MFMAs -- were built and measured: layers 5 / 6 79.3 / 103.4 us against 80.7 / 102.9 us
    //  alone, bench 53.96k against 54.37k: 64 more VGPRs buy no overlap that an L2 round trip of ~2 us against 576 matrix-pipe cycles could use.
    //  Batches of one k-step (137 - 165 VGPRs): 88 / 105 us, 53.5k; of four (212 - 238): 82 / 112 us, 53.4k; two it is: 81 / 102 us, 54.1k.)
#pragma unroll
    for (int k0 = 0; k0 < KS; k0 += KB) {
        Stage sg;
        load_stage(sg, k0);
        __builtin_amdgcn_sched_barrier(0);   // (all requests of the batch first: the scheduler would sink each load to its MFMA)
        run_stage(sg);
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int m = 0; m < MPU; ++m) {
        const int mt = tile0 + m;
        const int row0 = 32 * mt + 4 * (lane >> 5);
        float* sp = jb.slab + (size_t)row0 * jb.sld + hl * 32 + (lane & 31);
        const int* ae = jb.a_exp + (jb.mt0 + mt) * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            sp[dr * jb.sld] = __builtin_ldexpf(acc[m][r], ae[dr] + we);
        }
    }
}
template <int KS, int HG, int NJ>
__device__ __forceinline__ void ft_gemm_phase_g3(const FtJob<KS> (&jobs)[NJ], const uint4* __restrict__ wplanes, const int* __restrict__ wexp, int head0,
                                                 int wave, int lane) {
    // Layer 5's two-job phases (one head; 12 source M-tiles + 3 destination M-tiles) were FIVE three-tile units for four waves: wave 0 ran two of them, the
    // phase took six tile-times.  Round 5: the four source groups first, one per wave, then the destination job as three single-tile units on waves 0 - 2:
    // four tile-times (the weight tile of the small job is streamed three times instead of once).  Same products per accumulator: identical slabs.
    if constexpr (NJ == 2 && HG == 1) {
        if (jobs[0].MT == 12 && jobs[1].MT == 3) {     // (kernel-uniform)
            ft_gemm_unit<KS, 3>(jobs[0], 3 * wave, 0, head0 * 5 + jobs[0].p, wplanes, wexp, lane);
            if (wave < 3) ft_gemm_unit<KS, 1>(jobs[1], wave, 0, head0 * 5 + jobs[1].p, wplanes, wexp, lane);
            return;
        }
    }
    int total = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) total += (jobs[j].MT / 3) * HG;
    for (int t = wave; t < total; t += 4) {      // (wave-uniform)
        int j = 0, off = t;
#pragma unroll
        for (int q = 0; q < NJ - 1; ++q)
            if (j == q && off >= (jobs[q].MT / 3) * HG) { off -= (jobs[q].MT / 3) * HG; j = q + 1; }
        FtJob<KS> jb = jobs[0];
#pragma unroll
        for (int q = 1; q < NJ; ++q)
            if (j == q) jb = jobs[q];
        const int ng = jb.MT / 3, hl = off / ng, g = off - hl * ng;
        ft_gemm_unit<KS, 3>(jb, 3 * g, hl, (head0 + hl) * 5 + jb.p, wplanes, wexp, lane);
    }
}

// The same phase with the A fragments SHARED through LDS (round 5, layer 6 only: its jobs all read the SAME three M-tiles -- source and destination points coincide --
// and its units are (job, head) pairs, one per wave).  ft_gemm_phase_g3 lets every wave stream its own copy of the A batch from L2 (12 of its 16 KB per two k-steps):
// the L1 does not merge the four waves' misses, so the layer-6 kernels ask the L2 for 8.4 M requests per step, 8 % of everything it serves (profiles/r5_final/
// l2_requests_per_kernel.txt).  Here the four waves fetch a quarter of the batch each (three fragments) into a double-buffered 24 KB ring, one barrier per batch; the W
// fragments stay per wave.  Same products in the same order per accumulator: bit-identical slabs.  (All four waves run the loop -- a wave without a unit only feeds the ring.)
#ifndef LS_FT_SHARE
#define LS_FT_SHARE 1      // (0: every wave streams its own A copy -- dev A/B: attention layer 6 101 -> 97 us one step in flight, bench +0.1 .. +1.4 % on the same box, L2 requests of the layer-6 kernels -50 %)
#endif
template <int KS, int HG, int NJ>
__device__ __forceinline__ void ft_gemm_phase_shared(const FtJob<KS> (&jobs)[NJ], const uint4* __restrict__ wplanes, const int* __restrict__ wexp, int head0,
                                                     int wave, int lane, uint4* ring) {
    // unit of this wave: job wave / HG, head wave % HG (NJ * HG <= 4)
    const bool has_unit = wave < NJ * HG;
    const int j = has_unit ? wave / HG : 0, hl = wave % HG;
    FtJob<KS> jb = jobs[0];
#pragma unroll
    for (int q = 1; q < NJ; ++q)
        if (j == q) jb = jobs[q];
    const int T = (head0 + hl) * 5 + jb.p;
    const char* ab = reinterpret_cast<const char*>(jobs[0].a_planes + ((size_t)jobs[0].mt0 * KS * 2) * 64);     // (every job: the same three M-tiles)
    const char* wb = reinterpret_cast<const char*>(wplanes + ((size_t)T * KS * 2) * 64);
    const int we = wexp[T * 32 + (lane & 31)];
    ff16_t acc[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    unsigned voff = (unsigned)lane * 16u;
    // fragment f of a batch (two k-steps k0, k0 + 1): u = f / 6 (k-step), m = (f % 6) / 2 (M-tile), pl = f % 2 (hi | lo plane)
    auto a_addr = [&](int k0, int f) { const int u = f / 6, m = (f % 6) / 2, pl = f % 2; return ab + (size_t)((m * KS + k0 + u) * 2 + pl) * 1024 + voff; };
    // THREE batches of loads in flight per wave (the set consumed in step s was requested in step s - 3): with the A fragments shared a batch costs a wave only
    // 7 x 16 bytes per lane, so three sets are 84 VGPRs -- the private-copy form had room for one batch (64 VGPRs), and every one of a phase's eight batches waited
    // ~2 us for its L2 round trip in front of 0.24 us of MFMAs.  (Register sets indexed by compile-time constants of the fully unrolled loop, filled by a macro:
    // arrays written inside a lambda stay in scratch memory.)
    constexpr int NB = KS / 2;
    uint4 sA_r0, sA_r1, sA_r2, sA_w0, sA_w1, sA_w2, sA_w3, sB_r0, sB_r1, sB_r2, sB_w0, sB_w1, sB_w2, sB_w3, sC_r0, sC_r1, sC_r2, sC_w0, sC_w1, sC_w2, sC_w3;
#define LS_FT_LOAD_BATCH(S, K0)                                                                           \
    {                                                                                                     \
        asm volatile("" : "+v"(voff));                                                                    \
        S##_r0 = *reinterpret_cast<const uint4*>(a_addr((K0), wave * 3 + 0));                             \
        S##_r1 = *reinterpret_cast<const uint4*>(a_addr((K0), wave * 3 + 1));                             \
        S##_r2 = *reinterpret_cast<const uint4*>(a_addr((K0), wave * 3 + 2));                             \
        S##_w0 = *reinterpret_cast<const uint4*>(wb + (size_t)(((K0) + 0) * 2) * 1024 + voff);            \
        S##_w1 = *reinterpret_cast<const uint4*>(wb + (size_t)(((K0) + 0) * 2 + 1) * 1024 + voff);        \
        S##_w2 = *reinterpret_cast<const uint4*>(wb + (size_t)(((K0) + 1) * 2) * 1024 + voff);            \
        S##_w3 = *reinterpret_cast<const uint4*>(wb + (size_t)(((K0) + 1) * 2 + 1) * 1024 + voff);        \
    }
#define LS_FT_STEP(S, BI)                                                                                                                          \
    {                                                                                                                                              \
        uint4* rb = ring + ((BI) & 1) * 12 * 64;                                                                                                   \
        rb[(wave * 3 + 0) * 64 + lane] = S##_r0;                                                                                                   \
        rb[(wave * 3 + 1) * 64 + lane] = S##_r1;                                                                                                   \
        rb[(wave * 3 + 2) * 64 + lane] = S##_r2;                                                                                                   \
        const fh8_t bh0 = __builtin_bit_cast(fh8_t, S##_w0), bl0 = __builtin_bit_cast(fh8_t, S##_w1), bh1 = __builtin_bit_cast(fh8_t, S##_w2),     \
                    bl1 = __builtin_bit_cast(fh8_t, S##_w3);                                                                                       \
        if ((BI) + 3 < NB) LS_FT_LOAD_BATCH(S, 2 * ((BI) + 3))                                                                                     \
        __syncthreads();                                                                                                                           \
        if (has_unit) {                                                                                                                            \
            _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                                                                        \
                const fh8_t bh = u ? bh1 : bh0, bl = u ? bl1 : bl0;                                                                                \
                _Pragma("unroll") for (int m = 0; m < 3; ++m) {                                                                                    \
                    const fh8_t ah = __builtin_bit_cast(fh8_t, rb[(u * 6 + m * 2) * 64 + lane]), al = __builtin_bit_cast(fh8_t, rb[(u * 6 + m * 2 + 1) * 64 + lane]); \
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[m], 0, 0, 0);                                                      \
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[m], 0, 0, 0);                                                      \
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[m], 0, 0, 0);                                                      \
                }                                                                                                                                  \
            }                                                                                                                                      \
        }                                                                                                                                          \
    }
    LS_FT_LOAD_BATCH(sA, 0)
    if (1 < NB) LS_FT_LOAD_BATCH(sB, 2)
    if (2 < NB) LS_FT_LOAD_BATCH(sC, 4)
#pragma unroll
    for (int bi = 0; bi < NB; bi += 3) {
        LS_FT_STEP(sA, bi)
        if (bi + 1 < NB) LS_FT_STEP(sB, bi + 1)
        if (bi + 2 < NB) LS_FT_STEP(sC, bi + 2)
    }
#undef LS_FT_STEP
#undef LS_FT_LOAD_BATCH
    if (has_unit) {
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            const int row0 = 32 * m + 4 * (lane >> 5);
            float* sp = jb.slab + (size_t)row0 * jb.sld + hl * 32 + (lane & 31);
            const int* ae = jb.a_exp + (jb.mt0 + m) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                sp[dr * jb.sld] = __builtin_ldexpf(acc[m][r], ae[dr] + we);
            }
        }
    }
}
#define LS_FT_PHASE ft_gemm_phase_g3

// thread -> (point, head-local, quad lane, neighbour range) of the attention phases: 256 threads = 32 points x HG heads x 4 lanes x (2 / HG) neighbour halves
template <int HG>
struct FtMap {
    int n, hl, ql, k0, kn;
    __device__ __forceinline__ FtMap(int tid) {
        ql = tid & 3;
        const int idx = tid >> 2;   // 0 .. 63
        n = idx >> 1;
        if constexpr (HG == 2) { hl = idx & 1; k0 = 0; kn = FK; }
        else { hl = 0; k0 = (idx & 1) * (FK / 2); kn = FK / 2; }
    }
};

// ---------------------------------------------------------------------------------------------------------------- kernel 1: q and k
// grid = (Co / 16 / HG) x B workgroups, instance major by default (edge_ft_attn_launch).
// NS = source points per instance (32: destination set == source set, one A image; 128: destination points selected by dst_rows, own A image).
template <int CIN, int NS, int HG>
__global__ __launch_bounds__(256, 2) void edge_ft_qk_kernel(const uint4* __restrict__ a_p, const int* __restrict__ ae_p, const uint4* __restrict__ a_q,
                                                         const int* __restrict__ ae_q, const uint4* __restrict__ wplanes, const int* __restrict__ wexp,
                                                         const int32_t* __restrict__ knn, int B, int H, float oms, float* __restrict__ scores,
                                                         float* __restrict__ sskp, float* __restrict__ ssqp, int inst_major) {
    constexpr int KS = CIN / 16, MTP = NS * 3 / 32, MTQ = FND * 3 / 32, SLD = HG * 32 + 4;
    __shared__ __attribute__((aligned(16))) float slab_q[FND * 3 * SLD];    // q phase: Qq (lin | dir); k phase: QK (lin | dir) of the destination points
    __shared__ __attribute__((aligned(16))) float slab_p[NS * 3 * SLD];     // k phase: PK (lin | dir) of the source points
    __shared__ __attribute__((aligned(16))) uint4 a_ring[(LS_FT_SHARE && NS == 32) ? 2 * 12 * 64 : 1];   // shared A batches (ft_gemm_phase_shared)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int ngroups = H / HG;
    const int hg = inst_major ? logical % ngroups : logical / B, b = inst_major ? logical / ngroups : logical % B, head0 = hg * HG;
    const FtMap<HG> mp(tid);
    const int head = head0 + mp.hl;

    // ---- q = VecLNA_Q(dst_f[n]) of the workgroup's heads (vec_dgcnn_atten.py:207,210)
    {
        const FtJob<KS> jq[1] = {{a_q, ae_q, b * MTQ, MTQ, 4, slab_q, SLD}};
        if constexpr (LS_FT_SHARE && NS == 32) ft_gemm_phase_shared<KS, HG, 1>(jq, wplanes, wexp, head0, wave, lane, a_ring);
        else LS_FT_PHASE<KS, HG, 1>(jq, wplanes, wexp, head0, wave, lane);
    }
    __syncthreads();
    const int c4 = mp.hl * 32 + mp.ql * 4;
    FT43 qf = ft_lds43(slab_q + (size_t)(3 * mp.n) * SLD + c4, SLD);
    {
        const FT43 kd = ft_lds43(slab_q + (size_t)(3 * mp.n) * SLD + c4 + 16, SLD);
        ft_act43(qf, kd, oms);
    }
    const float ssq = fquad_sum(ft_dot43(qf, qf));
    if (mp.ql == 0 && mp.k0 == 0) ssqp[((size_t)b * H + head) * FND + mp.n] = ssq;
    int nb[FK];
    {
        const int4* kp = reinterpret_cast<const int4*>(knn + ((size_t)b * FND + mp.n) * FK + mp.k0);
#pragma unroll
        for (int u = 0; u < FK / 4; ++u)
            if (4 * u < mp.kn) { const int4 v = kp[u]; nb[4 * u] = v.x; nb[4 * u + 1] = v.y; nb[4 * u + 2] = v.z; nb[4 * u + 3] = v.w; }
    }
    __syncthreads();   // every thread holds its q: slab_q may be overwritten

    // ---- k = VecLNA_K(E[n, k]) = act(PK_lin[nbr] + QK_lin[n], PK_dir[nbr] + QK_dir[n])  (:206,209)
    {
        const FtJob<KS> jk[2] = {{a_p, ae_p, b * MTP, MTP, 1, slab_p, SLD}, {a_q, ae_q, b * MTQ, MTQ, 3, slab_q, SLD}};
        if constexpr (LS_FT_SHARE && NS == 32) ft_gemm_phase_shared<KS, HG, 2>(jk, wplanes, wexp, head0, wave, lane, a_ring);
        else LS_FT_PHASE<KS, HG, 2>(jk, wplanes, wexp, head0, wave, lane);
    }
    __syncthreads();
    {
        const FT43 ql = ft_lds43(slab_q + (size_t)(3 * mp.n) * SLD + c4, SLD), qd = ft_lds43(slab_q + (size_t)(3 * mp.n) * SLD + c4 + 16, SLD);
        float* so = scores + (((size_t)b * H + head) * FND + mp.n) * FK + mp.k0;
        float* ko = sskp + (((size_t)b * H + head) * FND + mp.n) * FK + mp.k0;
#pragma unroll
        for (int k = 0; k < FK; ++k) {
            if (k < mp.kn) {
                const float* pr = slab_p + (size_t)(3 * nb[k]) * SLD + c4;
                FT43 y = ft_add43(ft_lds43(pr, SLD), ql);
                const FT43 kd = ft_add43(ft_lds43(pr + 16, SLD), qd);
                ft_act43(y, kd, oms);
                const float s2 = fquad_sum(ft_dot43(y, y));
                const float a = fquad_sum(ft_dot43(y, qf));
                if (mp.ql == 0) { so[k] = a; ko[k] = s2; }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- between the two: the norms
// Frobenius norms over ALL channels (channel_equi_vec_normalize, vec_layers.py:24-31): the per-head partial sums of kernel 1 added in ascending
// head order (deterministic), once per instance -- inside kernel 2 every one of an instance's 16 workgroups repeated these 32-term sums, and a
// loop over a run-time head count serialised its loads (one L2 round trip per term: 45 us of a 91 us launch).
// invk [B][Nd][16] = 1 / max(|k(n, k)|_F, 1e-12), invq [B][Nd]; grid = B, 256 threads.
__global__ __launch_bounds__(256) void edge_ft_norms_kernel(const float* __restrict__ sskp, const float* __restrict__ ssqp, int H, float* __restrict__ invk,
                                                            float* __restrict__ invq) {
    LS_LATENCY_CRITICAL();
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int e = tid; e < FND * FK + FND; e += 256) {
        const bool isk = e < FND * FK;
        const int stride = isk ? FND * FK : FND;
        const float* sp = isk ? sskp + (size_t)b * H * FND * FK + e : ssqp + (size_t)b * H * FND + (e - FND * FK);
        float s = 0.f;
        int h = 0;
        for (; h + 8 <= H; h += 8) {        // eight loads in flight, added in ascending head order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sp[(size_t)(h + u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; h < H; ++h) s += sp[(size_t)h * stride];
        if (isk) invk[(size_t)b * FND * FK + e] = finv_fro(s);
        else invq[(size_t)b * FND + (e - FND * FK)] = finv_fro(s);
    }
}

// ---------------------------------------------------------------------------------------------------------------- kernel 2: soft-max and v
#ifndef LS_FT_V_WPE
#define LS_FT_V_WPE 2      // dev A/B: waves per SIMD the layer-6 instance (NS == 32, 52 KB of LDS) is compiled for
#endif
template <int CIN, int NS, int HG>
__global__ __launch_bounds__(256, (NS == 32 ? LS_FT_V_WPE : 2)) void edge_ft_v_kernel(const uint4* __restrict__ a_p, const int* __restrict__ ae_p, const uint4* __restrict__ a_q,
                                                        const int* __restrict__ ae_q, const uint4* __restrict__ wplanes, const int* __restrict__ wexp,
                                                        const int32_t* __restrict__ knn, int B, int H, float oms, float inv_sqrt_dk,
                                                        const float* __restrict__ scores, const float* __restrict__ invk, const float* __restrict__ invq,
                                                        float* __restrict__ out, int Co, float* __restrict__ rowmax, int rm_parts, int inst_major,
                                                        float* __restrict__ colsum) {
    // colsum (nullable) [B][3][Co]: the sums of `out` over the instance's 32 points (ascending) -- the mean of the residual global conv
    // (vec_dgcnn_atten.py:223) without a second pass over `out` (glob_mean_gemv_kernel, pointwise.hip)
    constexpr int KS = CIN / 16, MTP = NS * 3 / 32, MTQ = FND * 3 / 32, SLD = HG * 32 + 4;
    __shared__ __attribute__((aligned(16))) float slab_q[FND * 3 * SLD];    // QV (lin | dir) of the destination points
    __shared__ __attribute__((aligned(16))) float slab_p[NS * 3 * SLD];     // PV (lin | dir) of the source points
    __shared__ __attribute__((aligned(16))) uint4 a_ring[(LS_FT_SHARE && NS == 32) ? 2 * 12 * 64 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int ngroups = H / HG;
    const int hg = inst_major ? logical % ngroups : logical / B, b = inst_major ? logical / ngroups : logical % B, head0 = hg * HG;
    const FtMap<HG> mp(tid);
    const int head = head0 + mp.hl;

    // the point's neighbour list, raw scores and norms: issued before anything else, consumed after the v tables are formed (a load per
    // neighbour inside the weighted-sum loop would expose one L2 round trip each)
    int nb[FK];
    float4 sraw[FK / 4], ik4[FK / 4];
    float iq;
    {
        const int4* kp = reinterpret_cast<const int4*>(knn + ((size_t)b * FND + mp.n) * FK);
        const float4* sp = reinterpret_cast<const float4*>(scores + (((size_t)b * H + head) * FND + mp.n) * FK);
        const float4* ip = reinterpret_cast<const float4*>(invk + ((size_t)b * FND + mp.n) * FK);
#pragma unroll
        for (int u = 0; u < FK / 4; ++u) {
            const int4 v = kp[u];
            nb[4 * u] = v.x; nb[4 * u + 1] = v.y; nb[4 * u + 2] = v.z; nb[4 * u + 3] = v.w;
            sraw[u] = sp[u];
            ik4[u] = ip[u];
        }
        iq = invq[(size_t)b * FND + mp.n] * inv_sqrt_dk;
    }
    // ---- v tables of the workgroup's heads
    {
        const FtJob<KS> jv[2] = {{a_p, ae_p, b * MTP, MTP, 0, slab_p, SLD}, {a_q, ae_q, b * MTQ, MTQ, 2, slab_q, SLD}};
        if constexpr (LS_FT_SHARE && NS == 32) ft_gemm_phase_shared<KS, HG, 2>(jv, wplanes, wexp, head0, wave, lane, a_ring);
        else LS_FT_PHASE<KS, HG, 2>(jv, wplanes, wexp, head0, wave, lane);
    }
    __syncthreads();
    // ---- soft-max over the 16 neighbours of (point, head) (:211-215); every lane of the (point, head) group computes it for itself
    float wgt[FK];
    float sum = 0.f;
    {
        float mx = -INFINITY;
#pragma unroll
        for (int u = 0; u < FK / 4; ++u) {
            const float4 v = sraw[u], ik = ik4[u];
            wgt[4 * u] = v.x * iq * ik.x;
            wgt[4 * u + 1] = v.y * iq * ik.y;
            wgt[4 * u + 2] = v.z * iq * ik.z;
            wgt[4 * u + 3] = v.w * iq * ik.w;
        }
#pragma unroll
        for (int k = 0; k < FK; ++k) mx = fmaxf(mx, wgt[k]);
#pragma unroll
        for (int k = 0; k < FK; ++k) { wgt[k] = expf(wgt[k] - mx); sum += wgt[k]; }
    }
    // ---- out = sum_k softmax_k * VecLNA_V(E[n, k])  (:208,216-219)
    const int c4 = mp.hl * 32 + mp.ql * 4;
    FT43 acc;
    acc.x = acc.y = acc.z = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const FT43 ql = ft_lds43(slab_q + (size_t)(3 * mp.n) * SLD + c4, SLD), qd = ft_lds43(slab_q + (size_t)(3 * mp.n) * SLD + c4 + 16, SLD);
#pragma unroll
        for (int k = 0; k < FK; ++k) {
            if (k >= mp.k0 && k < mp.k0 + mp.kn) {
                const float* pr = slab_p + (size_t)(3 * nb[k]) * SLD + c4;
                FT43 y = ft_add43(ft_lds43(pr, SLD), ql);
                const FT43 kd = ft_add43(ft_lds43(pr + 16, SLD), qd);
                ft_act43(y, kd, oms);
                ft_fma43(acc, wgt[k], y);
            }
        }
    }
    if constexpr (HG == 1) {   // the two neighbour halves of a point sit four lanes apart: lower half + upper half (fixed order)
        auto comb = [&](float v) { const float o = __shfl_xor(v, 4, 64); return (tid & 4) ? o + v : v + o; };
        acc.x = make_float4(comb(acc.x.x), comb(acc.x.y), comb(acc.x.z), comb(acc.x.w));
        acc.y = make_float4(comb(acc.y.x), comb(acc.y.y), comb(acc.y.z), comb(acc.y.w));
        acc.z = make_float4(comb(acc.z.x), comb(acc.z.y), comb(acc.z.z), comb(acc.z.w));
    }
    const float inv = 1.0f / sum;
    const float4 ox = make_float4(acc.x.x * inv, acc.x.y * inv, acc.x.z * inv, acc.x.w * inv);
    const float4 oy = make_float4(acc.y.x * inv, acc.y.y * inv, acc.y.z * inv, acc.y.w * inv);
    const float4 oz = make_float4(acc.z.x * inv, acc.z.y * inv, acc.z.z * inv, acc.z.w * inv);
    const bool writer = HG == 2 || (tid & 4) == 0;
    if (writer) {
        float* op = out + ((size_t)b * FND + mp.n) * 3 * Co + head * 16 + mp.ql * 4;
        *reinterpret_cast<float4*>(op) = ox;
        *reinterpret_cast<float4*>(op + Co) = oy;
        *reinterpret_cast<float4*>(op + 2 * Co) = oz;
    }
    if (rowmax) {   // max |out[row, the workgroup's 16 HG channels]| -> part hg of the row's maxima (GemmAux: a_parts = Co / (16 HG))
        auto amax4 = [](const float4& v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); };
        float rx = fquad_max(amax4(ox)), ry = fquad_max(amax4(oy)), rz = fquad_max(amax4(oz));
        if constexpr (HG == 2) {   // the two heads of a point sit four lanes apart
            rx = fmaxf(rx, __shfl_xor(rx, 4, 64)); ry = fmaxf(ry, __shfl_xor(ry, 4, 64)); rz = fmaxf(rz, __shfl_xor(rz, 4, 64));
        }
        if ((tid & 7) == 0) {
            float* rp = rowmax + (((size_t)b * FND + mp.n) * 3) * rm_parts + hg;
            rp[0] = rx; rp[rm_parts] = ry; rp[2 * rm_parts] = rz;
        }
    }
    if (colsum) {   // kernel-uniform: [point][axis][HG x 16 channels] through the q slab, then one thread per column
        constexpr int CW = HG * 16;
        __syncthreads();   // every wave is done reading the slabs
        if (writer) {
            float* cp = slab_q + mp.n * 3 * CW + mp.hl * 16 + mp.ql * 4;
            *reinterpret_cast<float4*>(cp) = ox;
            *reinterpret_cast<float4*>(cp + CW) = oy;
            *reinterpret_cast<float4*>(cp + 2 * CW) = oz;
        }
        __syncthreads();
        if (tid < 3 * CW) {
            float a = 0.f;
#pragma unroll 8
            for (int n = 0; n < FND; ++n) a += slab_q[n * 3 * CW + tid];
            const int ax = tid / CW, c = tid - ax * CW;
            colsum[((size_t)b * 3 + ax) * Co + head0 * 16 + c] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- host side
bool edge_ft_supported(int Co, int Cin, int Ns, int Nd, int head_c, bool has_rows) {
    if (head_c != 16 || Nd != FND) return false;
    if (Cin == 128 && Ns == 128 && has_rows && Co % 16 == 0) return true;         // layer 5 of the released schedule
    if (Cin == 256 && Ns == 32 && !has_rows && Co % 32 == 0) return true;         // layer 6
    return false;
}
int edge_ft_heads_per_group(int Cin) { return Cin == 256 ? 2 : 1; }
// scratch of one fused layer call (inside the layer's table area): A planes + exponents (source rows, and the selected destination rows
// when the layer down-samples), raw scores, partial norms
size_t edge_ft_scratch_bytes(int B, int Ns, int Nd, int Cin, int Co, bool has_rows) {
    const size_t rp = (size_t)B * Ns * 3, rq = (size_t)B * Nd * 3, H = Co / 16;
    size_t s = rp * Cin * 4 + rp * 4 + 512;
    if (has_rows) s += rq * Cin * 4 + rq * 4 + 512;
    s += 2 * ((size_t)B * H * Nd * FK * 4 + 256) + (size_t)B * H * Nd * 4 + 256;
    s += (size_t)B * Nd * FK * 4 + 256 + (size_t)B * Nd * 4 + 256;       // the summed norms (invk, invq)
    return s;
}
struct FtScratch { uint4* a_p; int* ae_p; uint4* a_q; int* ae_q; float* scores; float* sskp; float* ssqp; float* invk; float* invq; };
static FtScratch ft_layout(void* scratch, int B, int Ns, int Nd, int Cin, int Co, bool has_rows) {
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t rp = (size_t)B * Ns * 3, rq = (size_t)B * Nd * 3, H = Co / 16;
    char* c = (char*)scratch;
    FtScratch s;
    size_t off = 0;
    s.a_p = (uint4*)(c + off); off = up(off + rp * Cin * 4);
    s.ae_p = (int*)(c + off); off = up(off + rp * 4);
    if (has_rows) {
        s.a_q = (uint4*)(c + off); off = up(off + rq * Cin * 4);
        s.ae_q = (int*)(c + off); off = up(off + rq * 4);
    } else { s.a_q = s.a_p; s.ae_q = s.ae_p; }
    s.scores = (float*)(c + off); off = up(off + (size_t)B * H * Nd * FK * 4);
    s.sskp = (float*)(c + off); off = up(off + (size_t)B * H * Nd * FK * 4);
    s.ssqp = (float*)(c + off); off = up(off + (size_t)B * H * Nd * 4);
    s.invk = (float*)(c + off); off = up(off + (size_t)B * Nd * FK * 4);
    s.invq = (float*)(c + off);
    return s;
}
// the layer's operand image (launched where the table GEMM was: it depends on the features only, not on the graph)
int edge_ft_prep_launch(const float* cur, const int32_t* dst_rows, int B, int Ns, int Nd, int Cin, int Co, void* scratch, hipStream_t st) {
    const bool has_rows = dst_rows != nullptr;
    LS_REQUIRE(edge_ft_supported(Co, Cin, Ns, Nd, 16, has_rows), "edge_ft_prep: unsupported shape (Co=%d Cin=%d Ns=%d Nd=%d)", Co, Cin, Ns, Nd);
    const FtScratch s = ft_layout(scratch, B, Ns, Nd, Cin, Co, has_rows);
    if (Cin == 128) {
        hipLaunchKernelGGL(edge_ft_prep_a_kernel<8>, dim3(B * Ns * 3 / 32), dim3(256), 0, st, cur, (const int32_t*)nullptr, Ns, Nd, s.a_p, s.ae_p);
        LS_LAUNCH_CHECK();
        hipLaunchKernelGGL(edge_ft_prep_a_kernel<8>, dim3(B * Nd * 3 / 32), dim3(256), 0, st, cur, dst_rows, Ns, Nd, s.a_q, s.ae_q);
    } else {
        hipLaunchKernelGGL(edge_ft_prep_a_kernel<16>, dim3(B * Ns * 3 / 32), dim3(256), 0, st, cur, (const int32_t*)nullptr, Ns, Nd, s.a_p, s.ae_p);
    }
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int edge_ft_attn_launch(const void* wplanes, const int32_t* knn, bool has_rows, int B, int Ns, int Nd, int Cin, int Co, float neg_slope, void* scratch,
                        float* out, float* rowmax, hipStream_t st, float* colsum) {
    LS_REQUIRE(edge_ft_supported(Co, Cin, Ns, Nd, 16, has_rows) && wplanes, "edge_ft_attn: unsupported shape (Co=%d Cin=%d Ns=%d Nd=%d)", Co, Cin, Ns, Nd);
    const FtScratch s = ft_layout(scratch, B, Ns, Nd, Cin, Co, has_rows);
    const int H = Co / 16, HG = edge_ft_heads_per_group(Cin), KS = Cin / 16;
    const uint4* wp = (const uint4*)wplanes;
    const int* we = (const int*)(wp + (size_t)H * 5 * KS * 128);
    const float oms = 1.0f - neg_slope, isd = 1.0f / sqrtf(3.0f * 16);
    const dim3 grid((H / HG) * B);
    // workgroup order: instance major (an instance's head groups side by side on an XCD: its A image stays in that XCD's L2 and is read by all of
    // them) or head-group major (the 64 instances of one W slice side by side).  Measured (us per launch, q/k | v): layer 5 36.9 | 41.6 instance
    // major vs 54.9 | 60.6 head major; layer 6 59.2 | 46.6 vs 65.9 | 48.1 -- the A fragments are the stream that matters.  LS_FT_ORDER=0 (dev library): head major.
    static const int im = dev_knob("LS_FT_ORDER", 1);
    if (Cin == 128) {
        hipLaunchKernelGGL((edge_ft_qk_kernel<128, 128, 1>), grid, dim3(256), 0, st, s.a_p, s.ae_p, s.a_q, s.ae_q, wp, we, knn, B, H, oms, s.scores, s.sskp, s.ssqp, im);
        LS_LAUNCH_CHECK();
        hipLaunchKernelGGL(edge_ft_norms_kernel, dim3(B), dim3(256), 0, st, s.sskp, s.ssqp, H, s.invk, s.invq);
        LS_LAUNCH_CHECK();
        hipLaunchKernelGGL((edge_ft_v_kernel<128, 128, 1>), grid, dim3(256), 0, st, s.a_p, s.ae_p, s.a_q, s.ae_q, wp, we, knn, B, H, oms, isd, s.scores, s.invk,
                           s.invq, out, Co, rowmax, H / HG, im, colsum);
    } else {
        hipLaunchKernelGGL((edge_ft_qk_kernel<256, 32, 2>), grid, dim3(256), 0, st, s.a_p, s.ae_p, s.a_q, s.ae_q, wp, we, knn, B, H, oms, s.scores, s.sskp, s.ssqp, im);
        LS_LAUNCH_CHECK();
        hipLaunchKernelGGL(edge_ft_norms_kernel, dim3(B), dim3(256), 0, st, s.sskp, s.ssqp, H, s.invk, s.invq);
        LS_LAUNCH_CHECK();
        hipLaunchKernelGGL((edge_ft_v_kernel<256, 32, 2>), grid, dim3(256), 0, st, s.a_p, s.ae_p, s.a_q, s.ae_q, wp, we, knn, B, H, oms, isd, s.scores, s.invk,
                           s.invq, out, Co, rowmax, H / HG, im, colsum);
    }
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int edge_ft_rowmax_parts(int Co, int Cin) { return Co / 16 / edge_ft_heads_per_group(Cin); }

}  // namespace ls
