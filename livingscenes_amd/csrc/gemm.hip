// gemm.hip -- out[M,N] = act(A[M,K] * W[N,K]^T + bias), fp32 in / fp32 out on the CDNA4 matrix cores.
//
// Kernels in this file (all 128 x 128 workgroup tiles, four waves as 2 x 2, 32 x 32 MFMA tiles):
//   gemm_f32_kernel<false>          v_mfma_f32_32x32x2_f32, exact fp32 FMA chains (LS_GEMM_BF16X3=0, and the caller-designated latency GEMMs)
//   gemm_f32_kernel<true, 3 | 2>    fp32 products as three (two) bf16 pieces, six (three) v_mfma_f32_32x32x16_bf16 per 16 k (LS_GEMM_MODE=bf16x3)
//   gemm_f32_kernel<true, 22>       fp32 products as two f16 pieces (h + residual), three v_mfma_f32_32x32x16_f16 into one accumulator: the DEFAULT arithmetic
//   gemm_h2_kernel                  the same arithmetic as a double-buffered software pipeline (K >= 128)
//   gemm_w2_kernel                  the same again on 256 x 256 tiles, eight waves of 4 x 2 MFMA tiles (chip-filling K >= 128 problems: the decoder)
//   gemm_h2_smallk_kernel           the same arithmetic, persistent over the M-tiles of an N-tile (K = 32 / 64 table GEMMs)
//   gemm_vn_kernel                  gemm_h2_kernel with the VN activation of the residual global conv as its epilogue
//   gemm_smallk_kernel              fp32-MFMA persistent small-K kernel (fp32 mode only)
// The paragraph below describes the fp32-MFMA kernel the others grew from (tile shape, LDS layout, k permutation).
//
// This is the ONLY MFMA-shaped work on the path: the VecLinear channel contraction
// (/root/reference/lib_shape_prior/core/lib/vec_sim3/vec_layers.py:121-136, F.linear at :134) applied to
// x-major feature rows [B*N*3, C_in], and the DeepSDF linears
// (/root/reference/lib_shape_prior/core/lib/implicit_func/deepsdf_decoder.py:98-121).
//
// v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, bitwise an fmaf chain; 157 TFLOP/s peak = fp32 vector peak,
// but it leaves the VALU free and needs one VGPR per operand).  Workgroup tile 128x128, BK=16, 4 waves as
// 2(M) x 2(N), each wave 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).  Both operands are K-contiguous
// ("A * W^T"), staged through LDS with rows padded to 20 floats: a 16-lane ds_read_b128 group then touches
// 16 distinct 16-byte bank slots (20*i mod 64 is a permutation of the multiples of 4).  The k index is
// permuted inside a BK block (lanes 0-31 take k = 8h+s, lanes 32-63 take k = 8h+4+s at MFMA step (h,s)) so
// that every lane fetches its four k-values with ONE ds_read_b128; A and B use the same permutation, so the
// product is unchanged.  Next-tile global loads are issued before the MFMA block (register double buffer).
#include <atomic>
#include "ls_common.h"
#include <string.h>
#include <algorithm>

namespace ls {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// ---- fp32 GEMM on the bf16 matrix cores ("3 x bf16 split", SPLIT = true).  An fp32 value is the exact sum of three bf16 pieces
// up to 2^-24 relative: a = a1 + a2 + a3 with a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2) (round to nearest; the
// residuals are exact in fp32).  The product a b is then a1b1 + (a1b2 + a2b1) + (a2b2 + a1b3 + a3b1) + O(2^-24 |a b|): six
// v_mfma_f32_32x32x16_bf16 (exact bf16 x bf16 products, fp32 accumulate) per 16 k instead of eight v_mfma_f32_32x32x2_f32 at
// 1/16 of the rate -> 2.7x less matrix-pipe time for a result that is as accurate as the fp32 FMA chain (measured against
// fp64: scripts/gemm_microbench.py --check).  The split runs on the VALU while the tile is staged (v_cvt_pk_bf16_f32: ~5.5
// instructions per value), each piece goes to its own LDS plane ([row][16 k] bf16 = 32 B rows: a lane's 16-byte operand
// read is contiguous over the wave, conflict-free).
__device__ __forceinline__ unsigned cvt2_bf16(f32x2_t v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t)); }
__device__ __forceinline__ f32x2_t expand2_bf16(unsigned u) { return f32x2_t{__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u)}; }
// four consecutive-k values -> their three bf16 pieces, packed as 4 x bf16 = 8 bytes per piece
__device__ __forceinline__ void split3_bf16(const float4& v, uint2& p1, uint2& p2, uint2& p3) {
    const f32x2_t lo = {v.x, v.y}, hi = {v.z, v.w};
    p1.x = cvt2_bf16(lo); p1.y = cvt2_bf16(hi);
    const f32x2_t r1l = lo - expand2_bf16(p1.x), r1h = hi - expand2_bf16(p1.y);
    p2.x = cvt2_bf16(r1l); p2.y = cvt2_bf16(r1h);
    const f32x2_t r2l = r1l - expand2_bf16(p2.x), r2h = r1h - expand2_bf16(p2.y);
    p3.x = cvt2_bf16(r2l); p3.y = cvt2_bf16(r2h);
}


// ---- fp32 GEMM on the f16 matrix cores ("2 x f16 split", PIECES = 22).  a = h + l with h = f16(a) (11 significant bits) and
// l = f16(a - h) (the residual is exact in fp32 and keeps 11 more bits while it is a normal f16): |a - (h + l)| <= 2^-22 |a|.  The
// matrix core takes f16 subnormals as they are (scripts/ubench/f16_denorm.hip), so a residual below 2^-14 degrades gracefully: its
// absolute error never exceeds 2^-25.  a b = h_a h_b + h_a l_b + l_a h_b + O(2^-21 |a b|): THREE v_mfma_f32_32x32x16_f16 per 16 k,
// accumulated -- in the order l_a h_b, h_a h_b, h_a l_b, in every kernel of this file and in the fused attention kernel, which is what
// keeps their results bit-identical to each other -- into ONE fp32 accumulator.  (Rounds 1-2 scaled the residual by 1024 and kept the
// cross terms in a second accumulator set: 64 more VGPRs per wave, which capped the wave tile at 2 x 2 MFMA tiles.)  Against fp64 the
// result is as close as an fp32 FMA chain's (scripts/gemm_microbench.py --check on badly scaled operands: 6.6 - 9.6 units of
// 2^-24 sum|a||w| at K = 32 .. 768; an fp32 FMA chain: 8 - 13 -- the fp32 accumulation error any fp32 GEMM carries) at half the
// matrix-pipe time of the three-piece bf16 split and three VALU instructions per split value.  The f16 range is made to follow every
// operand ROW by an exact power-of-two scale: "operand range of the f16 split" below.  LS_GEMM_MODE=bf16x3 keeps the six-MFMA split.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2_f16_pair(f32x2_t v, unsigned& h, unsigned& l) {
    const f16x2_t hv = __builtin_convertvector(v, f16x2_t);
    const f16x2_t lv = __builtin_convertvector(v - __builtin_convertvector(hv, f32x2_t), f16x2_t);
    h = __builtin_bit_cast(unsigned, hv);
    l = __builtin_bit_cast(unsigned, lv);
}
__device__ __forceinline__ void split2_f16(const float4& v, uint2& h, uint2& l) {
    split2_f16_pair(f32x2_t{v.x, v.y}, h.x, l.x);
    split2_f16_pair(f32x2_t{v.z, v.w}, h.y, l.y);
}
// ---- operand range of the f16 split.  f16 covers 2^-14 .. 65504, fp32 features and gradients do not stay there (a trained encoder's
// conv_c outputs sit at ~1.5e-5 because the heads multiply by scale_factor = 64000, vec_dgcnn_atten.py:234-250; the gradients of the
// pose refinement at 1e-6 .. 1e-8).  So every ROW of A and every row of W is multiplied by its own exact power of two before the split --
// s = 2^(14 - floor(log2 max|row|)): the row's largest element lands in [2^14, 2^15), elements down to 2^-17 of it keep the full 22
// bits (their residual is still a normal f16), below that the absolute error is at most 2^-39 of the row maximum -- 2^-15 of the
// fp32 rounding of the row's largest term -- and the product of the two inverse scales multiplies the fp32 accumulators in
// the epilogue.  Powers of two commute with every rounding in between, so for data that was in range before the result is
// BIT-IDENTICAL to the unscaled split, any finite fp32 input is handled, and a row's result depends on that row's data only.
// The row maxima come from (a) a pre-pass of the kernel over its own operand rows (default), or (b) caller-supplied arrays
// (GemmAux: the decoder chains them from the previous layer's epilogue, weights carry theirs from ls_model_create), which may be
// any upper bound: each factor of two of slack costs one bit at the bottom of the 17-binade window.
// (struct GemmAux: ls_common.h)
__device__ __forceinline__ float amax4(float m, const float4& v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
// exact powers of two: s * amax in [2^14, 2^15), inv = 1 / s.  amax = 0 (or fp32-subnormal) -> s = 2^126; Inf / NaN rows stay non-finite.
__device__ __forceinline__ void pow2_scale(float amax, float& s, float& inv) {
    unsigned be = (__float_as_uint(amax) >> 23) & 0xffu;
    be = be < 15u ? 15u : be;
    s = __uint_as_float((268u - be) << 23);
    inv = __uint_as_float((be - 14u) << 23);
}
// The epilogue's acc * s_a^-1 * s_w^-1: both inverse scales are exact NORMAL powers of two (pow2_scale), so their exponents are kept as
// integers (pow2_e), added, and applied by ONE v_ldexp_f32 -- exact wherever fp32 holds the result, rounded once into the subnormals, never
// an intermediate overflow.  (Until round 4 the two floats were multiplied first: that product flushes to 0 below 2^-149 -- two operand rows
// at ~1e-19 each -- although acc times it can be a normal number.)  In range the result is bit-identical to the multiply.
__device__ __forceinline__ int pow2_e(float p) { return (__float_as_int(p) >> 23) - 127; }
__device__ __forceinline__ float scale_pow2(float acc, int e) { return __builtin_ldexpf(acc, e); }
template <int CTRL>
__device__ __forceinline__ float dpp_fmax(float v) {
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false)));
}
// max over aligned groups of 8 lanes (the 8 staging threads of one operand row) / 16 lanes, in every lane of the group
__device__ __forceinline__ float max8(float v) { return dpp_fmax<0x141>(dpp_fmax<0x4E>(dpp_fmax<0xB1>(v))); }
__device__ __forceinline__ float max16(float v) { return dpp_fmax<0x140>(max8(v)); }
// split of s * v (s: the row's power of two); the multiply is spelled as packed fp32 (v_pk_mul_f32: the file is built without SLP vectorisation)
template <int WHICH>   // 0 = A operand, 1 = W operand (dev timing variants below)
__device__ __forceinline__ void split2_f16s(const float4& v, float s, uint2& h, uint2& l) {
    const f32x2_t sv = {s, s};
    split2_f16_pair(f32x2_t{v.x, v.y} * sv, h.x, l.x);
    split2_f16_pair(f32x2_t{v.z, v.w} * sv, h.y, l.y);
}

constexpr int GM = 128, GN = 128, GK = 16, GLD = 20;

// Write one wave's staged 32x64 half-tile (LDS rows of 68 floats) to out[gm0.., gn0..].  Interior tiles (the common case)
// take the predicate-free path: all eight LDS reads in flight, then eight 16-byte stores per lane (each store = four
// 256-byte row segments per wave); per-element branches around the LDS read -> store pairs cost 12 % on the K <= 64 GEMMs.
// `mask` (nullable, same shape and row stride as `out`): out = mask > 0 ? value : 0 -- the ReLU derivative of the decoder's backward
// pass fused into the store (ls_sdf_backward: dh_{l-1} = (dz_l W_l) [h_{l-1} > 0]; a separate pass cost 80 us per layer at 65 536 rows).
// `rowmax` (nullable): rowmax[gm * rm_parts + rm_part] = max|value| of row gm over this wave's 64 columns (after the mask) -- the next
// GEMM's operand range, see GemmAux.
__device__ __forceinline__ void store_half_tile(const float* stg, float* __restrict__ out, int ldc, int M, int N, int gm0, int gn0,
                                                int lane, bool full_tile, bool vec_ok, const float* __restrict__ mask,
                                                float* __restrict__ rowmax = nullptr, int rm_parts = 0, int rm_part = 0) {
    if (full_tile) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(&stg[(u * 4 + (lane >> 4)) * 68 + (lane & 15) * 4]);
        const size_t o0 = (size_t)(gm0 + (lane >> 4)) * ldc + gn0 + (lane & 15) * 4;
        if (mask) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float4 hm = *reinterpret_cast<const float4*>(mask + o0 + (size_t)(u * 4) * ldc);
                v[u].x = hm.x > 0.f ? v[u].x : 0.f; v[u].y = hm.y > 0.f ? v[u].y : 0.f;
                v[u].z = hm.z > 0.f ? v[u].z : 0.f; v[u].w = hm.w > 0.f ? v[u].w : 0.f;
            }
        }
        float* op = out + o0;
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<float4*>(op + (size_t)(u * 4) * ldc) = v[u];
        if (rowmax) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float rm = max16(amax4(0.f, v[u]));
                if ((lane & 15) == 0) rowmax[(size_t)(gm0 + u * 4 + (lane >> 4)) * rm_parts + rm_part] = rm;
            }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int idx = u * 64 + lane;           // 32 rows x 16 float4
        const int rr = idx >> 4, c4 = (idx & 15) * 4;
        const int gm = gm0 + rr, gn = gn0 + c4;
        float rm = 0.f;
        if (gm < M && gn < N) {
            float4 v = *reinterpret_cast<const float4*>(&stg[rr * 68 + c4]);
            float* op = out + (size_t)gm * ldc + gn;
            if (mask) {
                const float* mp = mask + (size_t)gm * ldc + gn;
                v.x = mp[0] > 0.f ? v.x : 0.f;
                if (gn + 1 < N) v.y = mp[1] > 0.f ? v.y : 0.f;
                if (gn + 2 < N) v.z = mp[2] > 0.f ? v.z : 0.f;
                if (gn + 3 < N) v.w = mp[3] > 0.f ? v.w : 0.f;
            }
            if (vec_ok && gn + 3 < N) {
                *reinterpret_cast<float4*>(op) = v;
                rm = amax4(0.f, v);
            } else {
                op[0] = v.x; rm = fabsf(v.x);
                if (gn + 1 < N) { op[1] = v.y; rm = fmaxf(rm, fabsf(v.y)); }
                if (gn + 2 < N) { op[2] = v.z; rm = fmaxf(rm, fabsf(v.z)); }
                if (gn + 3 < N) { op[3] = v.w; rm = fmaxf(rm, fabsf(v.w)); }
            }
        }
        if (rowmax) {   // wave-uniform
            rm = max16(rm);
            if ((lane & 15) == 0 && gm < M) rowmax[(size_t)gm * rm_parts + rm_part] = rm;
        }
    }
}

// store_half_tile for the SLICE-MAJOR layout (GemmAux::slice_cols / slice_rows; edge_staged.hip): the half tile's 64 columns are 64 / sc slices, and the 32
// rows x sc columns of one slice are CONTIGUOUS in memory (32 * sc floats: 1 KB at sc = 8) -- the lanes walk the half tile slice by slice, so every
// store instruction writes whole lines.  gm0 % 32 == 0 and slice_rows % 32 == 0: the 32 rows belong to one instance.
__device__ __forceinline__ void store_half_tile_sliced(const float* stg, float* __restrict__ out, int M, int N, int gm0, int gn0, int lane, int sc, int srows) {
    const int bi = gm0 / srows, r0 = gm0 - bi * srows;
    float* ob = out + (size_t)bi * srows * N + (size_t)r0 * sc;
    const size_t sstride = (size_t)srows * sc;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int idx = u * 64 + lane;           // float4 index inside the half tile, slice-major: [slice][row][float4 of the row's sc columns]
        int s_, rr, q;
        if (sc == 8) { s_ = idx >> 6; rr = (idx >> 1) & 31; q = idx & 1; }
        else { s_ = idx >> 5; rr = idx & 31; q = 0; }
        const int gn = gn0 + s_ * sc + q * 4;
        if (gm0 + rr < M && gn < N) {
            const float4 v = *reinterpret_cast<const float4*>(&stg[rr * 68 + s_ * sc + q * 4]);
            *reinterpret_cast<float4*>(ob + (size_t)(gn / sc) * sstride + (size_t)rr * sc + q * 4) = v;
        }
    }
}

// Optional row gather (down-sampled encoder layers): output row m = (b*gNd + n)*3 + x reads A row
// (b*gNs + a_rows[b*gNd + n])*3 + x, i.e. the GEMM runs only on the FPS-selected points of each instance.
// PIECES = 2 (opt-in, LS_SDF_BF16X2): a = a1 + a2 only, three MFMAs per 16 k (a1b1 + a1b2 + a2b1); products carry a 2^-16
// relative error instead of 2^-24 -- a decode mode for throughput, never the default.
template <bool SPLIT, int PIECES = 3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                       int ldw, const float* __restrict__ bias, float* __restrict__ out,
                                                       int ldc, int M, int N, int K, int relu, int ntiles_n,
                                                       const int32_t* __restrict__ a_rows, int gNd, int gNs, int kchunk,
                                                       size_t slab_stride, const float* __restrict__ mask, GemmAux aux) {
    constexpr int STG = 32 * 68;  // epilogue staging: 32 rows x (64 + 4) floats per wave
    constexpr int BK = SPLIT ? 32 : GK;          // k depth of one staged slab
    constexpr int NST = SPLIT ? 4 : 2;           // float4 per thread per operand slab
    constexpr int PLANE = GM * 64;               // SPLIT: bytes of one bf16 plane (128 rows x 32 k)
    constexpr int OPER_FLOATS = SPLIT ? 6 * PLANE / 4 : (GM + GN) * GLD;
    __shared__ __attribute__((aligned(16))) float smem[(4 * STG > OPER_FLOATS) ? 4 * STG : OPER_FLOATS];
    float* As = smem;
    float* Bs = smem + GM * GLD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    // consecutive logical ids walk the N tiles of one M tile: they share the A panel in one XCD's L2
    const int tm = logical / ntiles_n, tn = logical % ntiles_n;
    const int m0 = tm * GM, n0 = tn * GN;
    const int wm = wave >> 1, wn = wave & 1;
    // split-K (blockIdx.y = slice): this workgroup covers k in [kbeg, kend) and writes an un-activated partial slab
    const int kbeg = blockIdx.y * kchunk, kend = min(K, kbeg + kchunk);
    out += (size_t)blockIdx.y * slab_stride;

    constexpr bool H2 = PIECES == 22;   // two f16 pieces (see split2_f16)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // staging map.  fp32 path: 128 rows x 4 float4 per operand slab, two per thread (rows sr0, sr0 + 64).  SPLIT: 128 rows x 8
    // float4 (32 k), four per thread (rows sr0 + 32 u)
    const int sr0 = SPLIT ? (tid >> 3) : (tid >> 2), sk = SPLIT ? (tid & 7) * 4 : (tid & 3) * 4;
    constexpr int RSTEP = SPLIT ? 32 : 64;
    float4 ra[NST], rb[NST];
    float sa[NST], sw[NST];   // f16 split: power-of-two scale of this thread's staged A / W rows (GemmAux)
#pragma unroll
    for (int h = 0; h < NST; ++h) { sa[h] = 1.f; sw[h] = 1.f; }
    __shared__ int rsc[H2 ? GM + GN : 1];   // inverse scales of the tile's 128 A rows, then of its 128 W rows
    if constexpr (H2) { if (aux.noscale) rsc[tid] = 0; }   // (256 threads = GM + GN entries; otherwise the range block below writes every entry)
    long long arow[NST];  // source row of A for this thread's staged rows (-1 = out of range)
#pragma unroll
    for (int h = 0; h < NST; ++h) {
        const int gm = m0 + sr0 + h * RSTEP;
        long long r = gm < M ? gm : -1;
        if (a_rows && gm < M) {
            const int pt = gm / 3, x = gm - pt * 3;
            const int bb = pt / gNd;
            r = ((long long)bb * gNs + a_rows[pt]) * 3 + x;
        }
        arow[h] = r;
    }
    auto gload = [&](int k0) {
#pragma unroll
        for (int h = 0; h < NST; ++h) {
            const int r = sr0 + h * RSTEP;
            const int gn = n0 + r, gk = k0 + sk;
            ra[h] = (arow[h] >= 0 && gk < kend) ? *reinterpret_cast<const float4*>(A + (size_t)arow[h] * lda + gk) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[h] = (gn < N && gk < kend) ? *reinterpret_cast<const float4*>(W + (size_t)gn * ldw + gk) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // SPLIT: three bf16 planes per operand, plane = 128 rows x 64 bytes (32 k); the 16-byte slot index (k / 8) is XOR-ed with
    // bits 2-3 of the row, so that 16 consecutive rows reading the same logical slot hit 16 distinct bank slots
    char* Ap = reinterpret_cast<char*>(smem);
    char* Bp = Ap + 3 * PLANE;
    auto lstore = [&]() {
#pragma unroll
        for (int h = 0; h < NST; ++h) {
            const int r = sr0 + h * RSTEP;
            if constexpr (SPLIT) {
                const int swz = r * 64 + (((sk >> 3) ^ ((r >> 2) & 3)) << 4) + ((sk >> 2) & 1) * 8;
                uint2 p1, p2, p3;
                if constexpr (H2) {
                    split2_f16s<0>(ra[h], sa[h], p1, p2);
                    *reinterpret_cast<uint2*>(Ap + swz) = p1;
                    *reinterpret_cast<uint2*>(Ap + PLANE + swz) = p2;
                    split2_f16s<1>(rb[h], sw[h], p1, p2);
                    *reinterpret_cast<uint2*>(Bp + swz) = p1;
                    *reinterpret_cast<uint2*>(Bp + PLANE + swz) = p2;
                    continue;
                }
                split3_bf16(ra[h], p1, p2, p3);
                *reinterpret_cast<uint2*>(Ap + swz) = p1;
                *reinterpret_cast<uint2*>(Ap + PLANE + swz) = p2;
                if constexpr (PIECES == 3) *reinterpret_cast<uint2*>(Ap + 2 * PLANE + swz) = p3;
                split3_bf16(rb[h], p1, p2, p3);
                *reinterpret_cast<uint2*>(Bp + swz) = p1;
                *reinterpret_cast<uint2*>(Bp + PLANE + swz) = p2;
                if constexpr (PIECES == 3) *reinterpret_cast<uint2*>(Bp + 2 * PLANE + swz) = p3;
            } else {
                *reinterpret_cast<float4*>(&As[r * GLD + sk]) = ra[h];
                *reinterpret_cast<float4*>(&Bs[r * GLD + sk]) = rb[h];
            }
        }
    };

    if constexpr (H2) {
        // operand range (see GemmAux): per-row powers of two from a pre-pass over this tile's rows, or from the caller's row maxima
        if (!aux.noscale) {
            float ma[NST], mw[NST];
#pragma unroll
            for (int h = 0; h < NST; ++h) { ma[h] = 0.f; mw[h] = 0.f; }
            const bool pre_a = aux.a_rowmax == nullptr, pre_w = aux.w_rowmax == nullptr;
            if (pre_a || pre_w)
                for (int k0 = kbeg; k0 < kend; k0 += BK) {
                    gload(k0);
#pragma unroll
                    for (int h = 0; h < NST; ++h) { ma[h] = amax4(ma[h], ra[h]); mw[h] = amax4(mw[h], rb[h]); }
                }
#pragma unroll
            for (int h = 0; h < NST; ++h) {
                const int r = sr0 + h * RSTEP;
                if (pre_a) ma[h] = max8(ma[h]);
                else {
                    ma[h] = 0.f;
                    if (arow[h] >= 0) for (int q = 0; q < aux.a_parts; ++q) ma[h] = fmaxf(ma[h], aux.a_rowmax[(size_t)arow[h] * aux.a_parts + q]);
                }
                if (pre_w) mw[h] = max8(mw[h]);
                else mw[h] = n0 + r < N ? aux.w_rowmax[n0 + r] : 0.f;
                float ia, iw;
                pow2_scale(ma[h], sa[h], ia);
                pow2_scale(mw[h], sw[h], iw);
                if ((tid & 7) == 0) { rsc[r] = pow2_e(ia); rsc[GM + r] = pow2_e(iw); }
            }
        }
    }
    gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (k0 + BK < kend) gload(k0 + BK);  // in flight under the MFMA block
        const int lr = lane & 31, lk = (lane >> 5) * 4;
        if constexpr (SPLIT) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {   // the slab's two 16-k halves; lane (row lr, lane>>5) holds k = 16 s2 + 8 (lane>>5) .. +7
                const int q = s2 * 2 + (lane >> 5);
                if constexpr (H2) {
                    f16x8_t a[2][2], b[2][2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int rowa = wm * 64 + i * 32 + lr, rowb = wn * 64 + i * 32 + lr;
#pragma unroll
                        for (int pc = 0; pc < 2; ++pc) {
                            a[i][pc] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Ap + pc * PLANE + rowa * 64 + ((q ^ ((rowa >> 2) & 3)) << 4)));
                            b[i][pc] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Bp + pc * PLANE + rowb * 64 + ((q ^ ((rowb >> 2) & 3)) << 4)));
                        }
                    }
                    // term-major: eight independent accumulator chains
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                    continue;
                }
                bf16x8_t a[2][3], b[2][3];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int rowa = wm * 64 + i * 32 + lr, rowb = wn * 64 + i * 32 + lr;
#pragma unroll
                    for (int p3 = 0; p3 < (H2 ? 2 : PIECES); ++p3) {
                        a[i][p3] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Ap + p3 * PLANE + rowa * 64 + ((q ^ ((rowa >> 2) & 3)) << 4)));
                        b[i][p3] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Bp + p3 * PLANE + rowb * 64 + ((q ^ ((rowb >> 2) & 3)) << 4)));
                    }
                }
                // term-major order: consecutive MFMAs go to the four independent accumulators (a dependent back-to-back chain on
                // one accumulator does not issue at the 32-cycle rate); smallest terms first
#define LS_TERM(PA, PB)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                         \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA], b[j][PB], acc[i][j], 0, 0, 0);
                if constexpr (PIECES == 3) { LS_TERM(2, 0) LS_TERM(0, 2) LS_TERM(1, 1) }
                LS_TERM(1, 0) LS_TERM(0, 1) LS_TERM(0, 0)
#undef LS_TERM
            }
        } else
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const float4*>(&As[(wm * 64 + i * 32 + lr) * GLD + h * 8 + lk]);
                b[i] = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + i * 32 + lr) * GLD + h * 8 + lk]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }

    // epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Storing straight from
    // that layout costs 64 global_store_dword per lane (store-issue-bound for the K = 32/64 table GEMMs, whose
    // epilogue outweighs their main loop).  Instead each wave transposes its 64x64 sub-tile through LDS in two 32-row
    // halves (the operand buffers are free now) and writes full 256-byte row segments with 16 dwordx4 stores per lane.
    __syncthreads();  // all waves are done with As / Bs
    float* stg = smem + wave * STG;  // 8.7 KB per wave
    const int col_l = lane & 31, rowh = (lane >> 5) * 4;
    const bool vec_ok = (ldc % 4 == 0) && (((uintptr_t)out & 15) == 0);
    const bool full_tile = vec_ok && (m0 + GM <= M) && (n0 + GN <= N);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wn * 64 + j * 32 + col_l;
            const float bv = (bias && gn < N) ? bias[gn] : 0.0f;
            int ce = 0;
            if constexpr (H2) ce = rsc[GM + wn * 64 + j * 32 + col_l];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r];
                if constexpr (H2) v = scale_pow2(v, rsc[wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + rowh] + ce);
                v += bv;
                if (relu) v = fmaxf(v, 0.0f);
                stg[((r & 3) + 8 * (r >> 2) + rowh) * 68 + j * 32 + col_l] = v;
            }
        }
        // wave-local hand-off through LDS: same wave writes and reads, LDS ops of a wave complete in order
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        store_half_tile(stg, out, ldc, M, N, m0 + wm * 64 + i * 32, n0 + wn * 64, lane, full_tile, vec_ok, mask,
                        gridDim.y == 1 ? aux.out_rowmax : nullptr, 2 * ntiles_n, 2 * tn + wn);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The f16-split GEMM as a software pipeline.  In gemm_f32_kernel a slab goes barrier -> split + LDS store -> barrier -> LDS reads +
// MFMAs, one after the other (measured at M = 262 144, N = K = 512: 39 % / 44 % of a wave's time in the two halves, the matrix pipe
// 30 % busy with two workgroups per CU).  Here the LDS operand area is double-buffered: while the 24 MFMAs of slab t run out of
// buffer t & 1, the same wave splits the global data of slab t+1 (already in registers) into buffer (t+1) & 1 and issues the loads
// of slab t+2 -- VALU, LDS stores and global loads interleaved with the matrix instructions in one basic block, one barrier per slab.
// Same arithmetic as gemm_f32_kernel<true, 22> (same products, same accumulation order per accumulator): bit-identical results.
// WPL: the W operand arrives PRE-SPLIT (GemmAux::w_planes: both f16 pieces of the scaled rows, written once per weight matrix by
// gemm_presplit_w_kernel with the same split function): 16-byte loads straight into the LDS planes' layout, no VALU work for W in the
// loop -- every W element used to be split again by each of the M / 128 workgroups that read it (measured at the decoder shape: the
// kernel issues VALU instructions 56 % of the time, two thirds of them the split).  Bit-identical to the in-kernel split.
template <bool KAL, bool WPL = false>   // KAL: the K range of this launch is a whole number of 32-k slabs
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_h2_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, const float* __restrict__ bias, float* __restrict__ out, int ldc,
    int M, int N, int K, int relu, int ntiles_n, const int32_t* __restrict__ a_rows, int gNd, int gNs, int kchunk, size_t slab_stride,
    const float* __restrict__ mask, GemmAux aux) {
    constexpr int STG = 32 * 68;
    constexpr int PLANE = GM * 64;         // one f16 plane: 128 rows x 32 k
    constexpr int BUF = 4 * PLANE;         // A hi, A lo, B hi, B lo
    static_assert(2 * BUF >= 4 * STG * 4, "epilogue staging aliases the operand buffers");
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    __shared__ __attribute__((aligned(16))) int rsc[GM + GN];   // inverse power-of-two scales of the tile's A rows, then of its W rows (GemmAux)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = logical / ntiles_n, tn = logical % ntiles_n;
    const int m0 = tm * GM, n0 = tn * GN;
    const int wm = wave >> 1, wn = wave & 1;
    const int kbeg = blockIdx.y * kchunk, kend = min(K, kbeg + kchunk);
    out += (size_t)blockIdx.y * slab_stride;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // staging map: 128 rows x 8 float4 (32 k) per operand slab, four per thread (rows sr0 + 32 h).  Rows past M / N are clamped
    // (computed, never stored) and so is k past the end (KAL: whole slabs, the data is never used; otherwise per float4, zeroed by
    // a select) -- no predicated load, so that the k-loop stays one basic block the scheduler can interleave.
    const int sr0 = tid >> 3, sk = (tid & 7) * 4;
    float4 ra[4], rb[4];
    const float* arow[4];
    const float* brow[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int gm = min(m0 + sr0 + h * 32, M - 1), gn = min(n0 + sr0 + h * 32, N - 1);
        size_t r = (size_t)gm;
        if (a_rows) {
            const int pt = gm / 3, x = gm - pt * 3;
            const int bb = pt / gNd;
            r = ((size_t)bb * gNs + a_rows[pt]) * 3 + x;
        }
        arow[h] = A + r * lda;
        brow[h] = W + (size_t)gn * ldw;
    }
    auto kof = [&](int k0) { return KAL ? min(k0, kend - 32) + sk : min(k0 + sk, kend - 4); };
    auto gload_a = [&](int k0) {
        const int ko = kof(k0);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            ra[h] = *reinterpret_cast<const float4*>(arow[h] + ko);
            if (!KAL && k0 + sk >= kend) ra[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // WPL staging map: a row's slab is ONE 128-byte line [hi: 32 k | lo: 32 k]; thread -> 16-byte chunk tid & 7 of rows (tid >> 3) + 32 u
    // (a wave load = eight whole lines, as with the fp32 rows; the first layout -- the two pieces K f16 apart, 64 useful bytes per
    // line and load -- cost +50 %: the kernel is as much L1-fill-bound as anything else)
    const int pr0 = tid >> 3, pc8 = tid & 7;
    const char* prow[4];
    int pswz[4];
    float4 pb0, pb1, pb2, pb3;
    if constexpr (WPL) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = pr0 + 32 * u;
            prow[u] = static_cast<const char*>(aux.w_planes) + (size_t)min(n0 + r, N - 1) * ((size_t)K * 4) + pc8 * 16;
            pswz[u] = (pc8 >> 2) * PLANE + r * 64 + (((pc8 & 3) ^ ((r >> 2) & 3)) << 4);
        }
    }
    auto gload_b = [&](int k0) {
        if constexpr (WPL) {
            const int kb = min(k0, kend - 32) * 4;     // slab k0 / 32 at byte (k0 / 32) * 128
            pb0 = *reinterpret_cast<const float4*>(prow[0] + kb); pb1 = *reinterpret_cast<const float4*>(prow[1] + kb);
            pb2 = *reinterpret_cast<const float4*>(prow[2] + kb); pb3 = *reinterpret_cast<const float4*>(prow[3] + kb);
            return;
        }
        const int ko = kof(k0);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            rb[h] = *reinterpret_cast<const float4*>(brow[h] + ko);
            if (!KAL && k0 + sk >= kend) rb[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // plane = 128 rows x 64 bytes; the 16-byte slot index (k / 8) is XOR-ed with bits 2-3 of the row (conflict-free b128 reads)
    int swz[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int r = sr0 + h * 32;
        swz[h] = r * 64 + (((sk >> 3) ^ ((r >> 2) & 3)) << 4) + ((sk >> 2) & 1) * 8;
    }
    float sa[4] = {1.f, 1.f, 1.f, 1.f}, sw[4] = {1.f, 1.f, 1.f, 1.f};   // operand range: one exact power of two per staged row (GemmAux), set below
    auto lstore2 = [&](char* plane_hi, const float4& v, float sc, int off) {   // A rows
        uint2 ph, pl;
        split2_f16s<0>(v, sc, ph, pl);
        *reinterpret_cast<uint2*>(plane_hi + off) = ph;
        *reinterpret_cast<uint2*>(plane_hi + PLANE + off) = pl;
    };
    auto lstore2w = [&](char* plane_hi, const float4& v, float sc, int off) {   // W rows
        uint2 ph, pl;
        split2_f16s<1>(v, sc, ph, pl);
        *reinterpret_cast<uint2*>(plane_hi + off) = ph;
        *reinterpret_cast<uint2*>(plane_hi + PLANE + off) = pl;
    };
    // the W half of a slab into LDS, in two parts (part 0 / 1 sit between the MFMA groups of the slab's second half)
    auto stage_w = [&](char* Bp, int part) {
        if constexpr (WPL) {   // (named registers: as an array these four went to scratch)
            if (part == 0) { *reinterpret_cast<float4*>(Bp + pswz[0]) = pb0; *reinterpret_cast<float4*>(Bp + pswz[1]) = pb1; }
            else { *reinterpret_cast<float4*>(Bp + pswz[2]) = pb2; *reinterpret_cast<float4*>(Bp + pswz[3]) = pb3; }
        } else {
            lstore2w(Bp, rb[2 * part], sw[2 * part], swz[2 * part]);
            lstore2w(Bp, rb[2 * part + 1], sw[2 * part + 1], swz[2 * part + 1]);
        }
    };
    // operand range: one exact power of two per staged row (GemmAux), from the caller's row maxima or a pre-pass over the rows
    if (aux.noscale) rsc[tid] = 0;
    else {
        float ma[4] = {0.f, 0.f, 0.f, 0.f}, mw[4] = {0.f, 0.f, 0.f, 0.f};
        if (aux.a_rowmax) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float* rp = aux.a_rowmax + (size_t)((arow[h] - A) / lda) * aux.a_parts;
                for (int q = 0; q < aux.a_parts; ++q) ma[h] = fmaxf(ma[h], rp[q]);
            }
        } else {
            for (int k0 = kbeg; k0 < kend; k0 += 32) {
                gload_a(k0);
#pragma unroll
                for (int h = 0; h < 4; ++h) ma[h] = amax4(ma[h], ra[h]);
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) ma[h] = max8(ma[h]);
        }
        if (aux.w_rowmax) {
#pragma unroll
            for (int h = 0; h < 4; ++h) mw[h] = aux.w_rowmax[(brow[h] - W) / ldw];
        } else if constexpr (!WPL) {
            for (int k0 = kbeg; k0 < kend; k0 += 32) {
                gload_b(k0);
#pragma unroll
                for (int h = 0; h < 4; ++h) mw[h] = amax4(mw[h], rb[h]);
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) mw[h] = max8(mw[h]);
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float ia, iw;
            pow2_scale(ma[h], sa[h], ia);
            pow2_scale(mw[h], sw[h], iw);
            if ((tid & 7) == 0) { rsc[sr0 + h * 32] = pow2_e(ia); rsc[GM + sr0 + h * 32] = pow2_e(iw); }
        }
    }
    const int lr = lane & 31;
    int offa[2], offb[2];   // byte offset of this lane's operand row inside a plane, per 32-row MFMA tile (slot XOR applied per half)
#pragma unroll
    for (int i = 0; i < 2; ++i) { offa[i] = (wm * 64 + i * 32 + lr) * 64; offb[i] = (wn * 64 + i * 32 + lr) * 64; }
    const int xa[2] = {((wm * 64 + lr) >> 2) & 3, ((wm * 64 + 32 + lr) >> 2) & 3}, xb[2] = {((wn * 64 + lr) >> 2) & 3, ((wn * 64 + 32 + lr) >> 2) & 3};

    gload_a(kbeg); gload_b(kbeg);
#pragma unroll
    for (int h = 0; h < 4; ++h) lstore2(smem, ra[h], sa[h], swz[h]);
    stage_w(smem + 2 * PLANE, 0); stage_w(smem + 2 * PLANE, 1);
    gload_a(kbeg + 32); gload_b(kbeg + 32);
    __syncthreads();

    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += 32, cur ^= 1) {
        const char* Ac = smem + cur * BUF;
        const char* Bc = Ac + 2 * PLANE;
        char* An = smem + (cur ^ 1) * BUF;
        char* Bn = An + 2 * PLANE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {   // the slab's two 16-k halves; lane (row lr, lane >> 5) holds k = 16 s2 + 8 (lane >> 5) .. +7
            const int q = s2 * 2 + (lane >> 5);
            f16x8_t a[2][2], b[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    a[i][pc] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Ac + pc * PLANE + offa[i] + ((q ^ xa[i]) << 4)));
                    b[i][pc] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Bc + pc * PLANE + offb[i] + ((q ^ xb[i]) << 4)));
                }
            // term-major, eight independent accumulator chains; between the three groups of four MFMAs: the split of the NEXT slab
            // (first half: the A rows, second half: the W rows), then the loads of the slab after it
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                }
            if (s2 == 0) { lstore2(An, ra[0], sa[0], swz[0]); lstore2(An, ra[1], sa[1], swz[1]); } else stage_w(Bn, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
            if (s2 == 0) { lstore2(An, ra[2], sa[2], swz[2]); lstore2(An, ra[3], sa[3], swz[3]); gload_a(k0 + 64); }
            else { stage_w(Bn, 1); gload_b(k0 + 64); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();   // buffer cur^1 is complete, and every wave is done reading buffer cur
    }

    // epilogue: as gemm_f32_kernel (each wave transposes its 64x64 sub-tile through LDS in two 32-row halves)
    float* stg = reinterpret_cast<float*>(smem) + wave * STG;
    const int col_l = lane & 31, rowh = (lane >> 5) * 4;
    const bool vec_ok = (ldc % 4 == 0) && (((uintptr_t)out & 15) == 0);
    const bool full_tile = vec_ok && (m0 + GM <= M) && (n0 + GN <= N);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wn * 64 + j * 32 + col_l;
            const float bv = (bias && gn < N) ? bias[gn] : 0.0f;
            const int ce = rsc[GM + wn * 64 + j * 32 + col_l];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int4 rs = *reinterpret_cast<const int4*>(&rsc[wm * 64 + i * 32 + 8 * r4 + rowh]);
                const int rev[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
                for (int rl = 0; rl < 4; ++rl) {
                    const int r = r4 * 4 + rl;
                    float v = scale_pow2(acc[i][j][r], rev[rl] + ce) + bv;
                    if (relu) v = fmaxf(v, 0.0f);
                    stg[((r & 3) + 8 * (r >> 2) + rowh) * 68 + j * 32 + col_l] = v;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        store_half_tile(stg, out, ldc, M, N, m0 + wm * 64 + i * 32, n0 + wn * 64, lane, full_tile, vec_ok, mask,
                        gridDim.y == 1 ? aux.out_rowmax : nullptr, 2 * ntiles_n, 2 * tn + wn);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The wide form of gemm_h2_kernel for the long GEMMs of the decoder (M >= thousands of rows, N and K in the hundreds): 256 x 256
// workgroup tiles, EIGHT waves as 2 (M) x 4 (N), a wave owns 128 x 64 = 4 x 2 MFMA tiles.  Why: at 128 x 128 the pipelined kernel is
// bound by everything at once (measured at the decoder shape, scripts/diag/gemm_decoder_pmc.sh + the timing variants of
// scripts/dev/build_variants.py): matrix pipe 45 % busy, VALU issue 56 % (two thirds of it the operand split, which every workgroup
// redoes on data that 2 047 / 5 other workgroups also split), 32 KB of L1 fill and 96 KB of LDS traffic per 96 MFMAs.  A 256 x 256
// tile halves the global bytes, the LDS writes and the split work per MFMA, the 4 x 2 wave tile takes a quarter off the LDS operand
// reads.  The accumulators of a 4 x 2 wave tile fit only if the main and the cross terms share them: 128 VGPRs.
// PP ("ping-pong", round 4, LS_GEMM_W2_PP=1; OFF by default): the two waves that share a SIMD (w and w + 4) run half a slab step apart.  A
// step is cut into a MEMORY phase (the 12 ds_read_b128 of the step's operand fragments, the split + LDS stores of the next slab, the
// global loads of the slab after it) and a COMPUTE phase (the step's 24 MFMAs and nothing else), with a workgroup barrier after each;
// waves 4 - 7 pass one extra barrier before the loop and waves 0 - 3 one after it, so at any time one wave per SIMD is in its compute
// phase while the other fills its registers.  Same products in the same order: bit-identical.  Buffer hand-over: slab s + 1 is written in
// the four phases in which slab s is read (the last readers of the old content finished one barrier earlier) and first read after the
// barrier that closes them.
// MEASURED (decoder shape 262 144 x 768 x 768, scripts/dev/w2_variants.sh, w2_clock.sh; profiles/r4_final/gemm_w2_power.txt): 957 us
// against 928 us for the in-step loop -- and the reason the answer is not "fewer stalls": this GEMM runs at the SOCKET POWER CAP.
// rocm-smi while it runs back to back: 1 400 W (the TDP), shader clock 1.70 GHz (in-step) / 1.82 GHz (ping-pong) instead of 2.39 GHz;
// the timing variants that remove the MFMAs, the LDS reads or the producer work run at 2.39 GHz and 975 - 1 255 W.  Under the cap the
// firmware trades every removed stall for clock: ping-pong needs 10 % MORE cycles (four barriers per slab) at a 7 % higher clock.  What
// is left to gain is energy per tile (fewer bytes moved per MFMA), not issue slots; "matrix pipe 55 % busy" is 55 % of the cycles of a
// clock the matrix pipe itself pulled down.
template <bool MASKED, int WPLM, bool PP>   // WPLM: 0 = W rows split here, 1 = pre-split planes staged through registers, 2 = planes by LDS-direct loads
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) void gemm_w2_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, const float* __restrict__ bias, float* __restrict__ out, int ldc,
    int M, int N, int K, int relu, int ntiles_n, int ntiles, const float* __restrict__ mask, GemmAux aux) {
    constexpr bool WPL = WPLM != 0, WDIR = WPLM == 2;
    static_assert(!(WDIR && PP), "the LDS-direct W path belongs to the in-step loop");
    constexpr int TM = 256, TN = 256;
    constexpr int STG = 32 * 68;
    constexpr int PLANE = TM * 64;         // one f16 plane: 256 rows x 32 k
    constexpr int BUF = 4 * PLANE;         // A hi, A lo, W hi, W lo
    static_assert(2 * BUF >= 8 * STG * 4, "epilogue staging aliases the operand buffers");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 * BUF + (TM + TN) * 4 bytes
    int* rsc = reinterpret_cast<int*>(smem + 2 * BUF);          // inverse power-of-two scales: the tile's A rows, then its W rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    // a workgroup takes the tiles blockIdx.x, + gridDim.x, ... (default launch: one tile each; LS_GEMM_W2_PERSIST=1: one workgroup per CU);
    // the workgroups resident on an XCD cover a contiguous range of tiles, tiles of one M block next to each other
    for (int tile = xcd_remap(blockIdx.x, gridDim.x); tile < ntiles; tile += gridDim.x) {
    const int tm = tile / ntiles_n, tn = tile % ntiles_n;
    const int m0 = tm * TM, n0 = tn * TN;

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // staging map: 256 rows x 8 float4 (32 k) per operand slab over 512 threads: four per thread (rows sr0 + 64 h)
    const int sr0 = tid >> 3, sk = (tid & 7) * 4;
    // (operand addresses = a wave-uniform base + a 32-bit byte offset per staged row: the launch keeps operands below 4 GB; the A rows' power
    // of two as an exponent applied by v_ldexp_f32 -- a packed multiply keeps every scale twice; both save registers the ping-pong loop needs)
    float4 ra[4], rb[4];
    unsigned aoff[4], boff[4];
    float sw[4] = {1.f, 1.f, 1.f, 1.f};
    int ea[4] = {0, 0, 0, 0};
    int swz[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int r = sr0 + h * 64;
        swz[h] = r * 64 + (((sk >> 3) ^ ((r >> 2) & 3)) << 4) + ((sk >> 2) & 1) * 8;
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        aoff[h] = ((unsigned)min(m0 + sr0 + h * 64, M - 1) * (unsigned)lda + sk) * 4u;
        boff[h] = ((unsigned)min(n0 + sr0 + h * 64, N - 1) * (unsigned)ldw + sk) * 4u;
    }
    auto gload_a = [&](int k0) {
        const int ko = min(k0, K - 32);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            asm volatile("" : "+v"(aoff[h]));   // (keeps the zero-extension next to the load: hoisted out of the loop it becomes a 64-bit register pair per row instead of the scalar-base addressing mode)
            ra[h] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(A + ko) + aoff[h]);
        }
    };
    // WPL (GemmAux::w_planes): a W row's slab is one 128-byte line [hi | lo]; thread -> chunk tid & 7 of rows (tid >> 3) + 64 u
    unsigned poff[4];
    int pswz[4];
    float4 pb0, pb1, pb2, pb3;
    if constexpr (WPL) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = sr0 + 64 * u, c8 = tid & 7;
            poff[u] = (unsigned)min(n0 + r, N - 1) * ((unsigned)K * 4u) + c8 * 16;
            pswz[u] = (c8 >> 2) * PLANE + r * 64 + (((c8 & 3) ^ ((r >> 2) & 3)) << 4);
        }
    }
    // WDIR (round 4): the W planes go global -> LDS without passing through registers (global_load_lds_dwordx4: the wave's 64 lanes fill
    // 1 KB of LDS starting at M0, lane l -> LDS slot l, so the slot swizzle is applied on the GLOBAL side: lane l of the instruction for
    // plane p, rows r0 .. r0 + 15 fetches chunk (l & 3) ^ ((r >> 2) & 3) of row r = r0 + (l >> 2)).  Wave w owns rows 32 w .. 32 w + 31:
    // four instructions per slab instead of four 16-byte register loads + four ds_write_b128, and 16 VGPRs fewer.  A slab's loads are
    // issued during the slab BEFORE it (their buffer was released by the barrier just passed), right after that slab's A rows left their
    // registers and right before the next A rows are requested, and awaited by s_waitcnt vmcnt(4) in front of the closing barrier -- the
    // four younger loads are those A rows (a sched_barrier pins them above the wait).  The loads are INLINE ASM on purpose: told about an
    // LDS-direct load (the builtin), hipcc answers every workgroup barrier and every use of a register load with s_waitcnt vmcnt(0) --
    // i.e. waits for the A rows it has just requested, once per slab.  Unknown to the compiler they only make its own waits for the A
    // rows wait for older loads as well, which by then have been awaited here.
    unsigned doff[4];
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    if constexpr (WDIR) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = wave * 32 + (u & 1) * 16 + (lane >> 2), pl = u >> 1;
            doff[u] = (unsigned)min(n0 + r, N - 1) * ((unsigned)K * 4u) + pl * 64 + (((lane & 3) ^ ((r >> 2) & 3)) << 4);
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    auto dload_w = [&](char* Bp, int k0) {   // W slab k0 -> the planes at Bp.  Inline asm: see the ordering note above
        const int ko = min(k0, K - 32);
        const char* pk = static_cast<const char*>(aux.w_planes) + ko * 4;
        const unsigned l0 = lds0 + (unsigned)(Bp - smem) + wave_s * (32 * 64);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#pragma unroll
        for (int u = 0; u < 4; ++u)
            asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l0 + (u >> 1) * PLANE + (u & 1) * (16 * 64)), "v"(doff[u]), "s"(pk) : "memory", "m0");
#pragma clang diagnostic pop
    };
    auto gload_b = [&](int k0) {
        const int ko = min(k0, K - 32);
        if constexpr (WDIR) return;
        if constexpr (WPL) {
            const char* pk = static_cast<const char*>(aux.w_planes) + ko * 4;
#pragma unroll
            for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(poff[u]));
            pb0 = *reinterpret_cast<const float4*>(pk + poff[0]); pb1 = *reinterpret_cast<const float4*>(pk + poff[1]);
            pb2 = *reinterpret_cast<const float4*>(pk + poff[2]); pb3 = *reinterpret_cast<const float4*>(pk + poff[3]);
            return;
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            asm volatile("" : "+v"(boff[h]));
            rb[h] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(W + ko) + boff[h]);
        }
    };
    auto stage_w = [&](char* Bp, int h) {
        if constexpr (WDIR) return;
        if constexpr (WPL) *reinterpret_cast<float4*>(Bp + pswz[h]) = h == 0 ? pb0 : (h == 1 ? pb1 : (h == 2 ? pb2 : pb3));   // (named registers: as an array these went to scratch)
        else { uint2 ph, pl; split2_f16s<1>(rb[h], sw[h], ph, pl); *reinterpret_cast<uint2*>(Bp + swz[h]) = ph; *reinterpret_cast<uint2*>(Bp + PLANE + swz[h]) = pl; }
    };
    auto lstore2 = [&](char* plane_hi, const float4& v, int e, int off) {
        uint2 ph, pl;
        split2_f16_pair(f32x2_t{__builtin_ldexpf(v.x, e), __builtin_ldexpf(v.y, e)}, ph.x, pl.x);   // == v * 2^e, as split2_f16s
        split2_f16_pair(f32x2_t{__builtin_ldexpf(v.z, e), __builtin_ldexpf(v.w, e)}, ph.y, pl.y);
        *reinterpret_cast<uint2*>(plane_hi + off) = ph;
        *reinterpret_cast<uint2*>(plane_hi + PLANE + off) = pl;
    };
    if (aux.noscale) rsc[tid] = 0;
    else {
        float ma[4] = {0.f, 0.f, 0.f, 0.f}, mw[4] = {0.f, 0.f, 0.f, 0.f};
        if (aux.a_rowmax) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float* rp = aux.a_rowmax + (size_t)min(m0 + sr0 + h * 64, M - 1) * aux.a_parts;
                for (int q = 0; q < aux.a_parts; ++q) ma[h] = fmaxf(ma[h], rp[q]);
            }
        } else {
            for (int k0 = 0; k0 < K; k0 += 32) {
                gload_a(k0);
#pragma unroll
                for (int h = 0; h < 4; ++h) ma[h] = amax4(ma[h], ra[h]);
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) ma[h] = max8(ma[h]);
        }
        if (aux.w_rowmax) {
#pragma unroll
            for (int h = 0; h < 4; ++h) mw[h] = aux.w_rowmax[min(n0 + sr0 + h * 64, N - 1)];
        } else if constexpr (!WPL) {
            for (int k0 = 0; k0 < K; k0 += 32) {
                gload_b(k0);
#pragma unroll
                for (int h = 0; h < 4; ++h) mw[h] = amax4(mw[h], rb[h]);
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) mw[h] = max8(mw[h]);
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float ia, iw, sah;
            pow2_scale(ma[h], sah, ia);
            pow2_scale(mw[h], sw[h], iw);
            ea[h] = pow2_e(sah);
            if ((tid & 7) == 0) { rsc[sr0 + h * 64] = pow2_e(ia); rsc[TM + sr0 + h * 64] = pow2_e(iw); }
        }
    }
    const int lr = lane & 31;
    int offa[4], offb[2], xa[4], xb[2];   // byte offset of this lane's operand row inside a plane per 32-row MFMA tile, and its slot XOR
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int r = wm * 128 + i * 32 + lr; offa[i] = r * 64; xa[i] = (r >> 2) & 3; }
#pragma unroll
    for (int j = 0; j < 2; ++j) { const int r = wn * 64 + j * 32 + lr; offb[j] = r * 64; xb[j] = (r >> 2) & 3; }

    if constexpr (WDIR) dload_w(smem + 2 * PLANE, 0);
    gload_a(0); gload_b(0);
#pragma unroll
    for (int h = 0; h < 4; ++h) { lstore2(smem, ra[h], ea[h], swz[h]); stage_w(smem + 2 * PLANE, h); }
    gload_a(32); gload_b(32);
    if constexpr (WDIR) {   // vmcnt(4): everything but the A rows of slab 1
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0f74);
    }
    __syncthreads();

    int cur = 0;
    const bool late_half = __builtin_amdgcn_readfirstlane(wave) >= 4;   // (waves w and w + 4 share a SIMD: the other pairings measured 1 064 - 1 083 us)
    if constexpr (PP) { if (late_half) __builtin_amdgcn_s_barrier(); }
    for (int k0 = 0; k0 < K; k0 += 32, cur ^= 1) {
        const char* Ac = smem + cur * BUF;
        const char* Bc = Ac + 2 * PLANE;
        char* An = smem + (cur ^ 1) * BUF;
        char* Bn = An + 2 * PLANE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {   // the slab's two 16-k halves
            const int q = s2 * 2 + (lane >> 5);
            f16x8_t a[4][2], b[2][2];
#pragma unroll
            for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j][pc] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Bc + pc * PLANE + offb[j] + ((q ^ xb[j]) << 4)));
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i][pc] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Ac + pc * PLANE + offa[i] + ((q ^ xa[i]) << 4)));
            }
            if constexpr (PP) {
                // memory phase: the producer work of this half step, then the barrier; compute phase: MFMAs only
                if (s2 == 0) {
#pragma unroll
                    for (int h = 0; h < 4; ++h) lstore2(An, ra[h], ea[h], swz[h]);
                    gload_a(k0 + 64);
                } else {
#pragma unroll
                    for (int h = 0; h < 4; ++h) stage_w(Bn, h);
                    gload_b(k0 + 64);
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS stores have landed, its fragments have arrived
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
#ifdef LS_W2_MFMA16
            // dev TIMING variant (verdict r4 item 8a; wrong results -- the epilogue still assumes the 32 x 32 accumulator map): the same flops as
            // v_mfma_f32_16x16x32_f16 -- a wave tile = 8 x 4 tiles of 16 x 16, one instruction per 32-k slab and product term; this half step takes the
            // tiles of rows 64 s2 .. 64 s2 + 63.  Same LDS planes and operand bytes per slab (12 fragments of each piece), same accumulator registers.
            {
                typedef float f32x4w __attribute__((ext_vector_type(4)));
                const int l15 = lane & 15, q4 = lane >> 4;
                f16x8_t a16[4][2], b16[4][2];
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const int r = wn * 64 + j * 16 + l15; b16[j][pc] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Bc + pc * PLANE + r * 64 + ((q4 ^ ((r >> 2) & 3)) << 4))); }
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const int r = wm * 128 + (4 * s2 + i) * 16 + l15; a16[i][pc] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Ac + pc * PLANE + r * 64 + ((q4 ^ ((r >> 2) & 3)) << 4))); }
                }
                f32x4w c16[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) c16[i][j][e] = acc[2 * s2 + (i >> 1)][j >> 1][((i & 1) * 2 + (j & 1)) * 4 + e];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) c16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[i][1], b16[j][0], c16[i][j], 0, 0, 0);
                if (s2 == 0) { lstore2(An, ra[0], ea[0], swz[0]); lstore2(An, ra[1], ea[1], swz[1]); }
                else { stage_w(Bn, 0); stage_w(Bn, 1); }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) c16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[i][0], b16[j][0], c16[i][j], 0, 0, 0);
                if (s2 == 0) { lstore2(An, ra[2], ea[2], swz[2]); lstore2(An, ra[3], ea[3], swz[3]); gload_a(k0 + 64); }
                else { stage_w(Bn, 2); stage_w(Bn, 3); gload_b(k0 + 64); }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) c16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[i][0], b16[j][1], c16[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[2 * s2 + (i >> 1)][j >> 1][((i & 1) * 2 + (j & 1)) * 4 + e] = c16[i][j][e];
                continue;
            }
#endif
            // per accumulator and 16-k step: lo(a) hi(w), hi(a) hi(w), hi(a) lo(w) -- the order every unified-accumulator kernel uses
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
            if (s2 == 0) { lstore2(An, ra[0], ea[0], swz[0]); lstore2(An, ra[1], ea[1], swz[1]); }
            else { stage_w(Bn, 0); stage_w(Bn, 1); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
            if (s2 == 0) {
                lstore2(An, ra[2], ea[2], swz[2]); lstore2(An, ra[3], ea[3], swz[3]);
                if constexpr (WDIR) dload_w(Bn, k0 + 32);   // (after the A rows have been consumed, before the next ones are requested: see dload_w)
                gload_a(k0 + 64);
            }
            else { stage_w(Bn, 2); stage_w(Bn, 3); gload_b(k0 + 64); }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
        }
        if constexpr (WDIR) {   // vmcnt(4): this slab's LDS-direct W loads have landed (younger: the A rows of slab + 2)
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0f74);
        }
        if constexpr (!PP) __syncthreads();   // buffer cur^1 is complete, and every wave is done reading buffer cur
    }
    if constexpr (PP) { if (!late_half) __builtin_amdgcn_s_barrier(); }   // the barrier counts of the two halves match again

    // (Not kept: s_setprio(1) around the MFMA groups: 930 -> 1 055 us.  The operand fragments as an explicit four-sub-phase software pipeline -- every LDS read batch one sub-phase ahead of its
    // MFMAs, the barrier in front of the last sub-phase.  Unfenced, the scheduler sinks the loads back to their uses: same time; fenced
    // with sched_barrier: 256 VGPRs + spills, 1010 -> 1110 us at the decoder shape.)

    // epilogue: each wave transposes its 128 x 64 sub-tile through LDS in four 32-row pieces
    float* stg = reinterpret_cast<float*>(smem) + wave * STG;
    const int col_l = lane & 31, rowh = (lane >> 5) * 4;
    const bool vec_ok = (ldc % 4 == 0) && (((uintptr_t)out & 15) == 0);
    const bool full_tile = vec_ok && (m0 + TM <= M) && (n0 + TN <= N);
    const int rm_parts = 2 * ((N + GN - 1) / GN);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wn * 64 + j * 32 + col_l;
            const float bv = (bias && gn < N) ? bias[gn] : 0.0f;
            const int ce = rsc[TM + wn * 64 + j * 32 + col_l];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int4 rs = *reinterpret_cast<const int4*>(&rsc[wm * 128 + i * 32 + 8 * r4 + rowh]);
                const int rev[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
                for (int rl = 0; rl < 4; ++rl) {
                    const int r = r4 * 4 + rl;
                    float v = scale_pow2(acc[i][j][r], rev[rl] + ce) + bv;
                    if (relu) v = fmaxf(v, 0.0f);
                    stg[((r & 3) + 8 * (r >> 2) + rowh) * 68 + j * 32 + col_l] = v;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        const int gn0 = n0 + wn * 64;
        store_half_tile(stg, out, ldc, M, N, m0 + wm * 128 + i * 32, gn0, lane, full_tile, vec_ok, MASKED ? mask : nullptr,
                        4 * tn + wn < rm_parts ? aux.out_rowmax : nullptr, rm_parts, 4 * tn + wn);   // (a part without valid columns receives 0)
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();   // the staging area and rsc are rewritten by the next tile
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The residual global conv (vec_dgcnn_atten.py:222-225) as ONE launch: out[p][x][c] = VN-act(lin, dir) with
//   lin = f[p][x] . W[c] + G[b][x][2C + c],   dir = f[p][x] . W[C + c] + G[b][x][3C + c]
// -- gemm_h2_kernel (same pipeline, same products, same accumulation order) with the point-wise VN activation as its epilogue instead
// of a [rows, 2C] table that vn_act_rows_kernel read back.  Two re-mappings make the activation local to a wave:
//   rows   a 32-row MFMA tile carries 30 rows = TEN whole points (the last two rows of each M-tile are duplicates, never stored), so a
//          workgroup tile is 120 rows of f: the xyz triple of a point never straddles tiles;
//   cols   the 128 tile columns are [lin c0..c0+31 | dir c0..c0+31 | lin c0+32..c0+63 | dir c0+32..c0+63] of 64 channels, so wave
//          (wm, wn) holds lin AND dir of its 32 channels.
template <bool KAL>   // K is a whole number of 32-k slabs
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_vn_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, const float* __restrict__ G, int ldg, float* __restrict__ out,
    int M, int C, int K, int npts, float oms, int ntiles_n, GemmAux aux) {
    constexpr int TR = 30;                 // useful rows per 32-row MFMA tile
    constexpr int STG = 32 * 68;
    constexpr int PLANE = GM * 64;         // one f16 plane: 128 rows x 32 k
    constexpr int BUF = 4 * PLANE;         // A hi, A lo, B hi, B lo
    static_assert(2 * BUF >= 4 * STG * 4, "epilogue staging aliases the operand buffers");
    __shared__ __attribute__((aligned(16))) char smem[2 * BUF];
    __shared__ __attribute__((aligned(16))) int rsc[GM + GN];   // inverse power-of-two scales of the staged A rows, then of the staged W rows (GemmAux)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = logical / ntiles_n, tn = logical % ntiles_n;
    const int m0 = tm * 4 * TR, c0 = tn * 64;
    const int wm = wave >> 1, wn = wave & 1;
    const int kbeg = 0, kend = K;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // staging map: 128 rows x 8 float4 (32 k) per operand slab, four per thread (rows sr0 + 32 h).  Rows past M / N are clamped
    // (computed, never stored) and so is k past the end (KAL: whole slabs, the data is never used; otherwise per float4, zeroed by
    // a select) -- no predicated load, so that the k-loop stays one basic block the scheduler can interleave.
    const int sr0 = tid >> 3, sk = (tid & 7) * 4;
    float4 ra[4], rb[4];
    const float* arow[4];
    const float* brow[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        // staged row sr0 + 32 h = row sr0 of M-tile h; staged W row = tile column sr0 + 32 h = (wn = h >> 1, lin | dir = h & 1, channel sr0)
        const int gm = min(m0 + h * TR + min(sr0, TR - 1), M - 1);
        arow[h] = A + (size_t)gm * lda;
        brow[h] = W + (size_t)((h & 1) * C + c0 + 32 * (h >> 1) + sr0) * ldw;
    }
    auto kof = [&](int k0) { return KAL ? min(k0, kend - 32) + sk : min(k0 + sk, kend - 4); };
    auto gload_a = [&](int k0) {
        const int ko = kof(k0);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            ra[h] = *reinterpret_cast<const float4*>(arow[h] + ko);
            if (!KAL && k0 + sk >= kend) ra[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto gload_b = [&](int k0) {
        const int ko = kof(k0);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            rb[h] = *reinterpret_cast<const float4*>(brow[h] + ko);
            if (!KAL && k0 + sk >= kend) rb[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // plane = 128 rows x 64 bytes; the 16-byte slot index (k / 8) is XOR-ed with bits 2-3 of the row (conflict-free b128 reads)
    int swz[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int r = sr0 + h * 32;
        swz[h] = r * 64 + (((sk >> 3) ^ ((r >> 2) & 3)) << 4) + ((sk >> 2) & 1) * 8;
    }
    auto lstore2 = [&](char* plane_hi, const float4& v, float sc, int off) {   // A rows
        uint2 ph, pl;
        split2_f16s<0>(v, sc, ph, pl);
        *reinterpret_cast<uint2*>(plane_hi + off) = ph;
        *reinterpret_cast<uint2*>(plane_hi + PLANE + off) = pl;
    };
    auto lstore2w = [&](char* plane_hi, const float4& v, float sc, int off) {   // W rows
        uint2 ph, pl;
        split2_f16s<1>(v, sc, ph, pl);
        *reinterpret_cast<uint2*>(plane_hi + off) = ph;
        *reinterpret_cast<uint2*>(plane_hi + PLANE + off) = pl;
    };
    // operand range: one exact power of two per staged row (GemmAux), from the caller's row maxima or a pre-pass over the rows
    float sa[4] = {1.f, 1.f, 1.f, 1.f}, sw[4] = {1.f, 1.f, 1.f, 1.f};
    if (aux.noscale) rsc[tid] = 0;
    else {
        float ma[4] = {0.f, 0.f, 0.f, 0.f}, mw[4] = {0.f, 0.f, 0.f, 0.f};
        if (aux.a_rowmax) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const float* rp = aux.a_rowmax + (size_t)((arow[h] - A) / lda) * aux.a_parts;
                for (int q = 0; q < aux.a_parts; ++q) ma[h] = fmaxf(ma[h], rp[q]);
            }
        } else {
            for (int k0 = kbeg; k0 < kend; k0 += 32) {
                gload_a(k0);
#pragma unroll
                for (int h = 0; h < 4; ++h) ma[h] = amax4(ma[h], ra[h]);
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) ma[h] = max8(ma[h]);
        }
        if (aux.w_rowmax) {
#pragma unroll
            for (int h = 0; h < 4; ++h) mw[h] = aux.w_rowmax[(brow[h] - W) / ldw];
        } else {
            for (int k0 = kbeg; k0 < kend; k0 += 32) {
                gload_b(k0);
#pragma unroll
                for (int h = 0; h < 4; ++h) mw[h] = amax4(mw[h], rb[h]);
            }
#pragma unroll
            for (int h = 0; h < 4; ++h) mw[h] = max8(mw[h]);
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float ia, iw;
            pow2_scale(ma[h], sa[h], ia);
            pow2_scale(mw[h], sw[h], iw);
            if ((tid & 7) == 0) { rsc[sr0 + h * 32] = pow2_e(ia); rsc[GM + sr0 + h * 32] = pow2_e(iw); }
        }
    }
    const int lr = lane & 31;
    int offa[2], offb[2];   // byte offset of this lane's operand row inside a plane, per 32-row MFMA tile (slot XOR applied per half)
#pragma unroll
    for (int i = 0; i < 2; ++i) { offa[i] = (wm * 64 + i * 32 + lr) * 64; offb[i] = (wn * 64 + i * 32 + lr) * 64; }
    const int xa[2] = {((wm * 64 + lr) >> 2) & 3, ((wm * 64 + 32 + lr) >> 2) & 3}, xb[2] = {((wn * 64 + lr) >> 2) & 3, ((wn * 64 + 32 + lr) >> 2) & 3};

    gload_a(kbeg); gload_b(kbeg);
#pragma unroll
    for (int h = 0; h < 4; ++h) { lstore2(smem, ra[h], sa[h], swz[h]); lstore2w(smem + 2 * PLANE, rb[h], sw[h], swz[h]); }
    gload_a(kbeg + 32); gload_b(kbeg + 32);
    __syncthreads();

    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += 32, cur ^= 1) {
        const char* Ac = smem + cur * BUF;
        const char* Bc = Ac + 2 * PLANE;
        char* An = smem + (cur ^ 1) * BUF;
        char* Bn = An + 2 * PLANE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {   // the slab's two 16-k halves; lane (row lr, lane >> 5) holds k = 16 s2 + 8 (lane >> 5) .. +7
            const int q = s2 * 2 + (lane >> 5);
            f16x8_t a[2][2], b[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    a[i][pc] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Ac + pc * PLANE + offa[i] + ((q ^ xa[i]) << 4)));
                    b[i][pc] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Bc + pc * PLANE + offb[i] + ((q ^ xb[i]) << 4)));
                }
            // term-major, eight independent accumulator chains; between the three groups of four MFMAs: the split of the NEXT slab
            // (first half: the A rows, second half: the W rows), then the loads of the slab after it
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
            if (s2 == 0) { lstore2(An, ra[0], sa[0], swz[0]); lstore2(An, ra[1], sa[1], swz[1]); } else { lstore2w(Bn, rb[0], sw[0], swz[0]); lstore2w(Bn, rb[1], sw[1], swz[1]); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
            if (s2 == 0) { lstore2(An, ra[2], sa[2], swz[2]); lstore2(An, ra[3], sa[3], swz[3]); gload_a(k0 + 64); }
            else { lstore2w(Bn, rb[2], sw[2], swz[2]); lstore2w(Bn, rb[3], sw[3], swz[3]); gload_b(k0 + 64); }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
        }
        __syncthreads();   // buffer cur^1 is complete, and every wave is done reading buffer cur
    }

    // epilogue: each wave stages a 32-row M-tile x its 64 columns (lin 32 | dir 32) in LDS, then activates ten points x 32 channels
    float* stg = reinterpret_cast<float*>(smem) + wave * STG;
    const int col_l = lane & 31, rowh = (lane >> 5) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ce = rsc[GM + wn * 64 + j * 32 + col_l];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int4 rs = *reinterpret_cast<const int4*>(&rsc[wm * 64 + i * 32 + 8 * r4 + rowh]);
                const int rev[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
                for (int rl = 0; rl < 4; ++rl) {
                    const int r = r4 * 4 + rl;
                    stg[((r & 3) + 8 * (r >> 2) + rowh) * 68 + j * 32 + col_l] = scale_pow2(acc[i][j][r], rev[rl] + ce);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        const int row0 = m0 + (wm * 2 + i) * TR;       // first row of this M-tile
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            const int item = it * 64 + lane, pl = item >> 5, c = item & 31;   // point 0..9 of the tile, channel
            const int grow = row0 + 3 * pl;
            float y0 = 0.f, y1 = 0.f, y2 = 0.f;
            if (grow < M) {
                const int ch = c0 + 32 * wn + c;
                const float* g = G + (size_t)((grow / 3) / npts) * 3 * ldg + 2 * C + ch;
                const float* sp = stg + 3 * pl * 68 + c;
                y0 = sp[0] + g[0]; y1 = sp[68] + g[ldg]; y2 = sp[136] + g[2 * ldg];
                const float k0 = sp[32] + g[C], k1 = sp[68 + 32] + g[ldg + C], k2 = sp[136 + 32] + g[2 * ldg + C];
                vn_act(y0, y1, y2, k0, k1, k2, oms);
                float* op = out + (size_t)grow * C + ch;
                op[0] = y0; op[C] = y1; op[2 * C] = y2;
            }
            if (aux.out_rowmax) {   // wave-uniform; [M][C / 32]: max|out[row, this wave's 32 channels]| (the half-wave of a point)
                float m0_ = max16(fabsf(y0)), m1_ = max16(fabsf(y1)), m2_ = max16(fabsf(y2));
                m0_ = fmaxf(m0_, __shfl_xor(m0_, 16, 64)); m1_ = fmaxf(m1_, __shfl_xor(m1_, 16, 64)); m2_ = fmaxf(m2_, __shfl_xor(m2_, 16, 64));
                if (c == 0 && grow < M) {
                    float* rp = aux.out_rowmax + (size_t)grow * (2 * ntiles_n) + 2 * tn + wn;
                    rp[0] = m0_; rp[2 * ntiles_n] = m1_; rp[4 * ntiles_n] = m2_;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K = 32: the per-point table GEMMs of encoder layers 1-2 (M = B*N*3 up to 196 608 rows, N = 128..384 columns).
// With two to four k-steps per 128x128 tile the kernel above is all prologue and epilogue: measured per workgroup (K = 64)
// 6.6 k cycles waiting for the first operand tile + 21.6 k in the k-loop (8.2 k of MFMA) + 21 k issuing the stores, with
// nothing of one tile overlapping anything of the next.  Here a workgroup is PERSISTENT over the M-tiles of one N-tile:
// the W tile (128 x K) is staged once, the whole-K A tile of M-tile t+1 is in flight (registers) under the MFMA block and
// the stores of tile t, one barrier pair per tile, and the staging area of the epilogue aliases the A buffer.
// whole-K A tile of M-tile TM into registers ra0..ra7: thread -> rows TM*128 + srow + u*RSTEP, 16 bytes at column scol.  Rows
// past M are clamped (computed, never stored); GATHER maps output row (b*gNd + n)*3 + x to source row
// (b*gNs + a_rows[b*gNd + n])*3 + x.  Named scalars, not an array: hipcc kept a 16-/32-float prefetch ARRAY in scratch memory
// (it is written in one loop iteration and read in the next; promote-alloca gave up on it).
#define LS_LOAD_A1(RA, U, TM)                                                               \
    {                                                                                       \
        const int gm = min((TM) * GM + srow + (U) * RSTEP, M - 1);                          \
        size_t r_ = (size_t)gm;                                                             \
        if constexpr (GATHER) {                                                             \
            const int pt = gm / 3, x = gm - pt * 3, bb = pt / gNd;                          \
            r_ = ((size_t)bb * gNs + a_rows[pt]) * 3 + x;                                   \
        }                                                                                   \
        RA = *reinterpret_cast<const float4*>(A + r_ * lda + scol);                         \
    }
#define LS_LOAD_A_TILE(TM)                                                                  \
    {                                                                                       \
        LS_LOAD_A1(ra0, 0, TM) LS_LOAD_A1(ra1, 1, TM) LS_LOAD_A1(ra2, 2, TM) LS_LOAD_A1(ra3, 3, TM) \
        if constexpr (PER == 8) { LS_LOAD_A1(ra4, 4, TM) LS_LOAD_A1(ra5, 5, TM) LS_LOAD_A1(ra6, 6, TM) LS_LOAD_A1(ra7, 7, TM) } \
    }
#define LS_STORE_A1(RA, U) *reinterpret_cast<float4*>(&As[(srow + (U) * RSTEP) * LD + scol]) = RA;

template <int KK, bool GATHER>
__global__ __launch_bounds__(256) void gemm_smallk_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias, float* __restrict__ out, int ldc, int M,
                                                          int N, int relu, int ntiles_m, int per_n,
                                                          const int32_t* __restrict__ a_rows, int gNd, int gNs) {
    constexpr int LD = KK + 4;                       // 36 / 68 floats: odd multiple of 16 bytes -> conflict-free ds_read_b128
    constexpr int STG = 32 * 68;
    constexpr int AS_FLOATS = (GM * LD > 4 * STG) ? GM * LD : 4 * STG;
    constexpr int F4_ROW = KK / 4;                   // float4 per operand row
    constexpr int PER = GM * F4_ROW / 256;           // float4 per thread per operand tile (4 / 8)
    __shared__ __attribute__((aligned(16))) float smem[AS_FLOATS + GN * LD];
    float* As = smem;                                // A tile, then the epilogue staging area
    float* Bs = smem + AS_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tn = blockIdx.x / per_n, slot = blockIdx.x % per_n;   // N-tile, and which M-tiles (slot, slot + per_n, ...)
    const int n0 = tn * GN;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lk = (lane >> 5) * 4;
    const int srow = tid / F4_ROW, scol = (tid % F4_ROW) * 4;       // staging: rows srow + (256 / F4_ROW) u
    constexpr int RSTEP = 256 / F4_ROW;
    const bool vec_ok = (ldc % 4 == 0) && (((uintptr_t)out & 15) == 0);

    // W tile once
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int r = srow + u * RSTEP, gn = n0 + r;
        // columns past N: clamped row, computed and never stored (a conditional load would be turned into a pointer select
        // through scratch memory by the compiler)
        *reinterpret_cast<float4*>(&Bs[r * LD + scol]) = *reinterpret_cast<const float4*>(W + (size_t)min(gn, N - 1) * ldw + scol);
    }
    float bv[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int gn = n0 + wn * 64 + j * 32 + lr;
        bv[j] = (bias && gn < N) ? bias[gn] : 0.0f;
    }

    float4 ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7;
    ra4 = ra5 = ra6 = ra7 = make_float4(0.f, 0.f, 0.f, 0.f);
    int tm = slot;
    LS_LOAD_A_TILE(min(tm, ntiles_m - 1))
    for (; tm < ntiles_m; tm += per_n) {
        const int m0 = tm * GM;
        __syncthreads();                               // previous tile's staging reads are done (and Bs is written)
        LS_STORE_A1(ra0, 0) LS_STORE_A1(ra1, 1) LS_STORE_A1(ra2, 2) LS_STORE_A1(ra3, 3)
        if constexpr (PER == 8) { LS_STORE_A1(ra4, 4) LS_STORE_A1(ra5, 5) LS_STORE_A1(ra6, 6) LS_STORE_A1(ra7, 7) }
        __syncthreads();
        // next A tile in flight under the MFMA block and the stores
        // (unconditional: the last iteration re-reads its own tile rather than branching around the loads)
        LS_LOAD_A_TILE(min(tm + per_n, ntiles_m - 1))

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
        for (int h = 0; h < KK / 8; ++h) {
            float4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const float4*>(&As[(wm * 64 + i * 32 + lr) * LD + h * 8 + lk]);
                b[i] = *reinterpret_cast<const float4*>(&Bs[(wn * 64 + i * 32 + lr) * LD + h * 8 + lk]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();                               // every wave is done with As: it becomes the staging area
        float* stg = As + wave * STG;
        const int col_l = lane & 31, rowh = (lane >> 5) * 4;
        const bool full_tile = vec_ok && (m0 + GM <= M) && (n0 + GN <= N);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] + bv[j];
                    if (relu) v = fmaxf(v, 0.0f);
                    stg[((r & 3) + 8 * (r >> 2) + rowh) * 68 + j * 32 + col_l] = v;
                }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
            store_half_tile(stg, out, ldc, M, N, m0 + wm * 64 + i * 32, n0 + wn * 64, lane, full_tile, vec_ok, nullptr);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K = 32 / 64 with the f16 split, PERSISTENT over the M-tiles of one N-tile (the per-point table GEMMs of encoder layers 1 - 4: 24 or
// 48 MFMAs against a 64 KB store per tile -- all prologue and epilogue in the tiled kernel).  The W tile is split once per workgroup; the
// whole-K A tile of M-tile t+1 is in flight (registers) under the MFMAs and the stores of tile t.  Same products and accumulation
// order as gemm_f32_kernel<true, 22>: bit-identical.
template <int KK, bool GATHER>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_h2_smallk_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, const float* __restrict__ bias, float* __restrict__ out, int ldc, int M,
    int N, int relu, int ntiles_m, int per_n, const int32_t* __restrict__ a_rows, int gNd, int gNs, GemmAux aux) {
    constexpr int STG = 32 * 68;
    constexpr int NSL = KK / 32;                    // 32-k slabs
    constexpr int PLANE = GM * 64;                  // one f16 plane of one slab: 128 rows x 32 k
    constexpr int OPER = 2 * NSL * PLANE;           // hi + lo planes of all slabs of one operand
    constexpr int ABYTES = (OPER > 4 * STG * 4) ? OPER : 4 * STG * 4;   // the A planes double as the epilogue staging area
    __shared__ __attribute__((aligned(16))) char smem[ABYTES + OPER];
    __shared__ __attribute__((aligned(16))) int rsc[GM + GN];   // inverse power-of-two scales of the current tile's A rows, then of the W rows (GemmAux)
    char* Ap = smem;
    char* Bp = smem + ABYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tn = blockIdx.x / per_n, slot = blockIdx.x % per_n;
    const int n0 = tn * GN;
    const int wm = wave >> 1, wn = wave & 1;
    const int sr0 = tid >> 3, sk = (tid & 7) * 4;   // staging: rows sr0 + 32 h, 16 bytes at column sk of a 32-k slab
    const bool vec_ok = (ldc % 4 == 0) && (((uintptr_t)out & 15) == 0);
    int swz[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int r = sr0 + h * 32;
        swz[h] = r * 64 + (((sk >> 3) ^ ((r >> 2) & 3)) << 4) + ((sk >> 2) & 1) * 8;
    }
    auto lstore2 = [&](char* base, int slab, const float4& v, float sc, int off) {
        uint2 ph, pl;
        split2_f16s<0>(v, sc, ph, pl);
        *reinterpret_cast<uint2*>(base + (2 * slab) * PLANE + off) = ph;
        *reinterpret_cast<uint2*>(base + (2 * slab + 1) * PLANE + off) = pl;
    };
    const bool scaled = !aux.noscale;
    if (!scaled) rsc[tid] = 0;
    // W tile, once (columns past N: clamped row, computed and never stored); each row scaled by its own power of two (GemmAux: the
    // whole K of a row is in this thread group's registers, so the row maximum costs three DPP steps)
    {
        float4 rw[NSL][4];
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int gn = min(n0 + sr0 + h * 32, N - 1);
                rw[sl][h] = *reinterpret_cast<const float4*>(W + (size_t)gn * ldw + sl * 32 + sk);
            }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float sw = 1.f;
            if (scaled) {
                float mw = 0.f, iw;
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) mw = amax4(mw, rw[sl][h]);
                pow2_scale(max8(mw), sw, iw);
                if ((tid & 7) == 0) rsc[GM + sr0 + h * 32] = pow2_e(iw);
            }
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) lstore2(Bp, sl, rw[sl][h], sw, swz[h]);
        }
    }
    float bv[2] = {0.f, 0.f};
    const int lr = lane & 31;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int gn = n0 + wn * 64 + j * 32 + lr;
        bv[j] = (bias && gn < N) ? bias[gn] : 0.0f;
    }
    int offa[2], offb[2], xa[2], xb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra_ = wm * 64 + i * 32 + lr, rb_ = wn * 64 + i * 32 + lr;
        offa[i] = ra_ * 64; offb[i] = rb_ * 64; xa[i] = (ra_ >> 2) & 3; xb[i] = (rb_ >> 2) & 3;
    }

    float4 ra[NSL][4];
    auto load_tile = [&](int tmx) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int gm = min(tmx * GM + sr0 + h * 32, M - 1);
            size_t r_ = (size_t)gm;
            if constexpr (GATHER) {
                const int pt = gm / 3, x = gm - pt * 3, bb = pt / gNd;
                r_ = ((size_t)bb * gNs + a_rows[pt]) * 3 + x;
            }
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) ra[sl][h] = *reinterpret_cast<const float4*>(A + r_ * lda + sl * 32 + sk);
        }
    };
    int tm = slot;
    load_tile(min(tm, ntiles_m - 1));
    for (; tm < ntiles_m; tm += per_n) {
        const int m0 = tm * GM;
        __syncthreads();                               // previous tile's staging and scale reads are done (and the W planes are written)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float sa = 1.f;
            if (scaled) {
                float ma = 0.f, ia;
#pragma unroll
                for (int sl = 0; sl < NSL; ++sl) ma = amax4(ma, ra[sl][h]);
                pow2_scale(max8(ma), sa, ia);
                if ((tid & 7) == 0) rsc[sr0 + h * 32] = pow2_e(ia);
            }
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) lstore2(Ap, sl, ra[sl][h], sa, swz[h]);
        }
        __syncthreads();
        load_tile(min(tm + per_n, ntiles_m - 1));      // next A tile in flight under the MFMAs and the stores (last iteration: re-reads its own)

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int q = s2 * 2 + (lane >> 5);
                f16x8_t a[2][2], b[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        a[i][pc] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Ap + (2 * sl + pc) * PLANE + offa[i] + ((q ^ xa[i]) << 4)));
                        b[i][pc] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Bp + (2 * sl + pc) * PLANE + offb[i] + ((q ^ xb[i]) << 4)));
                    }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
            }
        __syncthreads();                               // every wave is done with the A planes: they become the staging area
        float* stg = reinterpret_cast<float*>(Ap) + wave * STG;
        const int col_l = lane & 31, rowh = (lane >> 5) * 4;
        const bool full_tile = vec_ok && (m0 + GM <= M) && (n0 + GN <= N);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // (the row scales are read BEFORE the staging writes below: stg aliases the A planes, not rsc, but the next tile's scales
            //  are only written after the loop-top barrier)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ce = rsc[GM + wn * 64 + j * 32 + col_l];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int4 rs = *reinterpret_cast<const int4*>(&rsc[wm * 64 + i * 32 + 8 * r4 + rowh]);
                    const int rev[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
                    for (int rl = 0; rl < 4; ++rl) {
                        const int r = r4 * 4 + rl;
                        float v = scale_pow2(acc[i][j][r], rev[rl] + ce) + bv[j];
                        if (relu) v = fmaxf(v, 0.0f);
                        stg[((r & 3) + 8 * (r >> 2) + rowh) * 68 + j * 32 + col_l] = v;
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
            if (aux.slice_cols) store_half_tile_sliced(stg, out, M, N, m0 + wm * 64 + i * 32, n0 + wn * 64, lane, aux.slice_cols, aux.slice_rows);
            else store_half_tile(stg, out, ldc, M, N, m0 + wm * 64 + i * 32, n0 + wn * 64, lane, full_tile, vec_ok, nullptr);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// gemm_vn_kernel for K = 32 / 64 (the residual global conv of encoder layers 2 / 3: 98 304 rows, 128 tile columns, TWO 32-k slabs),
// PERSISTENT over the M-tiles of one channel block like gemm_h2_smallk_kernel (round 3).  In the tiled kernel such a problem is all
// prologue and epilogue -- load, split, barrier, 48 MFMAs, stage, activate, store, one after the other with two workgroups per CU:
// 34 us for 50 MB of compulsory traffic.  Here the weight tile is split once per workgroup, the whole-K A tile of M-tile t+1 is in flight
// (registers) under the MFMAs, the activation and the stores of tile t.  Same row / column maps, same products, same accumulation order
// and the same epilogue as gemm_vn_kernel: bit-identical results.
template <int KK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_vn_smallk_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw, const float* __restrict__ G, int ldg, float* __restrict__ out,
    int M, int C, int npts, float oms, int ntiles_m, int per_n, int ntiles_n, GemmAux aux) {
    constexpr int TR = 30;
    constexpr int STG = 32 * 68;
    constexpr int NSL = KK / 32;
    constexpr int PLANE = GM * 64;
    constexpr int OPER = 2 * NSL * PLANE;
    constexpr int ABYTES = (OPER > 4 * STG * 4) ? OPER : 4 * STG * 4;   // the A planes double as the epilogue staging area
    __shared__ __attribute__((aligned(16))) char smem[ABYTES + OPER];
    __shared__ __attribute__((aligned(16))) int rsc[GM + GN];
    char* Ap = smem;
    char* Bp = smem + ABYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tn = blockIdx.x / per_n, slot = blockIdx.x % per_n;
    const int c0 = tn * 64;
    const int wm = wave >> 1, wn = wave & 1;
    const int sr0 = tid >> 3, sk = (tid & 7) * 4;
    int swz[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const int r = sr0 + h * 32;
        swz[h] = r * 64 + (((sk >> 3) ^ ((r >> 2) & 3)) << 4) + ((sk >> 2) & 1) * 8;
    }
    auto lstore2 = [&](char* base, int slab, const float4& v, float sc, int off) {
        uint2 ph, pl;
        split2_f16s<0>(v, sc, ph, pl);
        *reinterpret_cast<uint2*>(base + (2 * slab) * PLANE + off) = ph;
        *reinterpret_cast<uint2*>(base + (2 * slab + 1) * PLANE + off) = pl;
    };
    const bool scaled = !aux.noscale;
    if (!scaled) rsc[tid] = 0;
    // W tile, once: staged row sr0 + 32 h = tile column (wn = h >> 1, lin | dir = h & 1, channel sr0)
    {
        float4 rw[NSL][4];
        int wrow[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            wrow[h] = (h & 1) * C + c0 + 32 * (h >> 1) + sr0;
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) rw[sl][h] = *reinterpret_cast<const float4*>(W + (size_t)wrow[h] * ldw + sl * 32 + sk);
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float sw = 1.f;
            if (scaled) {
                float mw = 0.f, iw;
                if (aux.w_rowmax) mw = aux.w_rowmax[wrow[h]];
                else {
#pragma unroll
                    for (int sl = 0; sl < NSL; ++sl) mw = amax4(mw, rw[sl][h]);
                    mw = max8(mw);
                }
                pow2_scale(mw, sw, iw);
                if ((tid & 7) == 0) rsc[GM + sr0 + h * 32] = pow2_e(iw);
            }
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) lstore2(Bp, sl, rw[sl][h], sw, swz[h]);
        }
    }
    const int lr = lane & 31;
    int offa[2], offb[2], xa[2], xb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra_ = wm * 64 + i * 32 + lr, rb_ = wn * 64 + i * 32 + lr;
        offa[i] = ra_ * 64; offb[i] = rb_ * 64; xa[i] = (ra_ >> 2) & 3; xb[i] = (rb_ >> 2) & 3;
    }
    float4 ra[NSL][4];
    int arow[4];   // source row of this thread's four staged rows (staged row sr0 + 32 h = row sr0 of M-tile h; rows past the tile / past M: clamped)
    auto load_tile = [&](int tmx) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            arow[h] = min(tmx * 4 * TR + h * TR + min(sr0, TR - 1), M - 1);
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) ra[sl][h] = *reinterpret_cast<const float4*>(A + (size_t)arow[h] * lda + sl * 32 + sk);
        }
    };
    int tm = slot;
    load_tile(min(tm, ntiles_m - 1));
    for (; tm < ntiles_m; tm += per_n) {
        const int m0 = tm * 4 * TR;
        __syncthreads();                               // previous tile's staging and scale reads are done (and the W planes are written)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float sa = 1.f;
            if (scaled) {
                float ma = 0.f, ia;
                if (aux.a_rowmax) {
                    for (int q = 0; q < aux.a_parts; ++q) ma = fmaxf(ma, aux.a_rowmax[(size_t)arow[h] * aux.a_parts + q]);
                } else {
#pragma unroll
                    for (int sl = 0; sl < NSL; ++sl) ma = amax4(ma, ra[sl][h]);
                    ma = max8(ma);
                }
                pow2_scale(ma, sa, ia);
                if ((tid & 7) == 0) rsc[sr0 + h * 32] = pow2_e(ia);
            }
#pragma unroll
            for (int sl = 0; sl < NSL; ++sl) lstore2(Ap, sl, ra[sl][h], sa, swz[h]);
        }
        __syncthreads();
        load_tile(min(tm + per_n, ntiles_m - 1));      // next A tile in flight under the MFMAs, the activation and the stores

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
        for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const int q = s2 * 2 + (lane >> 5);
                f16x8_t a[2][2], b[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pc = 0; pc < 2; ++pc) {
                        a[i][pc] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Ap + (2 * sl + pc) * PLANE + offa[i] + ((q ^ xa[i]) << 4)));
                        b[i][pc] = __builtin_bit_cast(f16x8_t, *reinterpret_cast<const uint4*>(Bp + (2 * sl + pc) * PLANE + offb[i] + ((q ^ xb[i]) << 4)));
                    }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
            }
        __syncthreads();                               // every wave is done with the A planes: they become the staging area
        float* stg = reinterpret_cast<float*>(Ap) + wave * STG;
        const int col_l = lane & 31, rowh = (lane >> 5) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ce = rsc[GM + wn * 64 + j * 32 + col_l];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int4 rs = *reinterpret_cast<const int4*>(&rsc[wm * 64 + i * 32 + 8 * r4 + rowh]);
                    const int rev[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
                    for (int rl = 0; rl < 4; ++rl) {
                        const int r = r4 * 4 + rl;
                        stg[((r & 3) + 8 * (r >> 2) + rowh) * 68 + j * 32 + col_l] = scale_pow2(acc[i][j][r], rev[rl] + ce);
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
            const int row0 = m0 + (wm * 2 + i) * TR;       // first row of this M-tile
#pragma unroll
            for (int it = 0; it < 5; ++it) {
                const int item = it * 64 + lane, pl = item >> 5, c = item & 31;   // point 0..9 of the tile, channel
                const int grow = row0 + 3 * pl;
                float y0 = 0.f, y1 = 0.f, y2 = 0.f;
                if (grow < M) {
                    const int ch = c0 + 32 * wn + c;
                    const float* g = G + (size_t)((grow / 3) / npts) * 3 * ldg + 2 * C + ch;
                    const float* sp = stg + 3 * pl * 68 + c;
                    y0 = sp[0] + g[0]; y1 = sp[68] + g[ldg]; y2 = sp[136] + g[2 * ldg];
                    const float k0 = sp[32] + g[C], k1 = sp[68 + 32] + g[ldg + C], k2 = sp[136 + 32] + g[2 * ldg + C];
                    vn_act(y0, y1, y2, k0, k1, k2, oms);
                    float* op = out + (size_t)grow * C + ch;
                    op[0] = y0; op[C] = y1; op[2 * C] = y2;
                }
                if (aux.out_rowmax) {   // wave-uniform; [M][C / 32]
                    float m0_ = max16(fabsf(y0)), m1_ = max16(fabsf(y1)), m2_ = max16(fabsf(y2));
                    m0_ = fmaxf(m0_, __shfl_xor(m0_, 16, 64)); m1_ = fmaxf(m1_, __shfl_xor(m1_, 16, 64)); m2_ = fmaxf(m2_, __shfl_xor(m2_, 16, 64));
                    if (c == 0 && grow < M) {
                        float* rp = aux.out_rowmax + (size_t)grow * (2 * ntiles_n) + 2 * tn + wn;
                        rp[0] = m0_; rp[2 * ntiles_n] = m1_; rp[4 * ntiles_n] = m2_;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The residual global conv of encoder layers 2 / 3 (K = C = 64, 98 304 rows: 25 MB in, 25 MB out, 1.6 GFLOP) as a STREAMING kernel: no workgroup barrier
// in its tile loop and no LDS transpose of the accumulators (round 5).  gemm_vn_smallk_kernel moves a 120-row tile through split -> barrier -> 48 MFMAs ->
// barrier -> LDS staging -> activation -> store with two workgroups per CU: 28 - 31 us for 6 us of HBM traffic.  Here a WAVE owns 32 output channels (lin AND
// dir columns: two 32-column MFMA tiles) of a stream of M-tiles; both W tiles stay split in 64 VGPRs for the whole launch.  The trick that removes the LDS
// transpose of the epilogue: an M-tile is EIGHT points laid out as MFMA rows 4 p + axis (axis 3 = a copy of z, dropped), so that in the 32x32 C/D map
// (col = lane & 31, rows 8 g + 4 (lane >> 5) + 0..3) a lane holds x, y, z of FOUR whole points for its channel in both accumulators: the VN activation is
// register-local.  Operands: the 24 rows of a tile are 6 KB contiguous -- fetched COALESCED (six 16-byte loads per lane) and turned into MFMA fragments through
// a wave-private, XOR-swizzled LDS scratch (no barrier: a wave's LDS operations complete in order); the weight blocks go through LDS once per workgroup.  (A
// lane that reads ITS row of a matrix from global memory 16 bytes at a time costs one L2 transaction per piece: the first forms of this kernel did, and
// took 25 - 32 us.)  Same operands (split2_f16s of the power-of-two scaled rows, scales from GemmAux::a_rowmax / w_rowmax), same three products per 16 k in
// the same order into one accumulator, ascending k, same integer-exponent epilogue, same activation: BIT-IDENTICAL to gemm_vn_kernel /
// gemm_vn_smallk_kernel (tests/test_hip_fullbatch.py).  25 % of the MFMA rows are padding: irrelevant, the matrix pipes are idle 90 % of the launch either way.
// Measured: 24 us in the encoder (alone 16: prologue 4.1, + loads / transpose / split 2.6, + MFMAs 4, + epilogue 5 -- two waves per SIMD do not overlap the
// phases of a tile; docs/history.md 13.6).
// Work split: a workgroup = two M-streams x two channel halves of ONE instance (its tiles t = stream, stream + streams per instance, ...), so the instance's
// offsets G (the mean part of the conv) are six registers per lane for the whole launch; the A rows of the next TWO tiles are in flight under a tile's
// MFMAs, activation and stores (one tile = 6 KB per wave).
// cs != null: G is not read but COMPUTED here from the partial column sums the attention kernel left (edge.hip: attn_colsum; [instance][cs_rows][3][C]) --
// the mean (sixteen row slices summed each, then the slices ascending), then the lane's six dot products with the W_b rows, k ascending in one fma chain
// -- what glob_mean_gemv_kernel would hand over up to the order of its dot products (it sums a weight row over sixteen lanes), without its launch (10 - 12 us
// on the critical path of layers 2 and 3).  Every path of the encoder that reaches this kernel carries column sums, so the two forms are never mixed.
template <int C, bool ONEPART>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_vn_direct_kernel(
    const float* __restrict__ A, const float* __restrict__ W, int ldw, const float* __restrict__ G, int ldg, float* __restrict__ out, int npts, int wgs_per_inst,
    float oms, GemmAux aux, const float* __restrict__ cs, int cs_rows, float cs_inv) {
    static_assert(C == 64, "two waves per M-stream, K = C");
    constexpr int KS = C / 16, NWN = C / 32, SPW = 4 / NWN;
    __shared__ __attribute__((aligned(16))) float lmean[3 * C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave % NWN, lr = lane & 31, kh = lane >> 5;
    const int b = blockIdx.x / wgs_per_inst, wg = blockIdx.x - b * wgs_per_inst;
    const int spi = wgs_per_inst * SPW;                  // M-streams per instance
    const int tiles_inst = npts / 8;
    const int ch = 32 * wn + lr;
    const int rofs = (lr >> 2) * 3 + min(lr & 3, 2);      // this lane's A row inside a tile (MFMA row lr = point lr / 4, axis lr % 4)
    // ---- the first two tiles' rows: requested before anything else.  A tile (24 rows x 256 B = 6 KB contiguous) is fetched COALESCED -- six 16-byte loads per
    // lane, every cache line by one instruction -- and transposed into MFMA fragments through a wave-private LDS scratch.  (First form of this kernel: every
    // lane read its own row in eight 16-byte pieces straight from global memory -- each 128-byte line was requested by eight different instructions, the L1
    // does not merge those misses, and the kernel took 25 - 32 us for 50 MB.)
    __shared__ __attribute__((aligned(16))) float4 stage[4][24 * 16];
    // (plain arrays and macros, not a struct handed to lambdas by reference: hipcc kept such a struct in scratch memory -- every tile stored and re-loaded)
    float4 ra0[6], ra1[6];
    float ma0 = 0.f, ma1 = 0.f;
#define LS_VND_LOAD(RA, MA, T)                                                                                                        \
    {                                                                                                                                 \
        const size_t R0 = ((size_t)b * tiles_inst + (T)) * 24;                                                                        \
        if constexpr (ONEPART) MA = aux.a_rowmax[R0 + rofs];                                                                          \
        else { MA = 0.f; for (int q = 0; q < aux.a_parts; ++q) MA = fmaxf(MA, aux.a_rowmax[(R0 + rofs) * aux.a_parts + q]); }         \
        const float4* ap = reinterpret_cast<const float4*>(A + R0 * C) + lane;                                                        \
        _Pragma("unroll") for (int i = 0; i < 6; ++i) RA[i] = ap[i * 64];                                                             \
    }
    int t = wg * SPW + wave / NWN;
    LS_VND_LOAD(ra0, ma0, min(t, tiles_inst - 1))            // (unconditional, clamped: the last trips re-read the instance's last tile and drop it)
    LS_VND_LOAD(ra1, ma1, min(t + spi, tiles_inst - 1))
    // ---- the weights go through LDS too: a [2C rows][C] block is fetched coalesced (eight 16-byte loads per thread) and read back row-per-lane.  (A lane
    // reading ITS weight row straight from global memory -- 16 + 32 loads of 16 bytes per lane, each its own L2 transaction -- cost 400 MB of L2 traffic
    // per launch for 64 KB of weights: 30 us.)  Row r's 16-byte slot s is kept at position s ^ (r & 15): conflict-free both ways.
    __shared__ __attribute__((aligned(16))) float4 wst[2 * C * (C / 4)];
    auto stage_w = [&](const float* Wb) {     // rows 0 .. 2C - 1 of Wb (row stride ldw) -> wst
#pragma unroll
        for (int i = 0; i < 2 * C * (C / 4) / 256; ++i) {
            const int idx = i * 256 + threadIdx.x, row = idx / (C / 4), slot = idx % (C / 4);
            wst[row * (C / 4) + (slot ^ (row & 15))] = *reinterpret_cast<const float4*>(Wb + (size_t)row * ldw + slot * 4);
        }
    };
    auto wslot = [&](int row, int slot) -> const float4& { return wst[row * (C / 4) + (slot ^ (row & 15))]; };
    stage_w(W);
    // the partial column sums of this instance (cs path), requested together with the weights
    float csv[32];
    if (cs && threadIdx.x < 3 * C) {   // (cs: kernel-uniform)
        const float* pc = cs + (size_t)b * cs_rows * 3 * C + threadIdx.x;
#pragma unroll
        for (int n = 0; n < 32; ++n) csv[n] = pc[(size_t)min(n, cs_rows - 1) * 3 * C];      // (clamped address + select below: a conditional load becomes 32 branches, each with its own wait)
    }
    __syncthreads();
    // ---- this wave's W tiles: row j C + ch (j = 0 lin, 1 dir), 8 consecutive k per k-step, split once
    f16x8_t bh[2][KS], bl[2][KS];
    int we[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int wr = j * C + ch;
        float sw, iw;
        pow2_scale(aux.w_rowmax[wr], sw, iw);
        we[j] = pow2_e(iw);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint2 h0, l0, h1, l1;
            split2_f16s<1>(wslot(wr, 4 * ks + 2 * kh), sw, h0, l0);
            split2_f16s<1>(wslot(wr, 4 * ks + 2 * kh + 1), sw, h1, l1);
            bh[j][ks] = __builtin_bit_cast(f16x8_t, make_uint4(h0.x, h0.y, h1.x, h1.y));
            bl[j][ks] = __builtin_bit_cast(f16x8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
        }
    }
    // ---- the instance's per-channel offsets: lin part gl[x] = G[b][x][2C + ch], dir part gd[x] = G[b][x][3C + ch]
    float gl[3], gd[3];
    if (cs) {   // kernel-uniform
        // mean: sixteen row slices (rows rs, rs + 16, ...) summed each, then the slices ascending, then * 1 / points (glob_mean_gemv_kernel's order)
        // (cs_rows <= 32, model.hip; absent rows add 0, which changes nothing)
        if (threadIdx.x < 3 * C) {
            float tsum = 0.f;
#pragma unroll
            for (int rs = 0; rs < 16; ++rs) {
                const float ssum = (0.f + (rs < cs_rows ? csv[rs] : 0.f)) + (rs + 16 < cs_rows ? csv[rs + 16] : 0.f);
                tsum = rs == 0 ? ssum : tsum + ssum;
            }
            lmean[threadIdx.x] = tsum * cs_inv;
        }
        __syncthreads();                                  // every wave has its W tiles: wst may be overwritten; lmean is complete
        stage_w(W + (size_t)2 * C * ldw);                 // (W = the conv's [4C][C] matrix: rows 2C .. 4C - 1 multiply the mean)
        __syncthreads();
        float al[3] = {0.f, 0.f, 0.f}, ad[3] = {0.f, 0.f, 0.f};
#pragma unroll 4
        for (int k4 = 0; k4 < C / 4; ++k4) {              // k ascending, one fma chain per output
            const float4 w0 = wslot(ch, k4), w1 = wslot(C + ch, k4);
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                const float4 mv = *reinterpret_cast<const float4*>(&lmean[x * C + 4 * k4]);
                al[x] = __builtin_fmaf(mv.x, w0.x, al[x]); al[x] = __builtin_fmaf(mv.y, w0.y, al[x]); al[x] = __builtin_fmaf(mv.z, w0.z, al[x]); al[x] = __builtin_fmaf(mv.w, w0.w, al[x]);
                ad[x] = __builtin_fmaf(mv.x, w1.x, ad[x]); ad[x] = __builtin_fmaf(mv.y, w1.y, ad[x]); ad[x] = __builtin_fmaf(mv.z, w1.z, ad[x]); ad[x] = __builtin_fmaf(mv.w, w1.w, ad[x]);
            }
        }
#pragma unroll
        for (int x = 0; x < 3; ++x) { gl[x] = al[x]; gd[x] = ad[x]; }
    } else {
        const float* g = G + (size_t)b * 3 * ldg + 2 * C + ch;
#pragma unroll
        for (int x = 0; x < 3; ++x) { gl[x] = g[x * ldg]; gd[x] = g[x * ldg + C]; }
    }
    // ---- the M-tiles of this wave's stream
#ifndef LS_VND_SKIP
#define LS_VND_SKIP 0      // dev timing variants (wrong results): 1 = no epilogue, 2 = no MFMAs, 4 = no tiles, 8 = no LDS transpose
#endif
    if (LS_VND_SKIP & 4) { if (bh[0][0][0] == (_Float16)123.f && gl[0] == 5.f) out[0] = 1.f; return; }
    auto do_tile = [&](float4 v0_, float4 v1_, float4 v2_, float4 v3_, float4 v4_, float4 v5_, float rma, int tt) {
        const float4 rv[6] = {v0_, v1_, v2_, v3_, v4_, v5_};     // (by value: an array handed over by reference stayed in scratch memory)
        float sa, ia;
        pow2_scale(rma, sa, ia);
        const int ea_own = pow2_e(ia);
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] = 0.0f;
        // transpose: lane L holds the 16-byte slot L % 16 of rows 4 i + L / 16; row r's slot s is kept at position s ^ (r & 15), so that both the writes
        // (a wave instruction = four whole rows) and the fragment reads (sixteen lanes = twelve consecutive rows, one slot) are free of bank conflicts
        float4* sg = stage[wave];
        // (wave-private scratch: the fences order this tile's stores after the previous tile's fragment reads and before this tile's -- no instruction,
        //  but the compiler may not move a may-alias LDS access across them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int row = 4 * i + (lane >> 4);
            if (!(LS_VND_SKIP & 8)) sg[row * 16 + ((lane & 15) ^ (row & 15))] = rv[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            uint2 h0, l0, h1, l1;
            const float4 v0 = (LS_VND_SKIP & 8) ? rv[ks] : sg[rofs * 16 + ((4 * ks + 2 * kh) ^ (rofs & 15))], v1 = (LS_VND_SKIP & 8) ? rv[ks + 1] : sg[rofs * 16 + ((4 * ks + 2 * kh + 1) ^ (rofs & 15))];
            split2_f16s<0>(v0, sa, h0, l0);
            split2_f16s<0>(v1, sa, h1, l1);
            const f16x8_t ah = __builtin_bit_cast(f16x8_t, make_uint4(h0.x, h0.y, h1.x, h1.y)), al = __builtin_bit_cast(f16x8_t, make_uint4(l0.x, l0.y, l1.x, l1.y));
            if (LS_VND_SKIP & 2) { acc[0][ks] += (float)ah[0] + (float)al[1] + (float)bh[0][ks][0]; acc[1][ks] += (float)ah[2] + (float)bl[1][ks][0]; continue; }
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j][ks], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j][ks], acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j][ks], acc[j], 0, 0, 0);
        }
        if (LS_VND_SKIP & 1) { if (acc[0][0] + acc[1][0] + acc[0][4] + acc[1][5] + acc[0][8] + acc[1][12] == 12345.f) out[tt] = 1.f; return; }
        // epilogue: accumulator rows 4 g + axis of this lane = point 2 g + kh of the tile
        const int p0 = (b * tiles_inst + tt) * 8;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int pl = 2 * gq + kh, pt = p0 + pl;
            int ea[3];
#pragma unroll
            for (int x = 0; x < 3; ++x) ea[x] = __shfl(ea_own, 4 * pl + x, 64);      // the row's scale sits with the lane that loaded the row
            float y0 = scale_pow2(acc[0][4 * gq + 0], ea[0] + we[0]) + gl[0];
            float y1 = scale_pow2(acc[0][4 * gq + 1], ea[1] + we[0]) + gl[1];
            float y2 = scale_pow2(acc[0][4 * gq + 2], ea[2] + we[0]) + gl[2];
            const float k0 = scale_pow2(acc[1][4 * gq + 0], ea[0] + we[1]) + gd[0];
            const float k1 = scale_pow2(acc[1][4 * gq + 1], ea[1] + we[1]) + gd[1];
            const float k2 = scale_pow2(acc[1][4 * gq + 2], ea[2] + we[1]) + gd[2];
            vn_act(y0, y1, y2, k0, k1, k2, oms);
            float* op = out + (size_t)pt * 3 * C + ch;
            op[0] = y0; op[C] = y1; op[2 * C] = y2;
            if (aux.out_rowmax) {   // kernel-uniform; [M][C / 32]: max|out[row, this wave's 32 channels]| (the half-wave of a point)
                float m0_ = max16(fabsf(y0)), m1_ = max16(fabsf(y1)), m2_ = max16(fabsf(y2));
                m0_ = fmaxf(m0_, __shfl_xor(m0_, 16, 64)); m1_ = fmaxf(m1_, __shfl_xor(m1_, 16, 64)); m2_ = fmaxf(m2_, __shfl_xor(m2_, 16, 64));
                if (lr == 0) {
                    float* rp = aux.out_rowmax + (size_t)pt * 3 * NWN + wn;
                    rp[0] = m0_; rp[NWN] = m1_; rp[2 * NWN] = m2_;
                }
            }
        }
    };
    for (; t < tiles_inst; t += 2 * spi) {      // two tiles per trip: the buffers keep their names (no register copies)
        do_tile(ra0[0], ra0[1], ra0[2], ra0[3], ra0[4], ra0[5], ma0, t);
        LS_VND_LOAD(ra0, ma0, min(t + 2 * spi, tiles_inst - 1))
        if (t + spi < tiles_inst) {
            do_tile(ra1[0], ra1[1], ra1[2], ra1[3], ra1[4], ra1[5], ma1, t + spi);
            LS_VND_LOAD(ra1, ma1, min(t + 3 * spi, tiles_inst - 1))
        }
    }
#undef LS_VND_LOAD
}

// split-K combine: out[m][n] = act(sum_s slab[s][m][n] + bias[n]), slices summed in ascending order (deterministic)
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ slabs, size_t slab_stride, int nsplit,
                                                                const float* __restrict__ bias, float* __restrict__ out, int ldc,
                                                                int M, int N, int relu) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // one float4 of a row (N is a multiple of 4 here)
    const int n4 = N / 4;
    if (i >= (long long)M * n4) return;
    const int m = (int)(i / n4), n = (int)(i % n4) * 4;
    float4 a = *reinterpret_cast<const float4*>(slabs + (size_t)m * N + n);
    for (int s2 = 1; s2 < nsplit; ++s2) {
        const float4 v = *reinterpret_cast<const float4*>(slabs + (size_t)s2 * slab_stride + (size_t)m * N + n);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (bias) { a.x += bias[n]; a.y += bias[n + 1]; a.z += bias[n + 2]; a.w += bias[n + 3]; }
    if (relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    float* op = out + (size_t)m * ldc + n;
    op[0] = a.x; op[1] = a.y; op[2] = a.z; op[3] = a.w;
}

// out[r] = max_k |W[r][k]| (weights: once per model, ls_model_create) -- the w_rowmax of GemmAux
__global__ __launch_bounds__(256) void gemm_rowmax_kernel(const float* __restrict__ W, int rows, int K, int ldw, float* __restrict__ out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    float m = 0.f;
    for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(W[(size_t)r * ldw + k]));
    m = wave_max(m);
    if (lane == 0) out[r] = m;
}
int gemm_rowmax_launch(const float* W, int rows, int K, int ldw, float* out, hipStream_t st) {
    hipLaunchKernelGGL(gemm_rowmax_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, W, rows, K, ldw, out);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int gemm_rowmax_parts(int N) { return 2 * cdiv(N, GN); }   // parts per row of GemmAux::out_rowmax for an N-column output

// GemmAux::w_planes of a weight matrix: row n = K / 32 lines of 128 bytes, line t = [hi piece of k = 32 t .. 32 t + 31 | lo piece of the
// same k] of s_n W[n, :] -- the split every workgroup would otherwise redo on its W slab, done once (same split2_f16s, same scale:
// bit-identical operands), laid out so that a staged slab row is one cache line
__global__ __launch_bounds__(256) void gemm_presplit_w_kernel(const float* __restrict__ W, int rows, int K, int ldw, const float* __restrict__ rowmax,
                                                              char* __restrict__ planes) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // one float4 of W
    const int k4 = K / 4;
    if (i >= (long long)rows * k4) return;
    const int r = (int)(i / k4), k = (int)(i % k4) * 4;
    float sc, inv;
    pow2_scale(rowmax[r], sc, inv);
    uint2 ph, pl;
    split2_f16s<1>(*reinterpret_cast<const float4*>(W + (size_t)r * ldw + k), sc, ph, pl);
    char* line = planes + (size_t)r * ((size_t)K * 4) + (size_t)(k >> 5) * 128 + (k & 31) * 2;
    *reinterpret_cast<uint2*>(line) = ph;
    *reinterpret_cast<uint2*>(line + 64) = pl;
}
size_t gemm_w_planes_bytes(size_t rows, int K) { return rows * (size_t)K * 4; }
// the kernels that read planes: gemm_h2_kernel<true, true>, gemm_w2_kernel<., true>.  Measured (scripts/diag/gemm_wide_probe.py): -5 % at
// the decoder shape (992 -> 941 us wide, 1198 -> 1128 us narrow), neutral at K = 512, +8 .. 15 % on the K = 128 / 256 tables: K >= 512 only
bool gemm_w_planes_useful(int K) { return K >= 512 && K % 32 == 0; }
int gemm_presplit_w_launch(const float* W, int rows, int K, int ldw, const float* rowmax, void* planes, hipStream_t st) {
    LS_REQUIRE(K % 8 == 0 && ((uintptr_t)planes % 16) == 0, "gemm_presplit_w: K must be a multiple of 8 and the planes 16-byte aligned (K=%d)", K);
    hipLaunchKernelGGL(gemm_presplit_w_kernel, dim3(cdiv((long long)rows * (K / 4), 256)), dim3(256), 0, st, W, rows, K, ldw, rowmax,
                       static_cast<char*>(planes));
    LS_LAUNCH_CHECK();
    return LS_OK;
}

// How an fp32 product is formed, process-wide (read once): 0 = two f16 pieces with per-row power-of-two scaling, three MFMAs per 16 k (default);
// 1 = LS_GEMM_MODE=bf16x3: three bf16 pieces, six MFMAs, any finite fp32 range without scaling; 2 = LS_GEMM_MODE=fp32: exact fp32 FMA chains on
// v_mfma_f32_32x32x2_f32 (bit-for-bit comparison with an fp32 VALU GEMM).  The fused edge / global-conv kernels exist for mode 0 only.
int gemm_mode() {
    static const int mode = [] {
        const char* v = getenv("LS_GEMM_MODE");
        return !v ? 0 : (!strcmp(v, "bf16x3") ? 1 : (!strcmp(v, "fp32") ? 2 : 0));
    }();
    return mode;
}

// Under-filled grids with a long K loop (the per-instance "mean" rows of the residual global conv: M = 3B rows against
// K = C up to 512; conv_c) are pure latency: 32 workgroups x 16 dependent k-steps = 44 us for 0.2 GFLOP.  They are split
// along K into slices written as partial slabs and combined by a second launch.
static int gemm_choose_splits(int M, int N, int K) {
    const int tiles = cdiv(M, GM) * cdiv(N, GN);
    if (tiles >= 192 || K < 128 || N % 4 != 0) return 1;
    int s2 = 512 / tiles;
    if (s2 > K / 32) s2 = K / 32;
    return s2 < 2 ? 1 : s2;
}
size_t gemm_scratch_floats(int M, int N, int K) {
    const int s2 = gemm_choose_splits(M, N, K);
    return s2 > 1 ? (size_t)s2 * M * N : 0;
}

int gemm_dispatch_full(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M, int N, int K,
                       int relu, const int32_t* a_rows, int gNd, int gNs, float* scratch, hipStream_t st, bool latency_path = false,
                       int pieces = 3, const float* mask = nullptr, GemmAux aux = GemmAux()) {
    static const bool range_off = dev_knob("LS_GEMM_RANGE", 1) == 0;   // dev A/B: the unscaled round-2 split
    if (range_off) aux.noscale = 1;
    LS_REQUIRE(M > 0 && N > 0 && K > 0, "gemm: empty problem (M=%d N=%d K=%d)", M, N, K);
    LS_REQUIRE(K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0, "gemm: K, lda, ldw must be multiples of 4 (K=%d lda=%d ldw=%d)", K, lda, ldw);
    LS_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "gemm: A and W must be 16-byte aligned");
    const int tm = cdiv(M, GM), tn = cdiv(N, GN);
    const bool split_on = gemm_mode() != 2;      // LS_GEMM_MODE=fp32: exact fp32 FMA chains on v_mfma_f32_32x32x2_f32
    // (fp32 mode only: with three-piece bf16 products the tiled kernel below is faster on the K = 32 tables too -- 27.8 / 42.8 /
    // 34.5 us vs 28.8 / 47.2 / 38.8 us for the three layer-1/2 shapes -- and the arithmetic then depends on nothing but K)
    if (K == 32 && tm >= 16 && !split_on) {
        // persistent small-K kernel: ~3 resident workgroups per CU, spread evenly over the N-tiles.  (The K = 64 instantiation
        // needs 70 KB of LDS -> 2 workgroups per CU and measured SLOWER than the tiled kernel: 138 vs 110 us at the layer-3 shape.)
        int per_n = cdiv(768, tn);
        if (per_n > tm) per_n = tm;
#define LS_SMALLK(KK, G)                                                                                                          \
    hipLaunchKernelGGL((gemm_smallk_kernel<KK, G>), dim3(tn * per_n), dim3(256), 0, st, A, lda, W, ldw, bias, out, ldc, M, N, relu, tm, \
                       per_n, a_rows, gNd, gNs)
        if (a_rows) LS_SMALLK(32, true); else LS_SMALLK(32, false);
#undef LS_SMALLK
        LS_LAUNCH_CHECK();
        return LS_OK;
    }
    // LS_GEMM_BF16X3=0: exact fp32 FMA chains on v_mfma_f32_32x32x2_f32 (A/B timing, bit-for-bit comparison with earlier builds)
    // The arithmetic must not depend on M (a decode of one instance's points has to equal the same rows inside a batched decode),
    // so the choice is the CALLER's: latency_path = a handful of tiles by construction (the per-instance mean rows of the global
    // conv, M = 3B), where the fp32 kernel's shorter slab (16 k, no split arithmetic before the first MFMA) wins: 44 vs 112 us
    // at M = 192, N = 1024, K = 512
    const bool split = split_on && !latency_path;
    // how an fp32 product is formed on the 16-bit matrix cores: 22 = two f16 pieces (three MFMAs per 16 k, the
    // default), 3 = three bf16 pieces (six MFMAs, any fp32 range: LS_GEMM_MODE=bf16x3), 2 = two bf16 pieces (opt-in decode mode)
    static const bool h2_unpipelined = dev_knob("LS_GEMM_H2_SIMPLE", 0) != 0;   // dev A/B: the two-barrier kernel
    static const bool planes_off = dev_knob("LS_GEMM_WPLANES", 1) == 0;         // dev A/B: split W inside the kernel
    const bool wpl = aux.w_planes && aux.w_rowmax && !aux.noscale && !planes_off && K % 32 == 0 && K > 64 && !h2_unpipelined;
#define LS_H2_KERNEL ((h2_unpipelined || K <= 64) ? gemm_f32_kernel<true, 22> : (K % 32 == 0 ? (wpl ? gemm_h2_kernel<true, true> : gemm_h2_kernel<true, false>) : gemm_h2_kernel<false, false>))
    const int default_pieces = gemm_mode() == 1 ? 3 : 22;
    if (pieces == 3) pieces = default_pieces;
    const int nsplit = (scratch && !mask) ? gemm_choose_splits(M, N, K) : 1;   // (a split launch writes no out_rowmax: callers check gemm_scratch_floats)
    if (nsplit > 1) {
        const int kq = split ? 32 : GK;
        int kchunk = cdiv(cdiv(K, nsplit), kq) * kq;
        const int ns = cdiv(K, kchunk);
        const size_t slab = (size_t)M * N;
        if (split && pieces == 22)
            hipLaunchKernelGGL(LS_H2_KERNEL, dim3(tm * tn, ns), dim3(256), 0, st, A, lda, W, ldw, nullptr, scratch, N, M, N, K, 0, tn,
                               a_rows, gNd, gNs, kchunk, slab, (const float*)nullptr, aux);
        else if (split)
            hipLaunchKernelGGL(gemm_f32_kernel<true>, dim3(tm * tn, ns), dim3(256), 0, st, A, lda, W, ldw, nullptr, scratch, N, M, N, K, 0, tn,
                               a_rows, gNd, gNs, kchunk, slab, (const float*)nullptr, aux);
        else
            hipLaunchKernelGGL(gemm_f32_kernel<false>, dim3(tm * tn, ns), dim3(256), 0, st, A, lda, W, ldw, nullptr, scratch, N, M, N, K, 0, tn,
                               a_rows, gNd, gNs, kchunk, slab, (const float*)nullptr, aux);
        LS_LAUNCH_CHECK();
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(cdiv((long long)M * (N / 4), 256)), dim3(256), 0, st, scratch, slab, ns, bias, out,
                           ldc, M, N, relu);
        LS_LAUNCH_CHECK();
        return LS_OK;
    }
    static const bool persist = dev_knob("LS_GEMM_PERSIST", 1) != 0;   // dev A/B: K = 32 / 64 on the tiled kernel
    // 256 x 256 tiles once they fill the chip (LS_GEMM_WIDE=0 / 1: never / always -- same arithmetic, bit-identical results)
    static const int wide_mode = dev_knob("LS_GEMM_WIDE", -1);
    // measured (scripts/diag/gemm_wide_probe.py): the wide kernel wins when its grid fills whole rounds of the 256 CUs (one workgroup per
    // CU): 480 tiles 88 -> 73 us, 768 tiles 355 -> 280 us, 3072 tiles 1186 -> 1002 us; ties at 384 tiles, loses below one round
    const long long wtiles = (long long)cdiv(M, 256) * cdiv(N, 256);
    const bool wide_on = wide_mode >= 0 ? wide_mode != 0 : (wtiles >= 1024 || (wtiles >= 256 && wtiles * 100 >= 85 * 256 * cdiv(wtiles, 256)));
    const bool sliced = aux.slice_cols != 0;
    if (sliced)
        LS_REQUIRE(split && pieces == 22 && !mask && !a_rows && (K == 32 || K == 64) && nsplit == 1 && !aux.out_rowmax && !bias && (aux.slice_cols == 4 || aux.slice_cols == 8) &&
                   aux.slice_rows % 32 == 0 && M % aux.slice_rows == 0 && N % aux.slice_cols == 0 && ((uintptr_t)out & 15) == 0,
                   "gemm: slice-major output needs the K = 32 / 64 f16-split kernel (M=%d N=%d K=%d slice %d x %d)", M, N, K, aux.slice_rows, aux.slice_cols);
    if (split && pieces == 22 && (sliced || (persist && tm >= 16 && !h2_unpipelined)) && !mask && (K == 32 || K == 64) && !aux.out_rowmax) {
        static const int sk_wgs32 = dev_knob("LS_GEMM_PERSIST_WGS32", 512), sk_wgs64 = dev_knob("LS_GEMM_PERSIST_WGS64", 512);   // dev A/B
        int per_n = cdiv(K == 32 ? sk_wgs32 : sk_wgs64, tn);   // resident workgroups per CU x 256, spread evenly over the N-tiles
        if (per_n > tm) per_n = tm;
#define LS_H2SK(KK, G) hipLaunchKernelGGL((gemm_h2_smallk_kernel<KK, G>), dim3(tn * per_n), dim3(256), 0, st, A, lda, W, ldw, bias, out, ldc, M, N, relu, tm, per_n, a_rows, gNd, gNs, aux)
        if (K == 32) { if (a_rows) LS_H2SK(32, true); else LS_H2SK(32, false); }
        else { if (a_rows) LS_H2SK(64, true); else LS_H2SK(64, false); }
#undef LS_H2SK
    } else if (split && pieces == 22 && wide_on && !a_rows && K % 32 == 0 && K >= 128 && (unsigned long long)M * lda < (1ull << 30) &&
               (unsigned long long)N * std::max(ldw, K) < (1ull << 30)) {   // (the wide kernel addresses its operands by 32-bit byte offsets)
        const int wtm = cdiv(M, 256), wtn = cdiv(N, 256);
        const size_t lds = 2 * 4 * 256 * 64 + 512 * sizeof(float);
        // the dynamic-LDS opt-in is a per-DEVICE function attribute: one flag per device ordinal (a process may drive several GPUs)
        static std::atomic<unsigned long long> attr_devices{0};
        int dev_ord = 0;
        LS_HIP_CHECK(hipGetDevice(&dev_ord));
        const unsigned long long dev_bit = 1ull << (dev_ord & 63);
        if (!(attr_devices.load(std::memory_order_acquire) & dev_bit)) {
#define LS_W2_ATTR(MK, PL, PG) LS_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_w2_kernel<MK, PL, PG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
            LS_W2_ATTR(false, 0, false); LS_W2_ATTR(true, 0, false); LS_W2_ATTR(false, 1, false); LS_W2_ATTR(true, 1, false);
#ifdef LS_DEV_KNOBS
            LS_W2_ATTR(false, 0, true); LS_W2_ATTR(true, 0, true); LS_W2_ATTR(false, 1, true); LS_W2_ATTR(true, 1, true);
            LS_W2_ATTR(false, 2, false); LS_W2_ATTR(true, 2, false);
#endif
#undef LS_W2_ATTR
            attr_devices.fetch_or(dev_bit, std::memory_order_release);
        }
        // dev variants (-DLS_DEV_KNOBS; all bit-identical to the default, all measured and not kept -- docs/history.md): LS_GEMM_W2_PERSIST=1 one workgroup
        // per CU walking the tiles (942 - 944 vs 950 - 956 us at the decoder shape, 73 -> 79 us at 480 tiles); LS_GEMM_W2_PP=1 the two waves of a SIMD half
        // a slab step apart (957 vs 928 us); LS_GEMM_W2_DIRECT=1 the W planes by LDS-direct loads (907 / 933 vs 915 / 925 us: the kernel is power-bound)
        static const bool w2_persist = dev_knob("LS_GEMM_W2_PERSIST", 0) != 0;
        const int w2_grid = w2_persist ? std::min(wtm * wtn, 256) : wtm * wtn;
#define LS_W2(MK, PL, PG) hipLaunchKernelGGL((gemm_w2_kernel<MK, PL, PG>), dim3(w2_grid), dim3(512), lds, st, A, lda, W, ldw, bias, out, ldc, M, N, K, relu, wtn, wtm * wtn, mask, aux)
#ifdef LS_DEV_KNOBS
        static const bool w2_pp = dev_knob("LS_GEMM_W2_PP", 0) != 0, w2_direct = dev_knob("LS_GEMM_W2_DIRECT", 0) != 0;
#define LS_W2P(MK, PL) do { if (w2_pp) LS_W2(MK, PL, true); else if (PL == 1 && w2_direct) LS_W2(MK, 2, false); else LS_W2(MK, PL, false); } while (0)
#else
#define LS_W2P(MK, PL) LS_W2(MK, PL, false)
#endif
        if (mask) { if (wpl) LS_W2P(true, 1); else LS_W2P(true, 0); }
        else { if (wpl) LS_W2P(false, 1); else LS_W2P(false, 0); }
#undef LS_W2P
#undef LS_W2
    } else if (split && pieces == 22)
        hipLaunchKernelGGL(LS_H2_KERNEL, dim3(tm * tn), dim3(256), 0, st, A, lda, W, ldw, bias, out, ldc, M, N, K, relu, tn, a_rows,
                           gNd, gNs, K, (size_t)0, mask, aux);
    else if (split && pieces == 2)
        hipLaunchKernelGGL((gemm_f32_kernel<true, 2>), dim3(tm * tn), dim3(256), 0, st, A, lda, W, ldw, bias, out, ldc, M, N, K, relu, tn, a_rows,
                           gNd, gNs, K, (size_t)0, mask, aux);
    else if (split)
        hipLaunchKernelGGL(gemm_f32_kernel<true>, dim3(tm * tn), dim3(256), 0, st, A, lda, W, ldw, bias, out, ldc, M, N, K, relu, tn, a_rows,
                           gNd, gNs, K, (size_t)0, mask, aux);
    else
        hipLaunchKernelGGL(gemm_f32_kernel<false>, dim3(tm * tn), dim3(256), 0, st, A, lda, W, ldw, bias, out, ldc, M, N, K, relu, tn, a_rows,
                           gNd, gNs, K, (size_t)0, mask, aux);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int gemm_dispatch_gather(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M, int N,
                         int K, int relu, const int32_t* a_rows, int gNd, int gNs, hipStream_t st, GemmAux aux) {
    return gemm_dispatch_full(A, lda, W, ldw, bias, out, ldc, M, N, K, relu, a_rows, gNd, gNs, nullptr, st, false, 3, nullptr, aux);
}
int gemm_dispatch(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M, int N,
                  int K, int relu, hipStream_t st, GemmAux aux) {
    return gemm_dispatch_full(A, lda, W, ldw, bias, out, ldc, M, N, K, relu, nullptr, 0, 0, nullptr, st, false, 3, nullptr, aux);
}
int gemm_dispatch_ws(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M, int N,
                     int K, int relu, float* scratch, hipStream_t st, GemmAux aux) {
    return gemm_dispatch_full(A, lda, W, ldw, bias, out, ldc, M, N, K, relu, nullptr, 0, 0, scratch, st, false, 3, nullptr, aux);
}
// opt-in two-piece products (decoder throughput mode); never splits K
int gemm_dispatch_fast2(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M, int N,
                        int K, int relu, hipStream_t st) {
    return gemm_dispatch_full(A, lda, W, ldw, bias, out, ldc, M, N, K, relu, nullptr, 0, 0, nullptr, st, false, 2);
}
// out = (mask > 0) ? A W^T : 0 with `mask` laid out like `out` (never splits K); pieces = 3 | 2
int gemm_dispatch_masked(const float* A, int lda, const float* W, int ldw, float* out, int ldc, int M, int N, int K, const float* mask, int pieces,
                         hipStream_t st, GemmAux aux) {
    return gemm_dispatch_full(A, lda, W, ldw, nullptr, out, ldc, M, N, K, 0, nullptr, 0, 0, nullptr, st, false, pieces, mask, aux);
}
// out [M = B * npts * 3, C] = VN-act(A W[0:C]^T + G lin part, A W[C:2C]^T + G dir part): see gemm_vn_kernel.  false = shape / mode not
// supported (the caller runs GEMM + vn_act_rows instead)
bool gemm_vn_supported(int M, int C, int K) {
    return gemm_mode() == 0 && C % 64 == 0 && K % 4 == 0 && K >= 32 && M % 3 == 0;
}
// does gemm_vn_dispatch take the streaming kernel (gemm_vn_direct_kernel) for this problem?  Only then is GemmAux::cs honoured (model.hip: global_conv)
bool gemm_vn_streams(int M, int C, int K, int lda, int npts, const GemmAux& aux) {
    return gemm_vn_supported(M, C, K) && K == 64 && C == 64 && lda == K && npts % 8 == 0 && aux.a_rowmax && aux.a_parts > 0 && aux.w_rowmax && !aux.noscale &&
           M >= 24 * 64 && dev_knob("LS_GLOB_DIRECT", 1) != 0;
}
int gemm_vn_dispatch(const float* A, int lda, const float* W, int ldw, const float* G, int ldg, float* out, int M, int C, int K, int npts, float oms,
                     hipStream_t st, GemmAux aux) {
    static const bool range_off = dev_knob("LS_GEMM_RANGE", 1) == 0;
    if (range_off) aux.noscale = 1;
    LS_REQUIRE(gemm_vn_supported(M, C, K) && lda % 4 == 0 && ldw % 4 == 0, "gemm_vn: unsupported shape (M=%d C=%d K=%d)", M, C, K);
    const int tm = cdiv(M, 120), tn = C / 64;
    const bool direct = true;   // (dev A/B: LS_GLOB_DIRECT=0 inside gemm_vn_streams)
    if (direct && gemm_vn_streams(M, C, K, lda, npts, aux) && (G || aux.cs)) {
        const int B = M / (3 * npts), tiles_inst = npts / 8;
        int wpi = cdiv(512, B);                                   // ~512 workgroups (two per CU), every one inside one instance
        wpi = std::max(1, std::min(wpi, cdiv(tiles_inst, 2)));
#define LS_VND(ONE) hipLaunchKernelGGL((gemm_vn_direct_kernel<64, ONE>), dim3(B * wpi), dim3(256), 0, st, A, W, ldw, G, ldg, out, npts, wpi, oms, aux, aux.cs, aux.cs_rows, 1.0f / (float)npts)
        if (aux.a_parts == 1) LS_VND(true); else LS_VND(false);
#undef LS_VND
        LS_LAUNCH_CHECK();
        return LS_OK;
    }
    static const bool persist = dev_knob("LS_GLOB_PERSIST", 1) != 0;   // dev A/B: K = 32 / 64 on the tiled kernel
    if (persist && (K == 32 || K == 64) && tm >= 16 && lda == K) {
        static const int vn_wgs32 = dev_knob("LS_GLOB_PERSIST_WGS32", 512);   // dev A/B
        int per_n = cdiv(K == 32 ? vn_wgs32 : 512, tn);   // resident workgroups per CU x 256
        if (per_n > tm) per_n = tm;
        if (K == 32) hipLaunchKernelGGL(gemm_vn_smallk_kernel<32>, dim3(tn * per_n), dim3(256), 0, st, A, lda, W, ldw, G, ldg, out, M, C, npts, oms, tm, per_n, tn, aux);
        else hipLaunchKernelGGL(gemm_vn_smallk_kernel<64>, dim3(tn * per_n), dim3(256), 0, st, A, lda, W, ldw, G, ldg, out, M, C, npts, oms, tm, per_n, tn, aux);
        LS_LAUNCH_CHECK();
        return LS_OK;
    }
    if (K % 32 == 0) hipLaunchKernelGGL(gemm_vn_kernel<true>, dim3(tm * tn), dim3(256), 0, st, A, lda, W, ldw, G, ldg, out, M, C, K, npts, oms, tn, aux);
    else hipLaunchKernelGGL(gemm_vn_kernel<false>, dim3(tm * tn), dim3(256), 0, st, A, lda, W, ldw, G, ldg, out, M, C, K, npts, oms, tn, aux);
    LS_LAUNCH_CHECK();
    return LS_OK;
}
int gemm_dispatch_small(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M, int N,
                        int K, int relu, float* scratch, hipStream_t st) {
    return gemm_dispatch_full(A, lda, W, ldw, bias, out, ldc, M, N, K, relu, nullptr, 0, 0, scratch, st, true);
}

}  // namespace ls
