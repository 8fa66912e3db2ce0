// edge.hip -- the VN-DGCNN edge-conv message passing, fused: neighbour gather -> VN-Linear (as pre-computed
// per-point tables) -> VN-LeakyReLU -> mean-pool  OR  K/Q/V + head soft-max + weighted sum.
//
// Replaces, per layer i of VecDGCNN_att.forward
// (/root/reference/lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:196-219):
//   get_graph_feature's cat([nbr - ctr, ctr]) (:160; + cross term at layer 0, :154-158),
//   V_list[i] / K_list[i] / Q_list[i]  VecLNA (vec_layers.py:523-534 = VecLinear :121-136 + VecActivation :241-268),
//   mean pool (:204) or channel_equi_vec_normalize + QK soft-max attention (:209-219; vec_layers.py:24-31).
//
// The reference materialises E[n,k] = [src[nbr]-dst[n] ; dst[n]] as a [B,2C,3,N,K] tensor and runs two dense
// channel contractions (lin: 2C->Co, lin_dir: Co->Co) on every one of the N*K edges.  Both are linear, so
//     lin(E)[n,k]     = W1 src[nbr] + (W2-W1) dst[n]            =: P_lin[nbr] + Q_lin[n]
//     lin_dir(lin(E)) = (Wd W1) src[nbr] + (Wd (W2-W1)) dst[n]  =: P_dir[nbr] + Q_dir[n]
// i.e. ONE per-point GEMM (gemm.hip, folded weights from packing.py) produces a table T[b, point, xyz, cols]
// and the per-edge work that remains is a gather plus ~35 VALU ops per (edge, channel): K=16x fewer MFMA FLOPs
// and an HBM/L2-gather-bound kernel.  Nothing of size N*K is ever written to memory.
//
// Table columns (Co = layer width):  pool layers  [PV_lin | PV_dir | QV_lin | QV_dir]
//                                    attn layers  [PV_lin | PV_dir | PK_lin | PK_dir | QV_lin | QV_dir | QK_lin | QK_dir | Qq_lin | Qq_dir]
// Attention layers with C_out = 64 / 128 (released encoder: layers 2 - 4) do NOT materialise the Q* column groups: edge_attn_fq_kernel
// computes them per workgroup on the f16 matrix cores from the destination points' feature rows (see there); their table holds the
// four P* groups only.
// Thread mapping: one wave per destination point, lanes = channels (coalesced 256-B row segments per
// neighbour and xyz component), channel chunks of 64; attention heads are 16 consecutive channels = one
// 16-lane DPP row, so head sums / soft-max reductions are 4-step row shuffles.
#include "ls_common.h"

namespace ls {

constexpr int EK = 16;  // neighbours per point (num_knn)

// ---------------------------------------------------------------------------------------------- layer 0
// pts [B,N,3]; w0 [6][Co] = {W[:,0], W[:,1], W[:,2], (Wd W)[:,0], (Wd W)[:,1], (Wd W)[:,2]}; out [B,N,3,Co]
__global__ __launch_bounds__(256) void edge_l0_kernel(const float* __restrict__ pts, const int32_t* __restrict__ knn,
                                                      const float* __restrict__ w0, int N, int Co, float oms,
                                                      float* __restrict__ out, int total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 5, o = lane & 31;  // two points per wave, 32 lanes each (Co <= 32 per pass)
    int pid = (xcd_remap(blockIdx.x, gridDim.x) * 4 + wave) * 2 + sub;
    const bool live = pid < total;
    if (!live) pid = total - 1;
    const int b = pid / N;
    const float* P = pts + (size_t)b * N * 3;
    const float cx = pts[(size_t)pid * 3 + 0], cy = pts[(size_t)pid * 3 + 1], cz = pts[(size_t)pid * 3 + 2];
    const float inv = 1.0f / fmaxf(sqrtf(cx * cx + cy * cy + cz * cz), 1e-12f);
    const float ax = cx * inv, ay = cy * inv, az = cz * inv;
    const int32_t* ki = knn + (size_t)pid * EK;
    // Round 4: what an edge contributes BEFORE the weights -- the cross product with the centre direction and the difference to the centre, six
    // floats -- does not depend on the output channel, yet each of the 32 channel lanes of a point recomputed it (three loads, an index load and
    // ~14 of the ~50 VALU instructions per edge, in a kernel that runs at 98 % of the VALU issue rate).  Lane o < 16 of a point now does it once
    // for edge o and leaves it in a wave-private LDS slot; the edge loop reads it back with two broadcast loads.  Same operations per element.
    __shared__ __attribute__((aligned(16))) float l_edge[4][2][EK][8];
    if (o < EK) {
        const int r = ki[o];
        const float nx = P[(size_t)r * 3 + 0], ny = P[(size_t)r * 3 + 1], nz = P[(size_t)r * 3 + 2];
        float* le = l_edge[wave][sub][o];
        *reinterpret_cast<float4*>(le) = make_float4(ay * nz - az * ny, az * nx - ax * nz, ax * ny - ay * nx, nx - cx);
        *reinterpret_cast<float2*>(le + 4) = make_float2(ny - cy, nz - cz);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the slots are wave-private, LDS operations of a wave complete in order
    __builtin_amdgcn_wave_barrier();
    for (int c0 = 0; c0 < Co; c0 += 32) {
        const int oc = c0 + o;
        const bool on = oc < Co;
        const int ow = on ? oc : 0;
        const float a0 = w0[0 * Co + ow], a1 = w0[1 * Co + ow], a2 = w0[2 * Co + ow];
        const float d0 = w0[3 * Co + ow], d1 = w0[4 * Co + ow], d2 = w0[5 * Co + ow];
        // the centre-point terms do not depend on the edge
        const float yc0 = a2 * cx, yc1 = a2 * cy, yc2 = a2 * cz, kc0 = d2 * cx, kc1 = d2 * cy, kc2 = d2 * cz;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 4
        for (int k = 0; k < EK; ++k) {
            const float4 e0 = *reinterpret_cast<const float4*>(l_edge[wave][sub][k]);
            const float2 e1 = *reinterpret_cast<const float2*>(l_edge[wave][sub][k] + 4);
            const float crx = e0.x, cry = e0.y, crz = e0.z, dx = e0.w, dy = e1.x, dz = e1.y;
            // (two fused multiply-adds per component -- the centre term first -- instead of multiply, fma, add: this kernel runs at 99.9 % of the VALU
            //  issue rate, every instruction is time)
            float y0 = __builtin_fmaf(a0, crx, __builtin_fmaf(a1, dx, yc0)), y1 = __builtin_fmaf(a0, cry, __builtin_fmaf(a1, dy, yc1)),
                  y2 = __builtin_fmaf(a0, crz, __builtin_fmaf(a1, dz, yc2));
            const float k0 = __builtin_fmaf(d0, crx, __builtin_fmaf(d1, dx, kc0)), k1 = __builtin_fmaf(d0, cry, __builtin_fmaf(d1, dy, kc1)),
                        k2 = __builtin_fmaf(d0, crz, __builtin_fmaf(d1, dz, kc2));
            vn_act(y0, y1, y2, k0, k1, k2, oms);
            s0 += y0; s1 += y1; s2 += y2;
        }
        if (live && on) {
            float* op = out + (size_t)pid * 3 * Co + oc;
            op[0] = s0 * (1.0f / EK); op[Co] = s1 * (1.0f / EK); op[2 * Co] = s2 * (1.0f / EK);
        }
    }
}

// ---------------------------------------------------------------------------------------------- pool layers (i >= 1)
// P table: Tp[b, src point, xyz, ldp] (columns PV_lin | PV_dir [| PK_lin | PK_dir]);
// Q table: Tq[b, q point, xyz, ldq]   (columns QV_lin | QV_dir [| QK_lin | QK_dir | Qq_lin | Qq_dir]) with NQ rows per
// instance, indexed by dst_rows[pid] when q_via_rows (one combined table over the source points) else by the point id.
__global__ __launch_bounds__(256) void edge_pool_kernel(const float* __restrict__ T, int ldt, const float* __restrict__ Tq, int ldq,
                                                        int NQ, int q_via_rows, const int32_t* __restrict__ knn,
                                                        const int32_t* __restrict__ dst_rows, int Nd, int Ns, int Co,
                                                        float oms, float* __restrict__ out, int total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lpp = Co <= 32 ? 32 : 64;         // lanes per point
    const int ppw = 64 / lpp;                   // points per wave
    const int sub = lane / lpp, ol = lane % lpp;
    int pid = (xcd_remap(blockIdx.x, gridDim.x) * 4 + wave) * ppw + sub;
    const bool live = pid < total;
    if (!live) pid = total - 1;
    const int b = pid / Nd;
    const int drow = (dst_rows && q_via_rows) ? dst_rows[pid] : (pid % Nd);
    const float* Tb = T + (size_t)b * Ns * 3 * ldt;
    const float* Td = Tq + ((size_t)b * NQ + drow) * 3 * ldq;
    const int32_t* ki = knn + (size_t)pid * EK;
    for (int c0 = 0; c0 < Co; c0 += lpp) {
        const int oc = c0 + ol;
        const bool on = oc < Co;
        const int ow = on ? oc : 0;
        float ql[3], qd[3];
#pragma unroll
        for (int x = 0; x < 3; ++x) { ql[x] = Td[x * ldq + ow]; qd[x] = Td[x * ldq + Co + ow]; }
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 4
        for (int k = 0; k < EK; ++k) {
            const float* Tr = Tb + (size_t)ki[k] * 3 * ldt;
            float y0 = Tr[ow] + ql[0], y1 = Tr[ldt + ow] + ql[1], y2 = Tr[2 * ldt + ow] + ql[2];
            const float k0 = Tr[Co + ow] + qd[0], k1 = Tr[ldt + Co + ow] + qd[1], k2 = Tr[2 * ldt + Co + ow] + qd[2];
            vn_act(y0, y1, y2, k0, k1, k2, oms);
            s0 += y0; s1 += y1; s2 += y2;
        }
        if (live && on) {
            float* op = out + (size_t)pid * 3 * Co + oc;
            op[0] = s0 * (1.0f / EK); op[Co] = s1 * (1.0f / EK); op[2 * Co] = s2 * (1.0f / EK);
        }
    }
}

// ---------------------------------------------------------------------------------------------- attention layers
// dynamic LDS per wave: q feature [3][Co] | head scores [Co/16][16] | attention weights [Co/16][16]
__global__ __launch_bounds__(256) void edge_attn_kernel(const float* __restrict__ T, int ldt, const float* __restrict__ Tq, int ldq,
                                                        int NQ, int q_via_rows, const int32_t* __restrict__ knn,
                                                        const int32_t* __restrict__ dst_rows, int Nd, int Ns, int Co,
                                                        float oms, float inv_sqrt_dk, float* __restrict__ out, int total) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nh = Co / 16;
    const int per_wave = 3 * Co + 2 * nh * EK;
    float* lq = smem + (size_t)wave * per_wave;
    float* lscore = lq + 3 * Co;
    float* latt = lscore + nh * EK;
    int pid = xcd_remap(blockIdx.x, gridDim.x) * 4 + wave;  // consecutive points (one instance) share an XCD L2
    const bool live = pid < total;
    if (!live) pid = total - 1;
    const int b = pid / Nd;
    const int drow = (dst_rows && q_via_rows) ? dst_rows[pid] : (pid % Nd);
    const float* Tb = T + (size_t)b * Ns * 3 * ldt;
    const float* Td = Tq + ((size_t)b * NQ + drow) * 3 * ldq;
    const int32_t* ki = knn + (size_t)pid * EK;
    int nbr[EK];
#pragma unroll
    for (int k = 0; k < EK; ++k) nbr[k] = ki[k];

    // ---- A: q = cevn(VecLNA_Q(dst_f[n]))  (vec_dgcnn_atten.py:207,210)
    float ssq = 0.f;
    for (int c0 = 0; c0 < Co; c0 += 64) {
        const int oc = c0 + lane;
        if (oc < Co) {
            float y0 = Td[4 * Co + oc], y1 = Td[ldq + 4 * Co + oc], y2 = Td[2 * ldq + 4 * Co + oc];
            const float k0 = Td[5 * Co + oc], k1 = Td[ldq + 5 * Co + oc], k2 = Td[2 * ldq + 5 * Co + oc];
            vn_act(y0, y1, y2, k0, k1, k2, oms);
            lq[oc] = y0; lq[Co + oc] = y1; lq[2 * Co + oc] = y2;
            ssq += y0 * y0 + y1 * y1 + y2 * y2;
        }
    }
    const float inv_q = 1.0f / fmaxf(sqrtf(wave_sum(ssq)), 1e-12f);
    __syncthreads();

    // ---- B: k = cevn(VecLNA_K(E)),  head scores  sum_{c in head} <k_c, q_c>  (:206,209,211-215)
    float ssk[EK];
#pragma unroll
    for (int k = 0; k < EK; ++k) ssk[k] = 0.f;
    for (int c0 = 0; c0 < Co; c0 += 64) {
        const int oc = c0 + lane;
        const bool on = oc < Co;
        const int ow = on ? oc : 0;
        float ql[3], qd[3], qv[3];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            ql[x] = Td[x * ldq + 2 * Co + ow];
            qd[x] = Td[x * ldq + 3 * Co + ow];
            qv[x] = lq[x * Co + ow];
        }
#pragma unroll
        for (int k = 0; k < EK; ++k) {
            const float* Tr = Tb + (size_t)nbr[k] * 3 * ldt;
            float y0 = Tr[2 * Co + ow] + ql[0], y1 = Tr[ldt + 2 * Co + ow] + ql[1], y2 = Tr[2 * ldt + 2 * Co + ow] + ql[2];
            const float k0 = Tr[3 * Co + ow] + qd[0], k1 = Tr[ldt + 3 * Co + ow] + qd[1], k2 = Tr[2 * ldt + 3 * Co + ow] + qd[2];
            vn_act(y0, y1, y2, k0, k1, k2, oms);
            float a = y0 * qv[0] + y1 * qv[1] + y2 * qv[2];
            float s2 = y0 * y0 + y1 * y1 + y2 * y2;
            if (!on) { a = 0.f; s2 = 0.f; }
            ssk[k] += s2;
            const float hs = row16_sum(a);
            if (on && (lane & 15) == 0) lscore[(oc >> 4) * EK + k] = hs;
        }
    }
    // Frobenius norm of the K feature at every neighbour; lane l keeps the one for k = l & 15
    float my_invk = 0.f;
#pragma unroll
    for (int k = 0; k < EK; ++k) {
        const float f = 1.0f / fmaxf(sqrtf(wave_sum(ssk[k])), 1e-12f);
        if ((lane & 15) == k) my_invk = f;
    }
    __syncthreads();
    // soft-max over the K neighbours per head (:214-215)
    for (int e0 = 0; e0 < nh * EK; e0 += 64) {
        const int e = e0 + lane;
        const bool on = e < nh * EK;
        float v = on ? lscore[e] * inv_q * my_invk * inv_sqrt_dk : -INFINITY;
        const float m = row16_max(v);
        const float ex = on ? expf(v - m) : 0.f;
        const float s = row16_sum(ex);
        if (on) latt[e] = ex / s;
    }
    __syncthreads();

    // ---- C: out = sum_k atten * VecLNA_V(E)  (:208,216-219)
    for (int c0 = 0; c0 < Co; c0 += 64) {
        const int oc = c0 + lane;
        const bool on = oc < Co;
        const int ow = on ? oc : 0;
        float ql[3], qd[3];
#pragma unroll
        for (int x = 0; x < 3; ++x) { ql[x] = Td[x * ldq + ow]; qd[x] = Td[x * ldq + Co + ow]; }
        const float* aw = latt + (ow >> 4) * EK;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < EK; ++k) {
            const float* Tr = Tb + (size_t)nbr[k] * 3 * ldt;
            float y0 = Tr[ow] + ql[0], y1 = Tr[ldt + ow] + ql[1], y2 = Tr[2 * ldt + ow] + ql[2];
            const float k0 = Tr[Co + ow] + qd[0], k1 = Tr[ldt + Co + ow] + qd[1], k2 = Tr[2 * ldt + Co + ow] + qd[2];
            vn_act(y0, y1, y2, k0, k1, k2, oms);
            const float w = aw[k];
            s0 += w * y0; s1 += w * y1; s2 += w * y2;
        }
        if (live && on) {
            float* op = out + (size_t)pid * 3 * Co + oc;
            op[0] = s0; op[Co] = s1; op[2 * Co] = s2;
        }
    }
}

// ---------------------------------------------------------------------------------------------- attention layers, float4 lanes
// Same math as edge_attn_kernel, re-mapped for memory efficiency: a lane owns FOUR consecutive channels (16-byte
// gathers: 4x fewer load instructions, 256 B..1 KB contiguous per neighbour row), a point is LPP = Co/(4*NCH) lanes
// (Co = 64/128/256/512 -> 16/32/64/64 lanes, 4/2/1/1 points per wave), an attention head (16 channels) is one DPP
// quad, and the per-head scores for all 16 neighbours stay in the quad's registers, so the soft-max over neighbours
// needs no cross-lane traffic and no LDS at all.
// LS_DPP_NOP=n (dev builds only, scripts/diag/pk_hazard_repro.sh): n + 1 wait states between the instruction that produces a DPP operand and the DPP
// instruction that reads it from other lanes -- the s_nop sweep of the reproducibility defect described at edge_attn_v4_kernel
#ifdef LS_DPP_NOP
#define LS_DPP_STR2(x) #x
#define LS_DPP_STR(x) LS_DPP_STR2(x)
#define LS_DPP_FENCE(v) asm volatile("s_nop " LS_DPP_STR(LS_DPP_NOP) : "+v"(v))
#else
#define LS_DPP_FENCE(v)
#endif
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    LS_DPP_FENCE(v);
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_sum(float v) { return dpp_add<0x4E>(dpp_add<0xB1>(v)); }
template <int LPP>
__device__ __forceinline__ float group_sum(float v) {  // all-reduce over aligned groups of LPP lanes
    v = quad_sum(v);
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x140>(v);  // row_mirror -> 16-lane sum in every lane
    if constexpr (LPP >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (LPP >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

template <int CTRL>
__device__ __forceinline__ float dpp_maxf(float v) {
    LS_DPP_FENCE(v);
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false)));
}
template <int LPP>
__device__ __forceinline__ float group_max(float v) {  // all-reduce (max) over aligned groups of LPP lanes
    v = dpp_maxf<0x140>(dpp_maxf<0x141>(dpp_maxf<0x4E>(dpp_maxf<0xB1>(v))));
    if constexpr (LPP >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
    if constexpr (LPP >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ float amax_f4(const float4& v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }

// 1 / max(sqrt(ss), 1e-12) (channel_equi_vec_normalize's Frobenius norm, vec_layers.py:24-31) as ONE v_rsq_f32 on the clamped square
// instead of a correctly rounded sqrt + IEEE division (~18 issue slots per neighbour in the K branch); 1 ulp, far inside the tolerance
__device__ __forceinline__ float inv_fro(float ss) { return __builtin_amdgcn_rsqf(fmaxf(ss, 1e-24f)); }

struct F43 { float4 x, y, z; };  // one xyz triple for four channels
__device__ __forceinline__ F43 ld43(const float* p, int ldt) {
    F43 r;
    r.x = *reinterpret_cast<const float4*>(p);
    r.y = *reinterpret_cast<const float4*>(p + ldt);
    r.z = *reinterpret_cast<const float4*>(p + 2 * ldt);
    return r;
}
__device__ __forceinline__ F43 add43(const F43& a, const F43& b) {
    F43 r;
    r.x = make_float4(a.x.x + b.x.x, a.x.y + b.x.y, a.x.z + b.x.z, a.x.w + b.x.w);
    r.y = make_float4(a.y.x + b.y.x, a.y.y + b.y.y, a.y.z + b.y.z, a.y.w + b.y.w);
    r.z = make_float4(a.z.x + b.z.x, a.z.y + b.z.y, a.z.z + b.z.z, a.z.w + b.z.w);
    return r;
}
// VN activation on four channels in place (y := act(y, k))
__device__ __forceinline__ void act43(F43& y, const F43& k, float oms) {
    vn_act(y.x.x, y.y.x, y.z.x, k.x.x, k.y.x, k.z.x, oms);
    vn_act(y.x.y, y.y.y, y.z.y, k.x.y, k.y.y, k.z.y, oms);
    vn_act(y.x.z, y.y.z, y.z.z, k.x.z, k.y.z, k.z.z, oms);
    vn_act(y.x.w, y.y.w, y.z.w, k.x.w, k.y.w, k.z.w, oms);
}
// (explicit fma chain, in this order: see vn_act in ls_common.h)
__device__ __forceinline__ float dot43(const F43& a, const F43& b) {
    float s = a.x.x * b.x.x;
    s = __builtin_fmaf(a.y.x, b.y.x, s); s = __builtin_fmaf(a.z.x, b.z.x, s);
    s = __builtin_fmaf(a.x.y, b.x.y, s); s = __builtin_fmaf(a.y.y, b.y.y, s); s = __builtin_fmaf(a.z.y, b.z.y, s);
    s = __builtin_fmaf(a.x.z, b.x.z, s); s = __builtin_fmaf(a.y.z, b.y.z, s); s = __builtin_fmaf(a.z.z, b.z.z, s);
    s = __builtin_fmaf(a.x.w, b.x.w, s); s = __builtin_fmaf(a.y.w, b.y.w, s); s = __builtin_fmaf(a.z.w, b.z.w, s);
    return s;
}
// acc += w * y on an xyz triple of four channels
__device__ __forceinline__ void fma43(F43& acc, float w, const F43& y) {
    acc.x.x = __builtin_fmaf(w, y.x.x, acc.x.x); acc.x.y = __builtin_fmaf(w, y.x.y, acc.x.y); acc.x.z = __builtin_fmaf(w, y.x.z, acc.x.z); acc.x.w = __builtin_fmaf(w, y.x.w, acc.x.w);
    acc.y.x = __builtin_fmaf(w, y.y.x, acc.y.x); acc.y.y = __builtin_fmaf(w, y.y.y, acc.y.y); acc.y.z = __builtin_fmaf(w, y.y.z, acc.y.z); acc.y.w = __builtin_fmaf(w, y.y.w, acc.y.w);
    acc.z.x = __builtin_fmaf(w, y.z.x, acc.z.x); acc.z.y = __builtin_fmaf(w, y.z.y, acc.z.y); acc.z.z = __builtin_fmaf(w, y.z.z, acc.z.z); acc.z.w = __builtin_fmaf(w, y.z.w, acc.z.w);
}

// ---------------------------------------------------------------------------------------------- pool layers, float4 lanes (round 4)
// edge_pool_kernel with a lane owning FOUR consecutive channels (16-byte gathers), a point = LPP = Co / 4 lanes, 64 / LPP points per wave: the
// scalar kernel issues one 4-byte load per lane, channel group and row -- 256 bytes per wave instruction, 3.1 M load instructions per layer-1
// launch (805 MB), the vector-memory issue rate its bound; this form issues a quarter of them.  Same additions in the same order per channel
// (neighbours k = 0 .. 15 ascending, then x 1/16): bit-identical to edge_pool_kernel.
template <int LPP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void edge_pool_v4_kernel(const float* __restrict__ T, int ldt, const float* __restrict__ Tq, int ldq, int NQ,
                                                           int q_via_rows, const int32_t* __restrict__ knn, const int32_t* __restrict__ dst_rows,
                                                           int Nd, int Ns, float oms, float* __restrict__ out, int total) {
    constexpr int PPW = 64 / LPP, Co = 4 * LPP;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / LPP, ll = lane % LPP;
    int pid = (xcd_remap(blockIdx.x, gridDim.x) * 4 + wave) * PPW + sub;
    const bool live = pid < total;
    if (!live) pid = total - 1;
    const int b = pid / Nd;
    const int drow = (dst_rows && q_via_rows) ? dst_rows[pid] : (pid % Nd);
    const float* Td = Tq + ((size_t)b * NQ + drow) * 3 * ldq;
    const int c4 = ll * 4;
    int nb[EK];
    {
        const int4* kp = reinterpret_cast<const int4*>(knn + (size_t)pid * EK);
#pragma unroll
        for (int u = 0; u < EK / 4; ++u) { const int4 v = kp[u]; nb[4 * u] = v.x; nb[4 * u + 1] = v.y; nb[4 * u + 2] = v.z; nb[4 * u + 3] = v.w; }
    }
    const F43 ql = ld43(Td + c4, ldq), qd = ld43(Td + Co + c4, ldq);
    F43 acc;
    acc.x = acc.y = acc.z = make_float4(0.f, 0.f, 0.f, 0.f);
    // (row gathers from one scalar base + a 32-bit byte offset, as in edge_attn_fq_kernel; the launch checks the table size)
    const unsigned row_bytes = 3u * (unsigned)ldt * 4u, lane_off = (unsigned)c4 * 4u, inst_row = (unsigned)b * (unsigned)Ns;
    auto ldrow = [&](unsigned off, int col) {
        asm volatile("" : "+v"(off));
        F43 r;
        r.x = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T + col) + off);
        r.y = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T + ldt + col) + off);
        r.z = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T + 2 * ldt + col) + off);
        return r;
    };
    // two-deep gather pipeline, as edge_attn_fq_kernel (round 4): the rolled loop loaded a neighbour's three y rows, waited, loaded its three
    // direction rows, waited, computed -- two dependent L2 round trips per neighbour with only the other waves of the SIMD to hide them
    // (VALU issue 0.43, waves parked 62 % of their life)
    constexpr int DP = 2;
    F43 py[DP], pd[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) { const unsigned o = __umul24(inst_row + (unsigned)nb[d], row_bytes) + lane_off; py[d] = ldrow(o, 0); pd[d] = ldrow(o, Co); }
    // (What keeps the order: the loads come from read-only no-alias memory and the loop has no store, so neither a sched_barrier nor a memory clobber
    //  ties them to the arithmetic -- left free, hipcc issued all sixteen neighbours' loads first: 401 registers.  The empty asm below makes the
    //  offset of neighbour k + DP depend on the accumulator as neighbour k - 1 left it, and neighbour k's arithmetic on that asm.)
#pragma unroll
    for (int k = 0; k < EK; ++k) {
        F43 y = py[k % DP], kd = pd[k % DP];
        if (k + DP < EK) {
            unsigned o = __umul24(inst_row + (unsigned)nb[k + DP], row_bytes) + lane_off;
            py[k % DP] = ldrow(o, 0); pd[k % DP] = ldrow(o, Co);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        y = add43(y, ql);
        kd = add43(kd, qd);
        act43(y, kd, oms);
        acc = add43(acc, y);
        asm volatile("" : "+v"(acc.x.x), "+v"(acc.x.y), "+v"(acc.x.z), "+v"(acc.x.w), "+v"(acc.y.x), "+v"(acc.y.y), "+v"(acc.y.z), "+v"(acc.y.w), "+v"(acc.z.x), "+v"(acc.z.y), "+v"(acc.z.z), "+v"(acc.z.w) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    if (live) {
        float* op = out + (size_t)pid * 3 * Co + c4;
        const float s = 1.0f / EK;
        *reinterpret_cast<float4*>(op) = make_float4(acc.x.x * s, acc.x.y * s, acc.x.z * s, acc.x.w * s);
        *reinterpret_cast<float4*>(op + Co) = make_float4(acc.y.x * s, acc.y.y * s, acc.y.z * s, acc.y.w * s);
        *reinterpret_cast<float4*>(op + 2 * Co) = make_float4(acc.z.x * s, acc.z.y * s, acc.z.z * s, acc.z.w * s);
    }
}

// Partial column sums of an attention kernel's output over its workgroup's PW = 4 (64 / LPP) points (thread tid = (wave, point of the wave, LPP lanes x 4
// channels) holds ox / oy / oz), ascending point order -> cp [3][Co].  The score slots l_score[k][tid] are lane-private and the caller is done with its
// own: no barrier before the writes.  Shared by edge_attn_v4_kernel and edge_attn_fq_kernel so that the two paths hand the residual global conv the
// same mean bit for bit (glob_mean_gemv_kernel finishes the sum over the workgroups).
template <int LPP>
__device__ __forceinline__ void attn_colsum(float (*l_score)[256], const float4& ox, const float4& oy, const float4& oz, bool live, int tid, float* __restrict__ cp) {
    constexpr int PPW = 64 / LPP, PW = 4 * PPW, Co = 4 * LPP;
    const float z = live ? 1.f : 0.f;   // (a lane past the end recomputed the last point: it must not be counted twice)
    l_score[0][tid] = ox.x * z; l_score[1][tid] = ox.y * z; l_score[2][tid] = ox.z * z; l_score[3][tid] = ox.w * z;
    l_score[4][tid] = oy.x * z; l_score[5][tid] = oy.y * z; l_score[6][tid] = oy.z * z; l_score[7][tid] = oy.w * z;
    l_score[8][tid] = oz.x * z; l_score[9][tid] = oz.y * z; l_score[10][tid] = oz.z * z; l_score[11][tid] = oz.w * z;
    __syncthreads();
    for (int col = tid; col < 3 * Co; col += 256) {      // column (axis, channel) <- the PW lanes that hold it
        const int ax = col / Co, c = col - ax * Co;
        const float* sp = &l_score[ax * 4 + (c & 3)][c >> 2];
        float a = 0.f;
#pragma unroll
        for (int pw = 0; pw < PW; ++pw) a += sp[(pw / PPW) * 64 + (pw % PPW) * LPP];
        cp[col] = a;
    }
}

template <int LPP, int NCH>
__global__ __launch_bounds__(256) void edge_attn_v4_kernel(const float* __restrict__ T, int ldt, const float* __restrict__ Tq, int ldq,
                                                           int NQ, int q_via_rows, const int32_t* __restrict__ knn,
                                                           const int32_t* __restrict__ dst_rows, int Nd, int Ns, int Co,
                                                           float oms, float inv_sqrt_dk, float* __restrict__ out, int total,
                                                           float* __restrict__ rowmax, float* __restrict__ colsum) {
    // colsum (nullable, NCH == 1 only) [total / PW][3][Co]: partial column sums of `out`, see attn_colsum
    // rowmax (nullable) [total * 3]: max|out[row, :]| -- the operand range of the GEMM that reads `out` (gemm.hip, GemmAux)
    constexpr int PPW = 64 / LPP;
    // lane-private LDS slots ([neighbour][thread]: conflict-free without padding -> 32 KB per chunk pair, five workgroups per CU):
    // head scores per chunk and |k|^2 per neighbour.  Keeping these arrays out of VGPRs lets the neighbour loops stay rolled
    // (bounded registers, 2 neighbours of loads in flight).
    // REPRODUCIBILITY (round 2): this file is built with -fno-slp-vectorize (build.py).  With hipcc's SLP-formed packed fp32 math
    // (v_pk_mul_f32 / v_pk_fma_f32, 172 of them in this kernel) the outputs of a few points per launch moved by 1e-6..1e-5 whenever
    // the kernel's waves shared CUs with the bf16-MFMA GEMM of other streams (always in the last 16 lanes of a wave; 4..19 of 48
    // launches; never when run alone) -- found by tests/test_hip_fullbatch.py, isolated by scripts/diag/edge_determinism.py by
    // elimination: no LDS (arrays in registers / a one-pass online soft-max), no DPP, no IEEE division, wait states after every
    // transcendental or LDS store each still failed; the same source without packed ops never did (0 of 96) and needs fewer
    // registers (this kernel 129 -> 104 VGPRs).  The one-pass online-soft-max form measured slower (0.57 vs 0.44 ms per step).
    __shared__ float l_score[NCH][EK][256];
    __shared__ float l_ssk[EK][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane / LPP, ll = lane % LPP;
    int pid = (xcd_remap(blockIdx.x, gridDim.x) * 4 + wave) * PPW + sub;
    const bool live = pid < total;
    if (!live) pid = total - 1;
    const int b = pid / Nd;
    const int drow = (dst_rows && q_via_rows) ? dst_rows[pid] : (pid % Nd);
    const float* Tb = T + (size_t)b * Ns * 3 * ldt;
    const float* Td = Tq + ((size_t)b * NQ + drow) * 3 * ldq;
    const int32_t* ki = knn + (size_t)pid * EK;

    // ---- A: q = cevn(VecLNA_Q(dst_f[n]))
    F43 qf[NCH];
    float ssq = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c4 = (ch * LPP + ll) * 4;
        qf[ch] = ld43(Td + 4 * Co + c4, ldq);
        const F43 kd = ld43(Td + 5 * Co + c4, ldq);
        act43(qf[ch], kd, oms);
        ssq += dot43(qf[ch], qf[ch]);
    }
    const float inv_q = inv_fro(group_sum<LPP>(ssq));

    // ---- B: K branch -> per-head scores for the 16 neighbours, Frobenius norms of k
    // (rolled loops, two neighbours of loads per iteration: the software-pipelined form of edge_attn_fq_kernel was measured here too --
    //  layers 5 / 6 have 8 192 / 2 048 destination points, a handful of waves per CU whose serial chain is the time either way, and the
    //  48 prefetch registers cost a workgroup per CU: 48.6 / 33.8 -> 51.4 / 34.7 us.  Not kept.)
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c4 = (ch * LPP + ll) * 4;
        const F43 ql = ld43(Td + 2 * Co + c4, ldq), qd = ld43(Td + 3 * Co + c4, ldq);
#pragma unroll 2
        for (int k = 0; k < EK; ++k) {
            const float* Tr = Tb + (size_t)ki[k] * 3 * ldt;
            F43 y = add43(ld43(Tr + 2 * Co + c4, ldt), ql);
            const F43 kd = add43(ld43(Tr + 3 * Co + c4, ldt), qd);
            act43(y, kd, oms);
            const float s2 = dot43(y, y);
            l_ssk[k][tid] = (ch == 0) ? s2 : l_ssk[k][tid] + s2;
            l_score[ch][k][tid] = quad_sum(dot43(y, qf[ch]));
        }
    }
    float mx[NCH], sum[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) { mx[ch] = -INFINITY; sum[ch] = 0.f; }
#pragma unroll 4
    for (int k = 0; k < EK; ++k) {
        const float invk = inv_fro(group_sum<LPP>(l_ssk[k][tid]));
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const float v = l_score[ch][k][tid] * inv_q * invk * inv_sqrt_dk;
            l_score[ch][k][tid] = v;
            mx[ch] = fmaxf(mx[ch], v);
        }
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {  // soft-max numerators (the 1/sum is applied to the weighted sum below)
#pragma unroll 4
        for (int k = 0; k < EK; ++k) {
            const float ex = expf(l_score[ch][k][tid] - mx[ch]);
            l_score[ch][k][tid] = ex;
            sum[ch] += ex;
        }
    }

    // ---- C: V branch, weighted sum
    float rmx = 0.f, rmy = 0.f, rmz = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c4 = (ch * LPP + ll) * 4;
        const F43 ql = ld43(Td + c4, ldq), qd = ld43(Td + Co + c4, ldq);
        F43 acc;
        acc.x = acc.y = acc.z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 2
        for (int k = 0; k < EK; ++k) {
            const float* Tr = Tb + (size_t)ki[k] * 3 * ldt;
            F43 y = add43(ld43(Tr + c4, ldt), ql);
            const F43 kd = add43(ld43(Tr + Co + c4, ldt), qd);
            act43(y, kd, oms);
            const float w = l_score[ch][k][tid];
            fma43(acc, w, y);
        }
        const float inv = 1.0f / sum[ch];
        const float4 ox = make_float4(acc.x.x * inv, acc.x.y * inv, acc.x.z * inv, acc.x.w * inv);
        const float4 oy = make_float4(acc.y.x * inv, acc.y.y * inv, acc.y.z * inv, acc.y.w * inv);
        const float4 oz = make_float4(acc.z.x * inv, acc.z.y * inv, acc.z.z * inv, acc.z.w * inv);
        if (live) {
            float* op = out + (size_t)pid * 3 * Co + c4;
            *reinterpret_cast<float4*>(op) = ox;
            *reinterpret_cast<float4*>(op + Co) = oy;
            *reinterpret_cast<float4*>(op + 2 * Co) = oz;
        }
        rmx = fmaxf(rmx, amax_f4(ox)); rmy = fmaxf(rmy, amax_f4(oy)); rmz = fmaxf(rmz, amax_f4(oz));
        if constexpr (NCH == 1) {
            if (colsum) attn_colsum<LPP>(l_score[0], ox, oy, oz, live, tid, colsum + (size_t)xcd_remap(blockIdx.x, gridDim.x) * 3 * Co);   // kernel-uniform
        }
    }
    if (rowmax) {   // wave-uniform
        rmx = group_max<LPP>(rmx); rmy = group_max<LPP>(rmy); rmz = group_max<LPP>(rmz);
        if (live && ll == 0) { float* rp = rowmax + (size_t)pid * 3; rp[0] = rmx; rp[1] = rmy; rp[2] = rmz; }
    }
}

// ---------------------------------------------------------------------------------------------- attention layers, destination side fused
// edge_attn_v4_kernel with the SIX destination-side column groups (v_lin, v_dir, k_lin, k_dir of the destination point and q_lin,
// q_dir: 60 % of an attention layer's table) computed IN the kernel instead of being written by the table GEMM and read back once:
// -151 MB of writes and -151 MB of reads per pass at layers 2 and 3 (B = 64), -75 + 75 MB at layer 4.  A workgroup owns PW = 16 (Co = 64)
// or 8 (Co = 128) destination points = 48 / 24 feature rows [x, Cin]; before each of the three phases (q | k | v) it multiplies them
// by the 2 Co weight rows that phase needs -- a [64 | 32] x Cin x [128 | 256] product on the f16 matrix cores with the same two-piece
// split as gemm.hip (a = h + l / 1024, three MFMAs per product, main and cross terms in separate accumulators), 8 output tiles of
// 32 x 32, two per wave -- into an LDS slab the phase then reads where the v4 kernel read Tq.  Cost: 144 - 288 MFMAs per workgroup
// (4 - 8 us per launch chip-wide) and 33 KB of LDS (two workgroups per CU instead of five; measured neutral for the gather itself,
// DESIGN.md 9).  The neighbour-side tables (P) are unchanged.
typedef _Float16 eh8_t __attribute__((ext_vector_type(8)));
typedef _Float16 eh2_t __attribute__((ext_vector_type(2)));
typedef float ef2_t __attribute__((ext_vector_type(2)));
typedef float ef16_t __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void esplit_pair(ef2_t v, unsigned& h, unsigned& l) {
    const eh2_t hv = __builtin_convertvector(v, eh2_t);
    const eh2_t lv = __builtin_convertvector(v - __builtin_convertvector(hv, ef2_t), eh2_t);
    h = __builtin_bit_cast(unsigned, hv);
    l = __builtin_bit_cast(unsigned, lv);
}
// eight consecutive-k fp32 values -> the (hi, lo) MFMA operand fragments of this lane
__device__ __forceinline__ void esplit8(const float* p, eh8_t& h, eh8_t& l) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    uint4 hh, ll;
    esplit_pair(ef2_t{a.x, a.y}, hh.x, ll.x); esplit_pair(ef2_t{a.z, a.w}, hh.y, ll.y);
    esplit_pair(ef2_t{b.x, b.y}, hh.z, ll.z); esplit_pair(ef2_t{b.z, b.w}, hh.w, ll.w);
    h = __builtin_bit_cast(eh8_t, hh);
    l = __builtin_bit_cast(eh8_t, ll);
}

// Wq [rows = 6 Co][Cin] fp32 -> (hi, lo) f16 pieces in MFMA-fragment-major order: [rows / 32 tiles][Cin / 16 steps][hi, lo][64 lanes] x 16 B,
// lane = (row & 31) + 32 (k / 8 & 1), eight consecutive k: a wave fetches an operand fragment with ONE contiguous 1 KB load.  (Fetching
// the fragments straight from the row-major fp32 weights -- 32 rows x 32 bytes per load instruction -- made the fused kernel 2x
// slower than the un-fused one: every load instruction touched 32 cache lines.)  Built once per model (ls_model_create).
// Operand range (gemm.hip, "operand range of the f16 split"): every weight row is multiplied by its own power of two before the split;
// the inverse scales [rows] follow the planes.  The kernel scales its feature rows the same way and multiplies the accumulators by the
// product of the two inverses -- the same exact scaling the table GEMM applies, so the two paths stay bit-identical.
__device__ __forceinline__ void epow2_scale(float amax, float& s, float& inv) {
    unsigned be = (__float_as_uint(amax) >> 23) & 0xffu;
    be = be < 15u ? 15u : be;
    s = __uint_as_float((268u - be) << 23);
    inv = __uint_as_float((be - 14u) << 23);
}
template <int CTRL>
__device__ __forceinline__ float edpp_fmax(float v) {
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false)));
}
__global__ __launch_bounds__(64) void edge_presplit_wq_kernel(const float* __restrict__ W, int rows, int Cin, int KS, uint4* __restrict__ planes,
                                                               float* __restrict__ winv) {
    const int tile = blockIdx.x / KS, ks = blockIdx.x % KS, lane = threadIdx.x;
    const int n = tile * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5);
    const float* wr = W + (size_t)min(n, rows - 1) * Cin;
    float am = 0.f;
    for (int c = 0; c < Cin; ++c) am = fmaxf(am, fabsf(wr[c]));
    float sc, inv;
    epow2_scale(am, sc, inv);
    if (ks == 0 && lane < 32 && n < rows) winv[n] = inv;
    __attribute__((aligned(16))) float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = wr[k + c] * sc;
    eh8_t h, l;
    esplit8(v, h, l);
    planes[((size_t)blockIdx.x * 2) * 64 + lane] = __builtin_bit_cast(uint4, h);
    planes[((size_t)blockIdx.x * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, l);
}
static size_t edge_wq_plane_count(int Co, int Cin) { return (size_t)(6 * Co / 32) * (Cin / 16) * 2 * 64; }
size_t edge_wq_planes_bytes(int Co, int Cin) { return edge_wq_plane_count(Co, Cin) * sizeof(uint4) + (size_t)6 * Co * sizeof(float); }
int edge_presplit_wq_launch(const float* Wq, int Co, int Cin, void* planes, hipStream_t st) {
    hipLaunchKernelGGL(edge_presplit_wq_kernel, dim3((6 * Co / 32) * (Cin / 16)), dim3(64), 0, st, Wq, 6 * Co, Cin, Cin / 16, (uint4*)planes,
                       (float*)((uint4*)planes + edge_wq_plane_count(Co, Cin)));
    LS_LAUNCH_CHECK();
    return LS_OK;
}

// Timing probe (round 6; WRONG results, dev builds only): -DLS_FQ_HALF_GATHER drops the gathers of the two `dir` column groups (the
// neighbour's `lin` values stand in) -- the upper bound of what a half-width table could buy BEFORE paying for the per-edge C x C product that would have
// to recompute the directions (profiles/r6_final/attn_halfwidth_ab.txt).
#ifdef LS_FQ_HALF_GATHER
#define LS_FQ_DIR(off, col, lin) (lin)
#else
#define LS_FQ_DIR(off, col, lin) ldrow(off, col)
#endif
template <int LPP, int CIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void edge_attn_fq_kernel(const float* __restrict__ T, int ldt, const float* __restrict__ cur,
                                                           const uint4* __restrict__ Wp, const int32_t* __restrict__ knn,
                                                           const int32_t* __restrict__ dst_rows, int Nd, int Ns, float oms, float inv_sqrt_dk,
                                                           float* __restrict__ out, int total, float* __restrict__ rowmax, float* __restrict__ colsum) {
    // colsum (nullable) [total / PW][3][Co]: the column sums of `out` over this workgroup's PW points, in ascending point order -- the residual global
    // conv's mean over the points (vec_dgcnn_atten.py:223) is finished from these partial sums by glob_mean_gemv_kernel (pointwise.hip) instead of a
    // second pass over `out` (the launch requires Nd % PW == 0: a workgroup's points then belong to one instance)
    constexpr int PPW = 64 / LPP, PW = 4 * PPW, ROWS = 3 * PW, MT = (ROWS + 31) / 32, Co = LPP * 4, SC = 2 * Co, NT = SC / 32, SLD = SC + 4,
                  KS = CIN / 16, ASTR = CIN * 2 + 16;   // A plane row stride in bytes (+16: conflict-free 16-byte fragment reads)
    static_assert(MT * NT == 8, "two output tiles per wave");
    __shared__ float l_score[EK][256];
    __shared__ __attribute__((aligned(16))) float slab[ROWS * SLD];
    __shared__ __attribute__((aligned(16))) char a_pl[2][MT * 32 * ASTR];   // the workgroup's feature rows as (hi, lo) f16 planes
    __shared__ int a_inv[MT * 32];                                          // inverse power-of-two scale of each staged row
    const float* __restrict__ winv = reinterpret_cast<const float*>(Wp + (size_t)(6 * Co / 32) * KS * 128);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane / LPP, ll = lane % LPP;
    const int pid0 = xcd_remap(blockIdx.x, gridDim.x) * PW;
    const int pwl = wave * PPW + sub;              // this lane's point inside the workgroup
    int pid = pid0 + pwl;
    const bool live = pid < total;
    if (!live) pid = total - 1;
    const int b = pid / Nd;
    const float* Tb = T + (size_t)b * Ns * 3 * ldt;
    const int32_t* ki = knn + (size_t)pid * EK;
    const int c4 = ll * 4;

    uint4 wfh[2][KS], wfl[2][KS];
    auto qload = [&](int cb) {
#ifdef LS_FQ_SKIP_QGEMM      // timing probe (WRONG results): the three destination-side products skipped
        return;
#endif
        const uint4* wt = Wp + (size_t)(cb / 32) * KS * 128 + lane;
#pragma unroll
        for (int u = 0; u < (MT == 2 ? 1 : 2); ++u) {      // (MT == 2: both M-tiles of a wave meet the same weight tile)
            const int nt = MT == 2 ? wave : wave + 4 * u;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                wfh[u][ks] = wt[((size_t)(nt * KS + ks) * 2) * 64];
                wfl[u][ks] = wt[((size_t)(nt * KS + ks) * 2 + 1) * 64];
            }
        }
    };
    qload(4 * Co);        // the q product's weight fragments travel while the rows are staged
    // ---- stage the destination points' feature rows (coalesced: a row is CIN * 4 contiguous bytes), split once for the three products
    for (int c = tid; c < MT * 32 * (CIN / 4); c += 256) {
        const int r = c / (CIN / 4), kq = c - r * (CIN / 4);
        uint2 h = make_uint2(0u, 0u), l = make_uint2(0u, 0u);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < ROWS) {
            const int pw = r / 3, x = r - 3 * pw;
            const int pa = min(pid0 + pw, total - 1), ba = pa / Nd;
            const int sp = dst_rows ? dst_rows[pa] : pa - ba * Nd;
            v = *reinterpret_cast<const float4*>(cur + (((size_t)ba * Ns + sp) * 3 + x) * CIN + kq * 4);
        }
        {   // the CIN / 4 = 8 | 16 threads of a row are an aligned lane group: row maximum by DPP, then the row's power of two
            float am = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
            am = edpp_fmax<0x141>(edpp_fmax<0x4E>(edpp_fmax<0xB1>(am)));
            if constexpr (CIN == 64) am = edpp_fmax<0x140>(am);
            float sc, inv;
            epow2_scale(am, sc, inv);
            if (kq == 0) a_inv[r] = (__float_as_int(inv) >> 23) - 127;    // exponent of the exact power of two (gemm.hip: pow2_e / scale_pow2)
            esplit_pair(ef2_t{v.x * sc, v.y * sc}, h.x, l.x);
            esplit_pair(ef2_t{v.z * sc, v.w * sc}, h.y, l.y);
        }
        *reinterpret_cast<uint2*>(&a_pl[0][r * ASTR + kq * 8]) = h;
        *reinterpret_cast<uint2*>(&a_pl[1][r * ASTR + kq * 8]) = l;
    }
    __syncthreads();

    // ---- destination-side product for one phase: slab[row][0 .. 2 Co) = x_rows . Wq[cb .. cb + 2 Co)^T
    // Round 6: the three products cost 15 / 27 / 44 us of a 99 / 108 / 82 us launch (probe: -DLS_FQ_SKIP_QGEMM, profiles/r6_final/attn_qgemm_ab.txt) although
    // they are 72 MFMAs per wave: hipcc issued the weight-fragment loads of a phase two at a time with an s_waitcnt behind each pair -- four to five
    // DEPENDENT L2 round trips per phase with every wave of the workgroup waiting.  Now a phase's fragments (2 tiles x KS steps x (hi, lo) x 16 bytes per lane:
    // 32 - 64 registers, at a point where nothing else is live) are requested in ONE batch (qload), the MFMAs run behind a scheduling barrier (qmma), and the
    // caller puts independent work between the two: the row staging in front of the q product, the q activation in front of the k product, the soft-max in
    // front of the v product.  Same operands, same MFMA order: bit-identical.
    auto qmma = [&](int cb) {
#ifdef LS_FQ_SKIP_QGEMM
        return;
#endif
        __builtin_amdgcn_sched_barrier(0);      // (the loads above stay above: left free, the scheduler sinks each pair to its first use again)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int mt = MT == 2 ? u : 0, nt = MT == 2 ? wave : wave + 4 * u;
            const int aoff = (32 * mt + (lane & 31)) * ASTR + (lane >> 5) * 16;
            ef16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const eh8_t ah = __builtin_bit_cast(eh8_t, *reinterpret_cast<const uint4*>(&a_pl[0][aoff + ks * 32]));
                const eh8_t al = __builtin_bit_cast(eh8_t, *reinterpret_cast<const uint4*>(&a_pl[1][aoff + ks * 32]));
                // (MT == 2: both M-tiles of a wave use the same weight tile -- its fragments are loaded once, u = 0)
                const eh8_t bh = __builtin_bit_cast(eh8_t, wfh[MT == 2 ? 0 : u][ks]);
                const eh8_t bl = __builtin_bit_cast(eh8_t, wfl[MT == 2 ? 0 : u][ks]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            }
            // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); rows past the workgroup's points: dropped
            const int row0 = 32 * mt + 4 * (lane >> 5);
            float* sp = slab + row0 * SLD + 32 * nt + (lane & 31);
            const int ce = (__float_as_int(winv[cb + 32 * nt + (lane & 31)]) >> 23) - 127;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                if (row0 + dr < ROWS) sp[dr * SLD] = __builtin_ldexpf(acc[r], a_inv[row0 + dr] + ce);   // acc * s_a^-1 * s_w^-1, exponents added as integers (as gemm.hip)
            }
        }
    };
    const float* srow = slab + 3 * pwl * SLD + c4;   // this lane's destination-side values: rows x, y, z of its point, its four channels
    auto lds43 = [&](int col) {
        F43 r;
        r.x = *reinterpret_cast<const float4*>(srow + col);
        r.y = *reinterpret_cast<const float4*>(srow + SLD + col);
        r.z = *reinterpret_cast<const float4*>(srow + 2 * SLD + col);
        return r;
    };

    // ---- A: q = cevn(VecLNA_Q(dst_f[n]))
    qmma(4 * Co);
    __syncthreads();
    qload(2 * Co);        // the k product's fragments travel under the q activation
    F43 qf = lds43(0);
    {
        const F43 kd = lds43(Co);
        act43(qf, kd, oms);
    }
    const float inv_q = inv_fro(group_sum<LPP>(dot43(qf, qf)));
    __syncthreads();   // every wave has its q: the slab may be overwritten

    // Where the time goes (round 4, timing variants, 12 steps in flight; layers 2 / 3 / 4 = 111 / 110 / 81 us): with the gathers of both loops
    // removed 56 / 64 / 38 us (arithmetic, the three destination-side products, LDS); with the arithmetic removed and the gathers kept 120 / 124 / 77 us.
    // The gathers alone take as long as the kernel: it is bound by the vector-memory path, not by the VALU -- and NOT by L2 traffic either (handing
    // the points out in Morton order cuts the distinct neighbour rows per workgroup from ~230 to 62 - 90 and changes nothing): 1.6 GB per launch have
    // to be DELIVERED by the L1s, 6.1 MB per CU at 64 B/clk = 46 us, and the two-deep pipeline of 16-byte loads does not overlap that with the arithmetic.
    // Three workgroups per CU for the <16, 64> instance (hipcc's "desired occupancy was 3, final occupancy is 2": 172 VGPRs and 60 416 bytes of LDS)
    // were BUILT in round 4 -- planes cut to the ROWS rows that exist, 128-byte rows XOR-swizzled instead of padded: 54 208 bytes, 168 VGPRs with 11
    // spilled -- and measured SLOWER (116 vs 110 us at layer 3, 84 vs 81 us at layer 4 where the same edit cost 10 more spills): the third workgroup
    // does not buy overlap that the spills do not take back.  Kept at two.
    // ---- B: K branch -> per-head scores for the 16 neighbours, normalised by the Frobenius norm of k
    // The gathers are a two-deep software pipeline (round 3): the 16 neighbour indices sit in registers (four 16-byte loads up front
    // instead of a dependent index load in front of every row gather) and the rows of neighbours k+1, k+2 are in flight while
    // neighbour k is activated.  The rolled loop it replaces ran index load -> wait -> twelve row loads -> wait -> compute, two full
    // L2 round trips per pair of neighbours with nothing of one iteration overlapping the next (s_waitcnt vmcnt(0) at the loop top).
    // (round 6: two 16-bit indices per register -- the sixteen indices live across both gather loops and the soft-max; eight registers less takes the
    //  <32, 64> instance (layer 4, three workgroups per CU at 168 VGPRs) from seven spilled registers to one: 84.8 -> 82.4 us; the launch checks Ns < 65 536.
    //  The same round measured the <16, 64> instance at THREE workgroups per CU -- A planes cut to the 48 real rows and XOR-swizzled, 54 272 bytes of LDS,
    //  168 VGPRs with 7 spilled: 109.4 against 111.7 us alone, bench 59.4k against 59.9k -- occupancy is not what bounds layer 3 either; not kept.)
    unsigned nb[EK / 2];
    {
        const int4* kp = reinterpret_cast<const int4*>(ki);
#pragma unroll
        for (int u = 0; u < EK / 4; ++u) { const int4 v = kp[u]; nb[2 * u] = (unsigned)v.x | ((unsigned)v.y << 16); nb[2 * u + 1] = (unsigned)v.z | ((unsigned)v.w << 16); }
    }
    // Row gathers as ONE uniform base + a 32-bit byte offset per neighbour (round 4): off = (instance row + neighbour) x row bytes + this lane's
    // column bytes by a single full-rate v_mad_u32_u24, the x / y / z rows and the column groups as scalar bases / immediates.  The 64-bit
    // pointer arithmetic it replaces cost eleven VALU instructions per neighbour and phase, three of them quarter-rate 32-bit multiplies
    // (the launch checks that the table is below 4 GB and its rows below 2^24).
    const unsigned row_bytes = 3u * (unsigned)ldt * 4u;
    const unsigned lane_off = (unsigned)c4 * 4u;
    const unsigned inst_row = (unsigned)b * (unsigned)Ns;
    auto noff = [&](int k) { return __umul24(inst_row + ((k & 1) ? nb[k >> 1] >> 16 : nb[k >> 1] & 0xFFFFu), row_bytes) + lane_off; };
    auto ldrow = [&](unsigned off, int col) {   // rows x, y, z of table columns col .. col + 3 (+ this lane's column offset, inside `off`)
        asm volatile("" : "+v"(off));           // (keeps the zero-extension next to the loads: the scalar-base addressing mode, as in gemm.hip)
        F43 r;
        r.x = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T + col) + off);
        r.y = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T + ldt + col) + off);
        r.z = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(T + 2 * ldt + col) + off);
        return r;
    };
    qmma(2 * Co);
    __syncthreads();
    {
        // (DP = 3 / 4 -- 195 / 219 VGPRs, still two workgroups per CU -- measured in round 4: layers 2 / 3 / 4 at 120 / 114 / 81 and 120 / 113 / 82 us
        //  against 110 / 111 / 84 us: a deeper prefetch does not deliver the rows faster, the vector-memory path is at its throughput.  Nor is it L2
        //  channel camping on the 1 KB row stride: table rows padded by 128 / 256 bytes changed nothing, 105 / 108 / 81 -> 106 / 108 / 82 us.  Nor does
        //  locality help: every point's neighbours taken in ascending index order (a 63-comparator sort in registers), with and without the Morton
        //  processing order: layers 2 / 3 / 4 119 / 99 / 80 and 109 / 97 / 79 us against 103 / 102 / 79 us.)
        constexpr int DP = 2;
        F43 py[DP], pd[DP];
#pragma unroll
        for (int d = 0; d < DP; ++d) { const unsigned o = noff(d); py[d] = ldrow(o, 2 * Co); pd[d] = LS_FQ_DIR(o, 3 * Co, py[d]); }
        const F43 ql = lds43(0), qd = lds43(Co);
#pragma unroll
        for (int k = 0; k < EK; ++k) {
            F43 y = py[k % DP], kd = pd[k % DP];
            if (k + DP < EK) { const unsigned o = noff(k + DP); py[k % DP] = ldrow(o, 2 * Co); pd[k % DP] = LS_FQ_DIR(o, 3 * Co, py[k % DP]); }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this neighbour's arithmetic (the scheduler would sink it to its first use)
            y = add43(y, ql);
            kd = add43(kd, qd);
            act43(y, kd, oms);
            const float invk = inv_fro(group_sum<LPP>(dot43(y, y)));
            l_score[k][tid] = quad_sum(dot43(y, qf)) * inv_q * invk * inv_sqrt_dk;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // (the row addresses are re-derived from the indices in phase C: sixteen 64-bit addresses kept alive across the soft-max cost
    //  more registers than the two instructions that rebuild each)
#pragma unroll
    for (int k = 0; k < EK / 2; ++k) asm volatile("" : "+v"(nb[k]));
    __syncthreads();   // done with the k slab
    qload(0);             // the v product's fragments travel under the soft-max
    float mx = -INFINITY, sum = 0.f;
#pragma unroll 4
    for (int k = 0; k < EK; ++k) mx = fmaxf(mx, l_score[k][tid]);
#pragma unroll 4
    for (int k = 0; k < EK; ++k) {  // soft-max numerators (the 1/sum is applied to the weighted sum below)
        const float ex = expf(l_score[k][tid] - mx);
        l_score[k][tid] = ex;
        sum += ex;
    }
    qmma(0);
    __syncthreads();

    // ---- C: V branch, weighted sum
    {
        const F43 ql = lds43(0), qd = lds43(Co);
        F43 acc;
        acc.x = acc.y = acc.z = make_float4(0.f, 0.f, 0.f, 0.f);
        constexpr int DP = 2;
        F43 py[DP], pd[DP];
#pragma unroll
        for (int d = 0; d < DP; ++d) { const unsigned o = noff(d); py[d] = ldrow(o, 0); pd[d] = LS_FQ_DIR(o, Co, py[d]); }
#pragma unroll
        for (int k = 0; k < EK; ++k) {
            F43 y = py[k % DP], kd = pd[k % DP];
            if (k + DP < EK) { const unsigned o = noff(k + DP); py[k % DP] = ldrow(o, 0); pd[k % DP] = LS_FQ_DIR(o, Co, py[k % DP]); }
            __builtin_amdgcn_sched_barrier(0);
            y = add43(y, ql);
            kd = add43(kd, qd);
            act43(y, kd, oms);
            const float w = l_score[k][tid];
            fma43(acc, w, y);
            __builtin_amdgcn_sched_barrier(0);
        }
        const float inv = 1.0f / sum;
        const float4 ox = make_float4(acc.x.x * inv, acc.x.y * inv, acc.x.z * inv, acc.x.w * inv);
        const float4 oy = make_float4(acc.y.x * inv, acc.y.y * inv, acc.y.z * inv, acc.y.w * inv);
        const float4 oz = make_float4(acc.z.x * inv, acc.z.y * inv, acc.z.z * inv, acc.z.w * inv);
        if (live) {
            float* op = out + (size_t)pid * 3 * Co + c4;
            *reinterpret_cast<float4*>(op) = ox;
            *reinterpret_cast<float4*>(op + Co) = oy;
            *reinterpret_cast<float4*>(op + 2 * Co) = oz;
        }
        if (rowmax) {   // wave-uniform: max|out[row, :]| for the GEMM that reads `out` (gemm.hip, GemmAux)
            const float rmx = group_max<LPP>(amax_f4(ox)), rmy = group_max<LPP>(amax_f4(oy)), rmz = group_max<LPP>(amax_f4(oz));
            if (live && ll == 0) { float* rp = rowmax + (size_t)pid * 3; rp[0] = rmx; rp[1] = rmy; rp[2] = rmz; }
        }
        if (colsum) attn_colsum<LPP>(l_score, ox, oy, oz, live, tid, colsum + (size_t)(pid0 / PW) * 3 * Co);   // kernel-uniform
    }
}

// can this layer shape take the fused kernel?
bool edge_attn_fq_supported(int Co, int Cin) { return (Co == 64 && (Cin == 32 || Cin == 64)) || (Co == 128 && Cin == 64); }
bool edge_attn_fq_fits(int B, int Ns, int ldt) {   // the table is addressed by 32-bit byte offsets formed with a 24-bit multiply
    return (unsigned long long)B * Ns * 3ull * ldt * 4ull < (1ull << 32) && (unsigned long long)B * Ns < (1ull << 24) && 3ull * ldt * 4ull < (1ull << 24) &&
           Ns < 65536;      // (+ neighbour indices packed two per register)
}
// points per workgroup of the fused kernel = rows of `colsum` per Nd / this many points (0: shape not served)
int edge_attn_fq_points_per_wg(int Co) { return Co == 64 ? 16 : Co == 128 ? 8 : 0; }
int edge_attn_fq_launch(const float* T, int ldt, const float* cur, int Cin, const void* wq_planes, const int32_t* knn, const int32_t* dst_rows, int B,
                        int Nd, int Ns, int Co, int head_c, float neg_slope, float* out, hipStream_t st, float* rowmax, float* colsum) {
    LS_REQUIRE(!colsum || (edge_attn_fq_points_per_wg(Co) > 0 && Nd % edge_attn_fq_points_per_wg(Co) == 0), "edge_attn_fq: column sums need Nd %% %d == 0 (Nd=%d)",
               edge_attn_fq_points_per_wg(Co), Nd);
    LS_REQUIRE(head_c == 16 && edge_attn_fq_supported(Co, Cin) && ldt % 4 == 0 && wq_planes, "edge_attn_fq: unsupported shape (Co=%d Cin=%d ldt=%d)", Co, Cin, ldt);
    LS_REQUIRE(edge_attn_fq_fits(B, Ns, ldt), "edge_attn_fq: the table is addressed by 32-bit byte offsets (B=%d Ns=%d ldt=%d)", B, Ns, ldt);
    const float isd = 1.0f / sqrtf(3.0f * head_c), oms = 1.0f - neg_slope;
    const int total = B * Nd;
#define LS_FQ(LPP, CIN) hipLaunchKernelGGL((edge_attn_fq_kernel<LPP, CIN>), dim3(cdiv(total, 4 * (64 / LPP))), dim3(256), 0, st, T, ldt, cur, (const uint4*)wq_planes, knn, dst_rows, Nd, Ns, oms, isd, out, total, rowmax, colsum)
    if (Co == 64 && Cin == 32) LS_FQ(16, 32);
    else if (Co == 64) LS_FQ(16, 64);
    else LS_FQ(32, 64);
#undef LS_FQ
    LS_LAUNCH_CHECK();
    return LS_OK;
}

template <int LPP, int NCH>
static int launch_attn_v4(const float* T, int ldt, const float* Tq, int ldq, int NQ, int qvr, const int32_t* knn,
                          const int32_t* dst_rows, int B, int Nd, int Ns, int Co, float neg_slope, float isd, float* out,
                          hipStream_t st, float* rowmax, float* colsum = nullptr) {
    const int total = B * Nd, ppb = 4 * (64 / LPP);
    hipLaunchKernelGGL((edge_attn_v4_kernel<LPP, NCH>), dim3(cdiv(total, ppb)), dim3(256), 0, st, T, ldt, Tq, ldq, NQ, qvr, knn,
                       dst_rows, Nd, Ns, Co, 1.0f - neg_slope, isd, out, total, rowmax, colsum);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

int edge_l0_launch(const float* pts, const int32_t* knn, const float* w0, int B, int N, int Co, float neg_slope, float* out,
                   hipStream_t st) {
    const int total = B * N;
    hipLaunchKernelGGL(edge_l0_kernel, dim3(cdiv(total, 8)), dim3(256), 0, st, pts, knn, w0, N, Co, 1.0f - neg_slope, out, total);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

int edge_pool_launch(const float* T, int ldt, const float* Tq, int ldq, int NQ, int qvr, const int32_t* knn,
                     const int32_t* dst_rows, int B, int Nd, int Ns, int Co, float neg_slope, float* out, hipStream_t st) {
    const int total = B * Nd;
    static const bool scalar_pool = dev_knob("LS_EDGE_POOL_SCALAR", 0) != 0;   // dev A/B: the one-channel-per-lane kernel (bit-identical)
    const bool off32 = edge_attn_fq_fits(B, Ns, ldt);
    if (!scalar_pool && off32 && ldt % 4 == 0 && ldq % 4 == 0 && (Co == 32 || Co == 64)) {   // (the float4 kernel addresses the table by 32-bit byte offsets)
        if (Co == 32)
            hipLaunchKernelGGL((edge_pool_v4_kernel<8>), dim3(cdiv(total, 32)), dim3(256), 0, st, T, ldt, Tq, ldq, NQ, qvr, knn, dst_rows, Nd, Ns, 1.0f - neg_slope, out, total);
        else
            hipLaunchKernelGGL((edge_pool_v4_kernel<16>), dim3(cdiv(total, 16)), dim3(256), 0, st, T, ldt, Tq, ldq, NQ, qvr, knn, dst_rows, Nd, Ns, 1.0f - neg_slope, out, total);
        LS_LAUNCH_CHECK();
        return LS_OK;
    }
    const int ppb = (Co <= 32) ? 8 : 4;
    hipLaunchKernelGGL(edge_pool_kernel, dim3(cdiv(total, ppb)), dim3(256), 0, st, T, ldt, Tq, ldq, NQ, qvr, knn, dst_rows, Nd, Ns,
                       Co, 1.0f - neg_slope, out, total);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

// does edge_attn_launch write the row maxima of its output for this shape? (the float4-lane kernels do, the generic one does not)
bool edge_attn_emits_rowmax(int Co, int ldt, int ldq) { return ldt % 4 == 0 && ldq % 4 == 0 && (Co == 64 || Co == 128 || Co == 256 || Co == 512); }
int edge_attn_launch(const float* T, int ldt, const float* Tq, int ldq, int NQ, int qvr, const int32_t* knn,
                     const int32_t* dst_rows, int B, int Nd, int Ns, int Co, int head_c, float neg_slope, float* out,
                     hipStream_t st, float* rowmax, float* colsum) {
    // colsum (nullable): partial column sums, one row per edge_attn_fq_points_per_wg(Co) points (Co = 64 / 128 with 16-byte-aligned tables only)
    LS_REQUIRE(!colsum || (ldt % 4 == 0 && ldq % 4 == 0 && edge_attn_fq_points_per_wg(Co) > 0 && Nd % edge_attn_fq_points_per_wg(Co) == 0),
               "edge_attn: no column sums for this shape (Co=%d Nd=%d)", Co, Nd);
    LS_REQUIRE(!rowmax || edge_attn_emits_rowmax(Co, ldt, ldq), "edge_attn: no row maxima from the generic kernel (Co=%d)", Co);
    LS_REQUIRE(head_c == 16 && Co % 16 == 0, "edge_attn: head width must be 16 and divide Co (head_c=%d Co=%d)", head_c, Co);
    const float isd = 1.0f / sqrtf(3.0f * head_c);
    if (ldt % 4 == 0 && ldq % 4 == 0) {
        if (Co == 64) return launch_attn_v4<16, 1>(T, ldt, Tq, ldq, NQ, qvr, knn, dst_rows, B, Nd, Ns, Co, neg_slope, isd, out, st, rowmax, colsum);
        if (Co == 128) return launch_attn_v4<32, 1>(T, ldt, Tq, ldq, NQ, qvr, knn, dst_rows, B, Nd, Ns, Co, neg_slope, isd, out, st, rowmax, colsum);
        if (Co == 256) return launch_attn_v4<64, 1>(T, ldt, Tq, ldq, NQ, qvr, knn, dst_rows, B, Nd, Ns, Co, neg_slope, isd, out, st, rowmax);
        if (Co == 512) return launch_attn_v4<64, 2>(T, ldt, Tq, ldq, NQ, qvr, knn, dst_rows, B, Nd, Ns, Co, neg_slope, isd, out, st, rowmax);
    }
    const int total = B * Nd;
    const size_t smem = (size_t)4 * (3 * Co + 2 * (Co / 16) * EK) * sizeof(float);
    LS_REQUIRE(smem <= 64 * 1024, "edge_attn: Co=%d too wide", Co);
    hipLaunchKernelGGL(edge_attn_kernel, dim3(cdiv(total, 4)), dim3(256), smem, st, T, ldt, Tq, ldq, NQ, qvr, knn, dst_rows, Nd, Ns,
                       Co, 1.0f - neg_slope, isd, out, total);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

}  // namespace ls
