// pk_fp32_beside_mfma.hip -- bounded attempt at a minimal reproducer of the round-2 reproducibility defect: a kernel
// full of packed fp32 math (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32, formed here ON PURPOSE with 2-vectors) gave different results
// from launch to launch while its waves shared CUs with the f16 / bf16 MFMA GEMM of other streams (edge_attn_v4_kernel, hipcc SLP output).
// Here: a "victim" gather + packed-math kernel (per-lane gathers of 16-byte rows, DPP quad reductions, LDS round trip: the ingredients
// of the attention kernel) launched REPS times on one stream while a second stream keeps the matrix pipes busy; every output is compared
// with the victim's solo result.
//   hipcc --offload-arch=gfx950 -O3 -o pk_fp32_beside_mfma pk_fp32_beside_mfma.hip && ./pk_fp32_beside_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void victim(const float* __restrict__ T, const int* __restrict__ idx, int rows, float* __restrict__ out) {
    __shared__ float sc[16][256];
    const int tid = threadIdx.x, p = blockIdx.x * 256 + tid;
    f2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
    for (int k = 0; k < 16; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(T + (size_t)idx[(size_t)p * 16 + k] * 64 + (tid & 15) * 4);
        const f2 x = {v.x, v.y}, y = {v.z, v.w};
        const f2 m = x * y;                         // v_pk_mul_f32
        a0 = x * m + a0;                            // v_pk_fma_f32
        a1 = y * m + a1;
        float s = a0.x + a1.y;
        s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
        sc[k][tid] = s;
    }
    float mx = -1e30f;
    for (int k = 0; k < 16; ++k) mx = fmaxf(mx, sc[k][tid]);
    f2 acc = {0.f, 0.f};
    for (int k = 0; k < 16; ++k) { const float e = __expf(sc[k][tid] - mx); acc = acc + f2{e, e * 0.5f} * (a0 + a1); }   // v_pk_add / v_pk_mul
    out[(size_t)p * 2] = acc.x; out[(size_t)p * 2 + 1] = acc.y;
}
__global__ __launch_bounds__(256) void mfma_load(float* sink, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    f16v c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    if (s == 12345.f) sink[threadIdx.x] = s;
}
int main() {
    const int rows = 1 << 16, P = 64 * 512, REPS = 400;
    std::vector<float> hT((size_t)rows * 64);
    std::vector<int> hI((size_t)P * 16);
    unsigned s = 1;
    for (auto& v : hT) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    for (auto& v : hI) { s = s * 1664525u + 1013904223u; v = (s >> 10) % rows; }
    float *T, *out, *ref, *sink; int* I;
    hipMalloc(&T, hT.size() * 4); hipMalloc(&I, hI.size() * 4); hipMalloc(&out, (size_t)P * 8); hipMalloc(&ref, (size_t)P * 8); hipMalloc(&sink, 4096);
    hipMemcpy(T, hT.data(), hT.size() * 4, hipMemcpyHostToDevice); hipMemcpy(I, hI.data(), hI.size() * 4, hipMemcpyHostToDevice);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipLaunchKernelGGL(victim, dim3(P / 256), dim3(256), 0, s1, T, I, rows, ref);
    hipStreamSynchronize(s1);
    std::vector<float> hr((size_t)P * 2), ho((size_t)P * 2);
    hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost);
    int bad_launches = 0; long long bad_values = 0;
    for (int mode = 0; mode < 2; ++mode) {            // 0: victim alone (control), 1: beside the MFMA stream
        bad_launches = 0; bad_values = 0;
        for (int r = 0; r < REPS; ++r) {
            if (mode) hipLaunchKernelGGL(mfma_load, dim3(1024), dim3(256), 0, s2, sink, 4000);
            hipLaunchKernelGGL(victim, dim3(P / 256), dim3(256), 0, s1, T, I, rows, out);
            hipStreamSynchronize(s1);
            hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
            long long b = 0;
            for (size_t i = 0; i < ho.size(); ++i) b += memcmp(&ho[i], &hr[i], 4) != 0;
            bad_launches += b != 0; bad_values += b;
        }
        hipDeviceSynchronize();
        printf("%s: %d of %d launches differ from the solo result (%lld values)\n", mode ? "beside MFMA stream" : "alone", bad_launches, REPS, bad_values);
    }
    return 0;
}
