// How gfx950 treats f16 subnormals: (1) v_mfma_f32_32x32x16_f16 inputs, (2) v_cvt_pk_f16_f32 under MODE.fp_denorm(f16/f64) = 0.
// hipcc --offload-arch=gfx950 -O2 scripts/ubench/f16_denorm.hip -o /tmp/f16_denorm && /tmp/f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    // (1) every a = 2^-20 (an f16 subnormal), every b = 1: C[m][n] = 16 * 2^-20 if subnormal inputs are honoured
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1.0f; }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (lane == 0) out[0] = c[0];
    // (2) conversion of 1e-5 (f16 subnormal range) with default mode, then with f16/f64 denormals flushed
    f2 v = {1e-5f * (1 + lane), 3e-5f};
    h2 h = __builtin_convertvector(v, h2);
    if (lane == 0) out[1] = (float)h[0];
    __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 0);   // MODE[7:6] = 0: flush f16/f64 denormals (in and out)
    asm volatile("" ::: "memory");
    h2 g;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(g) : "v"(v[0]), "v"(v[1]));
    if (lane == 0) out[2] = __builtin_bit_cast(unsigned, g) & 0xFFFF ? 1.f : 0.f;
    // MFMA again under the flushing mode
    f16v d = {0};
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
    if (lane == 0) out[3] = d[0];
    __builtin_amdgcn_s_setreg(1 | (6 << 6) | (1 << 11), 3);
}
int main() {
    float* d; hipMalloc(&d, 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("mfma f16 subnormal inputs: C = %.6e (honoured = %.6e)\n", h[0], 16 * 9.5367431640625e-07);
    printf("cvt 1e-5 default mode -> %.6e; under flush mode nonzero = %g; mfma under flush mode: %.6e\n", h[1], h[2], h[3]);
    return 0;
}
