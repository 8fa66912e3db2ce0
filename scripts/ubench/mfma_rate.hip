// MFMA issue-rate micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 per SIMD for 1..4 waves per SIMD and 1 / 2 / 4
// independent accumulators per wave, no memory traffic at all (operands are register constants).  Tells whether the bf16x3
// GEMM's 56 % matrix-pipe busy figure is a scheduling loss or the pipe's real rate under this instruction mix.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_rate.hip -o scripts/ubench/mfma_rate && scripts/ubench/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters, unsigned seed, long long* cyc) {
    const uint4 av = make_uint4(0x3F803F80u + threadIdx.x, 0x3F803F80u, 0x3F803F80u, seed), bv = make_uint4(0x3F803F80u, seed, 0x3F803F80u, 0x3F803F80u);
    const bf16x8 a = __builtin_bit_cast(bf16x8, av), b = __builtin_bit_cast(bf16x8, bv);
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int NACC>
void run(int blocks_per_cu, float* d, long long* dc) {
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_kernel<NACC><<<grid, 256>>>(d, 10, 1u, dc);
    hipEventRecord(e0);
    mfma_kernel<NACC><<<grid, 256>>>(d, iters, 1u, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    const double n_per_simd = (double)iters * 24 * blocks_per_cu;   // one wave of every block per SIMD
    const double tf = (double)grid * 4 * iters * 24 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("acc=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s bf16  s_memtime cycles per MFMA per SIMD (wave 0) = %.1f  (wall-clock %.1f ns per MFMA per SIMD)\n", NACC,
           blocks_per_cu, ms, tf, (double)c / (iters * 24.0) / blocks_per_cu * 1.0, ms * 1e6 / n_per_simd);
}
int main() {
    float* d; long long* dc;
    hipMalloc(&d, 256 * 8 * 256 * 4); hipMalloc(&dc, 8);
    for (int w = 1; w <= 4; ++w) { run<1>(w, d, dc); run<2>(w, d, dc); run<4>(w, d, dc); }
    return 0;
}
