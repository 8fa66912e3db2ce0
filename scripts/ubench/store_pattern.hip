// Store-pattern micro-benchmark: how fast can 256-thread workgroups write a [M, N] fp32 matrix in 128x128 tiles,
// with the lane -> address mapping of gemm.hip's epilogue (4 rows x 256 B per wave store) vs 1 KB contiguous per wave store?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE, int NT>
__global__ __launch_bounds__(256) void store_kernel(float* out, int ldc, int M, int N, int ntn) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tm = blockIdx.x / ntn, tn = blockIdx.x % ntn;
    const int m0 = tm * 128, n0 = tn * 128;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)blockIdx.x);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    if (MODE == 0) {  // gemm.hip: wave (wm, wn) owns 64x64; per store 4 rows x 16 lanes x 16 B
        const int wm = wave >> 1, wn = wave & 1;
        for (int i = 0; i < 2; ++i)
            for (int u = 0; u < 8; ++u) {
                const int idx = u * 64 + lane, rr = idx >> 4, c4 = (idx & 15) * 4;
                float* op = out + (size_t)(m0 + wm * 64 + i * 32 + rr) * ldc + n0 + wn * 64 + c4;
                if (NT) { f32x4 vv = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(vv, (f32x4*)op); } else *(float4*)op = v;
            }
    } else {          // wave owns 32 rows x 128 cols; per store 2 rows x 32 lanes x 16 B (512 B contiguous per row)
        for (int u = 0; u < 16; ++u) {
            const int idx = u * 64 + lane, rr = idx >> 5, c4 = (idx & 31) * 4;
            float* op = out + (size_t)(m0 + wave * 32 + rr) * ldc + n0 + c4;
            if (NT) { f32x4 vv = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(vv, (f32x4*)op); } else *(float4*)op = v;
        }
    }
}
template <int MODE, int NT>
float run(float* d, int M, int N) {
    const int ntm = M / 128, ntn = N / 128;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((store_kernel<MODE, NT>), dim3(ntm * ntn), dim3(256), 0, 0, d, N, M, N, ntn);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((store_kernel<MODE, NT>), dim3(ntm * ntn), dim3(256), 0, 0, d, N, M, N, ntn);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20;
}
int main() {
    float* d; hipMalloc(&d, (size_t)1 << 30);
    const int shapes[][2] = {{196608, 128}, {196608, 256}, {98304, 640}, {24576, 1024}, {6144, 5120}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1]; const double gb = (double)M * N * 4 / 1e9;
        float a = run<0, 0>(d, M, N), b = run<0, 1>(d, M, N), c = run<1, 0>(d, M, N), e = run<1, 1>(d, M, N);
        printf("M=%6d N=%5d (%.0f MB): gemm-map %.1f us %.2f TB/s | gemm-map nt %.1f us %.2f TB/s | row-map %.1f us %.2f TB/s | row-map nt %.1f us %.2f TB/s\n",
               M, N, gb * 1e3, a * 1e3, gb / a, b * 1e3, gb / b, c * 1e3, gb / c, e * 1e3, gb / e);
    }
    return 0;
}
