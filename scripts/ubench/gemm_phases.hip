// Phase timing of the bf16x3 GEMM main loop (dev tool): compiles csrc/gemm.hip with -DLS_GEMM_PROF and reports, summed over all
// waves, the s_memtime cycles spent in: [0] barrier before the slab store, [1] split + LDS store (incl. the wait for the global
// loads), [2] barrier after the store, [3] LDS operand reads + MFMA issue.
//   hipcc --offload-arch=gfx950 -O3 -w -DLS_GEMM_PROF -Iinclude -Ilivingscenes_amd/csrc scripts/ubench/gemm_phases.hip -o scripts/ubench/gemm_phases
#include "../../livingscenes_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstring>
#include <cstdarg>
#include <vector>
namespace ls { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc(10, stderr); } }
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 262144, N = argc > 2 ? atoi(argv[2]) : 768, K = argc > 3 ? atoi(argv[3]) : 768;
    const int lda = argc > 4 ? atoi(argv[4]) : K;   // 0: every A row is row 0 (an L1/L2-resident A operand: separates the cache path from the rest)
    float *A, *W, *O;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&W, (size_t)N * K * 4); hipMalloc(&O, (size_t)M * N * 4);
    std::vector<float> h((size_t)M * K);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        unsigned long long z[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(ls::ls_gemm_prof), z, sizeof z);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        ls::gemm_dispatch(A, lda, W, K, nullptr, O, N, M, N, K, 0, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpyFromSymbol(z, HIP_SYMBOL(ls::ls_gemm_prof), sizeof z);
        const double tot = (double)(z[0] + z[1] + z[2] + z[3]);
        printf("M=%d N=%d K=%d: %.3f ms (%.1f TFLOP/s)  barrier1 %.1f%%  split+store %.1f%%  barrier2 %.1f%%  reads+MFMA %.1f%%  (sum %.3g wave-cycles)\n", M, N, K,
               ms, 2.0 * M * N * K / ms / 1e9, 100 * z[0] / tot, 100 * z[1] / tot, 100 * z[2] / tot, 100 * z[3] / tot, tot);
    }
    return 0;
}
