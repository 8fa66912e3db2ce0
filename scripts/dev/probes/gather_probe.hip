// dev probe: what does the vector-memory path of one MI355X deliver for row gathers out of an L2-resident table, by segment size?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_probe scripts/dev/probes/gather_probe.hip && /tmp/gather_probe
// Table: NI "instances" of R rows x ROWB bytes (default 1024 x 3072 B = 3 MB: one instance per XCD pass, as the attention layers have it).
// A wave issues 16-byte-per-lane loads; SEG lanes share one row segment of SEG * 16 bytes (SEG = 4 / 8 / 16 / 32 / 64), row indices pseudo-random
// inside the workgroup's instance; DEPTH independent loads are in flight per lane before their results are consumed (summed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int SEG, int DEPTH>
__global__ __launch_bounds__(256) void gather_kernel(const float4* __restrict__ T, const int* __restrict__ idx, int R, int row16, int iters, int wg_per_inst,
                                                     float4* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int logical = (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;   // XCD-aware order: an instance's workgroups on one XCD (its table in that L2)
    const int inst = logical / wg_per_inst;
    const float4* Tb = T + (size_t)inst * R * row16;
    const int grp = lane / SEG, l = lane % SEG;              // 64 / SEG row segments per instruction
    const int* ip = idx + ((size_t)blockIdx.x * 4 + wave) * iters * (64 / SEG) * DEPTH;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int segoff = (blockIdx.x * 7 + wave) % (row16 / SEG) * SEG;   // which segment of the row this wave reads
    for (int it = 0; it < iters; ++it) {
        float4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int r = ip[(it * DEPTH + d) * (64 / SEG) + grp];
            v[d] = Tb[(size_t)r * row16 + segoff + l];
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int DEPTH>
__global__ __launch_bounds__(256) void stream_kernel(const float4* __restrict__ T, size_t n16, int iters, float4* __restrict__ out) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    size_t base = ((size_t)blockIdx.x * iters * DEPTH) * 256 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        float4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = T[(base + (size_t)(it * DEPTH + d) * 256) % n16];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int SEG, int DEPTH>
static void run(const float4* T, const int* idx, int NI, int R, int row16, float4* out, int wgs, int iters) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int wpi = wgs / NI;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((gather_kernel<SEG, DEPTH>), dim3(wgs), dim3(256), 0, 0, T, idx, R, row16, iters, wpi, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    }
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)wgs * 256 * iters * DEPTH * 16;
    printf("gather  seg %4d B  depth %2d  wgs %5d : %8.1f us  %7.2f TB/s  %5.1f B/clk/CU (2.4 GHz)\n", SEG * 16, DEPTH, wgs, ms * 1e3, bytes / ms / 1e9,
           bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
    const int NI = 64, R = 1024, ROWB = 3072, row16 = ROWB / 16;
    const size_t n16 = (size_t)NI * R * row16;
    float4* T; CK(hipMalloc(&T, n16 * 16)); CK(hipMemset(T, 0, n16 * 16));
    const int wgs = 4096, iters = 32;
    std::vector<int> h((size_t)wgs * 4 * iters * 16 * 8);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (s >> 10) % R; }
    int* idx; CK(hipMalloc(&idx, h.size() * 4)); CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    float4* out; CK(hipMalloc(&out, (size_t)wgs * 256 * 16));
    // streaming reference (whole table, 201 MB: out of HBM / MALL) and an L2-sized one (24 MB)
    for (size_t span : {n16, (size_t)(24u << 20) / 16}) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int rep = 0; rep < 2; ++rep) { CK(hipEventRecord(a)); hipLaunchKernelGGL((stream_kernel<4>), dim3(wgs), dim3(256), 0, 0, T, span, iters, out); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); }
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double bytes = (double)wgs * 256 * iters * 4 * 16;
        printf("stream  span %6.0f MB depth 4 : %8.1f us  %7.2f TB/s  %5.1f B/clk/CU\n", span * 16 / 1e6, ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
    }
    run<64, 4>(T, idx, NI, R, row16, out, wgs, iters);
    run<32, 4>(T, idx, NI, R, row16, out, wgs, iters);
    run<16, 4>(T, idx, NI, R, row16, out, wgs, iters);
    run<16, 8>(T, idx, NI, R, row16, out, wgs, iters);
    run<16, 12>(T, idx, NI, R, row16, out, wgs, iters);
    run<8, 4>(T, idx, NI, R, row16, out, wgs, iters);
    run<8, 8>(T, idx, NI, R, row16, out, wgs, iters);
    run<4, 8>(T, idx, NI, R, row16, out, wgs, iters);
    return 0;
}
