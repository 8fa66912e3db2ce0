// mise.hip -- multi-resolution iso-surface extraction (MISE) bookkeeping on the device: which lattice points of the
// (R+1)^3 grid still have to be evaluated by the decoder, and the final dense value grid.
//
// Replaces the CPU octree of
//   /root/reference/lib_shape_prior/core/models/utils/occnet_utils/utils/libmise/mise.pyx            (class MISE)
// as driven by
//   /root/reference/lib_shape_prior/core/models/utils/occnet_utils/mesh_extractor2.py:116-131        (query / eval / update loop)
// (the consumer of the DeepSDF decoder; the reference copies every round's points and values between
// device and host and keeps a std::map keyed octree.)  The reference's pointer octree becomes dense flag arrays:
//   value / known / exists on the lattice, sub[l] = "level-l voxel has been subdivided" (a voxel is a leaf iff all its
//   ancestors are subdivided and it is not).  Per update (mise.pyx:188-236): every KNOWN lattice point marks the leaf voxel
//   of each of its 8 adjacent unit cells as next-to-positive (value >= threshold) and / or next-to-negative (<=); every leaf
//   below the finest level carrying both marks is subdivided and its 27 corner / edge / face / centre points come to exist.
// Everything is integer / flag work plus exact comparisons: the query sets per round and the dense grid are bit-identical to
// the reference's (tests/golden/mise.npz; the ORDER of a round's queries is ascending lattice index instead of the
// reference's insertion order -- the field is point-wise, so it cannot matter).
#include "ls_common.h"

namespace ls {

struct MiseLayout {
    int res0, depth, R, G;
    size_t o_val, o_known, o_exists, o_sub[8], o_pos[8], o_neg[8], o_blk, total;
    long long npts;
    int nblk;
};
constexpr int MISE_PER_BLOCK = 4096;   // lattice points per compaction workgroup (256 threads x 16)

static MiseLayout mise_layout(int res0, int depth) {
    MiseLayout L{};
    L.res0 = res0; L.depth = depth; L.R = res0 << depth; L.G = L.R + 1;
    L.npts = (long long)L.G * L.G * L.G;
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off = (off + b + 255) & ~(size_t)255; return o; };
    L.o_val = take((size_t)L.npts * 4);
    L.o_known = take((size_t)L.npts);
    L.o_exists = take((size_t)L.npts);
    for (int l = 0; l < depth; ++l) {
        const size_t n = (size_t)(res0 << l) * (res0 << l) * (res0 << l);
        L.o_sub[l] = take(n); L.o_pos[l] = take(n); L.o_neg[l] = take(n);
    }
    L.nblk = (int)((L.npts + MISE_PER_BLOCK - 1) / MISE_PER_BLOCK);
    L.o_blk = take((size_t)(L.nblk + 1) * 4);
    L.total = off;
    return L;
}

struct MiseDev {   // device view passed by value
    float* val; unsigned char* known; unsigned char* exists;
    unsigned char* sub[8]; unsigned char* pos[8]; unsigned char* neg[8];
    int* blk;
    int res0, depth, R, G;
};
static MiseDev mise_dev(void* state, const MiseLayout& L) {
    MiseDev d{};
    char* b = (char*)state;
    d.val = (float*)(b + L.o_val); d.known = (unsigned char*)(b + L.o_known); d.exists = (unsigned char*)(b + L.o_exists);
    for (int l = 0; l < L.depth; ++l) {
        d.sub[l] = (unsigned char*)(b + L.o_sub[l]); d.pos[l] = (unsigned char*)(b + L.o_pos[l]); d.neg[l] = (unsigned char*)(b + L.o_neg[l]);
    }
    d.blk = (int*)(b + L.o_blk);
    d.res0 = L.res0; d.depth = L.depth; d.R = L.R; d.G = L.G;
    return d;
}

// initial lattice: the (res0+1)^3 corner points of the coarse voxels exist (mise.pyx:74-85)
__global__ void mise_init_kernel(MiseDev d, long long npts) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    const int G = d.G, z = (int)(i % G), y = (int)((i / G) % G), x = (int)(i / ((long long)G * G));
    const int m = (1 << d.depth) - 1;
    d.exists[i] = ((x & m) | (y & m) | (z & m)) == 0;
    d.known[i] = 0;
    d.val[i] = 0.f;
}

// ---- query: ordered compaction of exists && !known
__global__ __launch_bounds__(256) void mise_count_kernel(MiseDev d, long long npts) {
    __shared__ int red[4];
    const long long base = (long long)blockIdx.x * MISE_PER_BLOCK;
    int c = 0;
    for (int u = 0; u < 16; ++u) {
        const long long i = base + (long long)threadIdx.x * 16 + u;
        if (i < npts) c += (d.exists[i] && !d.known[i]) ? 1 : 0;
    }
    c = (int)wave_sum((float)c);   // <= 4096: exact in fp32
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) d.blk[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(1024) void mise_scan_kernel(int* blk, int nblk, int* count_out) {   // exclusive scan, one workgroup
    __shared__ int part[1024];
    const int t = threadIdx.x, per = (nblk + 1023) / 1024;
    int s = 0;
    for (int u = 0; u < per; ++u) { const int i = t * per + u; if (i < nblk) s += blk[i]; }
    part[t] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = t >= o ? part[t - o] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t ? part[t - 1] : 0;
    for (int u = 0; u < per; ++u) { const int i = t * per + u; if (i < nblk) { const int c = blk[i]; blk[i] = run; run += c; } }
    if (t == 1023) { blk[nblk] = part[1023]; *count_out = part[1023]; }
}
__global__ __launch_bounds__(256) void mise_emit_kernel(MiseDev d, long long npts, float box_size, int32_t* idx_out, float* pts_out,
                                                        int cap) {
    __shared__ int wsum[4];
    const long long base = (long long)blockIdx.x * MISE_PER_BLOCK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned m = 0;
    for (int u = 0; u < 16; ++u) {
        const long long i = base + (long long)tid * 16 + u;
        if (i < npts && d.exists[i] && !d.known[i]) m |= 1u << u;
    }
    const int c = __builtin_popcount(m);
    int inc = c;   // inclusive scan over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (lane >= o) inc += v; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int off = d.blk[blockIdx.x] + inc - c;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    const int G = d.G;
    for (int u = 0; u < 16; ++u) {
        if (!(m & (1u << u))) continue;
        const long long i = base + (long long)tid * 16 + u;
        if (off < cap) {
            const int z = (int)(i % G), y = (int)((i / G) % G), x = (int)(i / ((long long)G * G));
            idx_out[off] = (int32_t)i;
            // mesh_extractor2.py:122-124: p / resolution (float32 division), then box_size * (p - 0.5)
            pts_out[(size_t)off * 3 + 0] = box_size * ((float)x / (float)d.R - 0.5f);
            pts_out[(size_t)off * 3 + 1] = box_size * ((float)y / (float)d.R - 0.5f);
            pts_out[(size_t)off * 3 + 2] = box_size * ((float)z / (float)d.R - 0.5f);
        }
        ++off;
    }
}

// ---- update
__global__ void mise_scatter_kernel(MiseDev d, const int32_t* __restrict__ idx, const float* __restrict__ values, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    d.val[idx[i]] = values[i];
    d.known[idx[i]] = 1;
}
__global__ void mise_clear_kernel(unsigned char* a, unsigned char* b, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = 0; b[i] = 0; }
}
__global__ void mise_mark_kernel(MiseDev d, long long npts, double threshold) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts || !d.known[i]) return;
    const int G = d.G, R = d.R, D = d.depth;
    const int z = (int)(i % G), y = (int)((i / G) % G), x = (int)(i / ((long long)G * G));
    const double v = (double)d.val[i];
    const bool ge = v >= threshold, le = v <= threshold;
    for (int di = -1; di <= 0; ++di)
        for (int dj = -1; dj <= 0; ++dj)
            for (int dk = -1; dk <= 0; ++dk) {
                const int px = x + di, py = y + dj, pz = z + dk;
                if (px < 0 || py < 0 || pz < 0 || px >= R || py >= R || pz >= R) continue;
                int l = 0;
                for (; l < D; ++l) {   // descend while the level-l voxel containing the cell is subdivided
                    const int s = D - l, n = d.res0 << l;
                    if (!d.sub[l][((size_t)(px >> s) * n + (py >> s)) * n + (pz >> s)]) break;
                }
                if (l == D) continue;   // finest-level leaves are never subdivided
                const int s = D - l, n = d.res0 << l;
                const size_t c = ((size_t)(px >> s) * n + (py >> s)) * n + (pz >> s);
                if (ge) d.pos[l][c] = 1;
                if (le) d.neg[l][c] = 1;
            }
}
__global__ void mise_subdivide_kernel(MiseDev d, int l) {
    const int n = d.res0 << l;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n * n * n) return;
    if (!(d.pos[l][i] && d.neg[l][i])) return;   // marks only ever land on leaves
    d.sub[l][i] = 1;
    const int z = (int)(i % n), y = (int)((i / n) % n), x = (int)(i / ((size_t)n * n));
    const int s = d.depth - l, ns = 1 << (s - 1), G = d.G;   // new_size, mise.pyx:245
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            for (int c = 0; c < 3; ++c)
                d.exists[((size_t)((x << s) + a * ns) * G + ((y << s) + b * ns)) * G + ((z << s) + c * ns)] = 1;
}

// ---- dense grid (mise.pyx:128-165)
__global__ void mise_dense_fill_kernel(MiseDev d, long long npts, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npts) out[i] = d.exists[i] ? d.val[i] : __builtin_nanf("");
}
__global__ void mise_dense_axis_kernel(float* __restrict__ out, int G, int axis) {   // one thread per line along `axis`
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= G * G) return;
    const int a = t / G, b = t % G;
    size_t base, stride;
    if (axis == 0) { base = (size_t)a * G + b; stride = (size_t)G * G; }        // (j,k) fixed, walk i
    else if (axis == 1) { base = (size_t)a * G * G + b; stride = (size_t)G; }   // (i,k) fixed, walk j
    else { base = ((size_t)a * G + b) * G; stride = 1; }                        // (i,j) fixed, walk k
    float prev = out[base];
    for (int s = 1; s < G; ++s) {
        const size_t o = base + (size_t)s * stride;
        const float v = out[o];
        if (v != v) out[o] = prev; else prev = v;
    }
}

}  // namespace ls

using namespace ls;

extern "C" {

size_t ls_mise_state_bytes(int res0, int depth) {
    if (res0 < 1 || depth < 0 || depth > 7 || ((long long)res0 << depth) > 1024) return 0;
    return mise_layout(res0, depth).total;
}
long long ls_mise_lattice_points(int res0, int depth) {
    if (res0 < 1 || depth < 0 || depth > 7 || ((long long)res0 << depth) > 1024) return 0;
    return mise_layout(res0, depth).npts;
}

int ls_mise_init(void* state, size_t state_bytes, int res0, int depth, void* stream) {
    LS_REQUIRE(state != nullptr, "mise: null state");
    LS_REQUIRE(res0 >= 1 && depth >= 0 && depth <= 7 && ((long long)res0 << depth) <= 1024, "mise: resolution_0=%d depth=%d unsupported", res0, depth);
    const MiseLayout L = mise_layout(res0, depth);
    if (state_bytes < L.total) { set_error("mise: state %zu < required %zu bytes", state_bytes, L.total); return LS_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    LS_HIP_CHECK(hipMemsetAsync(state, 0, L.total, st));
    const MiseDev d = mise_dev(state, L);
    hipLaunchKernelGGL(mise_init_kernel, dim3(cdiv(L.npts, 256)), dim3(256), 0, st, d, L.npts);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

// Unknown lattice points in ascending lattice order: idx_out[cap] (linear index (x*G + y)*G + z), pts_out[cap,3] (the
// decoder's query coordinates, mesh_extractor2.py:122-124), *count_out (device int; may exceed cap: nothing past cap is written).
int ls_mise_query(void* state, int res0, int depth, float box_size, int32_t* idx_out, float* pts_out, int cap, int32_t* count_out,
                  void* stream) {
    LS_REQUIRE(state && idx_out && pts_out && count_out && cap >= 0, "mise_query: null argument");
    LS_REQUIRE(res0 >= 1 && depth >= 0 && depth <= 7 && ((long long)res0 << depth) <= 1024, "mise: resolution_0=%d depth=%d unsupported", res0, depth);
    const MiseLayout L = mise_layout(res0, depth);
    const MiseDev d = mise_dev(state, L);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mise_count_kernel, dim3(L.nblk), dim3(256), 0, st, d, L.npts);
    hipLaunchKernelGGL(mise_scan_kernel, dim3(1), dim3(1024), 0, st, d.blk, L.nblk, count_out);
    hipLaunchKernelGGL(mise_emit_kernel, dim3(L.nblk), dim3(256), 0, st, d, L.npts, box_size, idx_out, pts_out, cap);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

// Store the values of the queried points, then subdivide every active leaf voxel (mise.pyx:87-102, 188-236).
int ls_mise_update(void* state, int res0, int depth, double threshold, const int32_t* idx, const float* values, int n, void* stream) {
    LS_REQUIRE(state && (n == 0 || (idx && values)) && n >= 0, "mise_update: null argument");
    LS_REQUIRE(res0 >= 1 && depth >= 0 && depth <= 7 && ((long long)res0 << depth) <= 1024, "mise: resolution_0=%d depth=%d unsupported", res0, depth);
    const MiseLayout L = mise_layout(res0, depth);
    const MiseDev d = mise_dev(state, L);
    hipStream_t st = (hipStream_t)stream;
    if (n > 0) hipLaunchKernelGGL(mise_scatter_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, d, idx, values, n);
    for (int l = 0; l < depth; ++l) {
        const size_t nv = (size_t)(res0 << l) * (res0 << l) * (res0 << l);
        hipLaunchKernelGGL(mise_clear_kernel, dim3(cdiv((long long)nv, 256)), dim3(256), 0, st, d.pos[l], d.neg[l], nv);
    }
    if (depth > 0) {
        hipLaunchKernelGGL(mise_mark_kernel, dim3(cdiv(L.npts, 256)), dim3(256), 0, st, d, L.npts, threshold);
        for (int l = 0; l < depth; ++l) {
            const size_t nv = (size_t)(res0 << l) * (res0 << l) * (res0 << l);
            hipLaunchKernelGGL(mise_subdivide_kernel, dim3(cdiv((long long)nv, 256)), dim3(256), 0, st, d, l);
        }
    }
    LS_LAUNCH_CHECK();
    return LS_OK;
}

// Dense (R+1)^3 value grid: known values, the rest completed along x, then y, then z (mise.pyx:128-165).
int ls_mise_to_dense(void* state, int res0, int depth, float* grid_out, void* stream) {
    LS_REQUIRE(state && grid_out, "mise_to_dense: null argument");
    LS_REQUIRE(res0 >= 1 && depth >= 0 && depth <= 7 && ((long long)res0 << depth) <= 1024, "mise: resolution_0=%d depth=%d unsupported", res0, depth);
    const MiseLayout L = mise_layout(res0, depth);
    const MiseDev d = mise_dev(state, L);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mise_dense_fill_kernel, dim3(cdiv(L.npts, 256)), dim3(256), 0, st, d, L.npts, grid_out);
    for (int axis = 0; axis < 3; ++axis)
        hipLaunchKernelGGL(mise_dense_axis_kernel, dim3(cdiv(L.G * L.G, 256)), dim3(256), 0, st, grid_out, L.G, axis);
    LS_LAUNCH_CHECK();
    return LS_OK;
}

}  // extern "C"
