/*
 * ls_oracle.c -- TEST INFRASTRUCTURE ONLY (CPU oracle; never linked into the product path).
 *
 * Canonical CPU restatement of the two integer-exact third-party ops that the
 * LivingScenes hot path reaches through pytorch3d 0.7.4 (pinned in
 * /root/reference/install.sh:6, NOT vendored under /root/reference):
 *
 *   knn_points            called at lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:139-141
 *   sample_farthest_points called at vec_dgcnn_atten.py:169, model_utils.py:205,
 *                          lib_more/more_solver.py:67,107,108,193,252,259
 *
 * pytorch3d is absent from this image, so this file restates the published
 * algorithm of its CPU kernels (knn_cpu.cpp / sample_farthest_points_cpu.cpp):
 *
 *   k-NN : dist(q,s) = sum_{j=0..D-1} (q_j - s_j)^2 accumulated sequentially in
 *          fp32 in ascending j; the K smallest under the lexicographic order
 *          (dist, index) are returned in ascending order (max-heap of
 *          (dist, idx) tuples, strict '<' admission).
 *   FPS  : start at index 0 (random_start_point=False), keep
 *          mind[j] = min(mind[j], ||p_last - p_j||^2) (same fp32 sequential
 *          sum), next = FIRST arg-max of mind.
 *
 * PARITY UNPINNED for these two ops: the reference holds no test/golden vector
 * for them and pytorch3d cannot be imported here.  The
 * arithmetic is therefore *defined* here; `contract` selects the two plausible
 * compilations of `dist += diff*diff`:
 *     contract = 0 : separate fp32 multiply and add  (x86-64 gcc build of knn_cpu.cpp, no -mfma)
 *     contract = 1 : fused  d = fmaf(diff, diff, d)  (nvcc default -fmad=true build of knn.cu)
 * The HIP kernels implement both and are tested bit-exact against both.
 *
 * Feature layout: a point's feature vector is given as `x-major` rows
 * feat[n][x][c] (x in 0..2, c in 0..C-1), D = 3*C, but the canonical summation
 * index is the reference's flattening of [B,C,3,N] -> [B,3C,N]
 * (vec_dgcnn_atten.py:138): j = c*3 + x.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

static inline float acc_sq(float d, float diff, int contract) {
    if (contract) return fmaf(diff, diff, d);
    volatile float p = diff * diff; /* volatile: forbid the compiler from contracting */
    return d + p;
}

/* dst: [Nd][3][C]  src: [Ns][3][C]  (one instance);  idx_out: [Nd][K] int32, dist_out: [Nd][K] (nullable) */
static void knn_one(const float* dst, const float* src, int Nd, int Ns, int C, int K,
                    int contract, int32_t* idx_out, float* dist_out) {
#pragma omp parallel
    {
        float* bd = (float*)malloc(sizeof(float) * (size_t)K);
        int32_t* bi = (int32_t*)malloc(sizeof(int32_t) * (size_t)K);
#pragma omp for schedule(static)
        for (int q = 0; q < Nd; ++q) {
            const float* a = dst + (size_t)q * 3 * C;
            int cnt = 0;
            for (int s = 0; s < Ns; ++s) {
                const float* b = src + (size_t)s * 3 * C;
                float d = 0.0f;
                for (int c = 0; c < C; ++c)
                    for (int x = 0; x < 3; ++x) {
                        float diff = a[x * C + c] - b[x * C + c];
                        d = acc_sq(d, diff, contract);
                    }
                /* sorted insertion under (dist, idx); s ascending so ties keep the earlier index */
                if (cnt < K || d < bd[K - 1]) {
                    int pos = cnt < K ? cnt : K - 1;
                    while (pos > 0 && d < bd[pos - 1]) { /* strict: equal dist keeps earlier idx first */
                        bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos;
                    }
                    bd[pos] = d; bi[pos] = s;
                    if (cnt < K) ++cnt;
                }
            }
            for (int k = 0; k < K; ++k) {
                idx_out[(size_t)q * K + k] = k < cnt ? bi[k] : -1;
                if (dist_out) dist_out[(size_t)q * K + k] = k < cnt ? bd[k] : INFINITY;
            }
        }
        free(bd); free(bi);
    }
}

void lso_knn(const float* dst, const float* src, int B, int Nd, int Ns, int C, int K,
             int contract, int32_t* idx_out, float* dist_out) {
    for (int b = 0; b < B; ++b)
        knn_one(dst + (size_t)b * Nd * 3 * C, src + (size_t)b * Ns * 3 * C, Nd, Ns, C, K, contract,
                idx_out + (size_t)b * Nd * K, dist_out ? dist_out + (size_t)b * Nd * K : NULL);
}

/* pts: [B][N][3]; lengths: [B] (nullable -> N); idx_out: [B][K] int32 */
void lso_fps(const float* pts, const int32_t* lengths, int B, int N, int K, int contract,
             int32_t* idx_out) {
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < B; ++b) {
        const float* p = pts + (size_t)b * N * 3;
        int n = lengths ? lengths[b] : N;
        float* mind = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
        for (int j = 0; j < n; ++j) mind[j] = INFINITY;
        int last = 0;
        int32_t* out = idx_out + (size_t)b * K;
        for (int k = 0; k < K; ++k) out[k] = -1;
        if (n > 0) out[0] = 0;
        int kk = K < n ? K : n;
        for (int k = 1; k < kk; ++k) {
            float best = -1.0f; int besti = 0;
            for (int j = 0; j < n; ++j) {
                float d = 0.0f;
                for (int x = 0; x < 3; ++x) {
                    float diff = p[last * 3 + x] - p[j * 3 + x];
                    d = acc_sq(d, diff, contract);
                }
                float m = mind[j] < d ? mind[j] : d;
                mind[j] = m;
                if (m > best) { best = m; besti = j; } /* strict '>' : first arg-max */
            }
            last = besti;
            out[k] = besti;
        }
        free(mind);
    }
}

int lso_version(void) { return 1; }
