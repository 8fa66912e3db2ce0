// model.hip -- the C ABI: model handle, leaf-operator exports and the two composite hot-path entry points
// (ls_encode = Shape_Prior.encode, ls_sdf_decode = FieldWrapper.forward).  Host code only enqueues work on the
// caller's stream (plus one library-owned side stream for the FPS chain); it never synchronises the device.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "ls_common.h"

namespace ls {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// kernels / launchers defined in the other translation units
int knn_dispatch(const float*, const float*, const int32_t*, int, int, int, int, int, int, unsigned, int32_t*, float*, void*, const int32_t*, int, int,
                 hipStream_t);
size_t knn_scratch_bytes(int B, int Nd, int dst_n, int Ns, int C, bool seeded, unsigned flags);
bool knn_would_sweep(int C, int Ns, unsigned flags);
int knn_sweep_stats_launch(const void* scratch, int B, int Nd, int dst_n, int Ns, unsigned long long* out, hipStream_t st);
int fps_dispatch(const float*, const int32_t*, int, int, int, unsigned, int32_t*, float*, void*, size_t, hipStream_t);
size_t fps_scratch_bytes_per_cloud(int N);
int gemm_dispatch(const float*, int, const float*, int, const float*, float*, int, int, int, int, int, hipStream_t, GemmAux aux = GemmAux());
int gemm_rowmax_launch(const float* W, int rows, int K, int ldw, float* out, hipStream_t st);
int gemm_rowmax_parts(int N);
size_t gemm_w_planes_bytes(size_t rows, int K);
bool gemm_w_planes_useful(int K);
int gemm_presplit_w_launch(const float* W, int rows, int K, int ldw, const float* rowmax, void* planes, hipStream_t st);
int sdf_affine_rowmax_parts(int out_dim);
int edge_l0_launch(const float*, const int32_t*, const float*, int, int, int, float, float*, hipStream_t);
int edge_pool_launch(const float*, int, const float*, int, int, int, const int32_t*, const int32_t*, int, int, int, int, float, float*, hipStream_t);
int edge_attn_launch(const float*, int, const float*, int, int, int, const int32_t*, const int32_t*, int, int, int, int, int, float, float*, hipStream_t, float* rowmax = nullptr,
                     float* colsum = nullptr);
bool edge_attn_emits_rowmax(int Co, int ldt, int ldq);
bool edge_attn_fq_supported(int Co, int Cin);
bool edge_attn_fq_fits(int B, int Ns, int ldt);
int edge_attn_fq_launch(const float*, int, const float*, int, const void*, const int32_t*, const int32_t*, int, int, int, int, int, float, float*, hipStream_t, float* rowmax = nullptr,
                        float* colsum = nullptr);
int edge_attn_fq_points_per_wg(int Co);
size_t edge_wq_planes_bytes(int Co, int Cin);
// edge_fused.hip: attention layers with 32 destination points (released layers 5 / 6) -- table slices formed and consumed in LDS
bool edge_ft_supported(int Co, int Cin, int Ns, int Nd, int head_c, bool has_rows);
size_t edge_ft_w_bytes(int Co, int Cin);
int edge_ft_presplit_w_launch(const float* W, int Co, int Cin, void* planes, hipStream_t st);
size_t edge_ft_scratch_bytes(int B, int Ns, int Nd, int Cin, int Co, bool has_rows);
int edge_ft_prep_launch(const float* cur, const int32_t* dst_rows, int B, int Ns, int Nd, int Cin, int Co, void* scratch, hipStream_t st);
int edge_ft_attn_launch(const void* wplanes, const int32_t* knn, bool has_rows, int B, int Ns, int Nd, int Cin, int Co, float neg_slope, void* scratch,
                        float* out, float* rowmax, hipStream_t st, float* colsum = nullptr);
int edge_ft_rowmax_parts(int Co, int Cin);
int edge_presplit_wq_launch(const float* Wq, int Co, int Cin, void* planes, hipStream_t st);
// edge_staged.hip: attention layers 2 - 4 with LDS-staged neighbour tiles (slice-major table)
int edge_st_variant(int Co, int Cin);
int edge_st_cs(int variant);
int edge_st_pts(int variant);
size_t edge_st_q_bytes(int Co, int Cin);
bool edge_st_fits(int Co, int Cin, int Ns, int Nd);
int edge_st_prepare_launch(const float* W, int Co, int Cin, float* Wp, void* qplanes, hipStream_t st);
int edge_attn_staged_launch(const float* T, const float* cur, int Cin, const void* qplanes, const int32_t* knn, const int32_t* dst_rows, int B, int Nd, int Ns,
                            int Co, int head_c, float neg_slope, float* out, hipStream_t st, float* rowmax);
int gemm_dispatch_gather(const float*, int, const float*, int, const float*, float*, int, int, int, int, int, const int32_t*, int, int, hipStream_t, GemmAux aux = GemmAux());
int gemm_dispatch_ws(const float*, int, const float*, int, const float*, float*, int, int, int, int, int, float*, hipStream_t, GemmAux aux = GemmAux());
int gemm_dispatch_small(const float*, int, const float*, int, const float*, float*, int, int, int, int, int, float*, hipStream_t);
bool gemm_vn_supported(int M, int C, int K);
bool gemm_vn_streams(int M, int C, int K, int lda, int npts, const GemmAux& aux);
int gemm_mode();
int gemm_vn_dispatch(const float*, int, const float*, int, const float*, int, float*, int, int, int, int, float, hipStream_t, GemmAux aux = GemmAux());
int gemm_dispatch_fast2(const float*, int, const float*, int, const float*, float*, int, int, int, int, int, hipStream_t);
int gemm_dispatch_masked(const float*, int, const float*, int, float*, int, int, int, int, const float*, int, hipStream_t, GemmAux aux = GemmAux());
size_t gemm_scratch_floats(int M, int N, int K);
int prologue_launch(const float*, int, int, float*, float*, float*, float*, hipStream_t);
size_t prologue_scratch_floats(int B);
int transpose_cloud_launch(const float*, int, int, float*, hipStream_t);
int mean_points_launch(const float*, int, int, int, float*, hipStream_t);
int glob_mean_gemv_launch(const float*, int, int, int, const float*, int, int, float*, int, hipStream_t, int npoints = 0);
int vn_act_rows_launch(const float*, int, const float*, int, int, int, int, float, float*, hipStream_t);
int tail_launch(const float*, int, int, int, int, const float*, const float*, const float*, float, float, int, int, const float*,
                const float*, float*, float*, float*, float*, hipStream_t);
int sdf_prep_launch(const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, float*,
                    float*, hipStream_t);
int sdf_affine_launch(const float*, const float*, const float*, const float*, const float*, int, int, int, int, int, float*,
                      hipStream_t, float* rowmax = nullptr);
int sdf_out_launch(const float*, int, int, const float*, const float*, long long, float*, hipStream_t);
int sdf_affine_rows_launch(const float*, const int32_t*, const float*, const float*, const float*, const float*, long long, int, int, int,
                           float*, hipStream_t, float* rowmax = nullptr);
int sdf_out_bwd_launch(const float*, const float*, const float*, const float*, int, int, long long, float*, hipStream_t, float* rowmax = nullptr,
                       const float* wmax = nullptr);
int relu_mask_launch(float*, const float*, long long, int, int, hipStream_t);
int sdf_affine_bwd_launch(const float*, const float*, const float*, const float*, const float*, int, int, int, int, int, float*, float*,
                          float*, bool, hipStream_t);
int sdf_code_grad_launch(const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                         int, int, int, float*, float*, hipStream_t);
int sdf_query_grad_launch(const float*, const float*, const float*, const float*, int, int, float*, float*, float*, hipStream_t);
int transpose_launch(const float*, int, int, float*, hipStream_t);
int cosine_scores_launch(const float*, const float*, int, int, int, float*, float*, hipStream_t);
int greedy_match_launch(float*, int, int, long long*, long long*, hipStream_t);
int assign_launch(const float*, int, int, int, float, float, int, float, long long*, long long*, hipStream_t);
int kabsch_launch(const float*, const float*, const float*, int, int, int, int, float*, float*, float*, float*, int32_t*, hipStream_t,
                  const float* off1 = nullptr, const float* off2 = nullptr, const long long* sel1 = nullptr, const long long* sel2 = nullptr);
size_t icp_workspace_bytes(int b, int n);
int icp_run(const float*, const float*, const float*, const float*, int, int, int, int, float, unsigned, float*, float*, float*,
            int32_t*, void*, size_t, hipStream_t);

}  // namespace ls

using namespace ls;

struct ProfRec { int kind, layer; hipEvent_t a, b; };

struct ls_model {
    ls_model_desc d;
    float* blob = nullptr;
    void* wt_planes[LS_MAX_LAYERS] = {};   // attention layers whose input has 128 / 256 channels (released layers 5, 6): ALL table weights as per-head f16
                                           // MFMA fragment tiles (edge_fused.hip); used when the call's point counts fit the fused kernels
    void* wq_planes[LS_MAX_LAYERS] = {};   // attention layers 2 - 4: destination-side weights as f16 MFMA fragments (edge.hip, edge_attn_fq_kernel)
    float* wst_w[LS_MAX_LAYERS] = {};      // attention layers 2 - 4, LDS-staged path (edge_staged.hip): neighbour-side weight rows in slice order ...
    void* wst_q[LS_MAX_LAYERS] = {};       // ... and the destination-side weights as per-Q-block f16 MFMA fragments
    bool fuse_q = true;                    // LS_OPT_EDGE_FUSE_Q: destination side of attention layers 2 - 4 inside the edge kernel (edge.hip: edge_attn_fq_kernel)
    bool fuse_t = true;                    // LS_OPT_EDGE_FUSE_T: table-free 32-point attention layers (edge_fused.hip)
    int glob_fuse = 1;                     // LS_OPT_GLOB_FUSE (0 / 1 / 2; 2 = 1 + the operator export ls_vn_lna_f32 chains exact row maxima like the encoder's producers do): residual global conv as mean + GEMV launch and GEMM + VN activation in one kernel (gemm.hip: gemm_vn_kernel)
    int debug_edge = 0;                    // LS_OPT_DEBUG_EDGE: ls_vn_edgeconv_* runs 0 = table GEMM + edge kernel, 1 = the table GEMM only, 2 = the edge kernel only on the
                                           // tables already in the workspace (per-operator counter passes: scripts/pmc_ops.py)
    int edge_staged = 0;                   // LS_OPT_EDGE_STAGED: 0 = never (default: measured slower than the gather kernel), 1 = when the grid fills the chip
                                           // (B * Nd / PTS >= 128 workgroups), 2 = whenever the shape fits
    float* dec_wt = nullptr;            // transposed decoder weights [kin_l][out_l], built by the first backward call
    size_t dec_wt_off[12] = {};
    // max|row| of every weight matrix a GEMM reads (gemm.hip, GemmAux::w_rowmax): saves the kernels their pre-pass over W
    struct WMax { const float* base; size_t rows; int K; const float* wmax; const char* planes; };   // planes: pre-split rows (GemmAux::w_planes) or null
    std::vector<WMax> wreg;
    float* wmax_pool = nullptr;         // blob matrices (ls_model_create)
    float* wmax_pool_t = nullptr;       // transposed decoder weights (first backward call)
    char* wplanes_pool = nullptr;       // pre-split f16 pieces of the K >= 128 matrices of the blob
    char* wplanes_pool_t = nullptr;     // ... of the transposed decoder weights
    hipStream_t side = nullptr;    // FPS chain
    hipStream_t side2 = nullptr;   // per-layer table GEMMs, concurrent with the k-NN of the same layer
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_fps[LS_MAX_LAYERS] = {};   // one per FPS level: a down-sampling layer waits for ITS level only (round 3: waiting for the whole chain
                                             // kept layer 2 idle for ~100 us while levels 1 and 2 were still running)
    hipEvent_t ev_feat[LS_MAX_LAYERS] = {}, ev_tab[LS_MAX_LAYERS] = {};
    bool train_splitk = true;      // ls_model_set_option(LS_OPT_SDF_TRAIN_SPLITK): split-K in the decoder's TRAINING-path GEMMs (under-filled
                                   // M = 1024 problems: 2.43 -> 1.6 ms per step); off = a row's result never depends on the batch it rides in
    bool sdf_bf16x2 = false;       // LS_SDF_BF16X2=1: decoder GEMMs with two-piece bf16 products (2^-16 per product, ~1.7x; opt-in)
    // captured launch sequences of ls_encode, one per (workspace, B, N, mode): OPT-IN (LS_ENCODE_GRAPH=1 / ls_model_set_option).
    // Measured on MI355X / ROCm 7.2 (round 2, bench.py, B = 64): replaying the ~170-node graph costs the host MORE than enqueueing
    // the kernels directly -- one step in flight 22.2k obj/s (2.53 ms of host time per step) vs 29.6k (1.98 ms) direct; eight steps in
    // flight 38.1k vs 39.4k -- so the direct path stays the default.
    struct EncGraph { void* ws; size_t ws_bytes; int B, N, pre; unsigned flags; hipStream_t st; hipGraph_t graph; hipGraphExec_t exec; };
    std::vector<EncGraph> graphs;
    bool use_graph = false;
    bool graph_broken = false;     // a capture / instantiate failed once on this handle: stay on the direct path
    int debug_layers = -1;         // LS_DEBUG_LAYERS=n: ls_encode stops after n layers (outputs undefined) and prints the workspace plan:
                                   // race hunting by comparing workspaces (scripts/diag/)
    bool fps_side = true;          // LS_FPS_SIDE=0 runs the FPS chain on the caller's stream (A/B timing, race hunting)
    bool overlap_gemm = true;      // LS_GEMM_OVERLAP=0 serialises the table GEMMs on the caller's stream (A/B timing)
    unsigned skip_mask = 0;        // (always 0 unless built with -DLS_DEV_KNOBS) LS_SKIP=knn,attn,...: dev timing knob -- after LS_SKIP_AFTER (default 3) ls_encode calls on this handle the named
    int skip_after = 3, calls = 0; // launches are skipped (their outputs keep the previous call's values): marginal cost of a kernel family
                                   // with many steps in flight.  Results are then STALE: never set outside scripts/dev.
    bool profiling = false;
    std::vector<ProfRec> prof;          // pending (un-collected) event pairs
    std::vector<hipEvent_t> ev_pool;    // recycled events
    unsigned long long* knn_stats = nullptr;   // device [LS_MAX_LAYERS][2]: exact-phase statistics of the sweep-path k-NN layers, filled by profiled ls_encode calls only
    float prof_ms[LS_K_COUNT][16];
    int prof_n[LS_K_COUNT][16];
    std::vector<hipStream_t> prof_streams;
};

// RAII bracket: records an event pair around one launch when profiling is on
struct ProfScope {
    ls_model* m; hipStream_t st; ProfRec r; bool on;
    ProfScope(ls_model* m_, int kind, int layer, hipStream_t st_) : m(m_), st(st_), on(m_ && m_->profiling) {
        if (!on) return;
        auto get = [&]() { hipEvent_t e; if (!m->ev_pool.empty()) { e = m->ev_pool.back(); m->ev_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
        r.kind = kind; r.layer = layer < 16 ? layer : 15; r.a = get(); r.b = get();
        (void)hipEventRecord(r.a, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.b, st);
        m->prof.push_back(r);
        bool seen = false;
        for (auto s : m->prof_streams) seen |= (s == st);
        if (!seen) m->prof_streams.push_back(st);
    }
};
#define PROF(kind, layer, st) ProfScope _ps_##__LINE__(m, kind, layer, st)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------------------------------------ plan
struct EncPlan {
    int L = 0;
    int Ns[LS_MAX_LAYERS], Nd[LS_MAX_LAYERS], Cin[LS_MAX_LAYERS], Co[LS_MAX_LAYERS], ncols[LS_MAX_LAYERS], level[LS_MAX_LAYERS];
    int nlevels = 0, levelN[LS_MAX_LAYERS + 1];
    int NP = 0, Cdp = 0;
    // workspace offsets (bytes)
    size_t o_pts[LS_MAX_LAYERS + 1], o_fps[LS_MAX_LAYERS + 1], o_centroid, o_scale0, o_pro, o_knn, o_knn2, o_knns, o_fA, o_fB, o_msg, o_T, o_TG, o_g, o_G, o_Tc, o_gws, o_xin, o_out, o_rm_msg, o_rm_out[2], o_cs, o_fpsws, fpsws_bytes, total;
};

static int make_plan(const ls_model_desc& d, int B, int N, EncPlan& p) {
    LS_REQUIRE(d.num_layers >= 2 && d.num_layers <= LS_MAX_LAYERS, "encoder: num_layers=%d unsupported", d.num_layers);
    p.L = d.num_layers;
    int cur = N;
    p.nlevels = 0;
    p.levelN[0] = N;
    size_t maxF = 0, maxT = 0, maxTG = 0, maxC = 0, maxKnn = 0, maxKs = 0, maxGws = 0, maxRm = 0, maxCs = 0;
    for (int i = 0; i < p.L; ++i) {
        p.Ns[i] = cur;
        const int f = d.down_factor[i] > 1 ? d.down_factor[i] : 1;
        p.level[i] = -1;
        if (f > 1) {
            LS_REQUIRE(i > 0, "encoder: down-sampling at layer 0 unsupported");
            p.level[i] = p.nlevels;
            p.nlevels++;
            cur = cur / f;
            p.levelN[p.nlevels] = cur;
        }
        p.Nd[i] = cur;
        LS_REQUIRE(cur >= 1, "encoder: N=%d too small for the down-sampling schedule", N);
        // a layer with fewer source points than neighbours would hand -1 padded lists to the gather kernels (and the reference's
        // behaviour there is pytorch3d's padding convention, unpinned): refuse instead of reading out of bounds
        LS_REQUIRE(p.Ns[i] >= d.num_knn, "encoder: layer %d has %d source points < num_knn=%d (N=%d too small for the schedule)", i,
                   p.Ns[i], d.num_knn, N);
        p.Cin[i] = i == 0 ? 1 : d.feat_dim[i - 1];
        p.Co[i] = d.feat_dim[i];
        const bool attn = i >= d.atten_start_layer;
        p.ncols[i] = i == 0 ? 0 : (attn ? 10 : 4) * p.Co[i];
        LS_REQUIRE(p.Co[i] % 16 == 0, "encoder: feat_dim[%d]=%d must be a multiple of 16", i, p.Co[i]);
        if (i > 0) LS_REQUIRE(p.Cin[i] % 32 == 0, "encoder: feat_dim[%d]=%d must be a multiple of 32 (k-NN chunking)", i - 1, p.Cin[i]);
        maxF = std::max(maxF, (size_t)p.Nd[i] * 3 * p.Co[i]);
        {   // combined table over the source points, or split P (source points) + Q (destination points) tables
            const int pc = (attn ? 4 : 2) * p.Co[i], qc = p.ncols[i] - pc;
            const size_t need = (f > 1) ? (size_t)p.Ns[i] * 3 * pc + (size_t)p.Nd[i] * 3 * qc : (size_t)p.Ns[i] * 3 * p.ncols[i];
            maxT = std::max(maxT, need);
        }
        maxTG = std::max(maxTG, (size_t)p.Nd[i] * 3 * 2 * p.Co[i]);
        maxC = std::max(maxC, (size_t)p.Co[i]);
        // row maxima of a layer's message / output: one part per 32 channels + 1, or -- where the table-free path of the 32-point layers may run
        // (edge_fused.hip) -- one per head group; sized per LAYER (round 4 sized Nd and the parts by separate maxima, which only held because
        // the released schedule has wide early layers: ADVICE r4)
        if (i >= d.res_global_start_layer) {
            size_t parts = (size_t)p.Co[i] / 32 + 1;
            if (i > 0 && attn && (p.Cin[i] == 128 || p.Cin[i] == 256)) parts = std::max(parts, (size_t)edge_ft_rowmax_parts(p.Co[i], p.Cin[i]));
            maxRm = std::max(maxRm, (size_t)p.Nd[i] * parts);
            // partial column sums of the message from the attention kernel that wrote it (global_conv): at most one row per 8 points
            maxCs = std::max(maxCs, (size_t)(p.Nd[i] / 8 + 1) * 3 * p.Co[i]);
        }
        // split-K slabs of the under-filled GEMMs (residual global conv: per-point part and per-instance mean part)
        maxGws = std::max(maxGws, std::max(gemm_scratch_floats(B * p.Nd[i] * 3, 2 * p.Co[i], p.Co[i]), gemm_scratch_floats(B * 3, 4 * p.Co[i], p.Co[i])));
        maxKnn = std::max(maxKnn, (size_t)p.Nd[i] * 16);
        maxKs = std::max(maxKs, std::max(knn_scratch_bytes(B, p.Nd[i], p.Ns[i], p.Ns[i], p.Cin[i], true, 0u),
                                        knn_scratch_bytes(B, p.Nd[i], p.Ns[i], p.Ns[i], p.Cin[i], false, 0u)));
    }
    LS_REQUIRE(d.num_knn == 16, "encoder: num_knn=%d unsupported (16)", d.num_knn);
    LS_REQUIRE(p.Ns[p.L - 1] >= 1, "encoder: bad schedule");
    p.NP = cur;
    p.Cdp = (int)align_up((size_t)d.c_dim + 1, 4);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    for (int l = 0; l <= p.nlevels; ++l) {
        p.o_pts[l] = take((size_t)B * p.levelN[l] * 3 * 4);
        p.o_fps[l] = take((size_t)B * p.levelN[l] * 4);
    }
    p.o_centroid = take((size_t)B * 3 * 4);
    p.o_scale0 = take((size_t)B * 4);
    p.o_pro = take(prologue_scratch_floats(B) * 4);
    p.o_knn = take((size_t)B * maxKnn * 4);
    p.o_knn2 = take((size_t)B * maxKnn * 4);
    p.o_knns = take(maxKs + 256);
    p.o_fA = take((size_t)B * maxF * 4);
    p.o_fB = take((size_t)B * maxF * 4);
    p.o_msg = take((size_t)B * maxF * 4);
    p.o_T = take((size_t)B * maxT * 4);
    p.o_TG = take((size_t)B * maxTG * 4);
    p.o_g = take((size_t)B * 3 * maxC * 4);
    p.o_G = take((size_t)B * 3 * 4 * maxC * 4);
    p.o_Tc = take((size_t)B * p.NP * 3 * p.Cdp * 4);
    maxGws = std::max(maxGws, gemm_scratch_floats(B * p.NP * 3, p.Cdp, p.Co[p.L - 1]));
    p.o_gws = take(maxGws * 4 + 256);
    // row maxima of the messages / layer outputs, chained into the GEMMs that read them (gemm.hip, GemmAux)
    p.o_rm_msg = take((size_t)B * maxRm * 3 * 4);      // maxRm = max over the layers of Nd x parts (above)
    p.o_rm_out[0] = take((size_t)B * maxRm * 3 * 4);
    p.o_rm_out[1] = take((size_t)B * maxRm * 3 * 4);
    p.o_cs = take((size_t)B * maxCs * 4);
    // FPS scratch of the encoder's own down-sampling chain (fps.hip: exact bucket pruning for 8 193 .. 65 536 source points; 0 bytes below that).  The
    // levels run one after the other on one stream and share it.  (the chain passed no workspace, so an encode with N > 8192 failed in the
    // MIDDLE of the enqueue, after the side stream had forked.)
    p.fpsws_bytes = 0;
    for (int l = 0; l < p.nlevels; ++l) {
        LS_REQUIRE(p.levelN[l] <= 65536, "encoder: %d points at down-sampling level %d (FPS handles at most 65536)", p.levelN[l], l);
        p.fpsws_bytes = std::max(p.fpsws_bytes, fps_scratch_bytes_per_cloud(p.levelN[l]) * (size_t)B);
    }
    p.o_fpsws = take(p.fpsws_bytes);
    // staging of the captured-graph path: the graph reads x from / writes the codes to FIXED addresses inside the workspace
    p.o_xin = take((size_t)B * 3 * N * 4);
    p.o_out = take((size_t)B * (4 * (size_t)d.c_dim + 4) * 4);
    p.total = off;
    return LS_OK;
}

static int dec_out(const ls_model_desc& d, int l);
// ------------------------------------------------------------------------------------------------ weight row maxima
// rows [W, W + N*K) of a registered matrix with the same K -> their maxima; nullptr = not registered (the GEMM then scans W itself)
static const float* wmax_for(const ls_model* m, const float* W, int N, int K, const void** planes = nullptr) {
    for (const auto& e : m->wreg) {
        if (e.K != K || W < e.base) continue;
        const size_t off = (size_t)(W - e.base);
        if (off % K == 0 && off / K + (size_t)N <= e.rows) {
            if (planes) *planes = e.planes ? e.planes + gemm_w_planes_bytes(off / K, K) : nullptr;
            return e.wmax + off / K;
        }
    }
    return nullptr;
}
static GemmAux aux_w(const ls_model* m, const float* W, int N, int K) {
    GemmAux a;
    a.w_rowmax = wmax_for(m, W, N, K, &a.w_planes);
    return a;
}
struct WSpec { const float* base; size_t rows; int K; };
static int wmax_register(ls_model* m, const std::vector<WSpec>& specs, float** pool, char** planes_pool, hipStream_t st) {
    size_t total = 0, pbytes = 0;
    for (const auto& sp : specs) {
        total += sp.rows;
        if (gemm_w_planes_useful(sp.K)) pbytes += gemm_w_planes_bytes(sp.rows, sp.K);
    }
    if (!total) return LS_OK;
    LS_HIP_CHECK(hipMalloc((void**)pool, total * sizeof(float)));
    if (pbytes) LS_HIP_CHECK(hipMalloc((void**)planes_pool, pbytes));
    size_t off = 0, poff = 0;
    for (const auto& sp : specs) {
        int rc = gemm_rowmax_launch(sp.base, (int)sp.rows, sp.K, sp.K, *pool + off, st);
        if (rc != LS_OK) return rc;
        const char* planes = nullptr;
        if (gemm_w_planes_useful(sp.K)) {
            planes = *planes_pool + poff;
            rc = gemm_presplit_w_launch(sp.base, (int)sp.rows, sp.K, sp.K, *pool + off, *planes_pool + poff, st);
            if (rc != LS_OK) return rc;
            poff += gemm_w_planes_bytes(sp.rows, sp.K);
        }
        m->wreg.push_back({sp.base, sp.rows, sp.K, *pool + off, planes});
        off += sp.rows;
    }
    return LS_OK;
}

// ------------------------------------------------------------------------------------------------ layer pieces
// (shared by ls_encode and the per-operator exports ls_vn_edgeconv_* / ls_vn_lna_f32 / ls_encoder_tail_f32)
static inline int layer_cin(const ls_model_desc& d, int i) { return i == 0 ? 1 : d.feat_dim[i - 1]; }
static inline int layer_ncols(const ls_model_desc& d, int i) { return i == 0 ? 0 : (i >= d.atten_start_layer ? 10 : 4) * d.feat_dim[i]; }
static inline int layer_pcols(const ls_model_desc& d, int i) { return (i >= d.atten_start_layer ? 4 : 2) * d.feat_dim[i]; }
// floats of the per-point table(s) of edge-conv layer i: one combined table over the source points, or (destination points
// selected by FPS rows) a neighbour-side table on the source points + a destination-side table on the selected points
static size_t edge_table_floats(const ls_model_desc& d, int i, int B, int Ns, int Nd, bool rows) {
    if (i == 0) return 0;
    const int nc = layer_ncols(d, i), pc = layer_pcols(d, i);
    return rows ? (size_t)B * 3 * ((size_t)Ns * pc + (size_t)Nd * (nc - pc)) : (size_t)B * Ns * 3 * nc;
}
struct EdgeTables { const float* Tq; int ldp, ldq, NQ, qvr; const float* cur = nullptr; const void* Wq = nullptr; int Cin = 0;
                    const void* Wt = nullptr; const void* Ws = nullptr; };   // Ws != null: slice-major table + staged kernel (edge_staged.hip), Ws = its Q planes   // Wt != null: no table at all (edge_fused.hip); the table area holds that path's scratch   // cur != null: destination side fused into the edge kernel

// where the table(s) of layer i live in T and how the edge kernel reads them (no launch)
static bool edge_fused(const ls_model* m, int i) {
    // LS_OPT_EDGE_FUSE_Q = 0: destination side as table columns.  The fused kernel forms its products from f16 pieces like the default GEMM
    // mode; under LS_GEMM_MODE=bf16x3 / fp32 (any fp32 range, or the fp32-MFMA kernels) the table path is kept so that every product of the layer
    // is formed the same way
    return m->fuse_q && gemm_mode() == 0 && m->wq_planes[i];
}
// table-free path of the 32-point attention layers (edge_fused.hip).  LS_OPT_EDGE_FUSE_T = 0: the table GEMM + edge_attn_v4_kernel pair; like the
// destination-side fusion it forms its products from f16 pieces, so the table path is kept under LS_GEMM_MODE=bf16x3 / fp32.
static bool edge_fused_t(const ls_model* m, int i, int B, int Ns, int Nd, bool has_rows) {
    const ls_model_desc& d = m->d;
    if (!m->fuse_t || gemm_mode() != 0 || !m->wt_planes[i] || i < d.atten_start_layer) return false;
    const int Cin = layer_cin(d, i), Co = d.feat_dim[i];
    return edge_ft_supported(Co, Cin, Ns, Nd, d.atten_head_c, has_rows) &&
           edge_ft_scratch_bytes(B, Ns, Nd, Cin, Co, has_rows) <= edge_table_floats(d, i, B, Ns, Nd, has_rows) * sizeof(float);
}
// LDS-staged path of attention layers 2 - 4 (edge_staged.hip): one 1024-thread workgroup per CU streams its instance's table through the LDS, so
// it pays once B * Nd / PTS workgroups fill the chip; below that the gather kernel (edge_attn_fq_kernel) has the shorter critical path
static bool edge_staged(const ls_model* m, int i, int B, int Ns, int Nd) {
    const ls_model_desc& d = m->d;
    if (!m->edge_staged || !m->wst_q[i] || !edge_fused(m, i) || i < d.atten_start_layer || d.atten_head_c != 16) return false;
    const int Cin = layer_cin(d, i), Co = d.feat_dim[i];
    if (!edge_st_fits(Co, Cin, Ns, Nd)) return false;
    return m->edge_staged >= 2 || (long long)B * (Nd / edge_st_pts(edge_st_variant(Co, Cin))) >= 128;
}
static EdgeTables edge_tables_layout(const ls_model* m, int i, const float* cur, const int32_t* dst_rows, int B, int Ns, int Nd, float* T) {
    const ls_model_desc& d = m->d;
    const int Cin = layer_cin(d, i), nc = layer_ncols(d, i), pc = layer_pcols(d, i);
    if (edge_fused_t(m, i, B, Ns, Nd, dst_rows != nullptr)) { EdgeTables e{nullptr, 0, 0, 0, 0, cur, nullptr, Cin}; e.Wt = m->wt_planes[i]; return e; }
    if (edge_staged(m, i, B, Ns, Nd)) { EdgeTables e{nullptr, pc, 0, 0, 0, cur, nullptr, Cin}; e.Ws = m->wst_q[i]; return e; }
    // (the fused kernel gathers table rows by 32-bit byte offsets: a batch whose neighbour-side table reaches 4 GB takes the table path)
    if (edge_fused(m, i) && edge_attn_fq_fits(B, Ns, pc)) return EdgeTables{nullptr, pc, 0, 0, 0, cur, m->wq_planes[i], Cin};
    if (dst_rows) return EdgeTables{T + (size_t)B * Ns * 3 * pc, pc, nc - pc, Nd, 0};
    return EdgeTables{T + pc, nc, nc, Ns, 1};
}
// the folded VN-Linear contraction of layer i >= 1 (edge.hip header): cur [B,Ns,3,Cin] -> table(s) in T
// a_rowmax (nullable) [B*Ns*3][a_parts]: row maxima of `cur` from the kernel that wrote it (GemmAux)
static int edge_tables(ls_model* m, int i, const float* cur, const int32_t* dst_rows, int B, int Ns, int Nd, float* T, hipStream_t gs,
                       EdgeTables& et, const float* a_rowmax = nullptr, int a_parts = 0) {
    const ls_model_desc& d = m->d;
    const int Cin = layer_cin(d, i), nc = layer_ncols(d, i), pc = layer_pcols(d, i), qc = nc - pc;
    const float* W = m->blob + d.off_edge[i];
    et = edge_tables_layout(m, i, cur, dst_rows, B, Ns, Nd, T);
    PROF(LS_K_GEMM_EDGE, i, gs);
    // 32-point attention layers (edge_fused.hip): no table -- only the operand image of the feature rows (f16 fragment planes) is formed here
    if (et.Wt) return edge_ft_prep_launch(cur, dst_rows, B, Ns, Nd, Cin, d.feat_dim[i], T, gs);
    // attention layers 2 - 4 (fused): only the neighbour-side table; the destination side is computed inside the edge kernel (edge.hip)
    GemmAux ax = aux_w(m, W, nc, Cin);
    ax.a_rowmax = a_rowmax; ax.a_parts = a_parts;
    if (et.Ws) {   // slice-major neighbour-side table from the slice-ordered weight rows (edge_staged.hip)
        GemmAux as = aux_w(m, m->wst_w[i], pc, Cin);
        as.a_rowmax = a_rowmax; as.a_parts = a_parts;
        as.slice_cols = 2 * edge_st_cs(edge_st_variant(d.feat_dim[i], Cin)); as.slice_rows = Ns * 3;
        return gemm_dispatch(cur, Cin, m->wst_w[i], Cin, nullptr, T, pc, B * Ns * 3, pc, Cin, 0, gs, as);
    }
    if (et.cur) return gemm_dispatch(cur, Cin, W, Cin, nullptr, T, pc, B * Ns * 3, pc, Cin, 0, gs, ax);
    if (dst_rows) {
        // down-sampled layer: P table on all source points, Q table only on the FPS-selected destination points
        int rc = gemm_dispatch(cur, Cin, W, Cin, nullptr, T, pc, B * Ns * 3, pc, Cin, 0, gs, ax);
        GemmAux aq = ax;
        if (aq.w_rowmax) aq.w_rowmax += pc;
        if (aq.w_planes) aq.w_planes = static_cast<const char*>(aq.w_planes) + gemm_w_planes_bytes((size_t)pc, Cin);   // rows pc.. of the same matrix
        if (rc == LS_OK) rc = gemm_dispatch_gather(cur, Cin, W + (size_t)pc * Cin, Cin, nullptr, const_cast<float*>(et.Tq), qc, B * Nd * 3, qc, Cin, 0, dst_rows, Nd, Ns, gs, aq);
        return rc;
    }
    return gemm_dispatch(cur, Cin, W, Cin, nullptr, T, nc, B * Ns * 3, nc, Cin, 0, gs, ax);
}
// gather + VN activation + mean-pool | attention of layer i >= 1 over the tables
// rm_out (nullable) [B*Nd*3]: receives max|out[row, :]| when the kernel taken can write it; *rm_written says whether it did
static int edge_apply(ls_model* m, int i, const float* T, const EdgeTables& et, const int32_t* knn, const int32_t* dst_rows, int B, int Nd,
                      int Ns, float* out, hipStream_t st, float* rm_out = nullptr, bool* rm_written = nullptr, int* rm_parts = nullptr,
                      float* cs_out = nullptr, int* cs_rows = nullptr) {
    // cs_out (nullable): receives partial column sums of `out` ([B][*cs_rows][3][Co], *cs_rows = 0: the kernel taken writes none) -- global_conv
    const ls_model_desc& d = m->d;
    const int Co = d.feat_dim[i];
    if (rm_written) *rm_written = false;
    if (rm_parts) *rm_parts = 1;
    if (cs_rows) *cs_rows = 0;
    if (!cs_rows || !m->glob_fuse) cs_out = nullptr;
    if (i >= d.atten_start_layer) {
        PROF(LS_K_EDGE_ATTN, i, st);
        if (et.Wt) {   // rm_out (if any) must hold [B*Nd*3][edge_ft_rowmax_parts] floats: one maximum per head group and row
            if (rm_written) *rm_written = rm_out != nullptr;
            if (rm_parts) *rm_parts = edge_ft_rowmax_parts(Co, et.Cin);
            if (cs_out) *cs_rows = 1;
            return edge_ft_attn_launch(et.Wt, knn, dst_rows != nullptr, B, Ns, Nd, et.Cin, Co, d.neg_slope, const_cast<float*>(T), out, rm_out, st, cs_out);
        }
        if (et.Ws) {
            if (rm_written) *rm_written = rm_out != nullptr;
            return edge_attn_staged_launch(T, et.cur, et.Cin, et.Ws, knn, dst_rows, B, Nd, Ns, Co, d.atten_head_c, d.neg_slope, out, st, rm_out);
        }
        if (et.cur) {
            if (rm_written) *rm_written = rm_out != nullptr;
            const int pw = edge_attn_fq_points_per_wg(Co);
            if (cs_out && pw > 0 && Nd % pw == 0) *cs_rows = Nd / pw; else cs_out = nullptr;
            return edge_attn_fq_launch(T, et.ldp, et.cur, et.Cin, et.Wq, knn, dst_rows, B, Nd, Ns, Co, d.atten_head_c, d.neg_slope, out, st, rm_out, cs_out);
        }
        if (!edge_attn_emits_rowmax(Co, et.ldp, et.ldq)) rm_out = nullptr;
        if (rm_written) *rm_written = rm_out != nullptr;
        const int pw = edge_attn_fq_points_per_wg(Co);   // (the float4-lane kernel of Co = 64 / 128 shares the fused kernel's point map and column sums)
        if (cs_out && pw > 0 && Nd % pw == 0 && et.ldp % 4 == 0 && et.ldq % 4 == 0) *cs_rows = Nd / pw; else cs_out = nullptr;
        return edge_attn_launch(T, et.ldp, et.Tq, et.ldq, et.NQ, et.qvr, knn, dst_rows, B, Nd, Ns, Co, d.atten_head_c, d.neg_slope, out, st, rm_out, cs_out);
    }
    PROF(LS_K_EDGE_POOL, i, st);
    return edge_pool_launch(T, et.ldp, et.Tq, et.ldq, et.NQ, et.qvr, knn, dst_rows, B, Nd, Ns, Co, d.neg_slope, out, st);
}
// residual global conv of layer i (vec_dgcnn_atten.py:222-225): out = VecLNA_G(cat(msg, mean_n msg))
static size_t global_conv_gws_floats(const ls_model_desc& d, int i, int B, int Nd) {
    const int Co = d.feat_dim[i];
    return std::max(gemm_scratch_floats(B * Nd * 3, 2 * Co, Co), gemm_scratch_floats(B * 3, 4 * Co, Co));
}
// rm_msg (nullable) [B*Nd*3]: row maxima of msg from the kernel that wrote it; rm_out (nullable) [B*Nd*3][Co/32]: receives those of `out`
// (*rm_written: whether the path taken wrote them) -- gemm.hip, GemmAux
static int global_conv(ls_model* m, int i, const float* msg, int B, int Nd, float* g, float* G, float* TG, float* gws, float* out, hipStream_t st,
                       const float* rm_msg = nullptr, float* rm_out = nullptr, bool* rm_written = nullptr, int rm_msg_parts = 1, const float* cs = nullptr,
                       int cs_rows = 0) {
    // cs (nullable) [B][cs_rows][3][Co]: partial column sums of msg from the kernel that wrote it (edge_apply): the mean is finished from them
    const ls_model_desc& d = m->d;
    const int Co = d.feat_dim[i];
    const float* Wg = m->blob + d.off_glob[i];
    if (rm_written) *rm_written = false;
    int rc;
    if (m->glob_fuse && gemm_vn_supported(B * Nd * 3, Co, Co)) {
        // per-instance part: mean over the points and its contraction with the W_b / Wd W_b rows in ONE launch (pointwise.hip) ...
        GemmAux ax = aux_w(m, Wg, 2 * Co, Co);
        ax.a_rowmax = rm_msg; ax.a_parts = rm_msg ? rm_msg_parts : 0;
        ax.out_rowmax = rm_out;
        // 64-channel layers with the producer's column sums: the streaming kernel finishes the mean and its contraction itself -- no mean launch
        const bool self_mean = cs && cs_rows > 0 && cs_rows <= 32 && gemm_vn_streams(B * Nd * 3, Co, Co, Co, Nd, ax);
        if (self_mean) { ax.cs = cs; ax.cs_rows = cs_rows; }
        else {
            PROF(LS_K_MEAN, i, st);
            rc = (cs && cs_rows > 0) ? glob_mean_gemv_launch(cs, B, cs_rows, Co, Wg, 2 * Co, 2 * Co, G, 4 * Co, st, Nd)
                                     : glob_mean_gemv_launch(msg, B, Nd, Co, Wg, 2 * Co, 2 * Co, G, 4 * Co, st);
            if (rc != LS_OK) return rc;
        }
        // ... then ONE launch for the per-point contraction + the VN activation (gemm.hip: gemm_vn_kernel / gemm_vn_smallk_kernel / gemm_vn_direct_kernel)
        PROF(LS_K_GEMM_GLOB, i, st);
        if (rm_written) *rm_written = rm_out != nullptr;
        return gemm_vn_dispatch(msg, Co, Wg, Co, G, 4 * Co, out, B * Nd * 3, Co, Co, Nd, 1.0f - d.neg_slope, st, ax);
    }
    { PROF(LS_K_MEAN, i, st); rc = mean_points_launch(msg, B, Nd, Co, g, st); }
    if (rc != LS_OK) return rc;
    {
        PROF(LS_K_GEMM_GLOB, i, st);
        GemmAux ax = aux_w(m, Wg, 2 * Co, Co);
        ax.a_rowmax = rm_msg; ax.a_parts = rm_msg ? rm_msg_parts : 0;
        rc = gemm_dispatch_ws(msg, Co, Wg, Co, nullptr, TG, 2 * Co, B * Nd * 3, 2 * Co, Co, 0, gws, st, ax);
        if (rc == LS_OK) rc = gemm_dispatch_small(g, Co, Wg, Co, nullptr, G, 4 * Co, B * 3, 4 * Co, Co, 0, gws, st);
    }
    if (rc != LS_OK) return rc;
    PROF(LS_K_VN_ACT, i, st);
    return vn_act_rows_launch(TG, 2 * Co, G, 4 * Co, B, Nd, Co, d.neg_slope, out, st);
}
// conv_c + pooling + heads (vec_dgcnn_atten.py:231-250) + the encode epilogue (model_utils.py:182-195)
static int encoder_tail(ls_model* m, const float* cur, int B, int NP, float* Tc, float* gws, const float* centroid, const float* scale0,
                        float* z_so3, float* z_inv, float* s_out, float* t_out, hipStream_t st, const float* rm_cur = nullptr, int rm_parts = 0) {
    const ls_model_desc& d = m->d;
    const int Cl = d.feat_dim[d.num_layers - 1], Cdp = (int)align_up((size_t)d.c_dim + 1, 4);
    const float* W = m->blob;
    int rc;
    { PROF(LS_K_GEMM_TAIL, 0, st);
      GemmAux ax = aux_w(m, W + d.off_convc, Cdp, Cl);
      ax.a_rowmax = rm_cur; ax.a_parts = rm_parts;
      rc = gemm_dispatch_ws(cur, Cl, W + d.off_convc, Cl, nullptr, Tc, Cdp, B * NP * 3, Cdp, Cl, 0, gws, st, ax); }
    if (rc != LS_OK) return rc;
    PROF(LS_K_TAIL, 0, st);
    return tail_launch(Tc, Cdp, B, NP, d.c_dim, W + d.off_inv_t, W + d.off_c_fc0_t, W + d.off_c_misc, d.neg_slope, d.scale_factor,
                       d.center_pred, d.center_pred_scale, centroid, scale0, z_so3, z_inv, s_out, t_out, st);
}

extern "C" {

int ls_version(void) { return LS_ABI_VERSION; }
const char* ls_last_error(void) { return g_err; }

int ls_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return LS_ERR_NO_DEVICE; }
    return n;
}

// ------------------------------------------------------------------------------------------------ leaf exports
size_t ls_knn_workspace_bytes(int B, int Nd, int dst_n, int Ns, int C, int seeded, unsigned flags) {
    if (B <= 0 || Nd <= 0 || Ns <= 0 || dst_n <= 0) return 0;
    return knn_scratch_bytes(B, Nd, dst_n, Ns, C, seeded != 0, flags);
}
int ls_knn_f32(const float* dst, const float* src, const int32_t* dst_rows, const int32_t* seed_idx, int B, int Nd, int dst_n, int Ns,
               int C, int K, unsigned flags, int32_t* idx_out, float* dist_out, void* workspace, size_t workspace_bytes, void* stream) {
    LS_REQUIRE(B > 0 && Nd > 0 && Ns > 0 && dst_n > 0, "knn: empty problem (B=%d Nd=%d Ns=%d)", B, Nd, Ns);
    LS_REQUIRE(K >= 1 && K <= 16, "knn: K=%d unsupported (1..16)", K);
    LS_REQUIRE(C == 1 || C % 32 == 0, "knn: C=%d must be 1 or a multiple of 32", C);
    const size_t sb = knn_scratch_bytes(B, Nd, dst_n, Ns, C, seed_idx != nullptr, flags);
    if (sb > workspace_bytes || (sb && !workspace)) {
        set_error("knn: workspace %zu < required %zu (ls_knn_workspace_bytes)", workspace_bytes, sb);
        return LS_ERR_WORKSPACE;
    }
    return knn_dispatch(dst, src, dst_rows, B, Nd, dst_n, Ns, C, K, flags, idx_out, dist_out, workspace, seed_idx, Nd, 0, (hipStream_t)stream);
}
size_t ls_fps_workspace_bytes(int B, int N, int K) {
    (void)K;
    return (B > 0 && N > 0) ? fps_scratch_bytes_per_cloud(N) * (size_t)B : 0;
}
int ls_fps_f32(const float* pts, const int32_t* lengths, int B, int N, int K, unsigned flags, int32_t* idx_out, float* pts_out,
               void* workspace, size_t workspace_bytes, void* stream) {
    return fps_dispatch(pts, lengths, B, N, K, flags, idx_out, pts_out, workspace, workspace_bytes, (hipStream_t)stream);
}
size_t ls_gemm_workspace_bytes(int M, int N, int K) {
    return (M > 0 && N > 0 && K > 0 && K % 4 == 0) ? gemm_scratch_floats(M, N, K) * sizeof(float) : 0;
}
int ls_gemm_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M, int N, int K,
                int relu, void* workspace, size_t workspace_bytes, void* stream) {
    const size_t sb = (lda % 4 == 0 && ldw % 4 == 0) ? ls_gemm_workspace_bytes(M, N, K) : 0;
    if (!sb) return gemm_dispatch(A, lda, W, ldw, bias, out, ldc, M, N, K, relu, (hipStream_t)stream);
    if (sb > workspace_bytes || !workspace) {   // split-K slabs of an under-filled, long-K problem
        set_error("gemm: workspace %zu < required %zu (ls_gemm_workspace_bytes)", workspace_bytes, sb);
        return LS_ERR_WORKSPACE;
    }
    return gemm_dispatch_ws(A, lda, W, ldw, bias, out, ldc, M, N, K, relu, (float*)workspace, (hipStream_t)stream);
}
int ls_gemm_rowmax_parts(int N) { return N > 0 ? gemm_rowmax_parts(N) : 0; }
int ls_gemm_f32_ex(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M, int N, int K,
                   int relu, const float* a_rowmax, int a_parts, const float* w_rowmax, float* out_rowmax, void* workspace,
                   size_t workspace_bytes, void* stream) {
    LS_REQUIRE(!a_rowmax || a_parts >= 1, "gemm_ex: a_rowmax needs a_parts >= 1");
    GemmAux ax;
    ax.a_rowmax = a_rowmax; ax.a_parts = a_rowmax ? a_parts : 0; ax.w_rowmax = w_rowmax; ax.out_rowmax = out_rowmax;
    const size_t sb = (workspace && lda % 4 == 0 && ldw % 4 == 0) ? ls_gemm_workspace_bytes(M, N, K) : 0;
    LS_REQUIRE(!(sb && out_rowmax), "gemm_ex: a split-K launch (M=%d N=%d K=%d with a workspace) writes no out_rowmax: pass workspace = NULL", M, N, K);
    if (!sb) return gemm_dispatch(A, lda, W, ldw, bias, out, ldc, M, N, K, relu, (hipStream_t)stream, ax);
    if (sb > workspace_bytes) { set_error("gemm_ex: workspace %zu < required %zu (ls_gemm_workspace_bytes)", workspace_bytes, sb); return LS_ERR_WORKSPACE; }
    return gemm_dispatch_ws(A, lda, W, ldw, bias, out, ldc, M, N, K, relu, (float*)workspace, (hipStream_t)stream, ax);
}
int ls_rowmax_f32(const float* X, int rows, int K, int ld, float* out, void* stream) {
    LS_REQUIRE(X && out && rows > 0 && K > 0 && ld >= K, "rowmax: bad argument");
    return gemm_rowmax_launch(X, rows, K, ld, out, (hipStream_t)stream);
}
size_t ls_gemm_w_planes_bytes(int N, int K) { return (N > 0 && K > 0 && gemm_w_planes_useful(K)) ? gemm_w_planes_bytes((size_t)N, K) : 0; }
int ls_gemm_presplit_w_f32(const float* W, int ldw, int N, int K, const float* w_rowmax, void* planes, size_t planes_bytes, void* stream) {
    LS_REQUIRE(W && w_rowmax && planes && N > 0 && K > 0 && ldw >= K && ldw % 4 == 0, "gemm_presplit_w: bad argument");
    LS_REQUIRE(gemm_w_planes_useful(K), "gemm_presplit_w: no kernel reads planes at K = %d (ls_gemm_w_planes_bytes returns 0)", K);
    if (planes_bytes < gemm_w_planes_bytes((size_t)N, K)) { set_error("gemm_presplit_w: planes %zu < required %zu (ls_gemm_w_planes_bytes)", planes_bytes, gemm_w_planes_bytes((size_t)N, K)); return LS_ERR_WORKSPACE; }
    return gemm_presplit_w_launch(W, N, K, ldw, w_rowmax, planes, (hipStream_t)stream);
}
int ls_gemm_f32_planes(const float* A, int lda, const float* W, int ldw, const void* w_planes, const float* bias, float* out, int ldc, int M, int N,
                       int K, int relu, const float* a_rowmax, int a_parts, const float* w_rowmax, float* out_rowmax, void* stream) {
    LS_REQUIRE(w_planes && w_rowmax, "gemm_planes: w_planes and w_rowmax are required (ls_gemm_presplit_w_f32)");
    LS_REQUIRE(!a_rowmax || a_parts >= 1, "gemm_planes: a_rowmax needs a_parts >= 1");
    LS_REQUIRE(gemm_w_planes_useful(K), "gemm_planes: no kernel reads planes at K = %d", K);
    GemmAux ax;
    ax.a_rowmax = a_rowmax; ax.a_parts = a_rowmax ? a_parts : 0; ax.w_rowmax = w_rowmax; ax.out_rowmax = out_rowmax; ax.w_planes = w_planes;
    return gemm_dispatch(A, lda, W, ldw, bias, out, ldc, M, N, K, relu, (hipStream_t)stream, ax);
}
int ls_encode_prologue_f32(const float* x, int B, int N, float* pts_out, float* centroid_out, float* scale0_out, void* stream) {
    LS_REQUIRE(B > 0, "prologue: empty batch");
    return prologue_launch(x, B, N, pts_out, centroid_out, scale0_out, nullptr, (hipStream_t)stream);
}
size_t ls_cosine_scores_workspace_bytes(int n, int m) { return (n > 0 && m > 0) ? (size_t)(n + m) * sizeof(float) : 0; }
int ls_cosine_scores_f32(const float* m0, const float* m1, int n, int m, int D, float* scores, void* workspace, size_t workspace_bytes,
                         void* stream) {
    LS_REQUIRE(n > 0 && m > 0 && D > 0, "cosine_scores: empty problem");
    LS_REQUIRE(m0 && m1 && scores, "cosine_scores: null argument");
    if (!workspace || workspace_bytes < ls_cosine_scores_workspace_bytes(n, m)) {   // the n + m inverse row norms
        set_error("cosine_scores: workspace %zu < required %zu", workspace_bytes, ls_cosine_scores_workspace_bytes(n, m));
        return LS_ERR_WORKSPACE;
    }
    return cosine_scores_launch(m0, m1, n, m, D, (float*)workspace, scores, (hipStream_t)stream);
}
int ls_greedy_match_f32(float* scores, int n, int m, int64_t* matches0, int64_t* matches1, void* stream) {
    LS_REQUIRE(n > 0 && m > 0, "greedy_match: empty problem");
    return greedy_match_launch(scores, n, m, (long long*)matches0, (long long*)matches1, (hipStream_t)stream);
}
int ls_nn_match_f32(const float* scores, int n, int m, int64_t* matches0, int64_t* matches1, void* stream) {
    LS_REQUIRE(n > 0 && m > 0 && scores && matches0 && matches1, "nn_match: empty problem or null argument");
    return assign_launch(scores, n, m, 0, 1.0f, 0.0f, 0, 0.0f, (long long*)matches0, (long long*)matches1, (hipStream_t)stream);
}
int ls_sinkhorn_match_f32(const float* scores, int n, int m, float score_divisor, float alpha, int iters, float match_threshold, int64_t* matches0,
                          int64_t* matches1, void* stream) {
    LS_REQUIRE(n > 0 && m > 0 && scores && matches0 && matches1, "sinkhorn_match: empty problem or null argument");
    LS_REQUIRE(iters >= 0 && score_divisor != 0.0f, "sinkhorn_match: iters=%d score_divisor=%g", iters, (double)score_divisor);
    return assign_launch(scores, n, m, 1, score_divisor, alpha, iters, match_threshold, (long long*)matches0, (long long*)matches1, (hipStream_t)stream);
}
int ls_kabsch_batched_f32(const float* x1, const float* x2, const float* weights, int b, int n, unsigned flags, float* R, float* t,
                          float* res, int32_t* flags_out, void* stream) {
    LS_REQUIRE(b > 0 && n > 0, "kabsch: empty problem");
    LS_REQUIRE(x1 && x2 && R && t, "kabsch: null argument");
    return kabsch_launch(x1, x2, weights, b, n, 0, (flags & LS_FLAG_KABSCH_RAW_WEIGHTS) ? 1 : 0, R, t, res, nullptr, flags_out,
                         (hipStream_t)stream);
}
int ls_kabsch_codes_f32(const float* x1, const float* off1, const int64_t* sel1, const float* x2, const float* off2, const int64_t* sel2, int b,
                        int n, float* R, float* t, float* res, int32_t* flags_out, void* stream) {
    LS_REQUIRE(b > 0 && n > 0, "kabsch_codes: empty problem");
    LS_REQUIRE(x1 && x2 && R && t, "kabsch_codes: null argument");
    return kabsch_launch(x1, x2, nullptr, b, n, 0, 0, R, t, res, nullptr, flags_out, (hipStream_t)stream, off1, off2, (const long long*)sel1,
                         (const long long*)sel2);
}
int ls_kabsch_residual_matrix_f32(const float* src, const float* tgt, int n, int m, int P, float* res, void* stream) {
    LS_REQUIRE(n > 0 && m > 0 && P > 0, "kabsch_residual_matrix: empty problem");
    return kabsch_launch(src, tgt, nullptr, n * m, P, m, 0, nullptr, nullptr, nullptr, res, nullptr, (hipStream_t)stream);
}
size_t ls_icp_workspace_bytes(int b, int n) { return icp_workspace_bytes(b, n); }
int ls_icp_f32(const float* X, const float* Y, const float* R0, const float* T0, int b, int n, int m, int max_iter,
               float rel_rmse_thr, unsigned flags, float* R, float* T, float* rmse, int32_t* iters_out, void* workspace,
               size_t workspace_bytes, void* stream) {
    return icp_run(X, Y, R0, T0, b, n, m, max_iter, rel_rmse_thr, flags, R, T, rmse, iters_out, workspace, workspace_bytes,
                   (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ model
int ls_model_create(const ls_model_desc* desc, const float* blob_host, ls_model_t** out) {
    LS_REQUIRE(desc && blob_host && out, "model_create: null argument");
    LS_REQUIRE(desc->blob_floats > 0, "model_create: empty blob");
    ls_model* m = new ls_model();
    m->d = *desc;
    if (const char* ev = getenv("LS_ENCODE_GRAPH")) m->use_graph = atoi(ev) != 0;
    if (const char* ev = getenv("LS_SDF_BF16X2")) m->sdf_bf16x2 = atoi(ev) != 0;
#ifdef LS_DEV_KNOBS   // only in the variant library scripts/dev/marginal_cost.sh builds (-DLS_DEV_KNOBS): the release library cannot be made to skip work
    if (const char* ev = getenv("LS_FPS_SIDE")) m->fps_side = atoi(ev) != 0;
    if (const char* ev = getenv("LS_DEBUG_LAYERS")) m->debug_layers = atoi(ev);
    if (const char* ev = getenv("LS_SKIP")) {
        const char* names[] = {"knn", "attn", "pool", "l0", "tables", "glob", "fps", "tail", "prologue", "hi32"};   // hi32: the attention of the 32-point layers only (with their operand images)
        for (int i = 0; i < 10; ++i) if (strstr(ev, names[i])) m->skip_mask |= 1u << i;
        if (const char* ea = getenv("LS_SKIP_AFTER")) m->skip_after = atoi(ea);
    }
#endif
    hipError_t e = hipMalloc((void**)&m->blob, (size_t)desc->blob_floats * sizeof(float));
    if (e != hipSuccess) { delete m; set_error("hipMalloc(model blob): %s", hipGetErrorString(e)); return LS_ERR_HIP; }
    e = hipMemcpy(m->blob, blob_host, (size_t)desc->blob_floats * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->side2, hipStreamNonBlocking);
    for (int i = 0; i < LS_MAX_LAYERS && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&m->ev_feat[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_fps[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_tab[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&m->knn_stats, sizeof(unsigned long long) * 2 * LS_MAX_LAYERS);
    if (e == hipSuccess) e = hipMemset(m->knn_stats, 0, sizeof(unsigned long long) * 2 * LS_MAX_LAYERS);
    if (e != hipSuccess) { set_error("model_create: %s", hipGetErrorString(e)); ls_model_destroy(m); return LS_ERR_HIP; }
    for (int i = desc->atten_start_layer; i < desc->num_layers && i >= 1; ++i) {   // table-free 32-point layers (edge_fused.hip): the shapes it has kernels for
        const int Co = desc->feat_dim[i], Cin = layer_cin(*desc, i);
        if (desc->atten_head_c != 16 || !((Cin == 128 && Co % 16 == 0) || (Cin == 256 && Co % 32 == 0))) continue;
        e = hipMalloc(&m->wt_planes[i], edge_ft_w_bytes(Co, Cin));
        if (e != hipSuccess) { set_error("model_create: %s", hipGetErrorString(e)); ls_model_destroy(m); return LS_ERR_HIP; }
        const int rc = edge_ft_presplit_w_launch(m->blob + desc->off_edge[i], Co, Cin, m->wt_planes[i], nullptr);
        if (rc != LS_OK || hipDeviceSynchronize() != hipSuccess) { ls_model_destroy(m); return LS_ERR_HIP; }
    }
    for (int i = desc->atten_start_layer; i < desc->num_layers && i >= 1; ++i) {
        const int Co = desc->feat_dim[i], Cin = layer_cin(*desc, i);
        if (desc->atten_head_c != 16 || !edge_attn_fq_supported(Co, Cin)) continue;
        e = hipMalloc(&m->wq_planes[i], edge_wq_planes_bytes(Co, Cin));
        if (e != hipSuccess) { set_error("model_create: %s", hipGetErrorString(e)); ls_model_destroy(m); return LS_ERR_HIP; }
        const int rc = edge_presplit_wq_launch(m->blob + desc->off_edge[i] + (size_t)layer_pcols(*desc, i) * Cin, Co, Cin, m->wq_planes[i], nullptr);
        if (rc != LS_OK || hipDeviceSynchronize() != hipSuccess) { ls_model_destroy(m); return LS_ERR_HIP; }
    }
    if (const char* ev = getenv("LS_EDGE_STAGED")) m->edge_staged = atoi(ev);
    for (int i = desc->atten_start_layer; i < desc->num_layers && i >= 1; ++i) {   // LDS-staged attention (edge_staged.hip): slice-ordered P rows + Q-block planes
        const int Co = desc->feat_dim[i], Cin = layer_cin(*desc, i);
        if (desc->atten_head_c != 16 || !edge_st_variant(Co, Cin) || !m->wq_planes[i]) continue;
        e = hipMalloc((void**)&m->wst_w[i], (size_t)4 * Co * Cin * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&m->wst_q[i], edge_st_q_bytes(Co, Cin));
        if (e != hipSuccess) { set_error("model_create: %s", hipGetErrorString(e)); ls_model_destroy(m); return LS_ERR_HIP; }
        const int rc = edge_st_prepare_launch(m->blob + desc->off_edge[i], Co, Cin, m->wst_w[i], m->wst_q[i], nullptr);
        if (rc != LS_OK || hipDeviceSynchronize() != hipSuccess) { ls_model_destroy(m); return LS_ERR_HIP; }
    }
    {   // row maxima of every matrix the GEMMs read (GemmAux::w_rowmax)
        std::vector<WSpec> specs;
        const ls_model_desc& d = *desc;
        for (int i = 1; i < d.num_layers; ++i) specs.push_back({m->blob + d.off_edge[i], (size_t)layer_ncols(d, i), layer_cin(d, i)});
        for (int i = 1; i < d.num_layers; ++i) if (m->wst_w[i]) specs.push_back({m->wst_w[i], (size_t)layer_pcols(d, i), layer_cin(d, i)});
        for (int i = d.res_global_start_layer; i < d.num_layers; ++i)
            if (i >= 0) specs.push_back({m->blob + d.off_glob[i], (size_t)4 * d.feat_dim[i], d.feat_dim[i]});
        if (d.num_layers >= 1) specs.push_back({m->blob + d.off_convc, align_up((size_t)d.c_dim + 1, 4), d.feat_dim[d.num_layers - 1]});
        if (d.dec_num_linear >= 3) {
            int kin = d.dec_width;
            for (int l = 1; l < d.dec_num_linear - 1; ++l) {
                const int outw = dec_out(d, l);
                specs.push_back({m->blob + d.off_dec_w[l], (size_t)outw, kin});
                kin = outw;
            }
            specs.push_back({m->blob + d.off_dec_w[d.dec_num_linear - 1], 1, kin});
        }
        const int rc = wmax_register(m, specs, &m->wmax_pool, &m->wplanes_pool, nullptr);
        if (rc != LS_OK || hipDeviceSynchronize() != hipSuccess) { ls_model_destroy(m); return LS_ERR_HIP; }
    }
    *out = m;
    return LS_OK;
}

void ls_model_destroy(ls_model_t* m) {
    if (!m) return;
    if (m->blob) (void)hipFree(m->blob);
    if (m->dec_wt) (void)hipFree(m->dec_wt);
    if (m->wmax_pool) (void)hipFree(m->wmax_pool);
    if (m->wmax_pool_t) (void)hipFree(m->wmax_pool_t);
    if (m->wplanes_pool) (void)hipFree(m->wplanes_pool);
    if (m->wplanes_pool_t) (void)hipFree(m->wplanes_pool_t);
    for (int i = 0; i < LS_MAX_LAYERS; ++i) {
        if (m->wq_planes[i]) (void)hipFree(m->wq_planes[i]);
        if (m->wt_planes[i]) (void)hipFree(m->wt_planes[i]);
        if (m->wst_w[i]) (void)hipFree(m->wst_w[i]);
        if (m->wst_q[i]) (void)hipFree(m->wst_q[i]);
    }
    if (m->side) (void)hipStreamDestroy(m->side);
    if (m->side2) (void)hipStreamDestroy(m->side2);
    for (int i = 0; i < LS_MAX_LAYERS; ++i) {
        if (m->ev_feat[i]) (void)hipEventDestroy(m->ev_feat[i]);
        if (m->ev_fps[i]) (void)hipEventDestroy(m->ev_fps[i]);
        if (m->ev_tab[i]) (void)hipEventDestroy(m->ev_tab[i]);
    }
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    if (m->knn_stats) (void)hipFree(m->knn_stats);
    for (auto& g : m->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    for (auto& r : m->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : m->ev_pool) (void)hipEventDestroy(e);
    delete m;
}

// captured launch sequences are keyed on (workspace, B, N, mode, stream) only: an option that selects kernels must drop them, or a replay would
// run the launch sequence of the OLD setting
static void drop_graphs(ls_model* m) {
    for (auto& g : m->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
    m->graphs.clear();
}

int ls_model_set_option(ls_model_t* m, int option, int value) {
    LS_REQUIRE(m, "model_set_option: null model");
    switch (option) {
        case LS_OPT_SDF_TRAIN_SPLITK: m->train_splitk = value != 0; return LS_OK;
        case LS_OPT_SDF_BF16X2: m->sdf_bf16x2 = value != 0; return LS_OK;
        case LS_OPT_ENCODE_GRAPH: m->use_graph = value != 0; return LS_OK;
        case LS_OPT_EDGE_STAGED: LS_REQUIRE(value >= 0 && value <= 2, "model_set_option: LS_OPT_EDGE_STAGED takes 0, 1 or 2"); m->edge_staged = value; drop_graphs(m); return LS_OK;
        case LS_OPT_EDGE_FUSE_Q: m->fuse_q = value != 0; drop_graphs(m); return LS_OK;
        case LS_OPT_EDGE_FUSE_T: m->fuse_t = value != 0; drop_graphs(m); return LS_OK;
        case LS_OPT_GLOB_FUSE: LS_REQUIRE(value >= 0 && value <= 2, "model_set_option: LS_OPT_GLOB_FUSE takes 0, 1 or 2"); m->glob_fuse = value; drop_graphs(m); return LS_OK;
        case LS_OPT_DEBUG_EDGE: LS_REQUIRE(value >= 0 && value <= 2, "model_set_option: LS_OPT_DEBUG_EDGE takes 0, 1 or 2"); m->debug_edge = value; drop_graphs(m); return LS_OK;
        case LS_OPT_GEMM_OVERLAP: m->overlap_gemm = value != 0; drop_graphs(m); return LS_OK;
        default: set_error("model_set_option: unknown option %d", option); return LS_ERR_INVALID;
    }
}

int ls_model_get_option(const ls_model_t* m, int option, int* value) {
    LS_REQUIRE(m && value, "model_get_option: null argument");
    switch (option) {
        case LS_OPT_SDF_TRAIN_SPLITK: *value = m->train_splitk ? 1 : 0; return LS_OK;
        case LS_OPT_SDF_BF16X2: *value = m->sdf_bf16x2 ? 1 : 0; return LS_OK;
        case LS_OPT_ENCODE_GRAPH: *value = m->use_graph ? 1 : 0; return LS_OK;
        case LS_OPT_EDGE_STAGED: *value = m->edge_staged; return LS_OK;
        case LS_OPT_EDGE_FUSE_Q: *value = m->fuse_q ? 1 : 0; return LS_OK;
        case LS_OPT_EDGE_FUSE_T: *value = m->fuse_t ? 1 : 0; return LS_OK;
        case LS_OPT_GLOB_FUSE: *value = m->glob_fuse; return LS_OK;
        case LS_OPT_DEBUG_EDGE: *value = m->debug_edge; return LS_OK;
        case LS_OPT_GEMM_OVERLAP: *value = m->overlap_gemm ? 1 : 0; return LS_OK;
        default: set_error("model_get_option: unknown option %d", option); return LS_ERR_INVALID;
    }
}

size_t ls_encoder_workspace_bytes(const ls_model_t* m, int B, int N) {
    if (!m) return 0;
    EncPlan p;
    if (make_plan(m->d, B, N, p) != LS_OK) return 0;
    return p.total;
}

}  // extern "C"

// the launch sequence of Shape_Prior.encode on `st` (+ the handle's side streams, forked from and joined back into `st`)
static int encode_enqueue(ls_model_t* m, const EncPlan& p, const float* x, int B, int N, int pre_normalised, unsigned flags, float* z_so3,
                          float* z_inv, float* s_out, float* t_out, int32_t* trace_knn, int32_t* trace_fps, void* workspace, hipStream_t st) {
    const ls_model_desc& d = m->d;
    int rc = LS_OK;
    char* ws = (char*)workspace;
    auto F = [&](size_t o) { return (float*)(ws + o); };
    auto I = [&](size_t o) { return (int32_t*)(ws + o); };
    const float* W = m->blob;
    float* pts0 = F(p.o_pts[0]);
    float* centroid = F(p.o_centroid);
    float* scale0 = F(p.o_scale0);
    const unsigned skip = (m->skip_mask && m->calls++ >= m->skip_after) ? m->skip_mask : 0u;   // dev timing knob (LS_SKIP), see ls_model
    enum { SK_KNN = 1, SK_ATTN = 2, SK_POOL = 4, SK_L0 = 8, SK_TABLES = 16, SK_GLOB = 32, SK_FPS = 64, SK_TAIL = 128, SK_PROLOGUE = 256, SK_HI32 = 512 };

    {
        PROF(LS_K_PROLOGUE, 0, st);
        if (skip & SK_PROLOGUE) rc = LS_OK;
        else if (pre_normalised) rc = transpose_cloud_launch(x, B, N, pts0, st);
        else rc = prologue_launch(x, B, N, pts0, centroid, scale0, F(p.o_pro), st);
    }
    if (rc != LS_OK) return rc;
    // ---- FPS chain on the side stream: depends on xyz only, overlaps with layers 0..first down-sample
    hipStream_t fs = m->fps_side ? m->side : st;
    if (p.nlevels > 0) {
        if (m->fps_side) {
            LS_HIP_CHECK(hipEventRecord(m->ev_fork, st));
            LS_HIP_CHECK(hipStreamWaitEvent(fs, m->ev_fork, 0));
        }
        size_t toff = 0;
        for (int l = 0; l < p.nlevels; ++l) {
            int32_t* idx = trace_fps ? trace_fps + toff : I(p.o_fps[l + 1]);
            toff += (size_t)B * p.levelN[l + 1];
            {
                PROF(LS_K_FPS, l, fs);
                rc = (skip & SK_FPS) ? LS_OK : fps_dispatch(F(p.o_pts[l]), nullptr, B, p.levelN[l], p.levelN[l + 1], flags, idx, F(p.o_pts[l + 1]),
                                                              p.fpsws_bytes ? (void*)(ws + p.o_fpsws) : nullptr, p.fpsws_bytes, fs);
            }
            if (rc != LS_OK) return rc;
            if (m->fps_side) LS_HIP_CHECK(hipEventRecord(m->ev_fps[l], fs));
        }
        if (m->fps_side) LS_HIP_CHECK(hipEventRecord(m->ev_join, fs));
    }

    float* cur = F(p.o_fA);
    float* nxt = F(p.o_fB);
    float* msg = F(p.o_msg);
    float* T = F(p.o_T);
    bool joined = false;
    size_t knn_off = 0, fps_off = 0;
    const float* cur_rm = nullptr;        // row maxima of `cur` ([rows][cur_rm_parts]) when the kernel that wrote it emitted them (GemmAux)
    int cur_rm_parts = 0;
    for (int i = 0; i < p.L; ++i) {
        if (i == m->debug_layers) {
            static bool printed = false;
            if (!printed) {
                printed = true;
                fprintf(stderr, "LS_PLAN knn=%zu knn2=%zu knns=%zu fA=%zu fB=%zu msg=%zu T=%zu TG=%zu g=%zu G=%zu Tc=%zu gws=%zu total=%zu\n",
                        p.o_knn, p.o_knn2, p.o_knns, p.o_fA, p.o_fB, p.o_msg, p.o_T, p.o_TG, p.o_g, p.o_G, p.o_Tc, p.o_gws, p.total);
            }
            if (p.nlevels > 0 && !joined && m->fps_side) LS_HIP_CHECK(hipStreamWaitEvent(st, m->ev_join, 0));
            return LS_OK;
        }
        const int Ns = p.Ns[i], Nd = p.Nd[i], Co = p.Co[i];
        const int32_t* dst_rows = nullptr;
        if (p.level[i] >= 0) {
            if (m->fps_side) LS_HIP_CHECK(hipStreamWaitEvent(st, m->ev_fps[p.level[i]], 0));   // this layer's level only
            joined = p.level[i] == p.nlevels - 1;                                            // the last level joins the whole side chain
            dst_rows = trace_fps ? trace_fps + fps_off : I(p.o_fps[p.level[i] + 1]);
            fps_off += (size_t)B * Nd;
        }
        int32_t* knn = trace_knn ? trace_knn + knn_off : I((i & 1) ? p.o_knn2 : p.o_knn);  // ping-pong: layer i+1 is seeded by layer i
        knn_off += (size_t)B * Nd * 16;
        const bool attn = i >= d.atten_start_layer;
        const bool glob = i >= d.res_global_start_layer;
        float* mp = glob ? msg : nxt;
        bool msg_rm = false;
        int msg_rm_parts = 1, msg_cs_rows = 0;
        if (i == 0) {
            { PROF(LS_K_KNN, i, st); rc = (skip & SK_KNN) ? LS_OK : knn_dispatch(pts0, pts0, nullptr, B, Nd, Ns, Ns, 1, 16, flags, knn, nullptr, ws + p.o_knns, nullptr, 0, 0, st); }
            if (rc != LS_OK) return rc;
            LS_REQUIRE(!attn, "encoder: attention at layer 0 unsupported (atten_start_layer >= 1)");
            { PROF(LS_K_EDGE_L0, i, st); rc = (skip & SK_L0) ? LS_OK : edge_l0_launch(pts0, knn, W + d.off_l0, B, Ns, Co, d.neg_slope, mp, st); }
            if (rc != LS_OK) return rc;
        } else {
            const int Cin = p.Cin[i];
            // fork: the table GEMM(s) depend only on the layer input, like the k-NN -> run them on side2 (matrix cores /
            // HBM writes) concurrently with the VALU-bound k-NN on the caller's stream; join before the edge kernel.
            hipStream_t gs = m->overlap_gemm ? m->side2 : st;
            if (m->overlap_gemm) {
                LS_HIP_CHECK(hipEventRecord(m->ev_feat[i], st));
                LS_HIP_CHECK(hipStreamWaitEvent(gs, m->ev_feat[i], 0));
            }
            EdgeTables et;
            if ((skip & SK_TABLES) || ((skip & SK_HI32) && Nd == 32)) { et = edge_tables_layout(m, i, cur, dst_rows, B, Ns, Nd, T); rc = LS_OK; }
            else rc = edge_tables(m, i, cur, dst_rows, B, Ns, Nd, T, gs, et, cur_rm, cur_rm_parts);
            if (rc != LS_OK) return rc;
            if (m->overlap_gemm) LS_HIP_CHECK(hipEventRecord(m->ev_tab[i], gs));
            {   // (no hints: the fused k-NN kernel derives its thresholds from its own sweep -- knn_mfma.hip)
                PROF(LS_K_KNN, i, st);
                rc = (skip & SK_KNN) ? LS_OK : knn_dispatch(cur, cur, dst_rows, B, Nd, Ns, Ns, Cin, 16, flags, knn, nullptr, ws + p.o_knns, nullptr, 0, 0, st);
            }
            if (rc != LS_OK) return rc;
            if (m->profiling && m->knn_stats && !(skip & SK_KNN) && knn_would_sweep(Cin, Ns, flags)) {
                // (outside the launch's event bracket) how many candidates got a canonical distance: ls_profile_knn_stats
                rc = knn_sweep_stats_launch(ws + p.o_knns, B, Nd, Ns, Ns, m->knn_stats + 2 * i, st);
                if (rc != LS_OK) return rc;
            }
            if (m->overlap_gemm) LS_HIP_CHECK(hipStreamWaitEvent(st, m->ev_tab[i], 0));
            if ((skip & ((i >= d.atten_start_layer) ? SK_ATTN : SK_POOL)) || ((skip & SK_HI32) && Nd == 32)) rc = LS_OK;
            else rc = edge_apply(m, i, T, et, knn, dst_rows, B, Nd, Ns, mp, st, glob ? F(p.o_rm_msg) : nullptr, &msg_rm, &msg_rm_parts,
                                 glob ? F(p.o_cs) : nullptr, &msg_cs_rows);
            if (rc != LS_OK) return rc;
        }
        cur_rm = nullptr; cur_rm_parts = 0;
        if (glob) {
            bool out_rm = false;
            if (skip & SK_GLOB) rc = LS_OK;
            else rc = global_conv(m, i, msg, B, Nd, F(p.o_g), F(p.o_G), F(p.o_TG), F(p.o_gws), nxt, st, msg_rm ? F(p.o_rm_msg) : nullptr,
                             F(p.o_rm_out[i & 1]), &out_rm, msg_rm_parts, F(p.o_cs), msg_cs_rows);
            if (rc != LS_OK) return rc;
            if (out_rm) { cur_rm = F(p.o_rm_out[i & 1]); cur_rm_parts = Co / 32; }
        }
        std::swap(cur, nxt);
    }
    if (p.nlevels > 0 && !joined && m->fps_side) LS_HIP_CHECK(hipStreamWaitEvent(st, m->ev_join, 0));

    // ---- tail
    if (skip & SK_TAIL) return LS_OK;
    return encoder_tail(m, cur, B, p.NP, F(p.o_Tc), F(p.o_gws), pre_normalised ? nullptr : centroid, pre_normalised ? nullptr : scale0,
                        z_so3, z_inv, s_out, t_out, st, cur_rm, cur_rm_parts);
}

namespace ls { int scatter_codes_launch(const float* packed, int B, int c, float* z_so3, float* z_inv, float* s, float* t, hipStream_t st); }

// One encode = ~170 kernel launches, 1.35 ms of host time per 64-instance step in round 1 (the GPU needs 1.6 ms).  The launch
// sequence depends only on (B, N, flags, workspace), so it is CAPTURED once per such key into a hipGraph (stream capture incl. the
// fork / join onto the handle's side streams) and replayed: a call is then one D2D copy of x into the workspace, one graph launch
// and one tiny kernel that scatters the packed codes to the caller's output tensors (the graph itself only ever touches fixed
// addresses inside the caller's workspace, so the caller's x / output pointers may change freely between calls).
static int encode_graphed(ls_model_t* m, const EncPlan& p, const float* x, int B, int N, int pre_normalised, unsigned flags, float* z_so3,
                          float* z_inv, float* s_out, float* t_out, void* workspace, size_t workspace_bytes, hipStream_t st) {
    char* ws = (char*)workspace;
    float* xin = (float*)(ws + p.o_xin);
    float* packed = (float*)(ws + p.o_out);
    const int c = m->d.c_dim;
    float* g_zso3 = packed;
    float* g_zinv = g_zso3 + (size_t)B * c * 3;
    float* g_s = g_zinv + (size_t)B * c;
    float* g_t = g_s + B;
    ls_model::EncGraph* hit = nullptr;
    for (auto& g : m->graphs)
        if (g.ws == workspace && g.ws_bytes == workspace_bytes && g.B == B && g.N == N && g.pre == pre_normalised && g.flags == flags && g.st == st) hit = &g;
    if (!hit) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) { (void)hipGetLastError(); m->graph_broken = true; return 1; }
        const int rc = encode_enqueue(m, p, xin, B, N, pre_normalised, flags, g_zso3, g_zinv, g_s, g_t, nullptr, nullptr, workspace, st);
        e = hipStreamEndCapture(st, &graph);
        if (rc != LS_OK || e != hipSuccess || !graph) { (void)hipGetLastError(); if (graph) (void)hipGraphDestroy(graph); m->graph_broken = true; return rc != LS_OK ? rc : 1; }
        e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); (void)hipGraphDestroy(graph); m->graph_broken = true; return 1; }
        if (m->graphs.size() >= 8) {   // a handle serves a few shapes; drop the oldest beyond that
            (void)hipGraphExecDestroy(m->graphs.front().exec);
            (void)hipGraphDestroy(m->graphs.front().graph);
            m->graphs.erase(m->graphs.begin());
        }
        m->graphs.push_back(ls_model::EncGraph{workspace, workspace_bytes, B, N, pre_normalised, flags, st, graph, exec});
        hit = &m->graphs.back();
    }
    LS_HIP_CHECK(hipMemcpyAsync(xin, x, (size_t)B * 3 * N * sizeof(float), hipMemcpyDeviceToDevice, st));
    LS_HIP_CHECK(hipGraphLaunch(hit->exec, st));
    return scatter_codes_launch(packed, B, c, z_so3, z_inv, s_out, t_out, st);
}

extern "C" {

int ls_encode(ls_model_t* m, const float* x, int B, int N, int pre_normalised, unsigned flags, float* z_so3, float* z_inv,
              float* s_out, float* t_out, int32_t* trace_knn, int32_t* trace_fps, void* workspace, size_t workspace_bytes,
              void* stream) {
    LS_REQUIRE(m && x && z_so3 && z_inv && s_out && t_out && workspace, "encode: null argument");
    LS_REQUIRE(B > 0 && N >= 16, "encode: need B>0, N>=16 (B=%d N=%d)", B, N);
    EncPlan p;
    int rc = make_plan(m->d, B, N, p);
    if (rc != LS_OK) return rc;
    if (workspace_bytes < p.total) { set_error("encode: workspace %zu < required %zu", workspace_bytes, p.total); return LS_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    // the captured path: not while per-launch profiling events or the index traces are requested (tests / bench's profiled pass), not
    // on the legacy NULL stream (it cannot be captured), and not with the debugging knobs
    if (m->use_graph && !m->graph_broken && !m->profiling && !trace_knn && !trace_fps && st != nullptr && m->debug_layers < 0) {
        rc = encode_graphed(m, p, x, B, N, pre_normalised, flags, z_so3, z_inv, s_out, t_out, workspace, workspace_bytes, st);
        if (rc <= 0) return rc;    // 0 = done, < 0 = error; 1 = capture unavailable on this runtime: fall through to the direct path
    }
    return encode_enqueue(m, p, x, B, N, pre_normalised, flags, z_so3, z_inv, s_out, t_out, trace_knn, trace_fps, workspace, st);
}

// ------------------------------------------------------------------------------------------------ per-operator exports
static int check_layer(const ls_model_t* m, int layer, bool want_attn, const char* who) {
    LS_REQUIRE(m, "%s: null model", who);
    LS_REQUIRE(layer >= 0 && layer < m->d.num_layers, "%s: layer %d out of range", who, layer);
    LS_REQUIRE((layer >= m->d.atten_start_layer) == want_attn, "%s: layer %d is %s layer", who, layer, want_attn ? "a mean-pool" : "an attention");
    return LS_OK;
}
size_t ls_vn_edgeconv_workspace_bytes(const ls_model_t* m, int layer, int B, int Ns, int Nd, int has_dst_rows) {
    if (!m || layer < 0 || layer >= m->d.num_layers || B <= 0) return 0;
    return align_up(edge_table_floats(m->d, layer, B, Ns, Nd, has_dst_rows != 0) * sizeof(float) + 16, 256);
}
static int edgeconv_export(ls_model_t* m, int layer, const float* src_f, const int32_t* knn, const int32_t* dst_rows, int B, int Ns, int Nd,
                           float* out, void* workspace, size_t workspace_bytes, hipStream_t st, const char* who) {
    LS_REQUIRE(src_f && knn && out, "%s: null argument", who);
    LS_REQUIRE(B > 0 && Ns >= m->d.num_knn && Nd > 0 && (dst_rows || Nd == Ns), "%s: bad sizes (B=%d Ns=%d Nd=%d)", who, B, Ns, Nd);
    LS_REQUIRE(m->d.num_knn == 16, "%s: num_knn=%d unsupported (16)", who, m->d.num_knn);
    if (layer == 0) {
        LS_REQUIRE(!dst_rows, "%s: layer 0 does not down-sample", who);
        PROF(LS_K_EDGE_L0, 0, st);
        return edge_l0_launch(src_f, knn, m->blob + m->d.off_l0, B, Ns, m->d.feat_dim[0], m->d.neg_slope, out, st);
    }
    const size_t need = ls_vn_edgeconv_workspace_bytes(m, layer, B, Ns, Nd, dst_rows != nullptr);
    if (!workspace || workspace_bytes < need) { set_error("%s: workspace %zu < required %zu", who, workspace_bytes, need); return LS_ERR_WORKSPACE; }
    EdgeTables et;
    int rc = LS_OK;
    if (m->debug_edge == 2) {   // LS_OPT_DEBUG_EDGE: the tables are already in the workspace
        et = edge_tables_layout(m, layer, src_f, dst_rows, B, Ns, Nd, (float*)workspace);
    } else {
        rc = edge_tables(m, layer, src_f, dst_rows, B, Ns, Nd, (float*)workspace, st, et);
    }
    if (rc != LS_OK) return rc;
    if (m->debug_edge == 1) return LS_OK;
    return edge_apply(m, layer, (const float*)workspace, et, knn, dst_rows, B, Nd, Ns, out, st);
}
int ls_vn_edgeconv_pool_f32(ls_model_t* m, int layer, const float* src_f, const int32_t* knn, const int32_t* dst_rows, int B, int Ns, int Nd,
                            float* out, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_layer(m, layer, false, "vn_edgeconv_pool");
    if (rc != LS_OK) return rc;
    return edgeconv_export(m, layer, src_f, knn, dst_rows, B, Ns, Nd, out, workspace, workspace_bytes, (hipStream_t)stream, "vn_edgeconv_pool");
}
int ls_vn_edgeconv_attn_f32(ls_model_t* m, int layer, const float* src_f, const int32_t* knn, const int32_t* dst_rows, int B, int Ns, int Nd,
                            float* out, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_layer(m, layer, true, "vn_edgeconv_attn");
    if (rc != LS_OK) return rc;
    return edgeconv_export(m, layer, src_f, knn, dst_rows, B, Ns, Nd, out, workspace, workspace_bytes, (hipStream_t)stream, "vn_edgeconv_attn");
}

struct LnaPlan { size_t o_g, o_G, o_TG, o_gws, o_rm, total; };
static LnaPlan lna_plan(const ls_model_desc& d, int layer, int B, int N) {
    const size_t Co = (size_t)d.feat_dim[layer];
    LnaPlan q{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    q.o_g = take((size_t)B * 3 * Co * 4);
    q.o_G = take((size_t)B * 3 * 4 * Co * 4);
    q.o_TG = take((size_t)B * N * 3 * 2 * Co * 4);
    q.o_gws = take(global_conv_gws_floats(d, layer, B, N) * 4 + 256);
    q.o_rm = take((size_t)B * N * 3 * 4);
    q.total = off;
    return q;
}
size_t ls_vn_lna_workspace_bytes(const ls_model_t* m, int layer, int B, int N) {
    if (!m || layer < m->d.res_global_start_layer || layer >= m->d.num_layers || B <= 0 || N <= 0) return 0;
    return lna_plan(m->d, layer, B, N).total;
}
int ls_vn_lna_f32(ls_model_t* m, int layer, const float* f, int B, int N, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    LS_REQUIRE(m && f && out && workspace, "vn_lna: null argument");
    LS_REQUIRE(layer >= m->d.res_global_start_layer && layer < m->d.num_layers, "vn_lna: layer %d has no residual global conv", layer);
    LS_REQUIRE(B > 0 && N > 0, "vn_lna: empty problem");
    const LnaPlan q = lna_plan(m->d, layer, B, N);
    if (workspace_bytes < q.total) { set_error("vn_lna: workspace %zu < required %zu", workspace_bytes, q.total); return LS_ERR_WORKSPACE; }
    char* ws = (char*)workspace;
    const float* rm = nullptr;
    if (m->glob_fuse == 2) {   // the message's exact row maxima, as the encoder's attention kernels hand them over (GemmAux::a_rowmax)
        const int Co = m->d.feat_dim[layer];
        const int rc = gemm_rowmax_launch(f, B * N * 3, Co, Co, (float*)(ws + q.o_rm), (hipStream_t)stream);
        if (rc != LS_OK) return rc;
        rm = (const float*)(ws + q.o_rm);
    }
    return global_conv(m, layer, f, B, N, (float*)(ws + q.o_g), (float*)(ws + q.o_G), (float*)(ws + q.o_TG), (float*)(ws + q.o_gws), out,
                       (hipStream_t)stream, rm);
}

size_t ls_encoder_tail_workspace_bytes(const ls_model_t* m, int B, int NP) {
    if (!m || B <= 0 || NP <= 0) return 0;
    const int Cl = m->d.feat_dim[m->d.num_layers - 1], Cdp = (int)align_up((size_t)m->d.c_dim + 1, 4);
    return align_up((size_t)B * NP * 3 * Cdp * 4, 256) + gemm_scratch_floats(B * NP * 3, Cdp, Cl) * 4 + 256;
}
int ls_encoder_tail_f32(ls_model_t* m, const float* f, const float* centroid, const float* scale0, int B, int NP, float* z_so3, float* z_inv,
                        float* s, float* t, void* workspace, size_t workspace_bytes, void* stream) {
    LS_REQUIRE(m && f && z_so3 && z_inv && s && t && workspace, "encoder_tail: null argument");
    LS_REQUIRE(B > 0 && NP > 0, "encoder_tail: empty problem");
    LS_REQUIRE((centroid == nullptr) == (scale0 == nullptr), "encoder_tail: centroid and scale0 go together");
    const size_t need = ls_encoder_tail_workspace_bytes(m, B, NP);
    if (workspace_bytes < need) { set_error("encoder_tail: workspace %zu < required %zu", workspace_bytes, need); return LS_ERR_WORKSPACE; }
    const int Cdp = (int)align_up((size_t)m->d.c_dim + 1, 4);
    float* Tc = (float*)workspace;
    float* gws = (float*)((char*)workspace + align_up((size_t)B * NP * 3 * Cdp * 4, 256));
    return encoder_tail(m, f, B, NP, Tc, gws, centroid, scale0, z_so3, z_inv, s, t, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ SDF decode
static int dec_out(const ls_model_desc& d, int l) {  // padded output width of linear layer l
    const int nl = d.dec_num_linear;
    if (l == nl - 1) return 1;
    if (d.dec_latent_in >= 0 && l + 1 == d.dec_latent_in) return (int)align_up((size_t)(d.dec_width - (2 * d.c_dim + 1)), 4);
    return d.dec_width;
}

// split-K scratch (floats) that covers every GEMM of the decoder forward and backward at `rows` query rows
static int sdf_rm_parts(int w) { return std::max(sdf_affine_rowmax_parts(w), gemm_rowmax_parts(w)); }
static size_t sdf_gemm_scratch(const ls_model_desc& d, long long rows) {
    size_t mx = 0;
    int kin = d.dec_width;
    for (int l = 1; l < d.dec_num_linear - 1; ++l) {
        const int outw = dec_out(d, l);
        mx = std::max(mx, gemm_scratch_floats((int)rows, outw, kin));   // forward: [rows,kin] x [outw,kin]^T
        mx = std::max(mx, gemm_scratch_floats((int)rows, kin, outw));   // backward: [rows,outw] x [kin,outw]^T
        kin = outw;
    }
    return mx;
}

size_t ls_sdf_workspace_bytes(const ls_model_t* m, int B, int M) {
    if (!m || m->d.dec_num_linear <= 0) return 0;
    const size_t w = (size_t)m->d.dec_width;
    size_t b = 0;
    b += 2 * align_up((size_t)B * w * 4 * 4, 256);   // A0, A4
    b += 2 * align_up((size_t)B * w * 4, 256);       // beff0, beff4
    b += 2 * align_up((size_t)B * M * w * 4, 256);   // ping-pong activations
    b += 2 * align_up((size_t)B * M * sdf_rm_parts((int)w) * 4, 256);   // row maxima of the activations (GemmAux)
    return b;
}
// training form: every layer's activations are kept for the backward pass, plus its scratch
size_t ls_sdf_train_workspace_bytes(const ls_model_t* m, int B, int M) {
    if (!m || m->d.dec_num_linear <= 0) return 0;
    const size_t w = (size_t)m->d.dec_width;
    const int nl = m->d.dec_num_linear;
    size_t b = 0;
    b += 2 * align_up((size_t)B * w * 4 * 4, 256) + 2 * align_up((size_t)B * w * 4, 256);      // A0, A4, beff0, beff4
    b += (size_t)(nl - 1) * align_up((size_t)B * M * w * 4, 256);                               // h_0 .. h_{nl-2}
    b += 2 * align_up((size_t)B * M * w * 4, 256);                                              // dz ping-pong
    b += 2 * align_up((size_t)B * w * 4 * 4, 256) + 2 * align_up((size_t)B * w * 4, 256);      // dA0, dA4, dbeff0, dbeff4
    b += align_up((size_t)B * M * 4 * 4, 256);                                                  // dQ
    b += align_up(sdf_gemm_scratch(m->d, (long long)B * M) * 4, 256) + 256;                      // split-K slabs (small M only)
    b += 2 * align_up((size_t)B * M * sdf_rm_parts((int)w) * 4, 256);                           // row maxima of the activations / gradients (GemmAux)
    return b;
}

struct SdfBuffers {
    float *A0, *A4, *b0, *b4;
    float* h[12];       // output of linear layer l (post-ReLU); inference: two ping-pong buffers
    float *dzA, *dzB, *dA0, *dA4, *db0, *db4, *dQ;
    float* gws;         // split-K scratch for the under-filled GEMMs (NULL when the grid is large enough)
    float* rm[2];       // row maxima of the activations / gradients, ping-pong: [rows][sdf_rm_parts(w)] (GemmAux)
};
static SdfBuffers sdf_buffers(const ls_model_desc& d, void* workspace, int B, int M, bool train, bool allow_splitk = true) {
    SdfBuffers sb{};
    const int w = d.dec_width, nl = d.dec_num_linear;
    char* ws = (char*)workspace;
    size_t off = 0;
    auto take = [&](size_t bytes) { float* p = (float*)(ws + off); off = align_up(off + bytes, 256); return p; };
    sb.A0 = take((size_t)B * w * 16);
    sb.A4 = take((size_t)B * w * 16);
    sb.b0 = take((size_t)B * w * 4);
    sb.b4 = take((size_t)B * w * 4);
    if (!train) {
        float* hA = take((size_t)B * M * w * 4);
        float* hB = take((size_t)B * M * w * 4);
        for (int l = 0; l < nl - 1; ++l) sb.h[l] = (l & 1) ? hB : hA;
        // inference never splits K: a query's SDF must not depend on how many other queries share the call (MISE evaluates
        // the same lattice point in calls of very different sizes; split-K changes the fp32 summation order)
        sb.gws = nullptr;
        sb.rm[0] = take((size_t)B * M * sdf_rm_parts(w) * 4);
        sb.rm[1] = take((size_t)B * M * sdf_rm_parts(w) * 4);
        return sb;
    }
    for (int l = 0; l < nl - 1; ++l) sb.h[l] = take((size_t)B * M * w * 4);
    sb.dzA = take((size_t)B * M * w * 4);
    sb.dzB = take((size_t)B * M * w * 4);
    sb.dA0 = take((size_t)B * w * 16);
    sb.dA4 = take((size_t)B * w * 16);
    sb.db0 = take((size_t)B * w * 4);
    sb.db4 = take((size_t)B * w * 4);
    sb.dQ = take((size_t)B * M * 16);
    sb.gws = sdf_gemm_scratch(d, (long long)B * M) ? take(sdf_gemm_scratch(d, (long long)B * M) * 4 + 256) : nullptr;
    if (!allow_splitk) sb.gws = nullptr;
    sb.rm[0] = take((size_t)B * M * sdf_rm_parts(w) * 4);
    sb.rm[1] = take((size_t)B * M * sdf_rm_parts(w) * 4);
    return sb;
}

// row_inst == nullptr: B instances x M rows each; else `M` = total rows R, row r belongs to instance row_inst[r]
static int sdf_forward(ls_model_t* m, const SdfBuffers& sb, const float* query, const float* z_so3, const float* z_inv, const float* s,
                       const float* t, int B, int M, float* sdf, hipStream_t st, const int32_t* row_inst = nullptr) {
    const long long rows = row_inst ? (long long)M : (long long)B * M;
    const ls_model_desc& d = m->d;
    const int w = d.dec_width, L = d.c_dim, nl = d.dec_num_linear, li = d.dec_latent_in;
    const float* W = m->blob;
    int rc;
    {
        PROF(LS_K_SDF_PREP, 0, st);
        rc = sdf_prep_launch(W + d.off_dec_inv_t[0], W + d.off_dec_so3_t[0], W + d.off_dec_len[0], W + d.off_dec_b[0], z_so3,
                             z_inv, B, L, w, sb.A0, sb.b0, st);
        if (rc == LS_OK && li >= 0)
            rc = sdf_prep_launch(W + d.off_dec_inv_t[li], W + d.off_dec_so3_t[li], W + d.off_dec_len[li], W + d.off_dec_b[li],
                                 z_so3, z_inv, B, L, w, sb.A4, sb.b4, st);
    }
    if (rc != LS_OK) return rc;
    // layer 0: pure affine in (q, |q|)
    // operand range of the GEMMs (gemm.hip, GemmAux): every kernel that writes an activation also writes its row maxima, the GEMM that
    // reads it takes them instead of scanning its A rows; the weights carry theirs from ls_model_create
    const bool chain = !m->sdf_bf16x2;
    bool have = chain;   // sb.rm[ri] holds the row maxima of h[l - 1], rm_parts per row
    int ri = 0, rm_parts = sdf_affine_rowmax_parts(w);
    { PROF(LS_K_SDF_AFFINE, 0, st);
      rc = row_inst ? sdf_affine_rows_launch(query, row_inst, s, t, sb.A0, sb.b0, rows, w, w, 0, sb.h[0], st, chain ? sb.rm[0] : nullptr)
                    : sdf_affine_launch(query, s, t, sb.A0, sb.b0, B, M, w, w, 0, sb.h[0], st, chain ? sb.rm[0] : nullptr); }
    if (rc != LS_OK) return rc;
    int kin = w;
    for (int l = 1; l < nl - 1; ++l) {
        const int outw = dec_out(d, l);
        const float* cur = sb.h[l - 1];
        float* nxt = sb.h[l];
        GemmAux ax = aux_w(m, W + d.off_dec_w[l], outw, kin);
        ax.a_rowmax = have ? sb.rm[ri] : nullptr;
        ax.a_parts = have ? rm_parts : 0;
        if (l == li) {
            { PROF(LS_K_GEMM_SDF, l, st);
              rc = (m->sdf_bf16x2 && !sb.gws) ? gemm_dispatch_fast2(cur, w, W + d.off_dec_w[l], kin, nullptr, nxt, w, (int)rows, outw, kin, 0, st)
                                              : gemm_dispatch_ws(cur, w, W + d.off_dec_w[l], kin, nullptr, nxt, w, (int)rows, outw, kin, 0, sb.gws, st, ax); }
            if (rc != LS_OK) return rc;
            PROF(LS_K_SDF_AFFINE, l, st);
            float* rmo = chain ? sb.rm[ri ^ 1] : nullptr;
            rc = row_inst ? sdf_affine_rows_launch(query, row_inst, s, t, sb.A4, sb.b4, rows, w, w, 1, nxt, st, rmo)
                          : sdf_affine_launch(query, s, t, sb.A4, sb.b4, B, M, w, w, 1, nxt, st, rmo);
            if (chain) { ri ^= 1; have = true; rm_parts = sdf_affine_rowmax_parts(w); }
        } else {
            PROF(LS_K_GEMM_SDF, l, st);
            const bool emit = chain && !(sb.gws && gemm_scratch_floats((int)rows, outw, kin) > 0);   // a split-K launch writes no row maxima
            if (emit) ax.out_rowmax = sb.rm[ri ^ 1];
            rc = (m->sdf_bf16x2 && !sb.gws)
                     ? gemm_dispatch_fast2(cur, w, W + d.off_dec_w[l], kin, W + d.off_dec_b[l], nxt, w, (int)rows, outw, kin, 1, st)
                     : gemm_dispatch_ws(cur, w, W + d.off_dec_w[l], kin, W + d.off_dec_b[l], nxt, w, (int)rows, outw, kin, 1, sb.gws, st, ax);
            if (emit) { ri ^= 1; rm_parts = gemm_rowmax_parts(outw); }
            have = emit;
        }
        if (rc != LS_OK) return rc;
        kin = outw;
    }
    PROF(LS_K_SDF_OUT, nl - 1, st);
    return sdf_out_launch(sb.h[nl - 2], w, kin, W + d.off_dec_w[nl - 1], W + d.off_dec_b[nl - 1], rows, sdf, st);
}

int ls_sdf_decode(ls_model_t* m, const float* query, const float* z_so3, const float* z_inv, const float* s, const float* t,
                  int B, int M, float* sdf, void* workspace, size_t workspace_bytes, void* stream) {
    LS_REQUIRE(m && query && z_so3 && z_inv && s && t && sdf && workspace, "sdf_decode: null argument");
    const ls_model_desc& d = m->d;
    LS_REQUIRE(d.dec_num_linear >= 3, "sdf_decode: model has no decoder packed");
    LS_REQUIRE(B > 0 && M > 0, "sdf_decode: empty problem");
    const size_t need = ls_sdf_workspace_bytes(m, B, M);
    if (workspace_bytes < need) { set_error("sdf_decode: workspace %zu < required %zu", workspace_bytes, need); return LS_ERR_WORKSPACE; }
    return sdf_forward(m, sdf_buffers(d, workspace, B, M, false), query, z_so3, z_inv, s, t, B, M, sdf, (hipStream_t)stream);
}

// Ragged batch: R query rows of B instances packed back to back (rows of one instance contiguous), row_inst[r] = instance of row r.
// Workspace: ls_sdf_rows_workspace_bytes(m, B, R).
size_t ls_sdf_rows_workspace_bytes(const ls_model_t* m, int B, long long R) {
    if (!m || m->d.dec_num_linear <= 0) return 0;
    const size_t w = (size_t)m->d.dec_width;
    size_t b = 2 * align_up((size_t)B * w * 4 * 4, 256) + 2 * align_up((size_t)B * w * 4, 256);
    b += 2 * align_up((size_t)R * w * 4, 256);
    b += 2 * align_up((size_t)R * sdf_rm_parts((int)w) * 4, 256);
    return b;
}
int ls_sdf_decode_rows(ls_model_t* m, const float* query, const int32_t* row_inst, const float* z_so3, const float* z_inv, const float* s,
                       const float* t, int B, long long R, float* sdf, void* workspace, size_t workspace_bytes, void* stream) {
    LS_REQUIRE(m && query && row_inst && z_so3 && z_inv && s && t && sdf && workspace, "sdf_decode_rows: null argument");
    const ls_model_desc& d = m->d;
    LS_REQUIRE(d.dec_num_linear >= 3, "sdf_decode_rows: model has no decoder packed");
    LS_REQUIRE(B > 0 && R > 0 && R < (1ll << 31), "sdf_decode_rows: empty or oversized problem");
    const size_t need = ls_sdf_rows_workspace_bytes(m, B, R);
    if (workspace_bytes < need) { set_error("sdf_decode_rows: workspace %zu < required %zu", workspace_bytes, need); return LS_ERR_WORKSPACE; }
    // same buffer layout as the dense form with "M" = R rows shared by all instances
    SdfBuffers sb{};
    const int w = d.dec_width, nl = d.dec_num_linear;
    char* ws = (char*)workspace;
    size_t off = 0;
    auto take = [&](size_t bytes) { float* p = (float*)(ws + off); off = align_up(off + bytes, 256); return p; };
    sb.A0 = take((size_t)B * w * 16); sb.A4 = take((size_t)B * w * 16);
    sb.b0 = take((size_t)B * w * 4); sb.b4 = take((size_t)B * w * 4);
    float* hA = take((size_t)R * w * 4);
    float* hB = take((size_t)R * w * 4);
    for (int l = 0; l < nl - 1; ++l) sb.h[l] = (l & 1) ? hB : hA;
    sb.gws = nullptr;   // batch-invariant: no split-K (see sdf_buffers)
    sb.rm[0] = take((size_t)R * sdf_rm_parts(w) * 4);
    sb.rm[1] = take((size_t)R * sdf_rm_parts(w) * 4);
    return sdf_forward(m, sb, query, z_so3, z_inv, s, t, B, (int)R, sdf, (hipStream_t)stream, row_inst);
}

// forward that keeps every layer's activations in `workspace` for ls_sdf_backward
int ls_sdf_decode_train(ls_model_t* m, const float* query, const float* z_so3, const float* z_inv, const float* s, const float* t,
                        int B, int M, float* sdf, void* workspace, size_t workspace_bytes, void* stream) {
    LS_REQUIRE(m && query && z_so3 && z_inv && s && t && sdf && workspace, "sdf_decode_train: null argument");
    const ls_model_desc& d = m->d;
    LS_REQUIRE(d.dec_num_linear >= 3, "sdf_decode_train: model has no decoder packed");
    LS_REQUIRE(B > 0 && M > 0, "sdf_decode_train: empty problem");
    const size_t need = ls_sdf_train_workspace_bytes(m, B, M);
    if (workspace_bytes < need) { set_error("sdf_decode_train: workspace %zu < required %zu", workspace_bytes, need); return LS_ERR_WORKSPACE; }
    return sdf_forward(m, sdf_buffers(d, workspace, B, M, true, m->train_splitk), query, z_so3, z_inv, s, t, B, M, sdf, (hipStream_t)stream);
}

static int build_dec_wt(ls_model_t* m, hipStream_t st) {   // transposed main weights of layers 1 .. nl-2
    if (m->dec_wt) return LS_OK;
    const ls_model_desc& d = m->d;
    const int nl = d.dec_num_linear, w = d.dec_width;
    size_t total = 0;
    int kin = w;
    for (int l = 1; l < nl - 1; ++l) { const int outw = dec_out(d, l); m->dec_wt_off[l] = total; total += (size_t)kin * outw; kin = outw; }
    LS_HIP_CHECK(hipMalloc((void**)&m->dec_wt, total * sizeof(float)));
    kin = w;
    for (int l = 1; l < nl - 1; ++l) {
        const int outw = dec_out(d, l);
        int rc = transpose_launch(m->blob + d.off_dec_w[l], outw, kin, m->dec_wt + m->dec_wt_off[l], st);   // W [out][kin] -> [kin][out]
        if (rc != LS_OK) return rc;
        kin = outw;
    }
    std::vector<WSpec> specs;
    kin = w;
    for (int l = 1; l < nl - 1; ++l) { const int outw = dec_out(d, l); specs.push_back({m->dec_wt + m->dec_wt_off[l], (size_t)kin, outw}); kin = outw; }
    return wmax_register(m, specs, &m->wmax_pool_t, &m->wplanes_pool_t, st);
}

// Gradients of sum(grad_sdf * sdf) w.r.t. the code and the query points, after ls_sdf_decode_train on the SAME arguments and
// workspace.  grad_query is optional; grad_z_so3 / grad_z_inv may be omitted TOGETHER (pose refinement with a fixed code).
int ls_sdf_backward(ls_model_t* m, const float* query, const float* z_so3, const float* z_inv, const float* s, const float* t, int B,
                    int M, const float* sdf, const float* grad_sdf, void* workspace, size_t workspace_bytes, float* grad_query,
                    float* grad_z_so3, float* grad_z_inv, float* grad_s, float* grad_t, void* stream) {
    LS_REQUIRE(m && query && z_so3 && z_inv && s && t && sdf && grad_sdf && workspace && grad_s && grad_t, "sdf_backward: null argument");
    LS_REQUIRE((grad_z_so3 != nullptr) == (grad_z_inv != nullptr), "sdf_backward: grad_z_so3 and grad_z_inv are given or omitted together");
    const bool need_code = grad_z_so3 != nullptr;   // a pose refinement with a fixed code skips the code-gradient reductions
    const ls_model_desc& d = m->d;
    LS_REQUIRE(d.dec_num_linear >= 3, "sdf_backward: model has no decoder packed");
    LS_REQUIRE(B > 0 && M > 0, "sdf_backward: empty problem");
    const size_t need = ls_sdf_train_workspace_bytes(m, B, M);
    if (workspace_bytes < need) { set_error("sdf_backward: workspace %zu < required %zu", workspace_bytes, need); return LS_ERR_WORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    int rc = build_dec_wt(m, st);
    if (rc != LS_OK) return rc;
    const SdfBuffers sb = sdf_buffers(d, workspace, B, M, true, m->train_splitk);
    const int w = d.dec_width, L = d.c_dim, nl = d.dec_num_linear, li = d.dec_latent_in;
    const float* W = m->blob;
    const long long rows = (long long)B * M;
    // widths: out_w[l] = padded output width of layer l
    int outw[12];
    for (int l = 0; l < nl; ++l) outw[l] = l == 0 ? w : dec_out(d, l);
    float* dz = sb.dzA;     // dz_l: gradient w.r.t. the pre-activation of layer l, row stride w
    float* other = sb.dzB;
    // last layer: dz_{nl-2}
    // operand range (gemm.hip, GemmAux): dz carries its row maxima from kernel to kernel like the activations of the forward pass
    const bool chain = !m->sdf_bf16x2;
    const float* w8max = wmax_for(m, W + d.off_dec_w[nl - 1], 1, outw[nl - 2]);
    bool have = chain && w8max;   // sb.rm[ri] holds the row maxima (or an upper bound) of dz, rm_parts per row
    int ri = 0, rm_parts = 1;
    rc = sdf_out_bwd_launch(grad_sdf, sdf, W + d.off_dec_w[nl - 1], sb.h[nl - 2], w, outw[nl - 2], rows, dz, st, have ? sb.rm[0] : nullptr, w8max);
    if (rc != LS_OK) return rc;
    bool dq_started = false;
    for (int l = nl - 2; l >= 1; --l) {
        const int kin = outw[l - 1];          // input width of layer l (= padded output width of layer l-1)
        if (l == li) {
            rc = sdf_affine_bwd_launch(query, s, t, dz, sb.A4, B, M, w, w, dq_started ? 1 : 0, sb.dA4, sb.db4, sb.dQ, need_code, st);
            if (rc != LS_OK) return rc;
            dq_started = true;
        }
        // dh_{l-1} [rows, kin] = dz_l [rows, out_l] . W_l [out_l][kin]  ==  dz_l . (Wt_l [kin][out_l])^T
        // the ReLU derivative [h_{l-1} > 0] is applied in the GEMM's store when the launch does not split K (h and dh share the row stride w)
        const bool fuse_mask = !sb.gws && kin % 4 == 0;
        GemmAux ax = aux_w(m, m->dec_wt + m->dec_wt_off[l], kin, outw[l]);
        ax.a_rowmax = have ? sb.rm[ri] : nullptr;
        ax.a_parts = have ? rm_parts : 0;
        if (fuse_mask) {
            if (chain) ax.out_rowmax = sb.rm[ri ^ 1];   // after the mask
            rc = gemm_dispatch_masked(dz, w, m->dec_wt + m->dec_wt_off[l], outw[l], other, w, (int)rows, kin, outw[l], sb.h[l - 1], m->sdf_bf16x2 ? 2 : 3, st, ax);
            if (chain) { ri ^= 1; rm_parts = gemm_rowmax_parts(kin); have = true; }
        } else {
            // (an un-masked, possibly split-K launch: its maxima would still bound the masked values, but a split launch writes none)
            const bool emit = chain && !(sb.gws && gemm_scratch_floats((int)rows, kin, outw[l]) > 0);
            if (emit) ax.out_rowmax = sb.rm[ri ^ 1];
            rc = gemm_dispatch_ws(dz, w, m->dec_wt + m->dec_wt_off[l], outw[l], nullptr, other, w, (int)rows, kin, outw[l], 0, sb.gws, st, ax);
            if (rc != LS_OK) return rc;
            if (emit) { ri ^= 1; rm_parts = gemm_rowmax_parts(kin); }
            have = emit;
            rc = relu_mask_launch(other, sb.h[l - 1], rows, kin, w, st);
        }
        if (rc != LS_OK) return rc;
        std::swap(dz, other);
    }
    rc = sdf_affine_bwd_launch(query, s, t, dz, sb.A0, B, M, w, w, dq_started ? 1 : 0, sb.dA0, sb.db0, sb.dQ, need_code, st);
    if (rc != LS_OK) return rc;
    if (need_code) rc = sdf_code_grad_launch(W + d.off_dec_so3_t[0], W + d.off_dec_inv_t[0], sb.dA0, sb.db0, li >= 0 ? W + d.off_dec_so3_t[li] : nullptr,
                              li >= 0 ? W + d.off_dec_inv_t[li] : nullptr, sb.dA4, sb.db4, B, L, w, grad_z_so3, grad_z_inv, st);
    if (rc != LS_OK) return rc;
    return sdf_query_grad_launch(query, s, t, sb.dQ, B, M, grad_query, grad_t, grad_s, st);
}

// ------------------------------------------------------------------------------------------------ profiling
static void prof_collect(ls_model* m) {
    for (auto s : m->prof_streams) (void)hipStreamSynchronize(s);
    for (auto& r : m->prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { m->prof_ms[r.kind][r.layer] += ms; m->prof_n[r.kind][r.layer] += 1; }
        m->ev_pool.push_back(r.a);
        m->ev_pool.push_back(r.b);
    }
    m->prof.clear();
}

int ls_profile_begin(ls_model_t* m) {
    LS_REQUIRE(m, "profile_begin: null model");
    prof_collect(m);
    memset(m->prof_ms, 0, sizeof(m->prof_ms));
    memset(m->prof_n, 0, sizeof(m->prof_n));
    if (m->knn_stats) LS_HIP_CHECK(hipMemset(m->knn_stats, 0, sizeof(unsigned long long) * 2 * LS_MAX_LAYERS));
    m->profiling = true;
    return LS_OK;
}

int ls_profile_knn_stats(ls_model_t* m, unsigned long long* out_host, int max_layers) {
    LS_REQUIRE(m && out_host && max_layers >= 1, "profile_knn_stats: null argument");
    LS_REQUIRE(m->knn_stats, "profile_knn_stats: the handle has no statistics buffer");
    for (auto s : m->prof_streams) (void)hipStreamSynchronize(s);
    unsigned long long h[2 * LS_MAX_LAYERS];
    LS_HIP_CHECK(hipMemcpy(h, m->knn_stats, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < std::min(max_layers, (int)LS_MAX_LAYERS); ++i) { out_host[2 * i] = h[2 * i]; out_host[2 * i + 1] = h[2 * i + 1]; }
    return LS_OK;
}

int ls_profile_end(ls_model_t* m, ls_profile_entry* out, int max_entries, int* n_out) {
    LS_REQUIRE(m && out && n_out, "profile_end: null argument");
    prof_collect(m);
    m->profiling = false;
    int n = 0;
    for (int k = 0; k < LS_K_COUNT; ++k)
        for (int l = 0; l < 16; ++l)
            if (m->prof_n[k][l] > 0 && n < max_entries) {
                out[n].kind = k; out[n].layer = l; out[n].launches = m->prof_n[k][l]; out[n].total_ms = m->prof_ms[k][l];
                ++n;
            }
    *n_out = n;
    return LS_OK;
}

}  // extern "C"
