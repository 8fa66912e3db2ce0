/*
 * livingscenes_hip.h -- C ABI of liblivingscenes_hip.so (MI355X / gfx950 only).
 *
 * The reference (GradientSpaces/LivingScenes) is pure Python/PyTorch: it has no FFI of its own.  The
 * boundary below is what a maintainer binds (ctypes) to replace the ATen /
 * pytorch3d kernels behind the reference's Python call surface for the per-instance inference path.
 * Every entry point cites the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; the caller (PyTorch) owns all
 *     input / output / workspace buffers; the library never frees or retains caller memory and never
 *     allocates device memory behind an operator call: every op that needs scratch takes
 *     (workspace, workspace_bytes) sized by its ls_<op>_workspace_bytes() query.  The only library-owned
 *     device allocations are the packed weights of an ls_model_t (and, on the first ls_sdf_backward of a
 *     model, their transposed copy), freed by ls_model_destroy.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises
 *     the device.  (ls_encoder_forward additionally forks onto a library-owned side stream and joins
 *     back with events; the join happens before it returns control of `stream`'s tail.)
 *   - return value: 0 on success, negative ls_status on failure; text via ls_last_error() (thread-local).
 *   - feature tensors use the library's point-major layout [B, N, 3, C] ("x-major rows": a point is
 *     three rows of C contiguous floats).  The reference layout is [B, C, 3, N]; the Python shim
 *     converts at the API edge only (inputs x[B,3,N], outputs z_so3[B,C,3] keep the reference layout).
 *   - all floating point is IEEE fp32 (the reference runs fp32: configs/room4cates.yaml:15).
 */
#ifndef LIVINGSCENES_HIP_H
#define LIVINGSCENES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    LS_OK = 0,
    LS_ERR_INVALID = -1,     /* bad argument / unsupported shape */
    LS_ERR_HIP = -2,         /* a HIP runtime call failed */
    LS_ERR_WORKSPACE = -3,   /* workspace too small */
    LS_ERR_NO_DEVICE = -4
} ls_status;

/* flags for the integer-exact ops: how `dist += diff*diff` is rounded (oracle/ls_oracle.c header) */
#define LS_FLAG_CONTRACT_FMA 1u /* d = fmaf(diff,diff,d); default (0) = separately rounded mul, add */
#define LS_FLAG_KNN_MFMA_FILTER 2u /* k-NN: accepted for compatibility, no effect: the MFMA sweep kernel (knn_mfma.hip) is the
                                     default wherever it applies (seed_idx given, C == 32 or 64); result is bit-identical */
#define LS_FLAG_KNN_VALU_ONLY 4u   /* k-NN: force the all-VALU kernel (knn.hip) even where the MFMA sweep applies (A/B work) */

#define LS_FLAG_KABSCH_RAW_WEIGHTS 8u /* Kabsch: `weights` are final (the caller applied pose_estimation.py:52-66 itself); no normalisation */

/* per-problem status written to ls_kabsch_batched_f32's flags_out */
#define LS_KABSCH_OK 0         /* covariance of rank >= 2: unique rotation */
#define LS_KABSCH_RANK1 1      /* rank 1: smallest rotation of the least-squares family (torch.svd also succeeds here) */
#define LS_KABSCH_RANK0 2      /* zero covariance: identity rotation, t = mu2 - mu1 */
#define LS_KABSCH_NONFINITE 3  /* NaN / Inf input: identity, zero t -- the reference's SVD-exception branch, flag = True */

#define LS_MAX_LAYERS 8

/* version of this C ABI: bumped whenever a signature or struct layout changes incompatibly (101: double-precision Adam hyper-parameters in
 * ls_adam_group / ls_adam_step_f32 / ls_se3_adam_step_f32, LS_OPT_EDGE_STAGED; 102: LS_OPT_EDGE_FUSE_Q / _T, LS_OPT_GLOB_FUSE, LS_OPT_DEBUG_EDGE; the
 * library reads no development switches from the environment any more).  ls_version() returns the value the LIBRARY was built with; a
 * binding compares it with the header it was written against and refuses a mismatch (livingscenes_amd/_lib.py: load). */
#define LS_ABI_VERSION 103
int ls_version(void);
const char* ls_last_error(void);
/* number of HIP devices visible, or a negative ls_status */
int ls_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Leaf operators
 * ---------------------------------------------------------------------------------------------- */

/* pytorch3d.ops.knn_points(dst, src, K, return_nn=...) as called at
 * lib_shape_prior/core/lib/vec_sim3/vec_dgcnn_atten.py:139-141 (feature-space dynamic graph).
 *   src      [B, Ns, 3, C]   candidate features
 *   dst      [B, Nd', 3, C]  query features; if dst_rows != NULL the query n of instance b is row
 *                            dst_rows[b*Nd + n] of `dst` (which then has Nd' = dst_n rows per instance)
 *   idx_out  [B, Nd, K] int32, ascending (dist, index); -1 padded when Ns < K
 *   dist_out [B, Nd, K] or NULL
 *   seed_idx [B, Nd, 16] int32 or NULL: optional HINTS, any candidate indices per query (-1 = none), e.g. the
 *            neighbour list of the same point in the previous encoder layer.  Their exact distances initialise
 *            the top-K lists (tighter admission threshold, fewer insertions); the RESULT DOES NOT DEPEND on them.
 * Distance = sum_j (a_j-b_j)^2, j = c*3+x ascending, fp32, rounding per `flags`.  K <= 16.
 * C must be 1 or a multiple of 32.  Kernel choice (never visible in the result): C == 1 -> wave-per-query kernel
 * (knn_xyz.hip); seed_idx given and C in {32, 64} -> seed / matrix-core sweep / finish (knn_mfma.hip; the sweep filters with
 * bf16 MFMA on centred rows for Ns <= 2048, fp32 MFMA above); Nd <= 32 -> small-problem kernel; otherwise the tiled all-VALU
 * kernel (knn.hip).  workspace: ls_knn_workspace_bytes(..., seeded = seed_idx != NULL, flags) bytes (0 for some shapes: then
 * workspace may be NULL). */
size_t ls_knn_workspace_bytes(int B, int Nd, int dst_n, int Ns, int C, int seeded, unsigned flags);
int ls_knn_f32(const float* dst, const float* src, const int32_t* dst_rows, const int32_t* seed_idx, int B, int Nd,
               int dst_n, int Ns, int C, int K, unsigned flags, int32_t* idx_out, float* dist_out, void* workspace,
               size_t workspace_bytes, void* stream);

/* pytorch3d.ops.sample_farthest_points(points, K=..., random_start_point=False) as called at
 * vec_dgcnn_atten.py:169, model_utils.py:205, lib_more/more_solver.py:107-108.
 *   pts [B, N, 3]; lengths [B] or NULL; idx_out [B, K] int32 (-1 padded when K > length);
 *   pts_out [B, K, 3] or NULL (gathered points).  Start index 0, running min, first arg-max.
 * workspace: ls_fps_workspace_bytes(B, N, K) bytes -- non-zero only for raw clouds of more than 8 192 points, where it holds the
 * bucketed copy of the clouds the pruned scan works on (fps.hip); NULL / too small there = the un-pruned scan (same result, ~6x slower). */
size_t ls_fps_workspace_bytes(int B, int N, int K);
int ls_fps_f32(const float* pts, const int32_t* lengths, int B, int N, int K, unsigned flags, int32_t* idx_out,
               float* pts_out, void* workspace, size_t workspace_bytes, void* stream);

/* out[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) -- the VecLinear channel contraction
 * (vec_layers.py:121-136, F.linear at :134) on x-major rows, and the DeepSDF linears
 * (lib_shape_prior/core/lib/implicit_func/deepsdf_decoder.py:98-121).  fp32 in, fp32 out; the products run on the f16
 * matrix cores as TWO-PIECE splits (a = h + l, h = f16(a), l = f16(a - h): 2^-22 |a|; three v_mfma_f32_32x32x16_f16 per 16 k into
 * one fp32 accumulator): measured against fp64 this is as accurate as an fp32 FMA chain (the fp32 accumulation error dominates
 * both) and exact on integer-valued operands below 2^22.
 * RANGE: any finite fp32 operands.  Every row of A and every row of W is multiplied by its own exact power of two before the split
 * (row maximum -> [2^14, 2^15)) and the accumulators by the inverse afterwards, so the f16 window follows each row: elements down to
 * 2^-17 of their row's maximum keep the full 22 bits, smaller ones an absolute error of at most 2^-39 of the row maximum (2^-15 of
 * the fp32 rounding of the row's largest term), nothing overflows; rows that hold
 * Inf / NaN give non-finite results in that row only.  A row's result depends on that row of A and on W only -- not on the other rows of
 * the call -- as long as the launch does not split K (split-K: M * N < 192 tiles of 128 x 128 and K >= 128 and a workspace is given;
 * it changes the fp32 summation order with M; ls_gemm_f32_ex with workspace = NULL never splits).
 * LS_GEMM_MODE=bf16x3 in the environment selects three-piece bf16 splits instead (six v_mfma_f32_32x32x16_bf16 per 16 k, ~1.5x the
 * time); LS_GEMM_BF16X3=0 the fp32-MFMA kernel (v_mfma_f32_32x32x2_f32, exact fp32 FMA chains); LS_GEMM_RANGE=0 the split
 * without the row scaling (A/B TIMING only: |a|, |w| < 65 504 required, operands below ~0.1 lose bits).
 * K % 4 == 0, lda/ldw/ldc % 4 == 0; bias may be NULL; relu in {0,1}.  workspace: ls_gemm_workspace_bytes(M, N, K) bytes
 * (split-K slabs of under-filled long-K problems; 0 -> may be NULL). */
size_t ls_gemm_workspace_bytes(int M, int N, int K);
int ls_gemm_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M,
                int N, int K, int relu, void* workspace, size_t workspace_bytes, void* stream);
/* The same GEMM for callers that chain several of them (an MLP): the row maxima the range scaling needs can be handed over instead
 * of being re-derived by a pre-pass over the operands (which costs ~30 % at K >= 128):
 *   a_rowmax [M][a_parts] or NULL: max over the parts >= max|A[row, :]| (any upper bound serves; a factor of two of slack costs one
 *                                  bit at the bottom of the 17-binade window);   w_rowmax [N] or NULL: likewise for W (ls_rowmax_f32 once
 *                                  per weight matrix);
 *   out_rowmax [M][ls_gemm_rowmax_parts(N)] or NULL: receives max|out[row, 64-column block]| -- the next layer's a_rowmax.  Not
 *                                  written by a launch that splits K: pass workspace = NULL when chaining.
 * Results are bit-identical to ls_gemm_f32 whenever the maxima handed over are the exact ones. */
int ls_gemm_rowmax_parts(int N);
int ls_gemm_f32_ex(const float* A, int lda, const float* W, int ldw, const float* bias, float* out, int ldc, int M, int N, int K,
                   int relu, const float* a_rowmax, int a_parts, const float* w_rowmax, float* out_rowmax, void* workspace,
                   size_t workspace_bytes, void* stream);
/* out[r] = max_k |X[r * ld + k]|, k < K  (X [rows, K]) */
int ls_rowmax_f32(const float* X, int rows, int K, int ld, float* out, void* stream);
/* A weight matrix that many GEMMs read can be split ONCE: planes = both f16 pieces of its range-scaled rows (4 * N * K bytes; row n =
 * K / 32 lines of [hi: 32 f16 | lo: 32 f16] of s_n W[n, :], s_n the power of two of w_rowmax[n], which must be the same array later
 * calls pass).  The
 * K >= 512 kernels then stage W with 16-byte copies instead of re-splitting it in every workgroup (decoder shape: -5 % time).
 * ls_gemm_w_planes_bytes = 0 -> no kernel reads planes at this K (K < 512 or K % 32 != 0): use ls_gemm_f32_ex.
 * ls_gemm_f32_planes: ls_gemm_f32_ex with the planes (W itself is still passed: same result, bit for bit, as without them);
 * never splits K.  F.linear of the reference (vec_layers.py:134, deepsdf_decoder.py:98-121) with a fixed weight. */
size_t ls_gemm_w_planes_bytes(int N, int K);
int ls_gemm_presplit_w_f32(const float* W, int ldw, int N, int K, const float* w_rowmax, void* planes, size_t planes_bytes, void* stream);
int ls_gemm_f32_planes(const float* A, int lda, const float* W, int ldw, const void* w_planes, const float* bias, float* out, int ldc,
                       int M, int N, int K, int relu, const float* a_rowmax, int a_parts, const float* w_rowmax, float* out_rowmax,
                       void* stream);

/* Shape_Prior.encode prologue, model_utils.py:166-177: centroid, scale_0 = mean of the 5 largest
 * entries of the N x N distance matrix, normalised cloud.
 *   x [B,3,N] (reference layout) -> pts_out [B,N,3], centroid_out [B,3], scale0_out [B] */
int ls_encode_prologue_f32(const float* x, int B, int N, float* pts_out, float* centroid_out, float* scale0_out,
                           void* stream);

/* sequential_matcher's score matrix, lib_more/matcher_new.py:110-120:
 * S = normalize(m0) @ normalize(m1)^T.   m0 [n,D], m1 [m,D] -> S [n,m];  workspace: (n + m) floats (the inverse row norms) */
size_t ls_cosine_scores_workspace_bytes(int n, int m);
int ls_cosine_scores_f32(const float* m0, const float* m1, int n, int m, int D, float* scores, void* workspace,
                         size_t workspace_bytes, void* stream);

/* The greedy assignment loop shared by sequential / sim3_seq / eq_seq matchers
 * (matcher_new.py:121-136, :166-181, :212-227): repeat min(n,m) times { S /= (max(S)+1e-5); take the
 * first row-major arg-max; record; delete row and column }.  `scores` [n,m] is DESTROYED.
 * matches0 [n], matches1 [m] int64, -1 = unmatched. */
int ls_greedy_match_f32(float* scores, int n, int m, int64_t* matches0, int64_t* matches1, void* stream);

/* nn_matcher after its score matrix, lib_more/matcher_new.py:89-98 (find_nn :73-83 without thresholds, mutual_check :100-105 twice):
 * matches0[i] = argmax_j S[i][j] if argmax_i S[i][that j] == i else -1; matches1 = mutual_check(argmax over rows, matches0).
 * First maximum on ties.  scores [n,m] (read only) -> matches0 [n], matches1 [m] int64.  One launch, one workgroup (LDS: about 190 x 190 at most). */
int ls_nn_match_f32(const float* scores, int n, int m, int64_t* matches0, int64_t* matches1, void* stream);

/* sinkhorn_matcher after its score matrix, matcher_new.py:49-71: couplings = [[S / score_divisor, alpha], [alpha, alpha]], `iters` log-space Sinkhorn
 * iterations (log_optimal_transport :20-40 with log_sinkhorn_iterations above it), arg-maxes of the inner block of Z, mutual checks, and
 * exp(max0) > match_threshold (:58-66).  The reference calls it with score_divisor = desc_dim ** 0.5, alpha = 1, iters = 100, threshold 0.
 * scores [n,m] (read only) -> matches0 [n], matches1 [m] int64, -1 = unmatched.  One launch, one workgroup. */
int ls_sinkhorn_match_f32(const float* scores, int n, int m, float score_divisor, float alpha, int iters, float match_threshold,
                          int64_t* matches0, int64_t* matches1, void* stream);

/* kabsch_transformation_estimation(x1, x2, weights, normalize_w=True, eps=1e-7),
 * lib_more/pose_estimation.py:29-102 (+ transformation_residuals :105-121).
 *   x1,x2 [b,n,3]; weights [b,n] or NULL (ones); flags: 0 = weights are normalised as :52-54 does, LS_FLAG_KABSCH_RAW_WEIGHTS =
 *   used as given (normalize_w=False, or the caller applied best_k / w_threshold of :58-66 to the normalised weights);
 *   R [b,3,3], t [b,3] (the reference's [b,3,1]), res [b,n] or NULL, flags_out [b] int32 or NULL = LS_KABSCH_* status
 *   (only LS_KABSCH_NONFINITE corresponds to the reference's SVD-failure branch at :79-88). */
int ls_kabsch_batched_f32(const float* x1, const float* x2, const float* weights, int b, int n, unsigned flags, float* R,
                          float* t, float* res, int32_t* flags_out, void* stream);
/* The same fit on More_Solver's pseudo-points (more_solver.py:114-116: kabsch(code1.z_so3 + code1.t, code2.z_so3 + code2.t)) with the
 * element-wise work around it inside the launch: set i of side k is x_k[i] + off_k[i] (off_k [*,3] or NULL), problem p reads set
 * sel_k[p] of side k (sel_k [b] int64 or NULL = p; a negative entry -- an unmatched row of matches0 -- reads set 0, as
 * matches0.clamp(min=0) does).  Unit weights.  Bit-identical to ls_kabsch_batched_f32 on the materialised sums. */
int ls_kabsch_codes_f32(const float* x1, const float* off1, const int64_t* sel1, const float* x2, const float* off2, const int64_t* sel2, int b,
                        int n, float* R, float* t, float* res, int32_t* flags_out, void* stream);

/* mean Kabsch residual of every (src i, tgt j) pair of equivariant codes: res_mat of
 * matcher_new.py:150-156 / :196-202.   src [n,P,3], tgt [m,P,3] -> res [n,m] */
int ls_kabsch_residual_matrix_f32(const float* src, const float* tgt, int n, int m, int P, float* res, void* stream);

/* pytorch3d.ops.iterative_closest_point(X, Y, init_transform=SimilarityTransform(R0,T0,1)) with default
 * arguments (100 iterations, relative_rmse_thr 1e-6), as called at lib_more/more_solver.py:182-187.
 * Row-vector convention Xt = X R + T.  X [b,n,3], Y [b,m,3], R0 [b,3,3], T0 [b,3] ->
 * R [b,3,3], T [b,3], rmse [b], iters_out [1] int32.  workspace >= ls_icp_workspace_bytes(b,n). */
size_t ls_icp_workspace_bytes(int b, int n);
int ls_icp_f32(const float* X, const float* Y, const float* R0, const float* T0, int b, int n, int m, int max_iter,
               float rel_rmse_thr, unsigned flags, float* R, float* T, float* rmse, int32_t* iters_out,
               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Model handle (packed device-resident weights: the library's only device allocation)
 * ---------------------------------------------------------------------------------------------- */
typedef struct ls_model ls_model_t;

typedef struct {
    /* encoder: VecDGCNN_att.__init__ arguments, vec_dgcnn_atten.py:23-45 */
    int32_t num_layers;
    int32_t feat_dim[LS_MAX_LAYERS];
    int32_t down_factor[LS_MAX_LAYERS]; /* 1 = no down-sampling before this layer */
    int32_t atten_start_layer;
    int32_t atten_head_c;
    int32_t res_global_start_layer;     /* >= num_layers disables the residual global conv */
    int32_t num_knn;
    int32_t c_dim;
    int32_t center_pred;
    int32_t center_pred_scale;
    float scale_factor;
    float neg_slope;
    /* decoder: DeepSDF_Decoder (deepsdf_decoder.py:12-57), decoder_type "inner_deepsdf" */
    int32_t dec_num_linear;             /* number of linear layers (9), 0 = no decoder packed */
    int32_t dec_width;                  /* hidden width (768) */
    int32_t dec_latent_in;              /* layer that re-concatenates the input (4), -1 = none */
    /* offsets (in floats) into the packed blob; layouts documented in livingscenes_amd/packing.py */
    int64_t off_l0;                     /* [6][feat_dim[0]] layer-0 folded rows */
    int64_t off_edge[LS_MAX_LAYERS];    /* layer i>=1: [ncols_i][feat_dim[i-1]] folded per-point weights */
    int64_t off_glob[LS_MAX_LAYERS];    /* layer i>=g0: [4*C][C] = {Wa;Wd*Wa;Wb;Wd*Wb} */
    int64_t off_convc;                  /* [c_dim+1 (padded to 4)][feat_dim[-1]] */
    int64_t off_inv_t;                  /* fc_inv^T [c_dim][c_dim] */
    int64_t off_c_fc0_t;                /* fc_center.fc0 {lin;dir*lin}^T : [c_dim][2*h] */
    int64_t off_c_misc;                 /* lin1 [h], shortcut [c_dim], act2 dir scalar [1] */
    int64_t off_dec_w[12];              /* decoder layer l: folded main weight [out_l][kin_l] */
    int64_t off_dec_b[12];              /* decoder bias [out_l] */
    int64_t off_dec_inv_t[12];          /* layers fed by the code (0 and latent_in): Wa^T [latent][out] */
    int64_t off_dec_so3_t[12];          /*   "    Wb^T [latent][out] */
    int64_t off_dec_len[12];            /*   "    w_len [out] */
    int64_t blob_floats;
} ls_model_desc;

int ls_model_create(const ls_model_desc* desc_host, const float* blob_host, ls_model_t** out);
void ls_model_destroy(ls_model_t* m);
/* per-handle switches (defaults in brackets) */
#define LS_OPT_SDF_TRAIN_SPLITK 1 /* [1] split-K in the GEMMs of ls_sdf_decode_train / ls_sdf_backward when the problem under-fills the chip
                                     (one 1024-point cloud); 0 = never: a row's values do not depend on the size of the call (batched == per pair) */
#define LS_OPT_SDF_BF16X2 2       /* [0, or LS_SDF_BF16X2 in the environment] decoder products as two-piece bf16 splits (2^-16 per product, ~1.5x) */
#define LS_OPT_ENCODE_GRAPH 3     /* [0, or LS_ENCODE_GRAPH in the environment] ls_encode replays a captured hipGraph of its ~170 launches (one per
                                     (workspace, B, N, flags, stream); needs a non-NULL stream; profiled / traced calls always enqueue directly).
                                     Off by default: on ROCm 7.2 the replay measured slower than direct enqueue (22.2k vs 29.6k obj/s, one step in flight) */
#define LS_OPT_EDGE_STAGED 4      /* [0, or LS_EDGE_STAGED in the environment] attention layers 2 - 4 (vec_dgcnn_atten.py:205-219) with LDS-staged neighbour tiles
                                     (edge_staged.hip): 0 = never (row gathers through L1: edge_attn_fq_kernel), 1 = when one workgroup per CU fills the chip
                                     (B * Nd / 128 >= 128), 2 = whenever the layer shape fits.  The two kernels agree to ~1e-6 of the tensor maximum.  Off by
                                     default: measured at B = 64 (round 5) 132 / 104 / 105 us against 96 / 107 / 82 us at layers 2 / 3 / 4 */
#define LS_OPT_EDGE_FUSE_Q 5      /* [1] attention layers 2 - 4: the destination-side column groups computed inside the edge kernel (edge.hip: edge_attn_fq_kernel);
                                     0 = as table columns written by the table GEMM (the general path; also taken under LS_GEMM_MODE=bf16x3 / fp32) */
#define LS_OPT_EDGE_FUSE_T 6      /* [1] the 32-point attention layers (released layers 5, 6) without a table (edge_fused.hip); 0 = table GEMM + edge kernel */
#define LS_OPT_GLOB_FUSE 7        /* [1] residual global conv as one mean + GEMV launch and one GEMM + VN-activation launch (gemm.hip: gemm_vn_kernel);
                                     0 = mean, GEMM -> table, per-instance GEMM, VN activation as separate launches (same products in the same order);
                                     2 = as 1, and the operator export ls_vn_lna_f32 first takes the exact row maxima of its input, as the encoder's attention
                                     kernels hand them to this conv (layers of 64 channels then run gemm_vn_direct_kernel: bit-identical, tests/) */
#define LS_OPT_DEBUG_EDGE 8       /* [0] ls_vn_edgeconv_*: 1 = run the table GEMM only, 2 = run the edge kernel only on the tables already in the workspace
                                     (per-operator counter passes: scripts/pmc_ops.py); 0 = both */
#define LS_OPT_GEMM_OVERLAP 9     /* [1] ls_encode runs a layer's table GEMM on a side stream beside its k-NN build (+5 % with one call in flight); 0 = on the caller's
                                     stream (better once several ls_encode calls overlap on different streams: bench.py turns it off, -3.5 % otherwise) */
int ls_model_set_option(ls_model_t* m, int option, int value);
/* the handle's CURRENT value of an option (what ls_model_create read from the environment, or the last ls_model_set_option): the only
 * way a caller can change an option temporarily and put back exactly what was there */
int ls_model_get_option(const ls_model_t* m, int option, int* value);

/* ------------------------------------------------------------------------------------------------
 * Composite hot path
 * ---------------------------------------------------------------------------------------------- */

/* Shape_Prior.encode(x), model_utils.py:165-197, = prologue + VecDGCNN_att.forward
 * (vec_dgcnn_atten.py:177-252) + epilogue (t = center + centroid, s = scale_0 * pred_scale).
 *   x [B,3,N] -> z_so3 [B,c_dim,3], z_inv [B,c_dim], s [B], t [B,3]
 * Host cost: ~170 launches (0.7 ms of host time per 64-instance call).  With LS_OPT_ENCODE_GRAPH the sequence is captured into a hipGraph
 * on first use per (workspace, B, N, flags, stream) and replayed afterwards (x and the outputs may move between calls).
 * trace_knn / trace_fps (nullable): per-layer k-NN / FPS indices for the parity tests, packed
 * back to back: knn layer i at offset sum_{j<i} B*Nd_j*K (int32), fps level l at sum B*Nd_l.
 * If pre_normalised != 0, x is taken as already centred/scaled (prologue skipped, centroid 0, scale_0 1):
 * this is VecDGCNN_att.forward alone and the outputs are (center, scale, z_so3, z_inv). */
size_t ls_encoder_workspace_bytes(const ls_model_t* m, int B, int N);
int ls_encode(ls_model_t* m, const float* x, int B, int N, int pre_normalised, unsigned flags, float* z_so3,
              float* z_inv, float* s, float* t, int32_t* trace_knn, int32_t* trace_fps, void* workspace,
              size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The encoder's layer operators on their own (the same code ls_encode runs), for callers that drive the layer loop
 * themselves and for isolated parity tests.  Features are [B, N, 3, C] rows; `layer` selects the weights.
 * ---------------------------------------------------------------------------------------------- */

/* One edge-conv layer of VecDGCNN_att.forward WITHOUT its residual global conv: get_graph_feature (vec_dgcnn_atten.py:124-161:
 * cat(nbr - ctr, ctr); + the cross-product channel at layer 0) -> V_list[layer] VecLNA (vec_layers.py:523-534) -> mean over the
 * 16 neighbours (:202-204, layers < atten_start_layer: ls_vn_edgeconv_pool_f32), or K / Q / V VecLNAs + channel_equi_vec_normalize
 * (vec_layers.py:24-31) + per-head soft-max attention (:205-219, ls_vn_edgeconv_attn_f32).
 *   src_f [B,Ns,3,C_in] (layer 0: the normalised cloud [B,Ns,3]); knn [B,Nd,16] int32 indices into the Ns source points;
 *   dst_rows [B,Nd] int32 or NULL: the FPS selection when the layer down-samples (destination point n = source row dst_rows[n]),
 *   NULL requires Nd == Ns;  out [B,Nd,3,C_out].  workspace holds the folded per-point tables (edge.hip header). */
size_t ls_vn_edgeconv_workspace_bytes(const ls_model_t* m, int layer, int B, int Ns, int Nd, int has_dst_rows);
int ls_vn_edgeconv_pool_f32(ls_model_t* m, int layer, const float* src_f, const int32_t* knn, const int32_t* dst_rows, int B,
                            int Ns, int Nd, float* out, void* workspace, size_t workspace_bytes, void* stream);
int ls_vn_edgeconv_attn_f32(ls_model_t* m, int layer, const float* src_f, const int32_t* knn, const int32_t* dst_rows, int B,
                            int Ns, int Nd, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* The residual global conv of layer `layer` >= res_global_start_layer (vec_dgcnn_atten.py:222-225):
 * out = VecLinearNormalizeActivate_G(cat(f, mean_n f))  (vec_layers.py:523-534 = VecLinear :121-136 + VecActivation :241-268).
 *   f [B,N,3,C] -> out [B,N,3,C] */
size_t ls_vn_lna_workspace_bytes(const ls_model_t* m, int layer, int B, int N);
int ls_vn_lna_f32(ls_model_t* m, int layer, const float* f, int B, int N, float* out, void* workspace, size_t workspace_bytes,
                  void* stream);

/* The encoder's heads (vec_dgcnn_atten.py:231-250): conv_c VecLNA (shared direction) -> mean over the NP points -> z_so3 =
 * channel_equi_vec_normalize, scale = mean_c |x_c| * scale_factor, z_inv = <cevn(fc_inv x), z_so3>, center = VecResBlock
 * (vec_layers.py:631-651) * scale_factor; then Shape_Prior.encode's epilogue (model_utils.py:182-185) when centroid / scale0
 * are given: t = center + centroid, s = scale0 * scale (both NULL: t = center, s = scale).
 *   f [B,NP,3,C_last] -> z_so3 [B,c_dim,3], z_inv [B,c_dim], s [B], t [B,3] */
size_t ls_encoder_tail_workspace_bytes(const ls_model_t* m, int B, int NP);
int ls_encoder_tail_f32(ls_model_t* m, const float* f, const float* centroid, const float* scale0, int B, int NP, float* z_so3,
                        float* z_inv, float* s, float* t, void* workspace, size_t workspace_bytes, void* stream);

/* FieldWrapper.forward(query, None, code, return_sdf=True), decoder_type "inner_deepsdf":
 * model_utils.py:230-263 + DeepSDF_Decoder.forward, deepsdf_decoder.py:78-123.
 *   query [B,M,3], z_so3 [B,c,3], z_inv [B,c], s [B], t [B,3] -> sdf [B,M] */
size_t ls_sdf_workspace_bytes(const ls_model_t* m, int B, int M);
int ls_sdf_decode(ls_model_t* m, const float* query, const float* z_so3, const float* z_inv, const float* s,
                  const float* t, int B, int M, float* sdf, void* workspace, size_t workspace_bytes, void* stream);

/* Ragged form of ls_sdf_decode: R query rows of B instances packed back to back (rows of one instance contiguous), row_inst [R]
 * int32 = instance of each row.  Used by the batched MISE rounds, where every instance asks for a different number of points
 * (the reference evaluates one instance at a time, occnet_utils/mesh_extractor2.py:116-131).  sdf [R]. */
size_t ls_sdf_rows_workspace_bytes(const ls_model_t* m, int B, long long R);
int ls_sdf_decode_rows(ls_model_t* m, const float* query, const int32_t* row_inst, const float* z_so3, const float* z_inv,
                       const float* s, const float* t, int B, long long R, float* sdf, void* workspace, size_t workspace_bytes,
                       void* stream);

/* SURVEY.md 8 (f-1), the autograd half: what `loss.backward()` computes in More_Solver._optimize_code
 * (lib_more/more_solver.py:191-228) through FieldWrapper.forward (model_utils.py:230-263) and DeepSDF_Decoder.forward
 * (deepsdf_decoder.py:78-123).  ls_sdf_decode_train = ls_sdf_decode keeping every layer's activations in `workspace`
 * (ls_sdf_train_workspace_bytes); ls_sdf_backward, called with the SAME arguments and workspace, returns the gradients of
 * sum(grad_sdf * sdf):  grad_query [B,M,3] (nullable), grad_z_so3 [B,c,3], grad_z_inv [B,c] (nullable TOGETHER: a pose refinement
 * with a fixed code -- more_solver.py:137-173 -- skips the code-gradient reductions), grad_s [B], grad_t [B,3]. */
size_t ls_sdf_train_workspace_bytes(const ls_model_t* m, int B, int M);
int ls_sdf_decode_train(ls_model_t* m, const float* query, const float* z_so3, const float* z_inv, const float* s,
                        const float* t, int B, int M, float* sdf, void* workspace, size_t workspace_bytes, void* stream);
int ls_sdf_backward(ls_model_t* m, const float* query, const float* z_so3, const float* z_inv, const float* s, const float* t,
                    int B, int M, const float* sdf, const float* grad_sdf, void* workspace, size_t workspace_bytes,
                    float* grad_query, float* grad_z_so3, float* grad_z_inv, float* grad_s, float* grad_t, void* stream);

/* SURVEY.md 8 (f-1), registration half: the log-domain softmin of entropic OT with cost |x-y|^2/2, the primitive of
 * geomloss.SamplesLoss('sinkhorn', p=2) as used at lib_more/more_solver.py:146,158 (geomloss itself is absent: the
 * epsilon-scaling loop around this primitive, livingscenes_amd/sinkhorn.py, is restated from memory -- parity unpinned):
 *   out[i] = -eps log sum_j exp(h[j] - |x_i - y_j|^2 / (2 eps));  grad_x[i] (nullable) = d out[i] / d x_i
 * x [N,3], y [M,3], h [M] (log-weight + potential / eps). */
int ls_sinkhorn_softmin_f32(const float* x, const float* y, const float* h, int N, int M, float eps, float* out, float* grad_x,
                            void* stream);

/* SURVEY.md 8 (f-1), registration half, BATCHED over P pairs (csrc/optim.hip): the device side of one step of
 * More_Solver._solve_pairwise_registration(optim=True) (lib_more/more_solver.py:118-189) for P pairs in lock-step.
 *   ls_se3_transform_f32         query[p] = g[p] . src[p]       g [P,3,4] (R | t), src / query [P,N,3]                  (:149)
 *   ls_smooth_l1_f32             loss[p] (+)= mean_i SmoothL1(sdf[p,i], 0);  grad_sdf[p,i] = d loss[p] / d sdf[p,i]       (:152-156)
 *   ls_sinkhorn_softmin_batched_f32   ls_sinkhorn_softmin_f32 with a pair index: out[p,i] = -eps_p log sum_j exp(logw + pot_y[p,j] / eps_p
 *                                - |x_i - y_j|^2 / (2 eps_p)) (averaged with prev[p,i] if `average`); eps_p <= 0: out = prev (schedule
 *                                of that pair has ended); grad_x optional; out must not alias prev                           (:146,158)
 *   ls_se3_adam_step_f32         tangent gradient (sum G, sum query x G) -> Adam moments m1 / m2 [P,6] -> g <- exp(-step) g; best-loss
 *                                snapshot best_g / min_loss (after the step, as the reference); geodesic angle to init_R [P,3,3] above
 *                                stop_angle -> active[p] = 0 (frozen from then on); query <- g . src for the next step             (:160-173)
 * torchlie / geomloss / roma are absent: retraction and Sinkhorn loop are this build's definitions (PARITY UNPINNED). */
int ls_se3_transform_f32(const float* g, const float* src, int P, int N, float* query, void* stream);
int ls_smooth_l1_f32(const float* sdf, int P, int N, int accumulate, float* loss, float* grad_sdf, void* stream);
int ls_sinkhorn_softmin_batched_f32(const float* x, const float* y, const float* pot_y, float logw, const float* eps, const float* prev,
                                    int average, int P, int N, int M, float* out, float* grad_x, void* stream);
/* Up to four independent batched softmins in one launch (the four potentials of one symmetric Sinkhorn iteration): the same
 * arithmetic as ls_sinkhorn_softmin_batched_f32 without the gradient, problem i on x [P,N,3], y [P,M,3], pot_y [P,M] or NULL. */
typedef struct ls_softmin_problem {
    const float* x;
    const float* y;
    const float* pot_y;
    const float* prev;
    float* out;
    float logw;
    int N, M;
} ls_softmin_problem;
int ls_sinkhorn_softmin_multi_f32(const ls_softmin_problem* problems, int count, const float* eps, int average, int P, void* stream);
/* More_Solver._optimize_code's loss and optimizer on the device (more_solver.py:199-221: MSELoss(sdf, 0), torch.optim.Adam with three
 * parameter groups -- z_inv 1e-5, t 1e-4, z_so3 5e-4 -- MultiStepLR([160]), best-loss bookkeeping):
 *   ls_mse_f32: loss[p] = mean_i sdf[p,i]^2, grad_sdf[p,i] = 2 sdf[p,i] / N; min_loss / improved (nullable, [P]): if loss[p] <
 *               min_loss[p] then min_loss[p] = loss[p], improved[p] = 1 (:219-221);
 *   ls_adam_step_f32: one Adam step (no weight decay, no amsgrad, torch's operation order) on up to four tensors, each with its own
 *               learning rate; step = 0-based step count (bias corrections use step + 1). */
typedef struct ls_adam_group {
    float* param;
    const float* grad;
    float* m;
    float* v;
    long long n;
    double lr;
} ls_adam_group;
int ls_mse_f32(const float* sdf, int P, int N, float* loss, float* grad_sdf, float* min_loss, int32_t* improved, void* stream);
/* betas, eps and learning rates are DOUBLES (torch.optim.Adam's Python floats): 1 - beta, 1 - beta^t, lr / (1 - beta1^t), sqrt(1 - beta2^t) are formed
 * in double on the host and rounded to fp32 once, as torch hands them to its kernels */
int ls_adam_step_f32(const ls_adam_group* groups, int count, double beta1, double beta2, double adam_eps, int step, void* stream);
int ls_se3_adam_step_f32(const float* src, const float* grad_query, const float* loss, int P, int N, double lr, double beta1, double beta2,
                         double adam_eps, int step, float stop_angle, float* g, float* m1, float* m2, float* min_loss, float* best_g,
                         const float* init_R, int32_t* active, float* query, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SURVEY.md 8 (f-2), first half: the MISE octree that decides WHICH lattice points of the (R+1)^3 grid the decoder has to
 * evaluate (R = resolution_0 << depth) and assembles the dense value grid -- class MISE of
 * occnet_utils/utils/libmise/mise.pyx, driven by the loop of occnet_utils/mesh_extractor2.py:116-131:
 *     init;  loop { query -> n points; if n == 0 break; values = decoder(points); update(points, values) };  to_dense
 * `state` is caller-allocated device memory of ls_mise_state_bytes(); lattice index = (x*(R+1) + y)*(R+1) + z.
 * (Marching cubes -- libmcubes -- follows below: ls_marching_cubes_*.)
 * ---------------------------------------------------------------------------------------------- */
size_t ls_mise_state_bytes(int resolution_0, int depth);
long long ls_mise_lattice_points(int resolution_0, int depth);   /* (R+1)^3: upper bound for a query */
/* MISE.__cinit__, mise.pyx:43-85 */
int ls_mise_init(void* state, size_t state_bytes, int resolution_0, int depth, void* stream);
/* MISE.query, mise.pyx:104-126 (+ the coordinate normalisation of mesh_extractor2.py:122-124, box_size = 1 + padding):
 * unknown points in ascending lattice order -> idx_out [cap] int32, pts_out [cap,3]; *count_out (DEVICE int32) = number of
 * unknown points (nothing is written past cap). */
int ls_mise_query(void* state, int resolution_0, int depth, float box_size, int32_t* idx_out, float* pts_out, int cap,
                  int32_t* count_out, void* stream);
/* MISE.update, mise.pyx:87-102 + subdivide_voxels :188-236; threshold = logit of the occupancy threshold (double, as the reference) */
int ls_mise_update(void* state, int resolution_0, int depth, double threshold, const int32_t* idx, const float* values, int n,
                   void* stream);
/* MISE.to_dense, mise.pyx:128-165 -> grid_out [(R+1)^3] float32 (the reference's float64 grid holds float32 values) */
int ls_mise_to_dense(void* state, int resolution_0, int depth, float* grid_out, void* stream);

/* SURVEY.md 8 (f-2), second half: libmcubes.marching_cubes(volume, isovalue) as called by Generator3D.extract_mesh
 * (occnet_utils/mesh_extractor2.py:161-176; occnet_utils/utils/libmcubes/marchingcubes.h:23-196, marchingcubes.cpp:290-326,
 * pywrapper.cpp:90-108): volume [nx,ny,nz] float64 -> vertices [nv,3] float64 (with the library's +0.5 offset) and faces
 * [nf,3] int64, in the reference's vertex and face ORDER.  counts_out = DEVICE long long[2] {nv, nf}; call once with
 * vertices = faces = NULL to size the outputs (nothing past cap_v vertices / cap_f faces is written).  isovalue: the
 * reference narrows it to float (mcubes.pyx:22) -- pass the narrowed value. */
size_t ls_mcubes_workspace_bytes(int nx, int ny, int nz);
int ls_marching_cubes_f64(const double* volume, int nx, int ny, int nz, double isovalue, double* vertices, long long cap_v,
                          long long* faces, long long cap_f, long long* counts_out, void* workspace, size_t workspace_bytes,
                          void* stream);

/* libsimplify.simplify_mesh(mesh, f_target, agressiveness) as called by Generator3D.extract_mesh with (mesh, simplify_nfaces, 5.0)
 * (occnet_utils/mesh_extractor2.py:205-208; occnet_utils/utils/libsimplify/__init__.py:7-17, simplify_mesh.pyx:34-88,
 * Simplify.h:345-445): quadric-error edge-collapse decimation down to `target_faces` triangles.  A sequential greedy algorithm the
 * reference runs on the CPU too: HOST arrays in and out.  vertices [nv,3] float64, faces [nf,3] int64 -> vertices_out (room for
 * nv rows), faces_out (room for nf rows), counts_out {nv', nf'}; vertices, faces and their ORDER are bit-identical to the reference.
 * initial_border: the value Vertex::border holds while the INITIAL edge costs are computed -- the reference never initialises it
 * (simplify_mesh.pyx:42 copies an uninitialised temporary; Simplify.h resets it only after the costs, :683): 1 = what every build of
 * the reference made here reads (non-zero: initial costs from the best of a, b, midpoint) and what the fixtures pin; 0 = as published. */
int ls_simplify_mesh_f64_host(const double* vertices_host, long long nv, const long long* faces_host, long long nf, int target_faces,
                              double aggressiveness, int initial_border, double* vertices_out_host, long long* faces_out_host,
                              long long* counts_out_host);

/* ------------------------------------------------------------------------------------------------
 * Live per-kernel timing (bench.py's roofline leg): while enabled, every kernel ls_encode / ls_sdf_decode
 * launches is bracketed by hipEvents on the stream it is launched on.  ls_profile_end synchronises those
 * streams and returns, per (kind, layer), the number of launches and their summed duration.
 * ---------------------------------------------------------------------------------------------- */
typedef enum {
    LS_K_PROLOGUE = 0, LS_K_FPS, LS_K_KNN, LS_K_GEMM_EDGE, LS_K_EDGE_L0, LS_K_EDGE_POOL, LS_K_EDGE_ATTN, LS_K_MEAN,
    LS_K_GEMM_GLOB, LS_K_VN_ACT, LS_K_GEMM_TAIL, LS_K_TAIL, LS_K_SDF_PREP, LS_K_SDF_AFFINE, LS_K_GEMM_SDF, LS_K_SDF_OUT,
    LS_K_COUNT
} ls_kernel_kind;

typedef struct {
    int32_t kind;      /* ls_kernel_kind */
    int32_t layer;     /* encoder layer / FPS level / decoder layer */
    int32_t launches;
    float total_ms;
} ls_profile_entry;

int ls_profile_begin(ls_model_t* m);
int ls_profile_end(ls_model_t* m, ls_profile_entry* out_host, int max_entries, int* n_out_host);
/* Exact-phase statistics of the k-NN graph builds of the profiled ls_encode calls since ls_profile_begin (call BEFORE ls_profile_end, or
 * after it: the counters persist until the next ls_profile_begin): for encoder layer i, out_host[2 i] = candidates that were given a
 * canonical distance beyond the hints, summed over queries and calls, out_host[2 i + 1] = queries (0 for layers that do not run the
 * filter / exact-phase path).  bench.py derives the EXECUTED fp32 VALU work of the dominant k-NN launch from it (roofline.frac_hw). */
int ls_profile_knn_stats(ls_model_t* m, unsigned long long* out_host, int max_layers);

#ifdef __cplusplus
}
#endif
#endif /* LIVINGSCENES_HIP_H */
