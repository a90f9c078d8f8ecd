/*
 * dgr_hip.h -- C ABI of libdgr_hip.so, the MI355X (gfx950) implementation of the
 * Deep Global Registration inference hot path.
 *
 * The reference (chrischoy/DeepGlobalRegistration) is pure Python with no FFI of
 * its own; its "operator interface" for this path is the set of Python callables
 * below.  Every entry point cites the reference interface it replaces
 * (paths relative to the reference repo).  INTEGRATION.md shows the ctypes
 * binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / STL types.
 *   - "dev" pointers are device (HBM) pointers valid on the ctx's device; the
 *     caller owns every input/output buffer, the library owns ctx, nets, hash
 *     tables, kernel maps and a grow-only workspace.
 *   - all work is enqueued on the caller's `stream` (a hipStream_t passed as
 *     void*; NULL = the default stream); functions that return host scalars
 *     synchronise that stream, all others are asynchronous.
 *   - return value: DGR_OK (0) or a negative DGR_E* code; dgr_last_error()
 *     returns a thread-local message for the last failure.
 *   - a ctx is bound to one device and is not thread-safe.
 */
#ifndef DGR_HIP_H
#define DGR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dgr_ctx dgr_ctx;
typedef struct dgr_net dgr_net;
typedef struct dgr_maps dgr_maps;
typedef void *dgr_stream; /* hipStream_t */

enum {
  DGR_OK = 0,
  DGR_EINVAL = -1,   /* bad argument / shape / duplicate coordinates */
  DGR_EHIP = -2,     /* HIP runtime error */
  DGR_ENOMEM = -3,   /* workspace exhausted (kernel-map capacity overflow) */
  DGR_ESVD = -4,     /* 3x3 SVD did not converge / non-finite input: maps to the
                        RuntimeError caught at core/deep_global_registration.py:295 */
  DGR_EINTERNAL = -5
};

/* per-pair status of the confidence gate, core/deep_global_registration.py:276-281 */
enum { DGR_STATUS_OK = 0, DGR_STATUS_LOW_CONFIDENCE = 1, DGR_STATUS_SVD_FAILED = 2,
       DGR_STATUS_SAFEGUARD = 3 /* gate failed, T from the safeguard RANSAC (dgr_params.safeguard) */,
       /* flag, OR-ed onto one of the codes above (status & DGR_STATUS_MASK): dgr_params.use_icp, but the final ICP could
          not run on this pair (no finite target point); T is the estimate the code names, before ICP */
       DGR_STATUS_FLAG_ICP_SKIPPED = 0x100, DGR_STATUS_MASK = 0xff,
       /* internal: the workgroups that share a pair's refinement lost each other (never returned to the caller as a
        * status: the call fails with DGR_EINTERNAL) */
       DGR_STATUS_EXCHANGE_TIMEOUT = 0x7f };

const char *dgr_last_error(void);
const char *dgr_version(void);

/* ---- context ------------------------------------------------------------------------ */
int dgr_ctx_create(int device, dgr_ctx **out);
void dgr_ctx_destroy(dgr_ctx *ctx);
/* bytes currently reserved by the grow-only workspace (diagnostics) */
int64_t dgr_ctx_workspace_bytes(dgr_ctx *ctx);
/* A stream on its own share of the GPU's compute units (no reference counterpart: the reference registers one pair at a
 * time on torch's current stream, core/deep_global_registration.py:238-324).  A process that keeps several contexts busy
 * (one per host thread, each with its own batches) lets every kernel of every context compete for all CUs: a persistent
 * 6-D conv kernel of one context holds every CU for milliseconds and another context's 10-us launch waits behind it.
 * This call creates a stream whose kernels run on share `part` of `nparts` (2 or 4) equal shares of the CUs
 * (hipExtStreamCreateWithCUMask; mask bit b is slot b / 8 of XCD b % 8, share p takes the slots with slot % nparts == p
 * of EVERY XCD, so each share keeps all eight L2s), sizes the context's persistent launches for that share, and returns
 * the stream in *out: pass it to every entry point called with this context.  The stream belongs to the context
 * (destroyed with it; a second call replaces it after a device synchronisation).  nparts = 1 drops the partition
 * (*out = NULL).  Results do not depend on it (grids only).  Measured on MI355X, BASELINE configs[1]: four contexts on
 * four shares 396 pairs/s against 378 for three contexts on the whole GPU and 360 for four (tools/r06_runs/run42.sh). */
int dgr_ctx_create_partition_stream(dgr_ctx *ctx, int part, int nparts, dgr_stream *out);

/* ---- voxelisation: replaces ME.utils.sparse_quantize(xyz / voxel, return_index=True) and
 * ME.utils.batched_coordinates at core/deep_global_registration.py:152,158 (preprocess, :134-161).
 * xyz: dev [M,3] float64 (is_f64=1) or float32 (is_f64=0); floor(xyz/voxel) is taken in that dtype.
 * sel_out: dev int64 [>=M]   indices of the first point of every voxel, ascending
 * coords_out: dev int32 [>=M,4] (batch_index, floor(xyz[sel]/voxel))
 * xyz_out: dev float32 [>=M,3] xyz[sel] cast to float32
 * n_out: host int64*, number of voxels (synchronises the stream). */
int dgr_voxelize(dgr_ctx *ctx, const void *xyz, int is_f64, int64_t M, double voxel_size,
                 int32_t batch_index, int64_t *sel_out, int32_t *coords_out, float *xyz_out,
                 int64_t *n_out, dgr_stream stream);

/* ---- network: replaces model.load_model('ResUNetBN2C')(in, out, bn_momentum, conv1_kernel_size,
 * normalize_feature, D) + load_state_dict + .eval() (core/deep_global_registration.py:96-131;
 * class at model/resunet.py:419-665).  Tensors are HOST float32 arrays in MinkowskiEngine
 * state_dict layout ("conv1.kernel" [K,Cin,Cout], "norm1.bn.weight", ..., "final.bias");
 * the library folds eval-mode batch norm into the kernels, re-tiles them for the MFMA
 * B-operand and copies them to HBM; the caller may free its arrays afterwards.
 * Offset axis of a kernel: index j = sum_d (delta_d + ks/2) ks^d, FIRST spatial axis fastest; a transposed
 * convolution (conv4_tr / conv3_tr / conv2_tr) pairs index and offset like the forward strided map it swaps -- the
 * library's reading of MinkowskiEngine 0.5.4 (unverifiable offline).  A checkpoint in another convention is
 * re-indexed by the caller before this call: deepglobalregistration_amd/model/me_conventions.py does it for the
 * Python host, tools/check_me_conventions.py tells the readings apart on a real pair. */
typedef struct {
  const char *name;
  const float *data; /* host (dgr_net_create) / device (dgr_net_create_device) */
  int64_t numel;
} dgr_weight_desc;

int dgr_net_create(dgr_ctx *ctx, int D, int in_channels, int out_channels, int conv1_kernel_size,
                   int normalize_feature, const dgr_weight_desc *weights, int n_weights,
                   dgr_net **out);
/* The same with every `data` a DEVICE pointer on the context's device ("host or device pointers"): the state
 * dict is already in HBM -- e.g. views of the flat RCCL broadcast buffer of a multi-GPU start (deepglobalregistration_amd/
 * dist.py) -- and batch-norm folding, the power-of-two weight scale, the f16 split and every MFMA operand layout are
 * produced by HIP kernels from there: nothing but the per-channel batch-norm vectors (cout floats each) travels to the host.
 * Bit-identical weight sets to dgr_net_create on the same values (tests/test_gpu_device_weights.py).  The library copies:
 * the caller may free its tensors when the call returns. */
int dgr_net_create_device(dgr_ctx *ctx, int D, int in_channels, int out_channels, int conv1_kernel_size,
                          int normalize_feature, const dgr_weight_desc *weights, int n_weights,
                          dgr_net **out);
void dgr_net_destroy(dgr_net *net);
int64_t dgr_net_param_bytes(const dgr_net *net);
/* A second net object bound to `ctx` (another context of the SAME device: one context per HIP stream / host thread)
 * over the weights `src` already holds in HBM -- the equivalent of several Python threads calling one torch module in
 * the reference (core/deep_global_registration.py:96-131 builds the models once).  The weights are immutable and
 * reference-counted: they are freed when the last net sharing them is destroyed; per-forward state (kernel maps,
 * intermediates) belongs to each net object.  dgr_net_sharers: how many net objects hold `net`'s weights;
 * dgr_net_param_bytes reports the shared set's bytes (count it once per distinct weight set). */
int dgr_net_share(dgr_ctx *ctx, const dgr_net *src, dgr_net **out);
int dgr_net_sharers(const dgr_net *net);

/* ---- sparse ResUNet forward: replaces ME.SparseTensor(feats, coordinates=coords) +
 * ResUNet2.forward (core/deep_global_registration.py:163-169, 210-217; model/resunet.py:598-649).
 * coords: dev int32 [N,1+D] (batch column first, unique rows); feats: dev f32 [N,Cin];
 * out: dev f32 [N,Cout], row i belongs to coords row i.  Asynchronous.
 * Precondition for a 3-D net with ONE input channel (the FCGF net): feats are finite numbers -- its first layer keeps
 * the input values in a dense grid whose empty cells hold the NaN bit pattern 0xffffffff, so an input with exactly those
 * bits would be read as an empty voxel (the reference feeds ones, core/deep_global_registration.py:160). */
int dgr_resunet_forward(dgr_ctx *ctx, dgr_net *net, const int32_t *coords, const float *feats,
                        int64_t N, float *out, dgr_stream stream);
/* after a forward: copy an intermediate activation ("s1","s2","s4","s8","s4_tr","s2_tr","s1_tr")
 * to a HOST buffer (post-ReLU values, like the reference's out_s* tensors); rows/cols out. */
int dgr_net_get_intermediate(dgr_ctx *ctx, dgr_net *net, const char *name, float *host_out,
                             int64_t capacity, int64_t *rows, int64_t *cols);
/* per-forward work statistics of the last forward (for the roofline report): for conv layer
 * `layer` (0..22): pairs, non-empty offsets, n_in, n_out, cin, cout.  Synchronises. */
int dgr_net_layer_stats(dgr_ctx *ctx, dgr_net *net, int layer, int64_t stats[8]);
int dgr_net_num_layers(const dgr_net *net);
/* kernel-tuning instrument: re-run conv layer `layer` of the last dgr_resunet_forward `reps` times on
 * the default stream and return the mean duration (ms) of its MFMA phase and its reduce phase. */
int dgr_net_rerun_layer(dgr_ctx *ctx, dgr_net *net, int layer, int reps, float *gemm_ms, float *reduce_ms);

/* ---- coordinate / kernel maps as a stand-alone object (inspection + parity tests):
 * the coordinate manager part of ME.SparseTensor / MinkowskiConvolution. */
int dgr_maps_create(dgr_ctx *ctx, const int32_t *coords, int64_t N, int D, int conv1_kernel_size,
                    dgr_maps **out, dgr_stream stream);
void dgr_maps_destroy(dgr_maps *maps);
/* coordinates of the map at tensor stride ts (1,2,4,8): host int32 [n,1+D]; returns n in *n */
int dgr_maps_get_coords(dgr_maps *maps, int ts, int32_t *host_out, int64_t capacity, int64_t *n);
/* kernel map: kind 0 = same-stride 3^D at ts, 1 = conv1 (ks^D at ts=1), 2 = strided ts -> 2ts; D = 3 only:
 * 3 / 4 / 5 = the dense neighbour tables of the output-stationary conv (same stride / ts -> 2ts / transposed
 * 2ts -> ts with out = the fine map), returned in the same (k, in, out) form.
 * host outputs: rule_ptr int32 [K+1], pair_in/pair_out int32 [P]; *K, *P returned. */
int dgr_maps_get_kernel_map(dgr_maps *maps, int kind, int ts, int32_t *rule_ptr, int64_t rule_cap,
                            int32_t *pair_in, int32_t *pair_out, int64_t pair_cap, int64_t *K,
                            int64_t *P);

/* ---- feature-space 1-NN: replaces core.knn.find_knn_gpu(F0, F1, nn_max_n, knn=1,
 * return_distance) (core/knn.py:23-74) incl. core.metrics.pdist (core/metrics.py:62-69).
 * F0 dev f32 [N0,C], F1 dev f32 [N1,C]; idx_out dev int64 [N0]; dist_out dev f32 [N0] or NULL.
 * squared=0: dist = sqrt(sum (a-b)^2 + 1e-7) (chunked branch, 'L2'); squared=1: sum (a-b)^2.
 * The [chunk,N1,C] temporary of the reference is never materialised. */
int dgr_knn1_l2(dgr_ctx *ctx, const float *F0, int64_t N0, const float *F1, int64_t N1, int C,
                int squared, int64_t *idx_out, float *dist_out, dgr_stream stream);

/* ---- the same search for every pair of a collated batch in ONE launch per kernel: replaces
 * core.knn.find_knn_gpu_batch(F0, F1, len_batch, ...) (core/knn.py:106-140, one find_knn_gpu call per
 * pair there).  F0 dev f32 [off0[npairs],C] / F1 dev f32 [off1[npairs],C] hold the pairs' rows back to back
 * (dataloader/base_loader.py:63-81); off0 / off1 HOST int64 [npairs+1], starting at 0, no empty pair.
 * idx_out dev int64 [off0[npairs]]: rows of the CONCATENATED F1 (the reference's `concat_results`
 * numbering, core/knn.py:131-134; subtract off1[p] for per-pair indices); dist_out as above or NULL. */
int dgr_knn1_l2_batch(dgr_ctx *ctx, const float *F0, const int64_t *off0, const float *F1,
                      const int64_t *off1, int npairs, int C, int squared, int64_t *idx_out,
                      float *dist_out, dgr_stream stream);

/* ---- 6-D inlier-network input: replaces the torch.cat at core/deep_global_registration.py:261-262
 * and inlier_feature_generation (:185-208).  idx1 dev int64 [N0] (corres_idx1; corres_idx0 is
 * arange(N0)).  feature_type 0='ones' -> feats [N0,1]; 1='coords' -> [N0,6] =
 * (cos(xyz0[i]), cos(xyz1[idx1[i]])).  coords6_out dev int32 [N0,7]. */
int dgr_inlier_inputs(dgr_ctx *ctx, const int32_t *coords0, const float *xyz0, int64_t N0,
                      const int32_t *coords1, const float *xyz1, int64_t N1, const int64_t *idx1,
                      int feature_type, int32_t *coords6_out, float *feats_out, dgr_stream stream);

/* ---- confidence gate: replaces logit.sigmoid(); weights[weights < clip] = 0; weights.sum().item()
 * (core/deep_global_registration.py:269-272).  weights_out dev f32 [N]; *wsum host (synchronises). */
int dgr_sigmoid_clip_sum(dgr_ctx *ctx, const float *logit, int64_t N, float clip, float *weights_out,
                         double *wsum, dgr_stream stream);

/* gather rows: X[i] = src[idx[i]] for [*,3] float32 (xyz1[corres_idx1] at :283-285) */
int dgr_gather_rows3(dgr_ctx *ctx, const float *src, const int64_t *idx, int64_t N, float *dst,
                     dgr_stream stream);

/* ---- weighted Procrustes: replaces core.registration.weighted_procrustes(X, Y, w, eps)
 * (core/registration.py:91-113).  X,Y dev f32 [N,3], w dev f32 [N]; R9 (row-major 3x3) and t3
 * are HOST outputs (the reference also lands on the host: .cpu() at :105,112).  3x3 SVD in f64
 * on the device.  Returns DGR_ESVD on non-finite input. */
int dgr_weighted_procrustes(dgr_ctx *ctx, const float *X, const float *Y, const float *w, int64_t N,
                            float eps, float *R9, float *t3, dgr_stream stream);

/* ---- robust SE(3) refinement: replaces core.registration.GlobalRegistration(points,
 * trans_points, weights, max_iter, max_break_count, break_threshold_ratio, quantization_size)
 * (core/registration.py:135-194) with HighDimSmoothL1Loss (core/loss.py:42-61), ortho2rotation
 * (:16-64), Adam(lr=0.1) + ExponentialLR(0.999) (:163-164): weighted-Procrustes initialisation
 * and the whole optimisation loop run in ONE persistent kernel (no per-iteration host syncs).
 * Host outputs: R9 row-major, t3, iterations, loss, break_count. */
int dgr_se3_refine(dgr_ctx *ctx, const float *X, const float *Y, const float *w, int64_t N,
                   float quantization_size, int max_iter, int max_break_count,
                   double break_threshold_ratio, float *R9, float *t3, int32_t *iterations,
                   float *loss, int32_t *break_count, dgr_stream stream);

/* ---- ICP: replaces o3d.pipelines.registration.registration_icp(source, target,
 * max_correspondence_distance = 2 * voxel, init = T) at core/deep_global_registration.py:317-322
 * (point-to-point without scaling; Open3D defaults max_iter 30, relative_fitness = relative_rmse = 1e-6).
 * src dev f32 [N0,3], dst dev f32 [N1,3]; T_init host f64[16] row-major or NULL (identity).
 * Host outputs: T_out f64[16]; stats_out f64[3] = {fitness, inlier_rmse, iterations} or NULL.  Synchronises. */
int dgr_icp_point_to_point(dgr_ctx *ctx, const float *src, int64_t N0, const float *dst, int64_t N1,
                           double max_correspondence_distance, const double *T_init, int max_iter,
                           double relative_fitness, double relative_rmse, double *T_out, double *stats_out,
                           dgr_stream stream);

/* ---- safeguard: replaces registration_ransac_based_on_correspondence(pcd0, pcd1, idx0, idx1,
 * distance_threshold, num_iterations) at core/deep_global_registration.py:50-64 (called from :302-315 when the
 * confidence gate fails): ransac_n = 4, point-to-point estimation, no checkers, every hypothesis evaluated,
 * the best 4-point hypothesis returned.  X = xyz0[idx0], Y = xyz1[idx1] dev f32 [N,3] (dgr_gather_rows3).
 * Samples come from a counter-based generator (seed) instead of Open3D's per-thread mt19937 streams and the
 * consensus test runs in f32 without fma, so that the result is reproducible (oracle/open3d_reg.py).
 * Host outputs: T_out f64[16]; stats_out f64[3] = {best hypothesis index, inlier count, inlier rmse}.  Synchronises. */
int dgr_ransac_correspondence(dgr_ctx *ctx, const float *X, const float *Y, int64_t N, double distance_threshold,
                              int64_t num_hypotheses, uint32_t seed, double *T_out, double *stats_out,
                              dgr_stream stream);

/* ---- fused pipeline: replaces DeepGlobalRegistration.register() steps 1-5 case 0
 * (core/deep_global_registration.py:248-300) for a batch of already voxelised pairs, without
 * intermediate host synchronisation.  Pair p uses rows [off0[p], off0[p+1]) of coords0/xyz0 and
 * [off1[p], off1[p+1]) of coords1/xyz1 (host offset arrays, npairs+1 entries); the batch column
 * of the coords must equal p.  Two harness-only overrides exist because no trained checkpoint is
 * available offline (both NULL in production, see DESIGN.md "Synthetic workload"):
 * override_idx1 (dev int64 [sum N0], batch-global fragment-1 row or -1 = keep) replaces 1-NN
 * results after the search ran; forced_logit (dev f32 [sum N0]) replaces the inlier network's
 * logits after it ran.
 * T_out host f32 [npairs,16] row-major 4x4, status_out host int32 [npairs],
 * stats_out host f32 [npairs,4] = (iterations, loss, break_count, wsum) or NULL. */
typedef struct {
  float clip_weight_thresh;     /* config.clip_weight_thresh, config.py:63 (0.05) */
  float voxel_size;             /* checkpoint config */
  int inlier_feature_type;      /* 0 'ones', 1 'coords' */
  int max_iter;                 /* 1000 */
  int max_break_count;          /* 20 */
  double break_threshold_ratio; /* 1e-4 as passed at :286 */
  int skip_refinement;          /* ablation (config C5): stop after weighted Procrustes */
  /* the two Open3D steps that end register() (:302-322), inside the same call (0 = leave them to the caller): */
  int safeguard;                /* pairs that fail the confidence gate: RANSAC over the putative correspondences (:302-315,
                                 * :50-64), status DGR_STATUS_SAFEGUARD; an SVD failure keeps T = identity (:295-300) */
  int64_t ransac_hypotheses;    /* 4000000 in the reference (:58) */
  uint32_t ransac_seed;
  int use_icp;                  /* point-to-point ICP from the estimate, max distance 2 voxel, 30 iterations (:317-322) */
} dgr_params;

int dgr_register_batch(dgr_ctx *ctx, dgr_net *fcgf, dgr_net *inlier, const int32_t *coords0,
                       const float *xyz0, const int64_t *off0, const int32_t *coords1,
                       const float *xyz1, const int64_t *off1, int npairs, const dgr_params *params,
                       const int64_t *override_idx1, const float *forced_logit, float *T_out,
                       int32_t *status_out,
                       float *stats_out, dgr_stream stream);
/* device-side intermediates of the last dgr_register_batch (valid until the next call on this
 * ctx): which = 0 idx1 (int64 [sumN0]), 1 logit (f32 [sumN0]), 2 weights (f32 [sumN0]),
 * 3 F0 (f32 [sumN0,C]), 4 F1 (f32 [sumN1,C]).  *numel receives the element count; when dst_dev is
 * not NULL the data is copied (device to device, on `stream`) into dst_dev (capacity in bytes). */
int dgr_register_batch_output(dgr_ctx *ctx, int which, void *dst_dev, int64_t capacity_bytes,
                              int64_t *numel, dgr_stream stream);
/* The transforms of the last dgr_register_batch on this context as float64, HOST [npairs,16] (what the reference's
 * register() returns, :290-291, 317-322: T is np.float64, Open3D's results are doubles).  The learned estimate is the f32
 * result widened; a pair that went through the safeguard RANSAC or the final ICP carries that stage's float64 result,
 * which dgr_register_batch's float T_out rounds.  T_out may be NULL (only *npairs is returned). */
int dgr_register_batch_f64(dgr_ctx *ctx, double *T_out, int64_t capacity_pairs, int64_t *npairs);

/* per-stage device time (ms, HIP events) of the last dgr_register_batch when profiling was
 * enabled with dgr_ctx_set_profiling(ctx, 1): [fcgf, knn, inlier_inputs, inlier_net, registration,
 * maps_3d, maps_6d, conv_kernels_total, safeguard RANSAC + ICP steps (dgr_params.safeguard / use_icp)].  Synchronises. */
int dgr_ctx_set_profiling(dgr_ctx *ctx, int enable);
#define DGR_NUM_STAGE_TIMES 9
/* writes min(capacity, DGR_NUM_STAGE_TIMES) values and the number written to *n (nullable): the explicit capacity is
 * what keeps a caller built against an older header from being overrun when the list grows. */
int dgr_ctx_stage_times_v2(dgr_ctx *ctx, float *times_ms, int capacity, int *n);
/* the version-0.1 entry point, kept under its name with its original contract: the first eight values into float[8] */
int dgr_ctx_stage_times(dgr_ctx *ctx, float *times_ms);
/* number of sparse-conv kernel launches covered by times_ms[7] */
int64_t dgr_ctx_conv_launches(dgr_ctx *ctx);
/* duration (ms) of every sparse-conv layer launch of the last profiled batch, in launch order (FCGF layers
 * 0..22, then the inlier net's): times_ms = MFMA phase + reduce phase, gemm_ms (nullable) = MFMA phase alone
 * (the sparse_conv_mfma kernel); *n = number written (<= capacity) */
int dgr_ctx_conv_launch_times(dgr_ctx *ctx, float *times_ms, float *gemm_ms, int64_t capacity, int64_t *n);
/* the same launches timed BY THE KERNEL ITSELF: execution span in microseconds (latest wave end - earliest wave start on
 * the device's 100-MHz wall clock) -- what rocprofv3 --kernel-trace reports as the kernel's duration, valid also while other
 * streams share the GPU (the HIP-event spans above then include the wait for compute units).  0 for launches whose kernel
 * is not instrumented (only the wide-layer kernel, the dominant one, is). */
int dgr_ctx_conv_launch_kernel_us(dgr_ctx *ctx, float *us, int64_t capacity, int64_t *n);
/* kernel variant that ran each of those launches (the kernel's name as rocprofv3 --kernel-trace prints it),
 * newline-separated and NUL-terminated in buf; *n = number of names written */
int dgr_ctx_conv_launch_kinds(dgr_ctx *ctx, char *buf, int64_t capacity, int64_t *n);


/* ---- debug entry points (parity tests): the device functions of the registration kernel on their own.
 * ortho2rotation (core/registration.py:16-64) forward for n parameter rows p6 [n,6] -> R9_out [n,9] (row-major 3x3) and,
 * when grad_R9 [n,9] and grad_p6_out [n,6] are given, its backward (what autograd computes for sum(R * grad_R)).
 * All pointers are device pointers. */
int dgr_debug_ortho2rotation(dgr_ctx *ctx, const float *p6, int64_t n, const float *grad_R9, float *R9_out,
                             float *grad_p6_out, dgr_stream stream);
/* The refinement loop of dgr_se3_refine (core/registration.py:168-190) RESUMED at iteration state_in[27] from a given
 * optimiser state and run up to iteration max_iter: 30 HOST doubles = prm[9] (rot6d, trans), Adam exp_avg[9],
 * exp_avg_sq[9], iteration, loss_prev, break count; state_out receives the state it ends with.  X, Y, w as for
 * dgr_se3_refine (device).  Parity instrumentation: lets a test run W steps of this kernel and W steps of the reference
 * algorithm (in f32 and in f64) from the SAME state, so that the comparison does not pass through the chaotic
 * amplification of a whole trajectory. */
int dgr_debug_se3_refine_from(dgr_ctx *ctx, const float *X, const float *Y, const float *w, int64_t N,
                              float quantization_size, int max_iter, int max_break_count,
                              double break_threshold_ratio, const double *state_in, double *state_out,
                              dgr_stream stream);
/* HighDimSmoothL1Loss (core/loss.py:51-61) per point of X, Y [n,3] (device) with quantization_size q -> per_point_out [n] */
int dgr_debug_smooth_l1(dgr_ctx *ctx, const float *X, const float *Y, int64_t n, float quantization_size,
                        float *per_point_out, dgr_stream stream);

/* One conv layer (index in forward order, 0..22; layers with a 3^D kernel and >= 32 input channels) of `net` applied to
 * a caller-supplied feature matrix `in` [N, Cin] (device; max(x, 0) applied first when in_relu) over the same-stride
 * 3^D kernel map of `coords` [N, 1+D]: out [N, Cout] (device) = folded batch-norm shift + sum over the map, through the
 * very kernels the forward runs for that layer (model/residual_block.py:118-134; MinkowskiConvolution.forward). */
int dgr_debug_conv_layer(dgr_ctx *ctx, dgr_net *net, int layer, const int32_t *coords, const float *in, int in_relu,
                         int64_t N, float *out, dgr_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* DGR_HIP_H */
