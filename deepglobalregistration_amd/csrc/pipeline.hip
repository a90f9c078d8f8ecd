// Stage glue of DeepGlobalRegistration.register() (core/deep_global_registration.py:238-300):
// 6-D inlier-network input assembly, confidence gate, row gather, and the fused batched pipeline
// dgr_register_batch (FCGF x2 -> 1-NN -> 6-D inputs -> inlier net -> gate + Procrustes + refinement)
// that runs without intermediate host synchronisation.
#include <string.h>

#include "dgr_internal.h"

int dgr_registration_launch_ctx(dgr_ctx *ctx, const float *xyz0, const float *xyz1, const int64_t *idx1,
                                const float *lw, int is_logit, float clip, const int64_t *off0_dev,
                                const int64_t *off0_host, int npairs, int64_t total_rows, float q, int max_iter, int max_break,
                                double ratio, int skip_refine, int gate, float eps, float *weights_out,
                                DgrRegResult *results_dev, hipStream_t stream);
int dgr_ctx_new_flag(dgr_ctx *ctx, hipStream_t stream);
int dgr_ctx_check_flag(dgr_ctx *ctx, hipStream_t stream);
int dgr_net_out_channels(const dgr_net *net);
void dgr_net_invalidate_runs(dgr_net *net);
int dgr_net_in_channels(const dgr_net *net);
int dgr_net_dim(const dgr_net *net);
const dgr_ctx *dgr_net_ctx(const dgr_net *net);

// ---- 6-D coordinates (:261-262) and inlier features (:185-208) ----------------------------------
__global__ void inlier_inputs_kernel(const int32_t *__restrict__ coords0, const float *__restrict__ xyz0,
                                     int64_t N0, const int32_t *__restrict__ coords1,
                                     const float *__restrict__ xyz1, const int64_t *__restrict__ idx1,
                                     int feature_type, int32_t *__restrict__ coords6,
                                     float *__restrict__ feats) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N0) return;
  const int64_t j = idx1[i];
#pragma unroll
  for (int d = 0; d < 4; ++d) coords6[i * 7 + d] = coords0[i * 4 + d];
#pragma unroll
  for (int d = 0; d < 3; ++d) coords6[i * 7 + 4 + d] = coords1[j * 4 + 1 + d];
  if (feature_type == 0) {
    feats[i] = 1.f;
  } else {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      feats[i * 6 + d] = cosf(xyz0[i * 3 + d]);
      feats[i * 6 + 3 + d] = cosf(xyz1[j * 3 + d]);
    }
  }
}

int dgr_inlier_inputs_impl(const int32_t *coords0, const float *xyz0, int64_t N0, const int32_t *coords1,
                           const float *xyz1, const int64_t *idx1, int feature_type, int32_t *coords6,
                           float *feats, hipStream_t stream) {
  inlier_inputs_kernel<<<(int)dgr_ceil_div(N0, 256), 256, 0, stream>>>(coords0, xyz0, N0, coords1, xyz1, idx1,
                                                                     feature_type, coords6, feats);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

extern "C" int dgr_inlier_inputs(dgr_ctx *ctx, const int32_t *coords0, const float *xyz0, int64_t N0,
                                 const int32_t *coords1, const float *xyz1, int64_t N1, const int64_t *idx1,
                                 int feature_type, int32_t *coords6_out, float *feats_out,
                                 dgr_stream stream) {
  DGR_REQUIRE(ctx && coords0 && xyz0 && coords1 && xyz1 && idx1 && coords6_out && feats_out,
              "dgr_inlier_inputs: NULL argument");
  DGR_REQUIRE(feature_type == 0 || feature_type == 1,
              "inlier_feature_type must be 'ones' (0) or 'coords' (1); 'feats' is inconsistent with the "
              "network input width in the reference (deep_global_registration.py:119)");
  DGR_REQUIRE(N0 > 0 && N1 > 0, "dgr_inlier_inputs: empty input");
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  return dgr_inlier_inputs_impl(coords0, xyz0, N0, coords1, xyz1, idx1, feature_type, coords6_out, feats_out,
                                (hipStream_t)stream);
}

// ---- sigmoid / clip / sum (:269-272) --------------------------------------------------------------
__global__ void sigmoid_clip_sum_kernel(const float *__restrict__ logit, int64_t N, float clip,
                                        float *__restrict__ w_out, double *__restrict__ sum) {
  __shared__ double part[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    float w = 1.f / (1.f + expf(-logit[i]));
    if (clip > 0.f && w < clip) w = 0.f;
    w_out[i] = w;
    s += (double)w;
  }
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sum, part[0] + part[1] + part[2] + part[3]);
}

extern "C" int dgr_sigmoid_clip_sum(dgr_ctx *ctx, const float *logit, int64_t N, float clip,
                                    float *weights_out, double *wsum, dgr_stream stream_) {
  DGR_REQUIRE(ctx && logit && weights_out && wsum, "dgr_sigmoid_clip_sum: NULL argument");
  DGR_REQUIRE(N > 0, "dgr_sigmoid_clip_sum: empty input");
  hipStream_t stream = (hipStream_t)stream_;
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  DGR_CHECK(ctx->arena.reset());
  double *sum;
  DGR_ALLOC(sum, ctx->arena, double, 1);
  DGR_HIP_CHECK(hipMemsetAsync(sum, 0, sizeof(double), stream));
  int blocks = (int)dgr_ceil_div(N, 256);
  if (blocks > 1024) blocks = 1024;
  sigmoid_clip_sum_kernel<<<blocks, 256, 0, stream>>>(logit, N, clip, weights_out, sum);
  DGR_LAUNCH_CHECK();
  DGR_HIP_CHECK(hipMemcpyAsync(wsum, sum, sizeof(double), hipMemcpyDeviceToHost, stream));
  DGR_HIP_CHECK(hipStreamSynchronize(stream));
  return DGR_OK;
}

// ---- xyz1[corres_idx1] (:283-285) -------------------------------------------------------------------
__global__ void gather_rows3_kernel(const float *__restrict__ src, const int64_t *__restrict__ idx, int64_t N,
                                    float *__restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int64_t j = idx[i];
  dst[i * 3] = src[j * 3]; dst[i * 3 + 1] = src[j * 3 + 1]; dst[i * 3 + 2] = src[j * 3 + 2];
}

extern "C" int dgr_gather_rows3(dgr_ctx *ctx, const float *src, const int64_t *idx, int64_t N, float *dst,
                                dgr_stream stream) {
  DGR_REQUIRE(ctx && src && idx && dst && N > 0, "dgr_gather_rows3: bad argument");
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  gather_rows3_kernel<<<(int)dgr_ceil_div(N, 256), 256, 0, (hipStream_t)stream>>>(src, idx, N, dst);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

__global__ void fill_kernel(float *p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void override_idx_kernel(int64_t *idx, const int64_t *__restrict__ ovr, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && ovr[i] >= 0) idx[i] = ovr[i];
}

// both fragments of every pair as ONE sparse tensor: fragment-0 rows keep batch 2p, fragment-1 rows
// get batch 2p+1 (the batch column only has to separate clouds; ME.utils.batched_coordinates layout)
__global__ void concat_coords_kernel(const int32_t *__restrict__ c0, int64_t n0, const int32_t *__restrict__ c1,
                                     int64_t n1, int32_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n0 + n1) return;
  const int32_t *src = i < n0 ? c0 + i * 4 : c1 + (i - n0) * 4;
  out[i * 4] = src[0] * 2 + (i < n0 ? 0 : 1);
  out[i * 4 + 1] = src[1]; out[i * 4 + 2] = src[2]; out[i * 4 + 3] = src[3];
}

// ---- fused batched pipeline --------------------------------------------------------------------------
struct StageTimer {
  dgr_ctx *ctx;
  hipStream_t stream;
  hipEvent_t e[9][2];
  bool on;
  int rec(int stage, int which) {
    if (!on) return DGR_OK;
    e[stage][which] = ctx->events.next();
    if (!e[stage][which]) return DGR_EHIP;
    DGR_HIP_CHECK(hipEventRecord(e[stage][which], stream));
    return DGR_OK;
  }
};

extern "C" int dgr_register_batch(dgr_ctx *ctx, dgr_net *fcgf, dgr_net *inlier, const int32_t *coords0,
                                  const float *xyz0, const int64_t *off0, const int32_t *coords1,
                                  const float *xyz1, const int64_t *off1, int npairs, const dgr_params *prm,
                                  const int64_t *override_idx1, const float *forced_logit, float *T_out,
                                  int32_t *status_out,
                                  float *stats_out, dgr_stream stream_) {
  DGR_REQUIRE(ctx && fcgf && inlier && coords0 && xyz0 && off0 && coords1 && xyz1 && off1 && prm && T_out &&
                  status_out,
              "dgr_register_batch: NULL argument");
  DGR_REQUIRE(npairs >= 1 && npairs <= 65535, "dgr_register_batch: npairs=%d out of range", npairs);
  DGR_REQUIRE(dgr_net_dim(fcgf) == 3 && dgr_net_dim(inlier) == 6 && dgr_net_in_channels(fcgf) == 1,
              "dgr_register_batch: expects the 3-D FCGF net (1 input channel) and the 6-D inlier net");
  const int ftype = prm->inlier_feature_type;
  DGR_REQUIRE((ftype == 0 && dgr_net_in_channels(inlier) == 1) || (ftype == 1 && dgr_net_in_channels(inlier) == 6),
              "inlier_feature_type %d does not match the inlier network input width %d", ftype,
              dgr_net_in_channels(inlier));
  for (int p = 0; p < npairs; ++p)
    DGR_REQUIRE(off0[p + 1] > off0[p] && off1[p + 1] > off1[p], "pair %d is empty", p);
  DGR_REQUIRE(dgr_net_ctx(fcgf) == ctx && dgr_net_ctx(inlier) == ctx,
              "dgr_register_batch: a net object belongs to another context (one dgr_net per context: dgr_net_share)");
  hipStream_t stream = (hipStream_t)stream_;
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  ctx->last_T64.clear();   // (a call that fails midway leaves nothing of an earlier batch behind for dgr_register_batch_f64)
  DGR_CHECK(ctx->arena.reset());
  dgr_ctx_begin_profile(ctx);
  DGR_CHECK(dgr_ctx_new_flag(ctx, stream));
  DgrArena &A = ctx->arena;
  const int64_t n0 = off0[npairs], n1 = off1[npairs];
  const int C = dgr_net_out_channels(fcgf);
  StageTimer tm{ctx, stream, {}, ctx->profiling};

  float *F0, *F1, *ones, *feats6, *logit, *weights;
  int64_t *idx1, *off0_dev;
  int32_t *coords6;
  DgrRegResult *res_dev;
  DGR_ALLOC(F0, A, float, (n0 + n1) * C);
  F1 = F0 + n0 * C;  // one forward over both fragments writes [F0; F1]
  DGR_ALLOC(ones, A, float, n0 + n1);
  DGR_ALLOC(idx1, A, int64_t, n0);
  DGR_ALLOC(coords6, A, int32_t, n0 * 7);
  DGR_ALLOC(feats6, A, float, n0 * 6);
  DGR_ALLOC(logit, A, float, n0);
  DGR_ALLOC(weights, A, float, n0);
  DGR_ALLOC(off0_dev, A, int64_t, npairs + 1);
  DGR_ALLOC(res_dev, A, DgrRegResult, npairs);
  DGR_HIP_CHECK(hipMemcpyAsync(off0_dev, off0, (size_t)(npairs + 1) * sizeof(int64_t), hipMemcpyHostToDevice, stream));
  fill_kernel<<<256, 256, 0, stream>>>(ones, n0 + n1, 1.f);

  // Step 1: FCGF features of both fragments (feats = ones[N,1], :160)
  DGR_CHECK(tm.rec(0, 0));
  {
    // the reference runs the two fragments one after the other (:250-251); their rows never interact
    // (different batch index), so one sparse tensor halves the number of launches and fills the GPU
    DgrArena::Mark mk = A.mark();
    int32_t *coords01;
    DGR_ALLOC(coords01, A, int32_t, (n0 + n1) * 4);
    concat_coords_kernel<<<(int)dgr_ceil_div(n0 + n1, 256), 256, 0, stream>>>(coords0, n0, coords1, n1, coords01);
    DGR_CHECK(dgr_resunet_forward_impl(ctx, fcgf, coords01, ones, n0 + n1, F0, stream));
    A.rewind(mk);
    dgr_net_invalidate_runs(fcgf);
  }
  DGR_CHECK(tm.rec(0, 1));
  // Step 2: coarse correspondences, per pair (corres_idx0 = arange)
  DGR_CHECK(tm.rec(1, 0));
  // one launch per kernel for all pairs; the indices come out as rows of the concatenated F1
  DGR_CHECK(dgr_knn1_batch_impl(ctx, F0, off0, F1, off1, npairs, C, 0, idx1, nullptr, stream));
  if (override_idx1)
    override_idx_kernel<<<(int)dgr_ceil_div(n0, 256), 256, 0, stream>>>(idx1, override_idx1, n0);
  DGR_CHECK(tm.rec(1, 1));
  // Step 3: 6-D coordinates + inlier features
  DGR_CHECK(tm.rec(2, 0));
  DGR_CHECK(dgr_inlier_inputs_impl(coords0, xyz0, n0, coords1, xyz1, idx1, ftype, coords6, feats6, stream));
  DGR_CHECK(tm.rec(2, 1));
  // Step 4: inlier likelihood
  DGR_CHECK(tm.rec(3, 0));
  {
    DgrArena::Mark mk = A.mark();
    DGR_CHECK(dgr_resunet_forward_impl(ctx, inlier, coords6, feats6, n0, logit, stream));
    A.rewind(mk);
    dgr_net_invalidate_runs(inlier);
  }
  DGR_CHECK(tm.rec(3, 1));
  // Step 5 case 0: gate + weighted Procrustes + robust refinement, one workgroup per pair
  DGR_CHECK(tm.rec(4, 0));
  const float eps = 1.1920928955078125e-07f;
  DGR_CHECK(dgr_registration_launch_ctx(ctx, xyz0, xyz1, idx1, forced_logit ? forced_logit : logit, 1,
                                        prm->clip_weight_thresh, off0_dev, off0, npairs, n0, 2.f * prm->voxel_size,
                                        prm->max_iter, prm->max_break_count, prm->break_threshold_ratio,
                                        prm->skip_refinement, 1, eps, weights, res_dev, stream));
  DGR_CHECK(tm.rec(4, 1));

  // The batch's results and its error flag land in PINNED host memory through two asynchronous copies enqueued in FRONT of
  // the batch's one wait.  (Round 6, first half: the wait came before a pageable result copy, because a device-to-host copy
  // into pageable memory blocks inside the runtime, busy-waiting, until the stream has drained -- issued first it WAS the
  // wait, and the driver threads spun at 98 % of the wall time.  That left two blocking copies + two synchronisations
  // behind the wait: ~0.2 ms per call with nothing on the GPU, profiles/r06_timeline_s1_b6.csv.gz.)
  unsigned char *pin;
  DGR_CHECK(dgr_ctx_pinned(ctx, 64 + (size_t)npairs * sizeof(DgrRegResult), &pin));
  const DgrRegResult *res = reinterpret_cast<const DgrRegResult *>(pin + 64);
  DGR_HIP_CHECK(hipMemcpyAsync(pin + 64, res_dev, (size_t)npairs * sizeof(DgrRegResult), hipMemcpyDeviceToHost, stream));
  DGR_HIP_CHECK(hipMemcpyAsync(pin, ctx->flag_dev, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  DGR_CHECK(dgr_ctx_wait(ctx, stream, (long)(ctx->batch_ns_per_row * (double)(n0 + n1))));
  ctx->batch_ns_per_row = (double)ctx->last_wait_ns / (double)(n0 + n1);
  DGR_CHECK(dgr_flag_error(*reinterpret_cast<volatile int32_t *>(pin)));
  for (int p = 0; p < npairs; ++p)
    if (res[p].status == DGR_STATUS_EXCHANGE_TIMEOUT) {
      dgr_set_error("registration of pair %d: the workgroups sharing the pair lost each other (exchange timed out)", p);
      return DGR_EINTERNAL;
    }
  for (int p = 0; p < npairs; ++p) {
    float *T = T_out + p * 16;
    const DgrRegResult &r = res[p];
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.f : 0.f;
    if (r.status == DGR_STATUS_OK) {
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[i * 4 + j] = r.R[i * 3 + j];
        T[i * 4 + 3] = r.t[i];
      }
    }
    status_out[p] = r.status;
    if (stats_out) {
      stats_out[p * 4 + 0] = (float)r.iterations;
      stats_out[p * 4 + 1] = r.loss;
      stats_out[p * 4 + 2] = (float)r.break_count;
      stats_out[p * 4 + 3] = r.wsum;
    }
  }
  // the same transforms in float64 (the reference returns np.float64, :290-291, and Open3D's results are doubles): the
  // network estimate is f32 arithmetic widened exactly; the RANSAC / ICP results below replace their pairs at full width
  ctx->last_T64.assign((size_t)npairs * 16, 0.0);
  for (int i = 0; i < npairs * 16; ++i) ctx->last_T64[i] = (double)T_out[i];
  ctx->last.ptr[0] = idx1;    ctx->last.numel[0] = n0;
  ctx->last.ptr[1] = logit;   ctx->last.numel[1] = n0;
  ctx->last.ptr[2] = weights; ctx->last.numel[2] = n0;
  ctx->last.ptr[3] = F0;      ctx->last.numel[3] = n0 * C;
  ctx->last.ptr[4] = F1;      ctx->last.numel[4] = n1 * C;
  ctx->last.generation = ctx->arena.generation;
  // The two Open3D steps that end register() (:302-322) for the whole batch, without leaving the library and with two
  // stream synchronisations in total: every failing pair's RANSAC and every pair's ICP set-up are enqueued back to
  // back; the ICP takes its initial T on the device (a RANSAC result record, or the uploaded estimate).  The batch
  // outputs above are complete at this point: a pair whose ICP cannot run keeps its estimate and gets its own status.
  DGR_CHECK(tm.rec(8, 0));
  if (prm->safeguard || prm->use_icp) {
    const DgrArena::Mark mk = A.mark();
    // host landing buffers of the asynchronous copies below: they must outlive every copy in flight, so an error in the
    // middle of the block synchronises the stream before this scope is left (the lambda's callers below)
    std::vector<double *> rs(npairs, nullptr);
    std::vector<double> Th((size_t)npairs * 16), rsh((size_t)npairs * DGR_RANSAC_RESULT_DOUBLES);
    std::vector<DgrIcpJob> jobs(prm->use_icp ? npairs : 0);
    auto tail = [&]() -> int {
      if (prm->safeguard)
        for (int p = 0; p < npairs; ++p) {
          if (res[p].status != DGR_STATUS_LOW_CONFIDENCE) continue;
          // Case 1 (:302-315): RANSAC over the putative correspondences xyz0[i] <-> xyz1[idx1[i]]
          const int64_t m0 = off0[p + 1] - off0[p];
          float *Y;
          DGR_ALLOC(Y, A, float, m0 * 3);
          gather_rows3_kernel<<<(int)dgr_ceil_div(m0, 256), 256, 0, stream>>>(xyz1, idx1 + off0[p], m0, Y);
          DGR_CHECK(dgr_ransac_begin(ctx, xyz0 + off0[p] * 3, Y, m0, 2.0 * prm->voxel_size,
                                     prm->ransac_hypotheses > 0 ? prm->ransac_hypotheses : 4000000, prm->ransac_seed, &rs[p],
                                     stream));
          DGR_HIP_CHECK(hipMemcpyAsync(&rsh[(size_t)p * DGR_RANSAC_RESULT_DOUBLES], rs[p],
                                       DGR_RANSAC_RESULT_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, stream));
          status_out[p] = DGR_STATUS_SAFEGUARD;
        }
      if (prm->use_icp) {   // :317-322: max correspondence distance 2 voxel, Open3D defaults (1e-6, 1e-6, 30)
        double *Tinit;
        DGR_ALLOC(Tinit, A, double, (int64_t)npairs * 16);
        for (int i = 0; i < npairs * 16; ++i) Th[i] = T_out[i];
        DGR_HIP_CHECK(hipMemcpyAsync(Tinit, Th.data(), Th.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        for (int p = 0; p < npairs; ++p)
          DGR_CHECK(dgr_icp_begin(ctx, xyz0 + off0[p] * 3, off0[p + 1] - off0[p], xyz1 + off1[p] * 3, off1[p + 1] - off1[p],
                                  2.0 * prm->voxel_size, rs[p] ? rs[p] : Tinit + (int64_t)p * 16, &jobs[p], stream));
      }
      DGR_HIP_CHECK(hipStreamSynchronize(stream));
      for (int p = 0; p < npairs; ++p)
        if (rs[p])
          for (int i = 0; i < 16; ++i) {
            ctx->last_T64[(size_t)p * 16 + i] = rsh[(size_t)p * DGR_RANSAC_RESULT_DOUBLES + i];
            T_out[p * 16 + i] = (float)rsh[(size_t)p * DGR_RANSAC_RESULT_DOUBLES + i];
          }
      if (prm->use_icp) {
        std::vector<char> ran(npairs, 0);
        for (int p = 0; p < npairs; ++p) {
          if (jobs[p].ncell < 1 || jobs[p].ncell > (4 << 20)) {
            // no finite target point: the estimate stands, the code keeps saying where it came from
            status_out[p] |= DGR_STATUS_FLAG_ICP_SKIPPED;
            continue;
          }
          DGR_CHECK(dgr_icp_run(ctx, &jobs[p], 30, 1e-6, 1e-6, stream));
          ran[p] = 1;
        }
        DGR_HIP_CHECK(hipStreamSynchronize(stream));
        for (int p = 0; p < npairs; ++p) {
          if (!ran[p]) continue;
          double Td[16];
          dgr_icp_finish(&jobs[p], Td, nullptr);
          for (int i = 0; i < 16; ++i) {
            ctx->last_T64[(size_t)p * 16 + i] = Td[i];
            T_out[p * 16 + i] = (float)Td[i];
          }
        }
      }
      return DGR_OK;
    };
    const int rc = tail();
    if (rc != DGR_OK) (void)hipStreamSynchronize(stream);   // nothing may still be copying into rsh / jobs[] / Th
    A.rewind(mk);   // the stream is synchronised on both paths
    if (rc != DGR_OK) return rc;
  }
  DGR_CHECK(tm.rec(8, 1));
  if (ctx->profiling) {
    memset(ctx->stage_ms, 0, sizeof(ctx->stage_ms));
    for (int s = 0; s < 5; ++s) DGR_HIP_CHECK(hipEventElapsedTime(&ctx->stage_ms[s], tm.e[s][0], tm.e[s][1]));
    DGR_HIP_CHECK(hipStreamSynchronize(stream));
    DGR_HIP_CHECK(hipEventElapsedTime(&ctx->stage_ms[8], tm.e[8][0], tm.e[8][1]));
    DGR_CHECK(dgr_ctx_collect_profile(ctx));
  }
  return DGR_OK;
}

extern "C" int dgr_register_batch_f64(dgr_ctx *ctx, double *T_out, int64_t capacity_pairs, int64_t *npairs) {
  DGR_REQUIRE(ctx && npairs, "dgr_register_batch_f64: NULL argument");
  const int64_t n = (int64_t)(ctx->last_T64.size() / 16);
  DGR_REQUIRE(n > 0, "no dgr_register_batch has run on this ctx");
  *npairs = n;
  if (T_out) {
    DGR_REQUIRE(capacity_pairs >= n, "destination buffer too small");
    memcpy(T_out, ctx->last_T64.data(), (size_t)n * 16 * sizeof(double));
  }
  return DGR_OK;
}

extern "C" int dgr_register_batch_output(dgr_ctx *ctx, int which, void *dst_dev, int64_t capacity_bytes,
                                         int64_t *numel, dgr_stream stream) {
  DGR_REQUIRE(ctx && numel && which >= 0 && which < 5, "dgr_register_batch_output: bad argument");
  DGR_REQUIRE(ctx->last.ptr[which] != nullptr, "no dgr_register_batch has run on this ctx");
  DGR_REQUIRE(ctx->last.generation == ctx->arena.generation,
              "the outputs of the last dgr_register_batch are gone: a later call on this context reused its workspace "
              "(fetch them before the next library call)");
  *numel = ctx->last.numel[which];
  if (dst_dev) {
    const int64_t bytes = ctx->last.numel[which] * (which == 0 ? 8 : 4);
    DGR_REQUIRE(capacity_bytes >= bytes, "destination buffer too small");
    DGR_HIP_CHECK(hipMemcpyAsync(dst_dev, ctx->last.ptr[which], (size_t)bytes, hipMemcpyDeviceToDevice,
                                 (hipStream_t)stream));
  }
  return DGR_OK;
}
