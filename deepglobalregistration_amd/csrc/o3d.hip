// The two Open3D routines around the learned path of DeepGlobalRegistration.register():
//   * point-to-point ICP            (core/deep_global_registration.py:317-322, registration_icp)
//   * safeguard RANSAC from the putative correspondences (:50-64, 302-315,
//     registration_ransac_based_on_correspondence, ransac_n = 4, no checkers, all hypotheses evaluated)
// restated from the published Open3D 0.17 algorithm (oracle/open3d_reg.py has the CPU restatement and the
// deviations that make RANSAC reproducible: counter-based sampling, f32 consensus test without fma).
// Everything is stream-ordered and free of host round trips until the final result copy.
#include "dgr_internal.h"
#include "svd3.h"

int dgr_icp_impl(dgr_ctx *ctx, const float *src, int64_t N0, const float *dst, int64_t N1, double max_dist,
                 const double *T_init, int max_iter, double rel_fitness, double rel_rmse, double *T_out,
                 double *stats_out, hipStream_t stream);
int dgr_ransac_impl(dgr_ctx *ctx, const float *X, const float *Y, int64_t N, double max_dist, int64_t num_hypotheses,
                    uint32_t seed, double *T_out, double *stats_out, hipStream_t stream);
#include <cstddef>
#include <cstring>

// ------------------------------------------------------------------------------------------------
// Umeyama / Kabsch without scaling from the sums  n, sum p, sum q, sum q p^T  (f64)
//   T = [R | t],  R = U diag(1,1,sign) V^T of sigma = sum q p^T / n - mu_q mu_p^T,  t = mu_q - R mu_p
// ------------------------------------------------------------------------------------------------
__device__ inline void umeyama_from_sums(double n, const double *Sp, const double *Sq, const double *Sqp,
                                         double R[9], double t[3]) {
  for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  t[0] = t[1] = t[2] = 0.0;
  if (!(n > 0.0)) return;
  double mp[3], mq[3], sig[9];
  for (int d = 0; d < 3; ++d) { mp[d] = Sp[d] / n; mq[d] = Sq[d] / n; }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) sig[i * 3 + j] = Sqp[i * 3 + j] / n - mq[i] * mp[j];
  double U[9], sv[3], V[9];
  svd3(sig, U, sv, V);
  const double sg = (det3(U) * det3(V) < 0.0) ? -1.0 : 1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      R[i * 3 + j] = U[i * 3 + 0] * V[j * 3 + 0] + U[i * 3 + 1] * V[j * 3 + 1] + sg * U[i * 3 + 2] * V[j * 3 + 2];
  for (int i = 0; i < 3; ++i) t[i] = mq[i] - (R[i * 3] * mp[0] + R[i * 3 + 1] * mp[1] + R[i * 3 + 2] * mp[2]);
}

// ================================================================================================
// ICP
// ================================================================================================
constexpr int ICP_THREADS = 256;
constexpr int ICP_NSUM = 17;   // n, sum p (3), sum q (3), sum q p^T (9), sum d^2

struct IcpState {
  double T[16];        // accumulated transformation (row-major 4x4)
  double upd[12];      // last update [R | t], applied by the next step
  double fitness, rmse, prev_fitness, prev_rmse;
  double gmin[3], cell;
  int32_t gdim[3];
  int32_t done, iters, have_prev, ncell;
  uint32_t bmin[3], bmax[3];   // ordered-uint bounding box of the target
};

__device__ __forceinline__ uint32_t ord_f32(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float unord_f32(uint32_t k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

__global__ void icp_init_kernel(IcpState *st, const double *T_init) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < 16; ++i) st->T[i] = T_init[i];
  for (int i = 0; i < 12; ++i) st->upd[i] = (i == 0 || i == 4 || i == 8) ? 1.0 : 0.0;   // [R | t] = identity
  st->fitness = st->rmse = st->prev_fitness = st->prev_rmse = 0.0;
  st->done = st->iters = st->have_prev = 0;
  for (int d = 0; d < 3; ++d) { st->bmin[d] = 0xffffffffu; st->bmax[d] = 0u; }
}

__global__ void __launch_bounds__(ICP_THREADS)
    icp_bbox_kernel(const float *__restrict__ dst, int64_t n, IcpState *st) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
  if (i < n && isfinite(dst[i * 3] + dst[i * 3 + 1] + dst[i * 3 + 2]))   // non-finite target points take no part
    for (int d = 0; d < 3; ++d) lo[d] = hi[d] = ord_f32(dst[i * 3 + d]);
  for (int s = 32; s >= 1; s >>= 1)
    for (int d = 0; d < 3; ++d) {
      lo[d] = min(lo[d], (uint32_t)__shfl_xor((int)lo[d], s, 64));
      hi[d] = max(hi[d], (uint32_t)__shfl_xor((int)hi[d], s, 64));
    }
  if ((threadIdx.x & 63) == 0)
    for (int d = 0; d < 3; ++d) { atomicMin(&st->bmin[d], lo[d]); atomicMax(&st->bmax[d], hi[d]); }
}

// uniform grid over the target's bounding box, cell edge >= max_dist (doubled until it fits the budget)
__global__ void icp_layout_kernel(IcpState *st, double max_dist, int32_t cell_cap) {
  if (threadIdx.x != 0) return;
  double lo[3], hi[3];
  for (int d = 0; d < 3; ++d) { lo[d] = (double)unord_f32(st->bmin[d]); hi[d] = (double)unord_f32(st->bmax[d]); }
  double cell = max_dist > 0.0 ? max_dist : 1.0;
  for (int tries = 0; tries < 64; ++tries) {
    double total = 1.0;
    for (int d = 0; d < 3; ++d) total *= floor((hi[d] - lo[d]) / cell) + 1.0;
    if (total <= (double)cell_cap) break;
    cell *= 2.0;
  }
  int64_t total = 1;
  for (int d = 0; d < 3; ++d) {
    st->gmin[d] = lo[d];
    st->gdim[d] = (int32_t)(floor((hi[d] - lo[d]) / cell) + 1.0);
    total *= st->gdim[d];
  }
  st->cell = cell;
  st->ncell = (int32_t)total;
}

__device__ __forceinline__ int icp_cell_of(const IcpState *st, double x, double y, double z) {
  const int cx = (int)floor((x - st->gmin[0]) / st->cell), cy = (int)floor((y - st->gmin[1]) / st->cell),
            cz = (int)floor((z - st->gmin[2]) / st->cell);
  return (cz * st->gdim[1] + cy) * st->gdim[0] + cx;
}

__global__ void icp_count_kernel(const float *__restrict__ dst, int64_t n, const IcpState *st, int32_t *counts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !isfinite(dst[i * 3] + dst[i * 3 + 1] + dst[i * 3 + 2])) return;
  atomicAdd(&counts[icp_cell_of(st, dst[i * 3], dst[i * 3 + 1], dst[i * 3 + 2])], 1);
}

__global__ void icp_fill_kernel(const float *__restrict__ dst, int64_t n, const IcpState *st,
                                const int32_t *__restrict__ starts, int32_t *cursor, double *__restrict__ sorted) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !isfinite(dst[i * 3] + dst[i * 3 + 1] + dst[i * 3 + 2])) return;
  const int c = icp_cell_of(st, dst[i * 3], dst[i * 3 + 1], dst[i * 3 + 2]);
  const int pos = starts[c] + atomicAdd(&cursor[c], 1);
  for (int d = 0; d < 3; ++d) sorted[(int64_t)pos * 3 + d] = (double)dst[i * 3 + d];
}

__global__ void icp_transform_init_kernel(const float *__restrict__ src, int64_t n, const IcpState *st, double *__restrict__ P) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = src[i * 3], y = src[i * 3 + 1], z = src[i * 3 + 2];
  for (int d = 0; d < 3; ++d)
    P[i * 3 + d] = st->T[d * 4] * x + st->T[d * 4 + 1] * y + st->T[d * 4 + 2] * z + st->T[d * 4 + 3];
}

// one ICP evaluation: apply the pending update to the source, find every point's nearest target point
// within max_dist (27 cells), and reduce the 17 Umeyama / fitness sums per block (fixed order)
__global__ void __launch_bounds__(ICP_THREADS)
    icp_step_kernel(double *__restrict__ P, int64_t n, const IcpState *st, const int32_t *__restrict__ starts,
                    const double *__restrict__ sorted, double max_dist, double *__restrict__ partial) {
  __shared__ double red[ICP_THREADS / 64][ICP_NSUM];
  if (st->done) return;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double s[ICP_NSUM];
  for (int k = 0; k < ICP_NSUM; ++k) s[k] = 0.0;
  if (i < n) {
    const double x = P[i * 3], y = P[i * 3 + 1], z = P[i * 3 + 2];
    const double *u = st->upd;
    const double px = u[0] * x + u[1] * y + u[2] * z + u[9];
    const double py = u[3] * x + u[4] * y + u[5] * z + u[10];
    const double pz = u[6] * x + u[7] * y + u[8] * z + u[11];
    P[i * 3] = px; P[i * 3 + 1] = py; P[i * 3 + 2] = pz;
    // cell of the point, clamped one cell outside the grid (farther points cannot have a neighbour)
    const double fx = floor((px - st->gmin[0]) / st->cell), fy = floor((py - st->gmin[1]) / st->cell),
                 fz = floor((pz - st->gmin[2]) / st->cell);
    const int cx = (int)fmin(fmax(fx, -2.0), (double)st->gdim[0] + 1.0), cy = (int)fmin(fmax(fy, -2.0), (double)st->gdim[1] + 1.0),
              cz = (int)fmin(fmax(fz, -2.0), (double)st->gdim[2] + 1.0);
    double best = max_dist * max_dist, q[3] = {0, 0, 0};
    bool found = false;
    for (int dz = -1; dz <= 1; ++dz) {
      const int zz = cz + dz;
      if (zz < 0 || zz >= st->gdim[2]) continue;
      for (int dy = -1; dy <= 1; ++dy) {
        const int yy = cy + dy;
        if (yy < 0 || yy >= st->gdim[1]) continue;
        for (int dx = -1; dx <= 1; ++dx) {
          const int xx = cx + dx;
          if (xx < 0 || xx >= st->gdim[0]) continue;
          const int c = (zz * st->gdim[1] + yy) * st->gdim[0] + xx;
          for (int p = starts[c]; p < starts[c + 1]; ++p) {
            const double ex = sorted[(int64_t)p * 3] - px, ey = sorted[(int64_t)p * 3 + 1] - py,
                         ez = sorted[(int64_t)p * 3 + 2] - pz;
            const double d2 = ex * ex + ey * ey + ez * ez;
            if (d2 <= best) {   // radius test and running minimum in one
              if (d2 < best || !found) { q[0] = sorted[(int64_t)p * 3]; q[1] = sorted[(int64_t)p * 3 + 1]; q[2] = sorted[(int64_t)p * 3 + 2]; }
              best = d2;
              found = true;
            }
          }
        }
      }
    }
    if (found) {
      s[0] = 1.0;
      s[1] = px; s[2] = py; s[3] = pz;
      s[4] = q[0]; s[5] = q[1]; s[6] = q[2];
      const double pv[3] = {px, py, pz};
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) s[7 + a * 3 + b] = q[a] * pv[b];
      s[16] = best;
    }
  }
  for (int k = 0; k < ICP_NSUM; ++k) {
    double v = s[k];
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < ICP_NSUM) {
    double v = 0.0;
    for (int w = 0; w < ICP_THREADS / 64; ++w) v += red[w][threadIdx.x];
    partial[(int64_t)blockIdx.x * ICP_NSUM + threadIdx.x] = v;
  }
}

// sums of all blocks (fixed order) -> fitness / rmse, convergence test (Registration.cpp: RegistrationICP),
// next update and accumulated transformation
__global__ void __launch_bounds__(ICP_THREADS)
    icp_solve_kernel(IcpState *st, const double *__restrict__ partial, int nblocks, int64_t n_src, int max_iter,
                     double rel_fitness, double rel_rmse) {
  __shared__ double red[ICP_THREADS][ICP_NSUM];
  if (st->done) return;
  for (int k = 0; k < ICP_NSUM; ++k) {
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += ICP_THREADS) v += partial[(int64_t)b * ICP_NSUM + k];
    red[threadIdx.x][k] = v;
  }
  __syncthreads();
  for (int s = ICP_THREADS / 2; s >= 1; s >>= 1) {
    if (threadIdx.x < s)
      for (int k = 0; k < ICP_NSUM; ++k) red[threadIdx.x][k] += red[threadIdx.x + s][k];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double *S = red[0];
  const double n = S[0];
  const double fitness = n / (double)(n_src > 0 ? n_src : 1);
  const double rmse = n > 0.0 ? sqrt(S[16] / n) : 0.0;
  st->fitness = fitness;
  st->rmse = rmse;
  if (st->have_prev && fabs(st->prev_fitness - fitness) < rel_fitness && fabs(st->prev_rmse - rmse) < rel_rmse) {
    st->done = 1;
    return;
  }
  if (st->iters >= max_iter) {
    st->done = 1;
    return;
  }
  double R[9], t[3];
  umeyama_from_sums(n, S + 1, S + 4, S + 7, R, t);
  double Tn[16];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 4; ++j)
      Tn[i * 4 + j] = R[i * 3] * st->T[j] + R[i * 3 + 1] * st->T[4 + j] + R[i * 3 + 2] * st->T[8 + j] + (j == 3 ? t[i] : 0.0);
  }
  Tn[12] = 0.0; Tn[13] = 0.0; Tn[14] = 0.0; Tn[15] = 1.0;
  for (int i = 0; i < 16; ++i) st->T[i] = Tn[i];
  for (int i = 0; i < 9; ++i) st->upd[i] = R[i];
  for (int i = 0; i < 3; ++i) st->upd[9 + i] = t[i];
  st->prev_fitness = fitness;
  st->prev_rmse = rmse;
  st->have_prev = 1;
  st->iters += 1;
}

extern "C" int dgr_icp_point_to_point(dgr_ctx *ctx, const float *src, int64_t N0, const float *dst, int64_t N1,
                                      double max_dist, const double *T_init, int max_iter, double rel_fitness,
                                      double rel_rmse, double *T_out, double *stats_out, dgr_stream stream_) {
  DGR_REQUIRE(ctx && src && dst && T_out, "dgr_icp_point_to_point: NULL argument");
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  DGR_CHECK(ctx->arena.reset());
  return dgr_icp_impl(ctx, src, N0, dst, N1, max_dist, T_init, max_iter, rel_fitness, rel_rmse, T_out, stats_out,
                      (hipStream_t)stream_);
}

// ICP in three host phases, so that a batch of pairs needs two stream synchronisations in total instead of two per pair
// (pipeline.hip): begin = scratch + target bounding box / grid layout (the cell count comes back to the host), run = grid
// build + all iterations + the state's way back, finish = results out of the job.  All scratch is taken behind the
// caller's arena allocations; the caller rewinds once the stream has been synchronised.
constexpr int32_t ICP_CELL_CAP = 4 << 20;

__global__ void icp_tinit_kernel(const double *T_init, double *Tdev) {
  if (threadIdx.x < 16) Tdev[threadIdx.x] = T_init ? T_init[threadIdx.x] : ((threadIdx.x % 5 == 0) ? 1.0 : 0.0);
}

int dgr_icp_begin(dgr_ctx *ctx, const float *src, int64_t N0, const float *dst, int64_t N1, double max_dist,
                  const double *T_init_dev, DgrIcpJob *job, hipStream_t stream) {
  DGR_REQUIRE(N0 > 0 && N1 > 0, "ICP: empty point cloud (N0=%lld, N1=%lld)", (long long)N0, (long long)N1);
  DGR_REQUIRE(max_dist > 0.0, "ICP: bad max_dist");
  DgrArena &A = ctx->arena;
  IcpState *st;
  double *Tdev;
  job->src = src; job->dst = dst; job->N0 = N0; job->N1 = N1; job->max_dist = max_dist;
  job->nblocks = (int)dgr_ceil_div(N0, ICP_THREADS);
  DGR_ALLOC(st, A, IcpState, 1);
  DGR_ALLOC(Tdev, A, double, 16);
  DGR_ALLOC(job->P, A, double, N0 * 3);
  DGR_ALLOC(job->sorted, A, double, N1 * 3);
  DGR_ALLOC(job->partial, A, double, (int64_t)job->nblocks * ICP_NSUM);
  job->st = st;
  icp_tinit_kernel<<<1, 64, 0, stream>>>(T_init_dev, Tdev);
  icp_init_kernel<<<1, 64, 0, stream>>>(st, Tdev);
  icp_bbox_kernel<<<(int)dgr_ceil_div(N1, ICP_THREADS), ICP_THREADS, 0, stream>>>(dst, N1, st);
  icp_layout_kernel<<<1, 64, 0, stream>>>(st, max_dist, ICP_CELL_CAP);
  DGR_LAUNCH_CHECK();
  // the actual cell count bounds the clears and the scan of the grid build -- a 3DMatch fragment at 10 cm cells has
  // ~10^5 cells, not the 4 M of the budget
  job->ncell = 0;
  DGR_HIP_CHECK(hipMemcpyAsync(&job->ncell, &st->ncell, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  return DGR_OK;
}

// after the stream has been synchronised behind dgr_icp_begin
int dgr_icp_run(dgr_ctx *ctx, DgrIcpJob *job, int max_iter, double rel_fitness, double rel_rmse, hipStream_t stream) {
  DGR_REQUIRE(max_iter >= 0 && max_iter <= 10000, "ICP: bad max_iter");
  DGR_REQUIRE(job->ncell >= 1 && job->ncell <= ICP_CELL_CAP, "ICP: target grid of %d cells (non-finite or empty target cloud?)",
              job->ncell);
  DgrArena &A = ctx->arena;
  IcpState *st = static_cast<IcpState *>(job->st);
  int32_t *counts, *starts, *cursor;
  DGR_ALLOC(counts, A, int32_t, job->ncell + 1);
  DGR_ALLOC(starts, A, int32_t, job->ncell + 1);
  DGR_ALLOC(cursor, A, int32_t, job->ncell + 1);
  DGR_HIP_CHECK(hipMemsetAsync(counts, 0, (size_t)(job->ncell + 1) * sizeof(int32_t), stream));
  DGR_HIP_CHECK(hipMemsetAsync(cursor, 0, (size_t)(job->ncell + 1) * sizeof(int32_t), stream));
  icp_count_kernel<<<(int)dgr_ceil_div(job->N1, 256), 256, 0, stream>>>(job->dst, job->N1, st, counts);
  DGR_CHECK(dgr_exclusive_scan_i32(A, counts, starts, (int64_t)job->ncell + 1, nullptr, stream));
  icp_fill_kernel<<<(int)dgr_ceil_div(job->N1, 256), 256, 0, stream>>>(job->dst, job->N1, st, starts, cursor, job->sorted);
  icp_transform_init_kernel<<<(int)dgr_ceil_div(job->N0, 256), 256, 0, stream>>>(job->src, job->N0, st, job->P);
  DGR_LAUNCH_CHECK();
  for (int it = 0; it <= max_iter; ++it) {
    icp_step_kernel<<<job->nblocks, ICP_THREADS, 0, stream>>>(job->P, job->N0, st, starts, job->sorted, job->max_dist, job->partial);
    icp_solve_kernel<<<1, ICP_THREADS, 0, stream>>>(st, job->partial, job->nblocks, job->N0, max_iter, rel_fitness, rel_rmse);
  }
  DGR_LAUNCH_CHECK();
  static_assert(sizeof(job->host_state) >= sizeof(IcpState), "host copy of the ICP state");
  DGR_HIP_CHECK(hipMemcpyAsync(job->host_state, st, sizeof(IcpState), hipMemcpyDeviceToHost, stream));
  return DGR_OK;
}

// after the stream has been synchronised behind dgr_icp_run
void dgr_icp_finish(const DgrIcpJob *job, double *T_out, double *stats_out) {
  IcpState host;
  memcpy(&host, job->host_state, sizeof(IcpState));
  memcpy(T_out, host.T, sizeof(double) * 16);
  if (stats_out) {
    stats_out[0] = host.fitness;
    stats_out[1] = host.rmse;
    stats_out[2] = (double)host.iters;
  }
}

// one ICP (the stand-alone entry point): T_init is a host matrix
int dgr_icp_impl(dgr_ctx *ctx, const float *src, int64_t N0, const float *dst, int64_t N1, double max_dist,
                 const double *T_init, int max_iter, double rel_fitness, double rel_rmse, double *T_out,
                 double *stats_out, hipStream_t stream) {
  DgrArena &A = ctx->arena;
  const DgrArena::Mark icp_mark = A.mark();
  double *Ti_dev = nullptr;
  if (T_init) {
    DGR_ALLOC(Ti_dev, A, double, 16);
    DGR_HIP_CHECK(hipMemcpyAsync(Ti_dev, T_init, 16 * sizeof(double), hipMemcpyHostToDevice, stream));
  }
  DgrIcpJob job;
  DGR_CHECK(dgr_icp_begin(ctx, src, N0, dst, N1, max_dist, Ti_dev, &job, stream));
  DGR_HIP_CHECK(hipStreamSynchronize(stream));
  DGR_CHECK(dgr_icp_run(ctx, &job, max_iter, rel_fitness, rel_rmse, stream));
  DGR_HIP_CHECK(hipStreamSynchronize(stream));
  dgr_icp_finish(&job, T_out, stats_out);
  A.rewind(icp_mark);   // the stream was synchronised above: nothing is in flight on the scratch
  return DGR_OK;
}

// ================================================================================================
// RANSAC from correspondences
// ================================================================================================
constexpr int RS_THREADS = 256;
constexpr int RS_TILE = 512;      // correspondences per LDS tile

__device__ __forceinline__ uint32_t rs_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}
// draw j of hypothesis h (oracle/open3d_reg.py: ransac_samples)
__device__ __forceinline__ int rs_sample(uint32_t seed, int64_t h, int j, int64_t n) {
  const uint32_t key = (uint32_t)((uint64_t)h * 4ull + (uint64_t)j + 0x9e3779b9ull * (uint64_t)seed);
  const uint32_t r = rs_mix32(rs_mix32(key) ^ 0x68bc21ebu);
  return (int)(((uint64_t)r * (uint64_t)n) >> 32);
}

// 4-point Umeyama of hypothesis h in f64 (R, t row-major)
__device__ inline void rs_hypothesis(const float *__restrict__ X, const float *__restrict__ Y, int64_t n, uint32_t seed,
                                     int64_t h, double R[9], double t[3]) {
  double Sp[3] = {0, 0, 0}, Sq[3] = {0, 0, 0}, Sqp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < 4; ++j) {
    const int64_t c = rs_sample(seed, h, j, n);
    const double p[3] = {X[c * 3], X[c * 3 + 1], X[c * 3 + 2]}, q[3] = {Y[c * 3], Y[c * 3 + 1], Y[c * 3 + 2]};
    for (int a = 0; a < 3; ++a) {
      Sp[a] += p[a];
      Sq[a] += q[a];
      for (int b = 0; b < 3; ++b) Sqp[a * 3 + b] += q[a] * p[b];
    }
  }
  umeyama_from_sums(4.0, Sp, Sq, Sqp, R, t);
}

struct RsBest {
  int32_t count;
  float err;
  int64_t h;
};
__device__ __forceinline__ bool rs_better(int32_t c, float e, int64_t h, int32_t c2, float e2, int64_t h2) {
  if (c != c2) return c > c2;
  if (e != e2) return e < e2;
  return h < h2;
}

// thread = hypothesis; the correspondences stream through LDS and are read as broadcasts
__global__ void __launch_bounds__(RS_THREADS)
    ransac_eval_kernel(const float *__restrict__ X, const float *__restrict__ Y, int64_t n, uint32_t seed, int64_t num_hyp,
                       float thr2, RsBest *__restrict__ block_best) {
  __shared__ __attribute__((aligned(16))) float tx[RS_TILE * 3], ty[RS_TILE * 3];
  __shared__ RsBest wbest[RS_THREADS / 64];
  const int64_t h = (int64_t)blockIdx.x * RS_THREADS + threadIdx.x;
  float R[9], t[3];
  {
    double Rd[9], td[3];
    rs_hypothesis(X, Y, n, seed, h < num_hyp ? h : 0, Rd, td);
    for (int i = 0; i < 9; ++i) R[i] = (float)Rd[i];
    for (int i = 0; i < 3; ++i) t[i] = (float)td[i];
  }
  int32_t count = 0;
  float err = 0.f;
  for (int64_t j0 = 0; j0 < n; j0 += RS_TILE) {
    const int m = (int)min((int64_t)RS_TILE, n - j0);
    __syncthreads();
    for (int e = threadIdx.x; e < m * 3; e += RS_THREADS) { tx[e] = X[j0 * 3 + e]; ty[e] = Y[j0 * 3 + e]; }
    __syncthreads();
    {
#pragma clang fp contract(off)   // the consensus test is DEFINED without fma (oracle/open3d_reg.py)
      for (int j = 0; j < m; ++j) {
        const float x = tx[j * 3], y = tx[j * 3 + 1], z = tx[j * 3 + 2];
        const float p0 = ((R[0] * x + R[1] * y) + R[2] * z) + t[0];
        const float p1 = ((R[3] * x + R[4] * y) + R[5] * z) + t[1];
        const float p2 = ((R[6] * x + R[7] * y) + R[8] * z) + t[2];
        const float e0 = p0 - ty[j * 3], e1 = p1 - ty[j * 3 + 1], e2 = p2 - ty[j * 3 + 2];
        const float d2 = (e0 * e0 + e1 * e1) + e2 * e2;
        const bool in = d2 < thr2;
        count += in ? 1 : 0;
        err = in ? err + d2 : err;   // sequential f32 sum over the inliers, correspondence order
      }
    }
  }
  if (h >= num_hyp) { count = -1; err = 0.f; }
  // block best
  int32_t bc = count; float be = err; int64_t bh = h;
  for (int o = 32; o >= 1; o >>= 1) {
    const int32_t oc = __shfl_xor(bc, o, 64);
    const float oe = __shfl_xor(be, o, 64);
    const int64_t oh = ((int64_t)__shfl_xor((int)(bh >> 32), o, 64) << 32) | (uint32_t)__shfl_xor((int)(bh & 0xffffffff), o, 64);
    if (rs_better(oc, oe, oh, bc, be, bh)) { bc = oc; be = oe; bh = oh; }
  }
  if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = {bc, be, bh};
  __syncthreads();
  if (threadIdx.x == 0) {
    RsBest b = wbest[0];
    for (int w = 1; w < RS_THREADS / 64; ++w)
      if (rs_better(wbest[w].count, wbest[w].err, wbest[w].h, b.count, b.err, b.h)) b = wbest[w];
    block_best[blockIdx.x] = b;
  }
}

struct RsResult {
  double T[16];
  double best_h, count, rmse;
};

__global__ void __launch_bounds__(256)
    ransac_final_kernel(const float *__restrict__ X, const float *__restrict__ Y, int64_t n, uint32_t seed,
                        const RsBest *__restrict__ block_best, int nblocks, RsResult *out) {
  __shared__ RsBest sb[256];
  RsBest b = {-1, 0.f, 0};
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    const RsBest c = block_best[i];
    if (rs_better(c.count, c.err, c.h, b.count, b.err, b.h)) b = c;
  }
  sb[threadIdx.x] = b;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (threadIdx.x < s && rs_better(sb[threadIdx.x + s].count, sb[threadIdx.x + s].err, sb[threadIdx.x + s].h,
                                     sb[threadIdx.x].count, sb[threadIdx.x].err, sb[threadIdx.x].h))
      sb[threadIdx.x] = sb[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  b = sb[0];
  double R[9], t[3];
  rs_hypothesis(X, Y, n, seed, b.h, R, t);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) out->T[i * 4 + j] = R[i * 3 + j];
    out->T[i * 4 + 3] = t[i];
  }
  out->T[12] = 0.0; out->T[13] = 0.0; out->T[14] = 0.0; out->T[15] = 1.0;
  out->best_h = (double)b.h;
  out->count = (double)b.count;
  out->rmse = b.count > 0 ? sqrt((double)b.err / (double)b.count) : 0.0;
}

extern "C" int dgr_ransac_correspondence(dgr_ctx *ctx, const float *X, const float *Y, int64_t N, double max_dist,
                                         int64_t num_hypotheses, uint32_t seed, double *T_out, double *stats_out,
                                         dgr_stream stream_) {
  DGR_REQUIRE(ctx && X && Y && T_out, "dgr_ransac_correspondence: NULL argument");
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  DGR_CHECK(ctx->arena.reset());
  return dgr_ransac_impl(ctx, X, Y, N, max_dist, num_hypotheses, seed, T_out, stats_out, (hipStream_t)stream_);
}

// enqueue only: the result record (T first: a device-side T_init for dgr_icp_begin) stays in the arena
int dgr_ransac_begin(dgr_ctx *ctx, const float *X, const float *Y, int64_t N, double max_dist, int64_t num_hypotheses,
                     uint32_t seed, double **result_dev, hipStream_t stream) {
  DGR_REQUIRE(N > 0 && N < (1ll << 31), "RANSAC: bad correspondence count %lld", (long long)N);
  DGR_REQUIRE(num_hypotheses > 0 && num_hypotheses <= (1ll << 30), "RANSAC: bad hypothesis count");
  DGR_REQUIRE(max_dist > 0.0, "RANSAC: bad distance threshold");
  const int nblocks = (int)dgr_ceil_div(num_hypotheses, RS_THREADS);
  RsBest *bb;
  RsResult *res;
  DGR_ALLOC(bb, ctx->arena, RsBest, nblocks);
  DGR_ALLOC(res, ctx->arena, RsResult, 1);
  const float md = (float)max_dist;
  ransac_eval_kernel<<<nblocks, RS_THREADS, 0, stream>>>(X, Y, N, seed, num_hypotheses, md * md, bb);
  ransac_final_kernel<<<1, 256, 0, stream>>>(X, Y, N, seed, bb, nblocks, res);
  DGR_LAUNCH_CHECK();
  static_assert(offsetof(RsResult, T) == 0 && sizeof(RsResult) == DGR_RANSAC_RESULT_DOUBLES * sizeof(double), "result record layout");
  *result_dev = reinterpret_cast<double *>(res);
  return DGR_OK;
}

int dgr_ransac_impl(dgr_ctx *ctx, const float *X, const float *Y, int64_t N, double max_dist, int64_t num_hypotheses,
                    uint32_t seed, double *T_out, double *stats_out, hipStream_t stream) {
  const DgrArena::Mark rs_mark = ctx->arena.mark();
  double *res;
  DGR_CHECK(dgr_ransac_begin(ctx, X, Y, N, max_dist, num_hypotheses, seed, &res, stream));
  double host[DGR_RANSAC_RESULT_DOUBLES];
  DGR_HIP_CHECK(hipMemcpyAsync(host, res, sizeof(host), hipMemcpyDeviceToHost, stream));
  DGR_HIP_CHECK(hipStreamSynchronize(stream));
  memcpy(T_out, host, sizeof(double) * 16);
  if (stats_out) {
    stats_out[0] = host[16];
    stats_out[1] = host[17];
    stats_out[2] = host[18];
  }
  ctx->arena.rewind(rs_mark);   // the stream was synchronised above
  return DGR_OK;
}
