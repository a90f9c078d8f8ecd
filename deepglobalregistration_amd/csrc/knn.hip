// Feature-space 1-nearest-neighbour search, exact L2, brute force.
// Replaces core.knn.find_knn_gpu (core/knn.py:23-74) + core.metrics.pdist (core/metrics.py:62-69).
//
// The reference materialises a [250, N1, 32] difference tensor per chunk (82 GB written and
// re-read per 26k x 24k pair).  Here nothing is materialised: a workgroup keeps QPT queries per
// thread in registers, streams tiles of F1 through LDS (broadcast ds_read_b128), evaluates the
// exact sum_c (a_c - b_c)^2 form in f32 on the vector ALUs (so the arg-min agrees with `pdist`
// instead of the cancellation-prone |a|^2+|b|^2-2ab expansion), keeps a running (min, argmin) per
// query, and merges the partial results of the F1 splits with one 64-bit atomicMin per query on
// the packed key (dist_bits << 32 | index): positive floats order like their bit patterns, and
// ties resolve to the smallest index like torch.min on the CPU.
#include "dgr_internal.h"

constexpr int KNN_THREADS = 256;
constexpr int KNN_TB = 64;  // F1 rows per LDS tile

template <int C, int QPT>
__global__ void __launch_bounds__(KNN_THREADS)
    knn1_kernel(const float *__restrict__ F0, int64_t N0, const float *__restrict__ F1, int64_t N1,
                int rows_per_split, unsigned long long *__restrict__ best) {
  __shared__ __attribute__((aligned(16))) float tile[KNN_TB * C];
  const int64_t q0 = ((int64_t)blockIdx.x * KNN_THREADS + threadIdx.x) * QPT;
  const int64_t j_begin = (int64_t)blockIdx.y * rows_per_split;
  const int64_t j_end = min(N1, j_begin + rows_per_split);
  float q[QPT][C];
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    const int64_t r = min(q0 + u, N0 - 1);
#pragma unroll
    for (int c = 0; c < C; c += 4) {
      const float4 v = *reinterpret_cast<const float4 *>(F0 + r * C + c);
      q[u][c] = v.x; q[u][c + 1] = v.y; q[u][c + 2] = v.z; q[u][c + 3] = v.w;
    }
  }
  float bd[QPT];
  int bi[QPT];
#pragma unroll
  for (int u = 0; u < QPT; ++u) { bd[u] = __builtin_inff(); bi[u] = 0x7fffffff; }

  for (int64_t j0 = j_begin; j0 < j_end; j0 += KNN_TB) {
    const int nrows = (int)min((int64_t)KNN_TB, j_end - j0);
    __syncthreads();
    for (int e = threadIdx.x; e < KNN_TB * C / 4; e += KNN_THREADS) {
      const int row = e / (C / 4);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < nrows) v = *reinterpret_cast<const float4 *>(F1 + (j0 + row) * C + (e % (C / 4)) * 4);
      *reinterpret_cast<float4 *>(tile + e * 4) = v;
    }
    __syncthreads();
    for (int jj = 0; jj < nrows; ++jj) {
      float d0[QPT], d1[QPT];
#pragma unroll
      for (int u = 0; u < QPT; ++u) { d0[u] = 0.f; d1[u] = 0.f; }
#pragma unroll
      for (int c = 0; c < C; c += 4) {
        const float4 b = *reinterpret_cast<const float4 *>(tile + jj * C + c);  // LDS broadcast
#pragma unroll
        for (int u = 0; u < QPT; ++u) {
          const float e0 = q[u][c] - b.x, e1 = q[u][c + 1] - b.y;
          const float e2 = q[u][c + 2] - b.z, e3 = q[u][c + 3] - b.w;
          d0[u] = fmaf(e0, e0, d0[u]);
          d1[u] = fmaf(e1, e1, d1[u]);
          d0[u] = fmaf(e2, e2, d0[u]);
          d1[u] = fmaf(e3, e3, d1[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < QPT; ++u) {
        const float d = d0[u] + d1[u];
        if (d < bd[u]) { bd[u] = d; bi[u] = (int)(j0 + jj); }  // strict <: first minimal index wins
      }
    }
  }
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    if (q0 + u < N0 && bi[u] != 0x7fffffff) {
      const unsigned long long key =
          ((unsigned long long)__float_as_uint(bd[u]) << 32) | (unsigned int)bi[u];
      atomicMin(best + q0 + u, key);
    }
  }
}

__global__ void knn1_finish(const unsigned long long *__restrict__ best, int64_t N0, int squared,
                            int64_t *__restrict__ idx_out, float *__restrict__ dist_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N0) return;
  const unsigned long long k = best[i];
  idx_out[i] = (k == ~0ull) ? 0 : (int64_t)(k & 0xffffffffull);  // all-NaN row: index 0
  if (dist_out) {
    const float d2 = __uint_as_float((unsigned int)(k >> 32));
    dist_out[i] = squared ? d2 : sqrtf(d2 + 1e-7f);  // pdist 'L2', core/metrics.py:64-65
  }
}

template <int C>
static int knn_launch(dgr_ctx *ctx, const float *F0, int64_t N0, const float *F1, int64_t N1,
                      unsigned long long *best, hipStream_t stream) {
  constexpr int QPT = (C <= 32) ? 4 : 2;
  const int qblocks = (int)dgr_ceil_div(N0, (int64_t)KNN_THREADS * QPT);
  // enough (query block, F1 split) workgroups to cover every CU a few times over
  int splits = (int)dgr_ceil_div((int64_t)ctx->num_cus * 4, qblocks);
  int64_t max_splits = dgr_ceil_div(N1, KNN_TB);
  if (splits > max_splits) splits = (int)max_splits;
  if (splits < 1) splits = 1;
  int rows_per_split = (int)dgr_ceil_div(dgr_ceil_div(N1, splits), KNN_TB) * KNN_TB;
  splits = (int)dgr_ceil_div(N1, rows_per_split);
  dim3 grid(qblocks, splits);
  knn1_kernel<C, QPT><<<grid, KNN_THREADS, 0, stream>>>(F0, N0, F1, N1, rows_per_split, best);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

int dgr_knn1_impl(dgr_ctx *ctx, const float *F0, int64_t N0, const float *F1, int64_t N1, int C,
                  int squared, int64_t *idx_out, float *dist_out, hipStream_t stream) {
  DGR_REQUIRE(N0 > 0 && N1 > 0, "find_knn: empty feature matrix (N0=%lld, N1=%lld)", (long long)N0,
              (long long)N1);
  DGR_REQUIRE(N1 < (1ll << 31), "find_knn: N1 too large");
  unsigned long long *best;
  DGR_ALLOC(best, ctx->arena, unsigned long long, N0);
  DGR_HIP_CHECK(hipMemsetAsync(best, 0xff, (size_t)N0 * sizeof(unsigned long long), stream));
  switch (C) {
    case 16: DGR_CHECK(knn_launch<16>(ctx, F0, N0, F1, N1, best, stream)); break;
    case 32: DGR_CHECK(knn_launch<32>(ctx, F0, N0, F1, N1, best, stream)); break;
    case 64: DGR_CHECK(knn_launch<64>(ctx, F0, N0, F1, N1, best, stream)); break;
    default:
      dgr_set_error("find_knn: feature width %d not supported (16, 32, 64)", C);
      return DGR_EINVAL;
  }
  knn1_finish<<<(int)dgr_ceil_div(N0, 256), 256, 0, stream>>>(best, N0, squared, idx_out, dist_out);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

extern "C" int dgr_knn1_l2(dgr_ctx *ctx, const float *F0, int64_t N0, const float *F1, int64_t N1, int C,
                           int squared, int64_t *idx_out, float *dist_out, dgr_stream stream) {
  DGR_REQUIRE(ctx && F0 && F1 && idx_out, "dgr_knn1_l2: NULL argument");
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  DGR_CHECK(ctx->arena.reset());
  return dgr_knn1_impl(ctx, F0, N0, F1, N1, C, squared, idx_out, dist_out, (hipStream_t)stream);
}
