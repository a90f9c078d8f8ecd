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
                int rows_per_split, unsigned long long *__restrict__ best, const int32_t *run_flag,
                const int32_t *__restrict__ qlist, const int32_t *qcount) {
  __shared__ __attribute__((aligned(16))) float tile[KNN_TB * C];
  if (run_flag && *run_flag == 0) return;  // fallback launch of the prefiltered path: nothing to redo
  // optional indirection: only the queries listed in qlist[0 .. *qcount) (prefilter slot overflow)
  const int64_t n_q = qlist ? (int64_t)*qcount : N0;
  if ((int64_t)blockIdx.x * KNN_THREADS * QPT >= n_q) return;
  const int64_t q0 = ((int64_t)blockIdx.x * KNN_THREADS + threadIdx.x) * QPT;
  const int64_t j_begin = (int64_t)blockIdx.y * rows_per_split;
  const int64_t j_end = min(N1, j_begin + rows_per_split);
  float q[QPT][C];
  int64_t qrow[QPT];
#pragma unroll
  for (int u = 0; u < QPT; ++u) {
    const int64_t li = min(q0 + u, n_q - 1);
    const int64_t r = qlist ? (int64_t)qlist[li] : li;
    qrow[u] = r;
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
    if (q0 + u < n_q && bi[u] != 0x7fffffff) {
      const unsigned long long key =
          ((unsigned long long)__float_as_uint(bd[u]) << 32) | (unsigned int)bi[u];
      atomicMin(best + qrow[u], key);
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
                      unsigned long long *best, const int32_t *run_flag, hipStream_t stream,
                      const int32_t *qlist = nullptr, const int32_t *qcount = nullptr) {
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
  knn1_kernel<C, QPT><<<grid, KNN_THREADS, 0, stream>>>(F0, N0, F1, N1, rows_per_split, best, run_flag, qlist, qcount);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}


// ------------------------------------------------------------------------------------------
// C = 32: bf16-MFMA prefilter + exact re-evaluation.  Same result as the brute-force kernel above,
// bit for bit, at ~1/6 of its time:
//   pack     every feature row is split x = hi + lo (+ r, |r| <= 2^-18 |x|) into two bf16 rows, stored
//            in MFMA operand order (32-row tiles); reference rows are pre-scaled by -2 (exact) and
//            carry their squared norm nb.
//   pass 1   d~'(i,j) = nb_i - 2 (hi.hi + hi.lo + lo.hi)  on v_mfma_f32_32x32x16_bf16 (6 per 32 x 32
//            block, accumulator initialised with nb through the C operand); per-query minimum m~_j.
//   pass 2   the same products again (identical bits); every (i, j) with d~' <= m~_j + tau_j goes to
//            a candidate list.  tau_j = 2 c (na_j + max nb), c = 4e-5, bounds twice the worst-case
//            difference between d~ and the f32 value the brute-force kernel computes (split residual
//            3 * 2^-18, f32 accumulation of 96 products, f32 norms), so the brute-force
//            arg-min -- including its first-index tie-break among equal f32 distances -- is always
//            in the list.
//   exact    one thread per candidate evaluates sum (a - b)^2 exactly like knn1_kernel and merges with
//            the same 64-bit atomicMin key.
// A query that collects more than KNN_SLOTS candidates (many near-ties, e.g. repeated structure) is redone by the
// brute-force kernel through a device-side query list; a non-finite / huge feature makes the brute-force kernel,
// launched behind, redo the whole search.  No host round trip either way.
// ------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
constexpr float KNN_TAU_C = 8e-5f;  // 2 c
constexpr int KNN_SLOTS = 8;        // candidate slots per query

__device__ __forceinline__ unsigned short knn_f2bf(float x) {  // round to nearest even
  uint32_t u = __float_as_uint(x);
  if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)((u >> 16) | ((u & 0xffffu) ? 0x40u : 0u));
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float knn_bf2f(unsigned short h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint32_t knn_ord(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float knn_unord(uint32_t k) {
  return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

// packed[(tile * 4 + f) * 64 + r + 32 g] = 8 bf16: dims 16 (f & 1) + 8 g .. + 7 of row 32 tile + r,
// f >> 1 = 0: hi, 1: lo.  One thread per (row, g, chunk); the (g = 0, chunk = 0) thread also writes the norm.
__global__ void __launch_bounds__(256)
    knn_pack_kernel(const float *__restrict__ F, int64_t N, int64_t n_pad, float scale, float pad_norm,
                    bf16x8 *__restrict__ packed, float *__restrict__ norms, uint32_t *norm_max, int32_t *fallback) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = t >> 2;
  if (row >= n_pad) return;
  const int g = (int)(t & 1), ch = (int)((t >> 1) & 1);
  bf16x8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) { hi[e] = 0; lo[e] = 0; }
  if (row < N) {
    const float *src = F + row * 32 + 16 * ch + 8 * g;
    const float4 v0 = *reinterpret_cast<const float4 *>(src), v1 = *reinterpret_cast<const float4 *>(src + 4);
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // non-finite or huge values (squared norms would overflow): leave the search to the exact kernel
      if (!(fabsf(x[e]) < 1e18f)) *fallback = 1;
      const unsigned short h = knn_f2bf(x[e]);
      const unsigned short l = knn_f2bf(x[e] - knn_bf2f(h));
      hi[e] = (short)knn_f2bf(knn_bf2f(h) * scale);  // scale is a power of two: exact
      lo[e] = (short)knn_f2bf(knn_bf2f(l) * scale);
    }
  }
  const int64_t tile = row >> 5;
  const int r = (int)(row & 31);
  packed[(tile * 4 + ch) * 64 + r + 32 * g] = hi;
  packed[(tile * 4 + 2 + ch) * 64 + r + 32 * g] = lo;
  if (g == 0 && ch == 0) {
    float n = pad_norm;
    if (row < N) {
      n = 0.f;
      for (int c = 0; c < 32; ++c) n = fmaf(F[row * 32 + c], F[row * 32 + c], n);
      if (norm_max) atomicMax(norm_max, __float_as_uint(n));  // n >= 0: bit patterns order like values
    }
    norms[row] = n;
  }
}

// The four waves of a workgroup need the same reference tiles: they are staged through LDS, KNN_ST tiles per
// stage (16.5 KB), double buffered -- one global read per workgroup instead of one per wave (the per-wave
// version ran the L1 at ~2/3 of its bandwidth with four identical request streams).
constexpr int KNN_ST = 4;
template <bool PASS2>
__global__ void __launch_bounds__(256, 2)
    knn_mfma_kernel(const bf16x8 *__restrict__ Q, const bf16x8 *__restrict__ R, const float *__restrict__ nb,
                    int n_qblocks, int n_rtiles, int tiles_per_split, uint32_t *__restrict__ mt,
                    const float *__restrict__ na, const uint32_t *__restrict__ nb_max, int64_t N0, int64_t N1,
                    int32_t *__restrict__ cand, int32_t *__restrict__ cand_cnt, int32_t *overflow) {
  __shared__ bf16x8 sA[2][KNN_ST * 4 * 64];
  __shared__ __attribute__((aligned(16))) float sNb[2][KNN_ST * 32];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5;
  const int qb0 = (blockIdx.x * 4 + wave) * 4;     // may lie beyond n_qblocks: clamped loads, guarded outputs
  bf16x8 bq[4][4];
  float m[4], thr[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int qb = min(qb0 + u, n_qblocks - 1);
#pragma unroll
    for (int f = 0; f < 4; ++f) bq[u][f] = Q[((int64_t)qb * 4 + f) * 64 + lane];
    m[u] = __builtin_inff();
    thr[u] = 0.f;
    if (PASS2) {
      const int64_t q = (int64_t)qb * 32 + (lane & 31);
      const float nmax = __uint_as_float(*nb_max);
      thr[u] = (q < N0 && qb0 + u < n_qblocks) ? knn_unord(mt[q]) + KNN_TAU_C * (na[q] + nmax) : -__builtin_inff();
    }
  }
  const int t_begin = blockIdx.y * tiles_per_split;
  const int t_end = min(n_rtiles, t_begin + tiles_per_split);
  if (t_begin >= t_end) return;   // block-uniform
  // stage loader: thread tid fetches piece tid + 256 j of tile t0 + j (contiguous 4 KB per tile) and one norm
  bf16x8 pre[KNN_ST];
  float pre_nb = 0.f;
  auto request = [&](int t0) {
#pragma unroll
    for (int j = 0; j < KNN_ST; ++j) pre[j] = R[(int64_t)min(t0 + j, n_rtiles - 1) * 256 + tid];
    if (tid < KNN_ST * 32) pre_nb = nb[(int64_t)min(t0 + (tid >> 5), n_rtiles - 1) * 32 + (tid & 31)];
  };
  auto deposit = [&](int buf) {
#pragma unroll
    for (int j = 0; j < KNN_ST; ++j) sA[buf][j * 256 + tid] = pre[j];
    if (tid < KNN_ST * 32) sNb[buf][tid] = pre_nb;
  };
  request(t_begin);
  deposit(0);
  __syncthreads();
  int buf = 0;
  for (int t0 = t_begin; t0 < t_end; t0 += KNN_ST) {
    if (t0 + KNN_ST < t_end) request(t0 + KNN_ST);   // lands behind this stage's MFMAs
#pragma unroll
    for (int j = 0; j < KNN_ST; ++j) {
      const int t = t0 + j;
      if (t >= t_end) break;   // block-uniform
      const bf16x8 a0 = sA[buf][(j * 4 + 0) * 64 + lane], a1 = sA[buf][(j * 4 + 1) * 64 + lane];
      const bf16x8 a2 = sA[buf][(j * 4 + 2) * 64 + lane], a3 = sA[buf][(j * 4 + 3) * 64 + lane];
      f32x16_t c0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = *reinterpret_cast<const float4 *>(&sNb[buf][j * 32 + 8 * g + 4 * h]);
        c0[4 * g] = v.x; c0[4 * g + 1] = v.y; c0[4 * g + 2] = v.z; c0[4 * g + 3] = v.w;
      }
      // the six MFMAs of a block form a dependent chain: the four blocks are interleaved step by step so that
      // every MFMA has three independent ones between itself and its predecessor
      f32x16_t acc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[u][0], c0, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[u][1], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[u][2], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[u][3], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bq[u][0], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, bq[u][1], acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float bm = fminf(fminf(acc[u][0], acc[u][1]), fminf(acc[u][2], acc[u][3]));
#pragma unroll
        for (int e = 4; e < 16; e += 4)
          bm = fminf(bm, fminf(fminf(acc[u][e], acc[u][e + 1]), fminf(acc[u][e + 2], acc[u][e + 3])));
        if (!PASS2) {
          m[u] = fminf(m[u], bm);
        } else {
          if (!(bm > thr[u])) {   // rare: some reference of this block is within tau of the query's minimum
            const int64_t q = (int64_t)(qb0 + u) * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int64_t i = (int64_t)t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
              if (!(acc[u][e] > thr[u]) && q < N0 && i < N1 && qb0 + u < n_qblocks) {
                const int slot = atomicAdd(cand_cnt + q, 1);   // per-query counters: no hot address
                if (slot < KNN_SLOTS) cand[q * KNN_SLOTS + slot] = (int32_t)i;   // beyond: knn_overflow_list
              }
            }
          }
        }
      }
    }
    if (t0 + KNN_ST < t_end) deposit(buf ^ 1);
    __syncthreads();   // the other buffer is complete; this one may be overwritten by the next deposit
    buf ^= 1;
  }
  if (!PASS2) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float o = __shfl_xor(m[u], 32, 64);
      const float mm = fminf(m[u], o);
      const int64_t q = (int64_t)(qb0 + u) * 32 + (lane & 31);
      if (lane < 32 && qb0 + u < n_qblocks && q < N0) atomicMin(mt + q, knn_ord(mm));
    }
  }
}

__global__ void __launch_bounds__(256)
    knn_exact_kernel(const float *__restrict__ F0, const float *__restrict__ F1, const int32_t *__restrict__ cand,
                     const int32_t *__restrict__ cand_cnt, int64_t N0, unsigned long long *__restrict__ best) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t q = t / KNN_SLOTS;
  const int slot = (int)(t % KNN_SLOTS);
  if (q >= N0 || slot >= min(cand_cnt[q], KNN_SLOTS)) return;
  const int i = cand[t];
  const float *a = F0 + q * 32, *b = F1 + (int64_t)i * 32;
  float d0 = 0.f, d1 = 0.f;  // the very chain of knn1_kernel
#pragma unroll
  for (int k = 0; k < 32; k += 4) {
    const float4 av = *reinterpret_cast<const float4 *>(a + k), bv = *reinterpret_cast<const float4 *>(b + k);
    const float e0 = av.x - bv.x, e1 = av.y - bv.y, e2 = av.z - bv.z, e3 = av.w - bv.w;
    d0 = fmaf(e0, e0, d0);
    d1 = fmaf(e1, e1, d1);
    d0 = fmaf(e2, e2, d0);
    d1 = fmaf(e3, e3, d1);
  }
  const float d = d0 + d1;
  if (d < __builtin_inff()) {
    const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)i;
    atomicMin(best + q, key);
  }
}

// queries that collected more candidates than slots (many near-ties): redone exactly by the brute-force kernel
__global__ void knn_overflow_list(const int32_t *__restrict__ cand_cnt, int64_t N0, int32_t *__restrict__ qlist,
                                  int32_t *qcount) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < N0 && cand_cnt[q] > KNN_SLOTS) qlist[atomicAdd(qcount, 1)] = (int32_t)q;
}

static int knn_prefiltered(dgr_ctx *ctx, const float *F0, int64_t N0, const float *F1, int64_t N1,
                           unsigned long long *best, hipStream_t stream) {
  DgrArena &arena = ctx->arena;
  const int n_qblocks = (int)dgr_ceil_div(N0, 32), n_rtiles = (int)dgr_ceil_div(N1, 32);
  bf16x8 *Qp, *Rp;
  float *na, *nb;
  uint32_t *mt, *nb_max;
  int32_t *cand, *cand_cnt;
  DGR_ALLOC(Qp, arena, bf16x8, (int64_t)n_qblocks * 256);
  DGR_ALLOC(Rp, arena, bf16x8, (int64_t)n_rtiles * 256);
  DGR_ALLOC(na, arena, float, (int64_t)n_qblocks * 32);
  DGR_ALLOC(nb, arena, float, (int64_t)n_rtiles * 32);
  DGR_ALLOC(mt, arena, uint32_t, N0);
  int32_t *qlist;
  DGR_ALLOC(qlist, arena, int32_t, N0);
  DGR_ALLOC(cand_cnt, arena, int32_t, N0 + 4);   // + [N0]: max nb bits, [N0 + 1]: fallback flag, [N0 + 2]: overflow count
  DGR_ALLOC(cand, arena, int32_t, N0 * KNN_SLOTS);
  nb_max = reinterpret_cast<uint32_t *>(cand_cnt + N0);
  int32_t *fallback = cand_cnt + N0 + 1;
  DGR_HIP_CHECK(hipMemsetAsync(cand_cnt, 0, (size_t)(N0 + 4) * sizeof(int32_t), stream));
  DGR_HIP_CHECK(hipMemsetAsync(mt, 0xff, (size_t)N0 * sizeof(uint32_t), stream));
  const int64_t q_pad = (int64_t)n_qblocks * 32, r_pad = (int64_t)n_rtiles * 32;
  knn_pack_kernel<<<(int)dgr_ceil_div(q_pad * 4, 256), 256, 0, stream>>>(F0, N0, q_pad, 1.f, 0.f, Qp, na, nullptr,
                                                                         fallback);
  knn_pack_kernel<<<(int)dgr_ceil_div(r_pad * 4, 256), 256, 0, stream>>>(F1, N1, r_pad, -2.f, __builtin_inff(), Rp,
                                                                         nb, nb_max, fallback);
  DGR_LAUNCH_CHECK();
  const int qgroups = (int)dgr_ceil_div(n_qblocks, 16);
  // reference splits chosen so that the grid fills the chip in whole rounds (one resident round when possible):
  // a grid of 1.3 x the resident capacity leaves the second round two thirds empty
  auto launch = [&](auto kernel) -> int {
    static int per_cu = 0;
    if (per_cu == 0) {
      int n = 0;
      DGR_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, 0));
      per_cu = n < 1 ? 1 : n;
    }
    const int capacity = ctx->num_cus * per_cu;
    int splits = capacity / qgroups;
    if (splits > n_rtiles / KNN_ST) splits = n_rtiles / KNN_ST;
    if (splits < 1) splits = 1;
    const int tps = (int)dgr_ceil_div(dgr_ceil_div(n_rtiles, splits), KNN_ST) * KNN_ST;
    splits = (int)dgr_ceil_div(n_rtiles, tps);
    dim3 grid(qgroups, splits);
    kernel<<<grid, 256, 0, stream>>>(Qp, Rp, nb, n_qblocks, n_rtiles, tps, mt, na, nb_max, N0, N1, cand, cand_cnt, fallback);
    return DGR_OK;
  };
  DGR_CHECK(launch(knn_mfma_kernel<false>));
  DGR_CHECK(launch(knn_mfma_kernel<true>));
  DGR_LAUNCH_CHECK();
  knn_exact_kernel<<<(int)dgr_ceil_div(N0 * KNN_SLOTS, 256), 256, 0, stream>>>(F0, F1, cand, cand_cnt, N0, best);
  DGR_LAUNCH_CHECK();
  // queries with more near-ties than slots are redone one by one by the brute-force kernel (its blocks beyond
  // the list length exit at once); non-finite / huge input: the brute-force kernel redoes the whole search
  int32_t *qcount = cand_cnt + N0 + 2;
  knn_overflow_list<<<(int)dgr_ceil_div(N0, 256), 256, 0, stream>>>(cand_cnt, N0, qlist, qcount);
  DGR_CHECK(knn_launch<32>(ctx, F0, N0, F1, N1, best, nullptr, stream, qlist, qcount));
  return knn_launch<32>(ctx, F0, N0, F1, N1, best, fallback, stream);
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
    case 16: DGR_CHECK(knn_launch<16>(ctx, F0, N0, F1, N1, best, nullptr, stream)); break;
    case 32: {
      static const bool brute = getenv("DGR_KNN_BRUTE") != nullptr;
      if (brute || N1 < 1024) DGR_CHECK(knn_launch<32>(ctx, F0, N0, F1, N1, best, nullptr, stream));
      else DGR_CHECK(knn_prefiltered(ctx, F0, N0, F1, N1, best, stream));
      break;
    }
    case 64: DGR_CHECK(knn_launch<64>(ctx, F0, N0, F1, N1, best, nullptr, stream)); break;
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
