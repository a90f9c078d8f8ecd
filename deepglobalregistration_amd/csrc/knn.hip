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
//
// Batched (round 5): every kernel below runs ONCE for all pairs of a batch -- blockIdx.z (packing: blockIdx.y) selects
// the pair, whose row ranges inside the concatenated feature matrices come from a by-value descriptor table -- instead
// of 13 launches per pair in a host loop; reference indices are written as rows of the concatenated F1.
#include "dgr_internal.h"

constexpr int KNN_THREADS = 256;
constexpr int KNN_TB = 64;  // F1 rows per LDS tile
constexpr int KNN_MAXP = 32;   // pairs per launch (descriptor table passed by value: no upload, no host buffer to keep alive)

struct KnnPair {
  int64_t q0, r0;      // first query row (of F0) / first reference row (of F1) of the pair
  int32_t n0, n1;      // queries / references
  int32_t qb0, rt0;    // first 32-row block of the pair in the packed query / reference arrays
};
struct KnnBatch {
  KnnPair p[KNN_MAXP];
  int np;
};

template <int C, int QPT>
__global__ void __launch_bounds__(KNN_THREADS)
    knn1_kernel(const float *__restrict__ F0, const float *__restrict__ F1, KnnBatch B, int splits,
                unsigned long long *__restrict__ best, const int32_t *run_flag,
                const int32_t *__restrict__ qlist, const int32_t *qcount) {
  __shared__ __attribute__((aligned(16))) float tile[KNN_TB * C];
  // blockIdx.z = pair: its rows of F0 / F1 / best (and of qlist), its flag and its list length
  const KnnPair d = B.p[blockIdx.z];
  if (run_flag && run_flag[blockIdx.z] == 0) return;  // fallback launch of the prefiltered path: nothing to redo
  F0 += d.q0 * C;
  F1 += d.r0 * C;
  best += d.q0;
  const int64_t N0 = d.n0, N1 = d.n1;
  // optional indirection: only the pair's queries listed in qlist[q0 .. q0 + qcount[pair]) (prefilter slot overflow)
  if (qlist) qlist += d.q0;
  const int64_t n_q = qlist ? (int64_t)qcount[blockIdx.z] : N0;
  if ((int64_t)blockIdx.x * KNN_THREADS * QPT >= n_q) return;
  const int64_t q0 = ((int64_t)blockIdx.x * KNN_THREADS + threadIdx.x) * QPT;
  const int rows_per_split = (int)(((N1 + splits - 1) / splits + KNN_TB - 1) / KNN_TB) * KNN_TB;
  const int64_t j_begin = (int64_t)blockIdx.y * rows_per_split;
  const int64_t j_end = min(N1, j_begin + rows_per_split);
  if (j_begin >= j_end) return;
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
      const unsigned long long key =   // the index as a row of the concatenated F1
          ((unsigned long long)__float_as_uint(bd[u]) << 32) | (unsigned int)(bi[u] + (int)d.r0);
      atomicMin(best + qrow[u], key);
    }
  }
}

__global__ void knn1_finish(const unsigned long long *__restrict__ best, KnnBatch B, int squared,
                            int64_t *__restrict__ idx_out, float *__restrict__ dist_out) {
  const KnnPair d = B.p[blockIdx.y];
  const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= d.n0) return;
  const int64_t i = d.q0 + li;
  const unsigned long long k = best[i];
  idx_out[i] = (k == ~0ull) ? d.r0 : (int64_t)(k & 0xffffffffull);  // all-NaN row: the pair's reference 0
  if (dist_out) {
    const float d2 = __uint_as_float((unsigned int)(k >> 32));
    dist_out[i] = squared ? d2 : sqrtf(d2 + 1e-7f);  // pdist 'L2', core/metrics.py:64-65
  }
}

template <int C>
static int knn_launch(dgr_ctx *ctx, const float *F0, const float *F1, const KnnBatch &B,
                      unsigned long long *best, const int32_t *run_flag, hipStream_t stream,
                      const int32_t *qlist = nullptr, const int32_t *qcount = nullptr) {
  constexpr int QPT = (C <= 32) ? 4 : 2;
  int64_t n0_max = 0, n1_max = 0, qblocks_all = 0;
  for (int p = 0; p < B.np; ++p) {
    n0_max = std::max<int64_t>(n0_max, B.p[p].n0);
    n1_max = std::max<int64_t>(n1_max, B.p[p].n1);
    qblocks_all += dgr_ceil_div(B.p[p].n0, (int64_t)KNN_THREADS * QPT);
  }
  const int qblocks = (int)dgr_ceil_div(n0_max, (int64_t)KNN_THREADS * QPT);
  // enough (query block, F1 split) workgroups to cover every CU a few times over; a pair with fewer rows than the
  // largest leaves its surplus blocks / splits empty (they exit at once)
  int splits = (int)dgr_ceil_div((int64_t)ctx->num_cus * 4, qblocks_all);
  int64_t max_splits = dgr_ceil_div(n1_max, KNN_TB);
  if (splits > max_splits) splits = (int)max_splits;
  if (splits < 1) splits = 1;
  dim3 grid(qblocks, splits, B.np);
  knn1_kernel<C, QPT><<<grid, KNN_THREADS, 0, stream>>>(F0, F1, B, splits, best, run_flag, qlist, qcount);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}


// ------------------------------------------------------------------------------------------
// C = 32: bf16-MFMA prefilter + exact re-evaluation.  Same result as the brute-force kernel above,
// bit for bit, at ~1/8 of its time:
//   pack     every feature row is split x = hi + lo (+ r, |r| <= 2^-18 |x|) into two bf16 rows, stored
//            in MFMA operand order (32-row tiles); reference rows are pre-scaled by -2 (exact) and
//            carry their squared norm nb.
//   pass 1   d~'(i,j) = nb_i - 2 (hi.hi + hi.lo + lo.hi)  on v_mfma_f32_32x32x16_bf16 (6 per 32 x 32
//            block, accumulator initialised with nb through the C operand); per-query minimum m~_j -- over a SAMPLE of
//            the reference tiles (every KNN_SUB-th stage; round 5): any upper bound of the true minimum will do for the
//            threshold below, and the minimum over half of the references has expected rank 2 among all of them.
//            Measured per 4-pair batch (BASELINE configs[1], one box): every stage 0.92 ms, every 2nd 0.83, every 4th
//            1.16 -- the second pass slows down with the number of candidates it has to emit (0.52 -> 0.66 ms) and
//            0.5-2 % of the queries overflow their slots, so the sampling stops paying at a half.
//   pass 2   the same products for ALL tiles (identical bits where pass 1 ran); every (i, j) with d~' <= m~_j + tau_j
//            goes to a candidate list.  tau_j = 2 c (na_j + max nb), c = 4e-5, bounds twice the worst-case
//            difference between d~ and the f32 value the brute-force kernel computes (split residual
//            3 * 2^-18, f32 accumulation of 96 products, f32 norms), so the brute-force
//            arg-min -- including its first-index tie-break among equal f32 distances -- is always
//            in the list (m~_j >= the true minimum of d~': the list only grows with the sampling, ~2 entries per query).
//   exact    one thread per candidate evaluates sum (a - b)^2 exactly like knn1_kernel and merges with
//            the same 64-bit atomicMin key.
// A query that collects more than KNN_SLOTS candidates (the sample minimum ranks low, or many near-ties, e.g. repeated
// structure) is redone exactly through a device-side query list (a few: knn_query_scan_kernel; many: knn1_kernel);
// a non-finite / huge feature makes the brute-force kernel, launched behind, redo the pair's whole search.  No host
// round trip either way.
// ------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
constexpr float KNN_TAU_C = 8e-5f;  // 2 c
constexpr int KNN_SLOTS = 32;       // candidate slots per query (every 2nd stage sampled: at most 16 seen on the benchmark's features)
#ifndef DGR_KNN_SUB
#define DGR_KNN_SUB 2
#endif
constexpr int KNN_SUB = DGR_KNN_SUB;   // pass 1 visits every KNN_SUB-th group of KNN_ST reference tiles (1: all of them)

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
// blockIdx.y = 2 pair + side (0: queries, 1: references -- pre-scaled by -2, padded with infinite norms, maximum norm
// of the pair left in nb_max[pair]).
// Reference rows are INTERLEAVED over the tiles: slot s of tile t holds row s n_tiles + t, so that every tile -- and
// every subset of tiles, the sample of pass 1 in particular -- is spread evenly over the cloud.  (Consecutive rows are
// neighbouring voxels with similar descriptors: a sample of whole 128-row stages in row order misses whole
// neighbourhoods, and then every member of the true neighbour's cluster lies under the sampled minimum.)
__global__ void __launch_bounds__(256)
    knn_pack_kernel(const float *__restrict__ F0, const float *__restrict__ F1, KnnBatch B,
                    bf16x8 *__restrict__ Qp, bf16x8 *__restrict__ Rp, float *__restrict__ na, float *__restrict__ nb,
                    uint32_t *__restrict__ nb_max, int32_t *__restrict__ fallback) {
  const int pair = blockIdx.y >> 1, side = blockIdx.y & 1;
  const KnnPair d = B.p[pair];
  const int64_t N = side ? d.n1 : d.n0;
  const int64_t n_pad = (N + 31) / 32 * 32;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t row = t >> 2;
  if (row >= n_pad) return;
  const float *F = side ? F1 + d.r0 * 32 : F0 + d.q0 * 32;
  const float scale = side ? -2.f : 1.f;
  const int64_t prow = row;                                   // position in the packed array
  if (side) row = (prow & 31) * (n_pad >> 5) + (prow >> 5);   // the reference row that sits there
  bf16x8 *packed = side ? Rp + (int64_t)d.rt0 * 256 : Qp + (int64_t)d.qb0 * 256;
  float *norms = side ? nb + (int64_t)d.rt0 * 32 : na + (int64_t)d.qb0 * 32;
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
      // non-finite or huge values (squared norms would overflow): leave the pair's search to the exact kernel
      if (!(fabsf(x[e]) < 1e18f)) fallback[pair] = 1;
      const unsigned short h = knn_f2bf(x[e]);
      const unsigned short l = knn_f2bf(x[e] - knn_bf2f(h));
      hi[e] = (short)knn_f2bf(knn_bf2f(h) * scale);  // scale is a power of two: exact
      lo[e] = (short)knn_f2bf(knn_bf2f(l) * scale);
    }
  }
  const int64_t tile = prow >> 5;
  const int r = (int)(prow & 31);
  packed[(tile * 4 + ch) * 64 + r + 32 * g] = hi;
  packed[(tile * 4 + 2 + ch) * 64 + r + 32 * g] = lo;
  // squared norm of the row: the four threads of a row (consecutive lanes) each sum their eight values, fixed order
  float n = 0.f;
  if (row < N) {
    const float *src = F + row * 32 + 16 * ch + 8 * g;
#pragma unroll
    for (int e = 0; e < 8; ++e) n = fmaf(src[e], src[e], n);
  }
  n += __shfl_xor(n, 1, 64);
  n += __shfl_xor(n, 2, 64);
  if (g == 0 && ch == 0) {
    if (row >= N) n = side ? __builtin_inff() : 0.f;   // padding rows: never a minimum / never a query
    else if (side) atomicMax(nb_max + pair, __float_as_uint(n));  // n >= 0: bit patterns order like values
    norms[prow] = n;
  }
}

// The four waves of a workgroup need the same reference tiles: they are staged through LDS, KNN_ST tiles per
// stage (16.5 KB), double buffered -- one global read per workgroup instead of one per wave (the per-wave
// version ran the L1 at ~2/3 of its bandwidth with four identical request streams).
// Grid: x = groups of 16 query blocks (of the largest pair), y = reference splits, z = pair.  PASS2 = false walks
// every KNN_SUB-th stage of its split only (the sample), PASS2 = true every stage.
constexpr int KNN_ST = 4;
template <bool PASS2>
__global__ void __launch_bounds__(256, 2)
    knn_mfma_kernel(const bf16x8 *__restrict__ Qp, const bf16x8 *__restrict__ Rp, const float *__restrict__ nbp,
                    KnnBatch B, int splits, uint32_t *__restrict__ mt, const float *__restrict__ nap,
                    const uint32_t *__restrict__ nb_max, int32_t *__restrict__ cand, int32_t *__restrict__ cand_cnt) {
  __shared__ bf16x8 sA[2][KNN_ST * 4 * 64];
  __shared__ __attribute__((aligned(16))) float sNb[2][KNN_ST * 32];
  const KnnPair d = B.p[blockIdx.z];
  const int n_qblocks = (d.n0 + 31) >> 5, n_rtiles = (d.n1 + 31) >> 5;
  if ((int)blockIdx.x * 16 >= n_qblocks) return;   // a smaller pair than the grid's largest
  const int64_t N0 = d.n0, N1 = d.n1;
  const bf16x8 *Q = Qp + (int64_t)d.qb0 * 256, *R = Rp + (int64_t)d.rt0 * 256;
  const float *nb = nbp + (int64_t)d.rt0 * 32, *na = nap + (int64_t)d.qb0 * 32;
  mt += d.q0;
  cand += d.q0 * KNN_SLOTS;
  cand_cnt += d.q0;
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
      const float nmax = __uint_as_float(nb_max[blockIdx.z]);
      thr[u] = (q < N0 && qb0 + u < n_qblocks) ? knn_unord(mt[q]) + KNN_TAU_C * (na[q] + nmax) : -__builtin_inff();
    }
  }
  // stages (KNN_ST tiles) of this split; pass 1 takes every KNN_SUB-th of them, offset by the split index so that the
  // sample does not alias with the split length
  const int n_stages = (n_rtiles + KNN_ST - 1) / KNN_ST;
  const int sps = (n_stages + splits - 1) / splits;   // stages per split
  const int s_begin = blockIdx.y * sps, s_end = min(n_stages, s_begin + sps);
  constexpr int STEP = PASS2 ? 1 : KNN_SUB;
  // (a pair with fewer than KNN_SUB stages per split still gets one sampled stage per split)
  const int s_first = PASS2 ? s_begin : s_begin + min((int)(blockIdx.y % KNN_SUB), max(s_end - s_begin - 1, 0));
  if (s_first >= s_end) return;   // block-uniform
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
  request(s_first * KNN_ST);
  deposit(0);
  __syncthreads();
  int buf = 0;
  for (int st = s_first; st < s_end; st += STEP) {
    const int t0 = st * KNN_ST;
    if (st + STEP < s_end) request((st + STEP) * KNN_ST);   // lands behind this stage's MFMAs
#pragma unroll
    for (int j = 0; j < KNN_ST; ++j) {
      const int t = t0 + j;
      if (t >= n_rtiles) break;   // block-uniform
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
          if (!(bm > thr[u])) {   // some reference of this block is within tau of the query's (sampled) minimum
            const int64_t q = (int64_t)(qb0 + u) * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              const int64_t i = (int64_t)((e & 3) + 8 * (e >> 2) + 4 * h) * n_rtiles + t;   // slot s of tile t = row s n_tiles + t
              if (!(acc[u][e] > thr[u]) && q < N0 && i < N1 && qb0 + u < n_qblocks) {
                const int slot = atomicAdd(cand_cnt + q, 1);   // per-query counters: no hot address
                if (slot < KNN_SLOTS) cand[q * KNN_SLOTS + slot] = (int32_t)(i + d.r0);   // beyond: knn_overflow_list
              }
            }
          }
        }
      }
    }
    if (st + STEP < s_end) deposit(buf ^ 1);
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

// one thread per query of the batch walks its candidate slots (two on average); candidates are rows of the concatenated F1
__global__ void __launch_bounds__(256)
    knn_exact_kernel(const float *__restrict__ F0, const float *__restrict__ F1, const int32_t *__restrict__ cand,
                     const int32_t *__restrict__ cand_cnt, int64_t q_begin, int64_t q_end,
                     unsigned long long *__restrict__ best) {
  const int64_t q = q_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= q_end) return;
  const int cnt = min(cand_cnt[q], KNN_SLOTS);
  if (cnt <= 0) return;
  float a[32];
#pragma unroll
  for (int k = 0; k < 32; k += 4) {
    const float4 v = *reinterpret_cast<const float4 *>(F0 + q * 32 + k);
    a[k] = v.x; a[k + 1] = v.y; a[k + 2] = v.z; a[k + 3] = v.w;
  }
  unsigned long long key = ~0ull;
  for (int slot = 0; slot < cnt; ++slot) {
    const int i = cand[q * KNN_SLOTS + slot];
    const float *b = F1 + (int64_t)i * 32;
    float d0 = 0.f, d1 = 0.f;  // the very chain of knn1_kernel
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
      const float4 bv = *reinterpret_cast<const float4 *>(b + k);
      const float e0 = a[k] - bv.x, e1 = a[k + 1] - bv.y, e2 = a[k + 2] - bv.z, e3 = a[k + 3] - bv.w;
      d0 = fmaf(e0, e0, d0);
      d1 = fmaf(e1, e1, d1);
      d0 = fmaf(e2, e2, d0);
      d1 = fmaf(e3, e3, d1);
    }
    const float d = d0 + d1;
    if (d < __builtin_inff()) {   // (distance bits, index): the order of the slots does not matter
      const unsigned long long k2 = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned int)i;
      key = k2 < key ? k2 : key;
    }
  }
  if (key != ~0ull) atomicMin(best + q, key);
}

// queries that collected more candidates than slots (many near-ties): redone exactly by the brute-force kernel.
// blockIdx.y = pair; the pair's list (row numbers inside the pair) starts at qlist[q0]
__global__ void knn_overflow_list(const int32_t *__restrict__ cand_cnt, KnnBatch B, int32_t *__restrict__ qlist,
                                  int32_t *qcount) {
  const KnnPair d = B.p[blockIdx.y];
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q < d.n0 && cand_cnt[d.q0 + q] > KNN_SLOTS) qlist[d.q0 + atomicAdd(qcount + blockIdx.y, 1)] = (int32_t)q;
}

// ... SHORT lists (<= KNN_SCAN_MAX queries of a pair) by parallelism over the references: workgroup (x, pair) owns
// the x-th of KNN_SCAN_SPLITS slices of the pair's references and, for every listed query in turn, evaluates its slice
// one reference per thread (the very chain of knn1_kernel per distance), reduces the (distance bits, index) keys over
// the workgroup and issues ONE atomicMin.  (The brute-force kernel keeps 4 queries per THREAD: a handful of listed
// queries would cost it one thread's walk over a whole reference split, ~0.1-0.2 ms.)  Longer lists -- many near-ties,
// e.g. repeated structure -- go to the brute-force kernel, which amortises its tiles over 1024 queries per workgroup.
constexpr int KNN_SCAN_MAX = 64, KNN_SCAN_SPLITS = 16;
__global__ void __launch_bounds__(256)
    knn_query_scan_kernel(const float *__restrict__ F0, const float *__restrict__ F1, KnnBatch B,
                          const int32_t *__restrict__ qlist, const int32_t *__restrict__ qcount,
                          unsigned long long *__restrict__ best) {
  __shared__ unsigned long long wkey[4];
  const KnnPair d = B.p[blockIdx.y];
  const int n_q = qcount[blockIdx.y];
  if (n_q <= 0 || n_q > KNN_SCAN_MAX) return;
  const int per = (d.n1 + KNN_SCAN_SPLITS - 1) / KNN_SCAN_SPLITS;
  const int j_begin = blockIdx.x * per, j_end = min(d.n1, j_begin + per);
  for (int li = 0; li < n_q; ++li) {
    const int64_t q = d.q0 + qlist[d.q0 + li];
    float a[32];
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
      const float4 v = *reinterpret_cast<const float4 *>(F0 + q * 32 + k);
      a[k] = v.x; a[k + 1] = v.y; a[k + 2] = v.z; a[k + 3] = v.w;
    }
    unsigned long long key = ~0ull;
    for (int j = j_begin + (int)threadIdx.x; j < j_end; j += 256) {
      const float *b = F1 + (d.r0 + j) * 32;
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int k = 0; k < 32; k += 4) {
        const float4 bv = *reinterpret_cast<const float4 *>(b + k);
        const float e0 = a[k] - bv.x, e1 = a[k + 1] - bv.y, e2 = a[k + 2] - bv.z, e3 = a[k + 3] - bv.w;
        d0 = fmaf(e0, e0, d0);
        d1 = fmaf(e1, e1, d1);
        d0 = fmaf(e2, e2, d0);
        d1 = fmaf(e3, e3, d1);
      }
      const float dd = d0 + d1;
      if (dd < __builtin_inff()) {
        const unsigned long long k2 = ((unsigned long long)__float_as_uint(dd) << 32) | (unsigned int)(j + (int)d.r0);
        key = k2 < key ? k2 : key;   // (distance bits, index): equal distances -> the smallest index
      }
    }
#pragma unroll
    for (int s2 = 32; s2 > 0; s2 >>= 1) {
      const unsigned long long o = __shfl_xor(key, s2, 64);
      key = o < key ? o : key;
    }
    if ((threadIdx.x & 63) == 0) wkey[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long k3 = wkey[0];
      for (int w = 1; w < 4; ++w) k3 = wkey[w] < k3 ? wkey[w] : k3;
      if (k3 != ~0ull) atomicMin(best + q, k3);
    }
    __syncthreads();
  }
}

// brute-force kernel behind the LONG lists: its run flag per pair
__global__ void knn_long_list_flags(const int32_t *__restrict__ qcount, int np, int32_t *__restrict__ flags) {
  if ((int)threadIdx.x < np) flags[threadIdx.x] = qcount[threadIdx.x] > KNN_SCAN_MAX;
}

// the pairs of B (all with at least KNN_MIN_REFS references); best is initialised by the caller
static int knn_prefiltered(dgr_ctx *ctx, const float *F0, const float *F1, KnnBatch B, unsigned long long *best,
                           hipStream_t stream) {
  DgrArena &arena = ctx->arena;
  int64_t q_begin = B.p[0].q0, q_end = 0;
  int n_qb = 0, n_rt = 0, qb_max = 0, rt_max = 0;
  for (int p = 0; p < B.np; ++p) {
    KnnPair &d = B.p[p];
    d.qb0 = n_qb;
    d.rt0 = n_rt;
    const int qb = (d.n0 + 31) / 32, rt = (d.n1 + 31) / 32;
    n_qb += qb;
    n_rt += rt;
    qb_max = std::max(qb_max, qb);
    rt_max = std::max(rt_max, rt);
    q_begin = std::min(q_begin, d.q0);
    q_end = std::max(q_end, d.q0 + d.n0);
  }
  // Query rows SPANNED by the pairs of B in the concatenated F0 (not the sum of their rows): the per-query scratch below is
  // addressed by the row number itself.  Pairs too small for the prefilter (n1 < KNN_MIN_REFS, handled by the brute-force
  // kernel) that sit between large ones are covered as well -- 4 x (3 + KNN_SLOTS) bytes per such row, their counts stay
  // zero and the per-row kernels return at once for them; DGR_ALLOC fails with DGR_ENOMEM if the span does not fit the
  // arena (the span is at most the batch's N0, which the arena is sized for).
  const int64_t nq = q_end - q_begin;
  bf16x8 *Qp, *Rp;
  float *na, *nb;
  uint32_t *mt, *nb_max;
  int32_t *cand, *cand_cnt, *qlist;
  DGR_ALLOC(Qp, arena, bf16x8, (int64_t)n_qb * 256);
  DGR_ALLOC(Rp, arena, bf16x8, (int64_t)n_rt * 256);
  DGR_ALLOC(na, arena, float, (int64_t)n_qb * 32);
  DGR_ALLOC(nb, arena, float, (int64_t)n_rt * 32);
  DGR_ALLOC(mt, arena, uint32_t, nq);
  DGR_ALLOC(qlist, arena, int32_t, nq);
  DGR_ALLOC(cand_cnt, arena, int32_t, nq + 4 * KNN_MAXP);   // + per pair: max nb bits, fallback flag, overflow count, long-list flag
  DGR_ALLOC(cand, arena, int32_t, nq * KNN_SLOTS);
  nb_max = reinterpret_cast<uint32_t *>(cand_cnt + nq);
  int32_t *fallback = cand_cnt + nq + KNN_MAXP, *qcount = cand_cnt + nq + 2 * KNN_MAXP;
  DGR_HIP_CHECK(hipMemsetAsync(cand_cnt, 0, (size_t)(nq + 4 * KNN_MAXP) * sizeof(int32_t), stream));
  DGR_HIP_CHECK(hipMemsetAsync(mt, 0xff, (size_t)nq * sizeof(uint32_t), stream));
  // per-query arrays are addressed by the row of the concatenated F0: shift them so that row q_begin is element 0 (device
  // addresses: the shifted pointers are only ever dereferenced at rows in [q_begin, q_end))
  mt -= q_begin; qlist -= q_begin; cand_cnt -= q_begin; cand -= q_begin * KNN_SLOTS;
  {
    const int rows_max = std::max(qb_max, rt_max) * 32;
    dim3 grid((unsigned)dgr_ceil_div((int64_t)rows_max * 4, 256), 2 * B.np);
    knn_pack_kernel<<<grid, 256, 0, stream>>>(F0, F1, B, Qp, Rp, na, nb, nb_max, fallback);
    DGR_LAUNCH_CHECK();
  }
  int qgroups_all = 0;
  for (int p = 0; p < B.np; ++p) qgroups_all += (int)dgr_ceil_div((B.p[p].n0 + 31) / 32, 16);
  const int qgroups = (int)dgr_ceil_div(qb_max, 16);
  // reference splits chosen so that the grid fills the chip in whole rounds (one resident round when possible):
  // a grid of 1.3 x the resident capacity leaves the second round two thirds empty
  auto launch = [&](auto kernel, int sub) -> int {
    static int per_cu = 0;
    if (per_cu == 0) {
      int n = 0;
      DGR_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, 0));
      per_cu = n < 1 ? 1 : n;
    }
    const int capacity = ctx->num_cus * per_cu;
    const int stages = (int)dgr_ceil_div(rt_max, KNN_ST);
    // the split count (<= 16) whose grid fills whole rounds of resident workgroups best; ties -> more splits
    int splits = 1;
    double best_fill = 0.;
    for (int sp = 1; sp <= std::min(16, std::max(1, stages / sub)); ++sp) {
      const int64_t blocks = (int64_t)qgroups_all * sp;
      const double fill = (double)blocks / (double)(dgr_ceil_div(blocks, (int64_t)capacity) * capacity);
      if (fill >= best_fill) { best_fill = fill; splits = sp; }
    }
    dim3 grid(qgroups, splits, B.np);
    kernel<<<grid, 256, 0, stream>>>(Qp, Rp, nb, B, splits, mt, na, nb_max, cand, cand_cnt);
    return DGR_OK;
  };
  DGR_CHECK(launch(knn_mfma_kernel<false>, KNN_SUB));
  DGR_CHECK(launch(knn_mfma_kernel<true>, 1));
  DGR_LAUNCH_CHECK();
  knn_exact_kernel<<<(int)dgr_ceil_div(nq, 256), 256, 0, stream>>>(F0, F1, cand, cand_cnt, q_begin, q_end, best);
  DGR_LAUNCH_CHECK();
  // queries with more candidates than slots are redone exactly, one workgroup each (knn_query_scan_kernel);
  // non-finite / huge input: the brute-force kernel redoes the pair's whole search
  {
    int64_t n0_max = 0;
    for (int p = 0; p < B.np; ++p) n0_max = std::max<int64_t>(n0_max, B.p[p].n0);
    dim3 grid((unsigned)dgr_ceil_div(n0_max, 256), B.np);
    knn_overflow_list<<<grid, 256, 0, stream>>>(cand_cnt, B, qlist, qcount);
    DGR_LAUNCH_CHECK();
  }
  {
    dim3 grid(KNN_SCAN_SPLITS, B.np);   // short lists (the normal case: empty)
    knn_query_scan_kernel<<<grid, 256, 0, stream>>>(F0, F1, B, qlist, qcount, best);
    int32_t *long_flags = cand_cnt + q_begin + nq + 3 * KNN_MAXP;
    knn_long_list_flags<<<1, 64, 0, stream>>>(qcount, B.np, long_flags);
    DGR_LAUNCH_CHECK();
    DGR_CHECK(knn_launch<32>(ctx, F0, F1, B, best, long_flags, stream, qlist, qcount));   // long lists
  }
  return knn_launch<32>(ctx, F0, F1, B, best, fallback, stream);
}

// 1-NN of every pair of `B` (row ranges of the concatenated F0 / F1): idx_out[q] = row of F1 (concatenated numbering)
static int knn_batch(dgr_ctx *ctx, const float *F0, const float *F1, const KnnBatch &B, int C, int squared,
                     unsigned long long *best, int64_t *idx_out, float *dist_out, hipStream_t stream) {
  int64_t n0_max = 0;
  for (int p = 0; p < B.np; ++p) n0_max = std::max<int64_t>(n0_max, B.p[p].n0);
  switch (C) {
    case 16: DGR_CHECK(knn_launch<16>(ctx, F0, F1, B, best, nullptr, stream)); break;
    case 32: {
      static const bool brute = getenv("DGR_KNN_BRUTE") != nullptr;
      // small reference sets: the brute-force kernel alone (the prefilter's fixed passes would cost more)
      KnnBatch big, small;
      big.np = small.np = 0;
      for (int p = 0; p < B.np; ++p) {
        if (brute || B.p[p].n1 < 1024) small.p[small.np++] = B.p[p];
        else big.p[big.np++] = B.p[p];
      }
      if (small.np) DGR_CHECK(knn_launch<32>(ctx, F0, F1, small, best, nullptr, stream));
      if (big.np) DGR_CHECK(knn_prefiltered(ctx, F0, F1, big, best, stream));
      break;
    }
    case 64: DGR_CHECK(knn_launch<64>(ctx, F0, F1, B, best, nullptr, stream)); break;
    default:
      dgr_set_error("find_knn: feature width %d not supported (16, 32, 64)", C);
      return DGR_EINVAL;
  }
  dim3 grid((unsigned)dgr_ceil_div(n0_max, 256), B.np);
  knn1_finish<<<grid, 256, 0, stream>>>(best, B, squared, idx_out, dist_out);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

// Batched entry of the fused pipeline: pair p = rows off0[p] .. off0[p + 1] of F0 against rows off1[p] .. off1[p + 1] of
// F1; idx_out [off0[npairs]] = rows of F1 in the concatenated numbering (what the 6-D assembly gathers with)
int dgr_knn1_batch_impl(dgr_ctx *ctx, const float *F0, const int64_t *off0, const float *F1, const int64_t *off1,
                        int npairs, int C, int squared, int64_t *idx_out, float *dist_out, hipStream_t stream) {
  const int64_t n0 = off0[npairs];
  DGR_REQUIRE(off0[0] == 0 && off1[0] == 0, "find_knn: the row offsets start at 0");
  DGR_REQUIRE(off1[npairs] < (1ll << 31) && n0 < (1ll << 31), "find_knn: N0 / N1 too large");
  unsigned long long *best;
  DGR_ALLOC(best, ctx->arena, unsigned long long, n0);
  DGR_HIP_CHECK(hipMemsetAsync(best, 0xff, (size_t)n0 * sizeof(unsigned long long), stream));
  for (int p0 = 0; p0 < npairs; p0 += KNN_MAXP) {
    KnnBatch B;
    B.np = std::min(KNN_MAXP, npairs - p0);
    for (int p = 0; p < B.np; ++p) {
      KnnPair &d = B.p[p];
      d.q0 = off0[p0 + p]; d.r0 = off1[p0 + p];
      d.n0 = (int32_t)(off0[p0 + p + 1] - off0[p0 + p]); d.n1 = (int32_t)(off1[p0 + p + 1] - off1[p0 + p]);
      d.qb0 = d.rt0 = 0;
      DGR_REQUIRE(d.n0 > 0 && d.n1 > 0, "find_knn: empty feature matrix (N0=%d, N1=%d)", d.n0, d.n1);
    }
    const DgrArena::Mark mk = ctx->arena.mark();
    DGR_CHECK(knn_batch(ctx, F0, F1, B, C, squared, best, idx_out, dist_out, stream));
    ctx->arena.rewind(mk);
  }
  return DGR_OK;
}

int dgr_knn1_impl(dgr_ctx *ctx, const float *F0, int64_t N0, const float *F1, int64_t N1, int C,
                  int squared, int64_t *idx_out, float *dist_out, hipStream_t stream) {
  DGR_REQUIRE(N0 > 0 && N1 > 0, "find_knn: empty feature matrix (N0=%lld, N1=%lld)", (long long)N0,
              (long long)N1);
  DGR_REQUIRE(N1 < (1ll << 31) && N0 < (1ll << 31), "find_knn: N0 / N1 too large");
  unsigned long long *best;
  DGR_ALLOC(best, ctx->arena, unsigned long long, N0);
  DGR_HIP_CHECK(hipMemsetAsync(best, 0xff, (size_t)N0 * sizeof(unsigned long long), stream));
  KnnBatch B;
  B.np = 1;
  B.p[0] = KnnPair{0, 0, (int32_t)N0, (int32_t)N1, 0, 0};
  return knn_batch(ctx, F0, F1, B, C, squared, best, idx_out, dist_out, stream);
}

extern "C" int dgr_knn1_l2(dgr_ctx *ctx, const float *F0, int64_t N0, const float *F1, int64_t N1, int C,
                           int squared, int64_t *idx_out, float *dist_out, dgr_stream stream) {
  DGR_REQUIRE(ctx && F0 && F1 && idx_out, "dgr_knn1_l2: NULL argument");
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  DGR_CHECK(ctx->arena.reset());
  return dgr_knn1_impl(ctx, F0, N0, F1, N1, C, squared, idx_out, dist_out, (hipStream_t)stream);
}

extern "C" int dgr_knn1_l2_batch(dgr_ctx *ctx, const float *F0, const int64_t *off0, const float *F1,
                                 const int64_t *off1, int npairs, int C, int squared, int64_t *idx_out,
                                 float *dist_out, dgr_stream stream) {
  DGR_REQUIRE(ctx && F0 && F1 && off0 && off1 && idx_out, "dgr_knn1_l2_batch: NULL argument");
  DGR_REQUIRE(npairs >= 1, "dgr_knn1_l2_batch: npairs=%d", npairs);
  for (int p = 0; p < npairs; ++p)
    DGR_REQUIRE(off0[p + 1] > off0[p] && off1[p + 1] > off1[p], "find_knn: pair %d has an empty feature matrix", p);
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  DGR_CHECK(ctx->arena.reset());
  return dgr_knn1_batch_impl(ctx, F0, off0, F1, off1, npairs, C, squared, idx_out, dist_out, (hipStream_t)stream);
}
