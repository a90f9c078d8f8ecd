// Coordinate maps: COO voxel hashing, first-occurrence unique, strided maps, voxelisation.
// Replaces the coordinate-manager part of MinkowskiEngine (ME.SparseTensor construction,
// stride-2 output maps) and ME.utils.sparse_quantize (core/deep_global_registration.py:152).
// Integer / HBM-latency-bound work: one thread per row, tables sized 2x rows (L2-resident).
#include <algorithm>

#include "dgr_internal.h"
#include "hash.h"

// ------------------------------------------------------------------------------------------
// exclusive scan (int32): block-local scan + single-block scan of the block sums
// ------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ int block_exclusive_scan_256(int v, int *lds /*>=4 ints*/, int *total) {
  // wave-level inclusive scan with DPP-free shuffles (wave64), then 4 wave totals through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) lds[wave] = x;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    int s = lds[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + x - v;
}

// one workgroup, 1024 threads x 4 items per round with a running carry: ONE launch instead of three
// for the many small scans of the map build (n <= a few 100 k), where launch boundaries dominate
__global__ void __launch_bounds__(1024) scan_single_block(const int32_t *__restrict__ in, int64_t n,
                                                          int32_t *__restrict__ out, int32_t *total_out) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 4096) {
    const int64_t i0 = base + (int64_t)threadIdx.x * 4;
    int v[4], s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = (i0 + j < n) ? in[i0 + j] : 0; s += v[j]; }
    int x = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { const int t = wsum[w]; if (w < wave) wbase += t; tot += t; }
    int ex = carry_s + wbase + x - s;
#pragma unroll
    for (int j = 0; j < 4; ++j) { if (i0 + j < n) out[i0 + j] = ex; ex += v[j]; }
    __syncthreads();
    if (threadIdx.x == 0) carry_s += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry_s;
}

// Up to DGR_SCAN_MAX independent scans in the SAME three launches (blockIdx.y = array): the map builders need two or
// three scans at the same point (row pointers of the out-major and the in-major CSR, cell bases), and at these
// sizes a scan is three ~4 us kernels + two launch boundaries, i.e. pure launch overhead.
struct ScanJobs {
  const int32_t *in[DGR_SCAN_MAX];
  int32_t *out[DGR_SCAN_MAX], *sums[DGR_SCAN_MAX], *total[DGR_SCAN_MAX];
  long long n[DGR_SCAN_MAX];
  int nblocks[DGR_SCAN_MAX];
};

__global__ void __launch_bounds__(SCAN_THREADS) scan_block_sums_multi(ScanJobs j) {
  __shared__ int lds[4];
  const int a = blockIdx.y;
  if ((int)blockIdx.x >= j.nblocks[a]) return;
  const int32_t *in = j.in[a];
  const long long n = j.n[a];
  long long base = (long long)blockIdx.x * SCAN_BLOCK + (long long)threadIdx.x * SCAN_ITEMS;
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) s += in[base + i];
  int tot;
  block_exclusive_scan_256(s, lds, &tot);
  if (threadIdx.x == 0) j.sums[a][blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_sums_serial_multi(ScanJobs j) {
  __shared__ int lds[4];
  const int a = blockIdx.x;
  int32_t *sums = j.sums[a];
  const int nblocks = j.nblocks[a];
  int carry = 0;
  for (int b0 = 0; b0 < nblocks; b0 += SCAN_THREADS) {
    int i = b0 + threadIdx.x;
    int v = i < nblocks ? sums[i] : 0;
    int tot;
    int ex = block_exclusive_scan_256(v, lds, &tot);
    if (i < nblocks) sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0 && j.total[a]) *j.total[a] = carry;
}

__global__ void __launch_bounds__(SCAN_THREADS) scan_final_multi(ScanJobs j) {
  __shared__ int lds[4];
  const int a = blockIdx.y;
  if ((int)blockIdx.x >= j.nblocks[a]) return;
  const int32_t *in = j.in[a];
  int32_t *out = j.out[a];
  const long long n = j.n[a];
  long long base = (long long)blockIdx.x * SCAN_BLOCK + (long long)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0;
    s += v[i];
  }
  int tot;
  int ex = block_exclusive_scan_256(s, lds, &tot) + j.sums[a][blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = ex;
    ex += v[i];
  }
}

int dgr_exclusive_scan_multi(DgrArena &arena, int count, const int32_t *const *in, int32_t *const *out, const int64_t *n,
                             int32_t *const *total_out, hipStream_t stream) {
  DGR_REQUIRE(count >= 1 && count <= DGR_SCAN_MAX, "scan: %d arrays", count);
  ScanJobs j;
  int maxb = 0;
  for (int a = 0; a < count; ++a) {
    DGR_REQUIRE(n[a] > 0, "scan: empty array");
    j.in[a] = in[a]; j.out[a] = out[a]; j.total[a] = total_out ? total_out[a] : nullptr; j.n[a] = n[a];
    j.nblocks[a] = (int)dgr_ceil_div(n[a], SCAN_BLOCK);
    DGR_ALLOC(j.sums[a], arena, int32_t, j.nblocks[a]);
    maxb = j.nblocks[a] > maxb ? j.nblocks[a] : maxb;
  }
  for (int a = count; a < DGR_SCAN_MAX; ++a) { j.in[a] = nullptr; j.out[a] = j.sums[a] = j.total[a] = nullptr; j.n[a] = 0; j.nblocks[a] = 0; }
  scan_block_sums_multi<<<dim3(maxb, count), SCAN_THREADS, 0, stream>>>(j);
  scan_sums_serial_multi<<<count, SCAN_THREADS, 0, stream>>>(j);
  scan_final_multi<<<dim3(maxb, count), SCAN_THREADS, 0, stream>>>(j);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

int dgr_exclusive_scan_i32(DgrArena &arena, const int32_t *in, int32_t *out, int64_t n,
                           int32_t *total_out, hipStream_t stream) {
  if (n <= 0) {
    if (total_out) DGR_HIP_CHECK(hipMemsetAsync(total_out, 0, sizeof(int32_t), stream));
    return DGR_OK;
  }
  if (n <= (1 << 14)) {
    scan_single_block<<<1, 1024, 0, stream>>>(in, n, out, total_out);
    DGR_LAUNCH_CHECK();
    return DGR_OK;
  }
  const int32_t *ins[1] = {in};
  int32_t *outs[1] = {out}, *tots[1] = {total_out};
  const int64_t ns[1] = {n};
  return dgr_exclusive_scan_multi(arena, 1, ins, outs, ns, tots, stream);
}

// ------------------------------------------------------------------------------------------
// first-occurrence unique over int32 key rows
// ------------------------------------------------------------------------------------------
// table[slot] ends up holding the SMALLEST row index among the rows that share a key, so the
// surviving row ("first occurrence") does not depend on thread scheduling.
template <int NC>
__global__ void unique_insert(const int32_t *__restrict__ keys, const int32_t *n_dev, int64_t n_cap,
                              int32_t *table, uint32_t mask) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t n = n_dev ? *n_dev : n_cap;
  if (r >= n) return;
  int32_t me[NC];
#pragma unroll
  for (int d = 0; d < NC; ++d) me[d] = keys[r * NC + d];
  uint32_t slot = dgr_hash_row<NC>(me) & mask;
  while (true) {
    int cur = __hip_atomic_load(&table[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == DGR_EMPTY) {
      int old = atomicCAS(&table[slot], DGR_EMPTY, (int)r);
      if (old == DGR_EMPTY) return;
      cur = old;
    }
    if (dgr_rows_equal<NC>(keys + (int64_t)cur * NC, me)) {
      atomicMin(&table[slot], (int)r);
      return;
    }
    slot = (slot + 1) & mask;
  }
}

template <int NC>
__global__ void unique_flag(const int32_t *__restrict__ keys, const int32_t *n_dev, int64_t n_cap,
                            const int32_t *__restrict__ table, uint32_t mask,
                            int32_t *__restrict__ first_flag, int32_t *dup_flag) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_cap) return;
  int64_t n = n_dev ? *n_dev : n_cap;
  if (r >= n) {
    if (first_flag) first_flag[r] = 0;
    return;
  }
  int32_t me[NC];
#pragma unroll
  for (int d = 0; d < NC; ++d) me[d] = keys[r * NC + d];
  int v = dgr_lookup<NC>(table, mask, keys, me);
  int first = (v == (int)r) ? 1 : 0;
  if (first_flag) first_flag[r] = first;
  if (!first && dup_flag) *dup_flag = 1;
}

template <int NC>
__global__ void unique_compact(const int32_t *__restrict__ keys, const int32_t *n_dev, int64_t n_cap,
                               const int32_t *__restrict__ first_flag,
                               const int32_t *__restrict__ rank, int32_t *__restrict__ out_coords,
                               int64_t *__restrict__ sel_out) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t n = n_dev ? *n_dev : n_cap;
  if (r >= n || !first_flag[r]) return;
  int64_t o = rank[r];
#pragma unroll
  for (int d = 0; d < NC; ++d) out_coords[o * NC + d] = keys[r * NC + d];
  if (sel_out) sel_out[o] = r;
}

__global__ void table_relabel(int32_t *table, uint32_t cap, const int32_t *__restrict__ rank) {
  uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= cap) return;
  int v = table[s];
  if (v >= 0) table[s] = rank[v];
}

template <int NC>
__global__ void stride_keys(const int32_t *__restrict__ coords, const int32_t *n_dev, int ts_new,
                            int32_t *__restrict__ keys) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *n_dev) return;
  const int32_t m = ~(ts_new - 1);  // ts_new is a power of two: floor(c / ts) * ts == c & ~(ts-1)
  keys[r * NC] = coords[r * NC];
#pragma unroll
  for (int d = 1; d < NC; ++d) keys[r * NC + d] = coords[r * NC + d] & m;
}

static inline int grid_for(int64_t n, int threads = 256) { return (int)dgr_ceil_div(n > 0 ? n : 1, threads); }

// keys [n,nc] -> table (row index of the surviving row per key), first_flag, rank (exclusive scan
// of first_flag), n_unique.  All device-side; n may be a device count (n_dev) bounded by n_cap.
template <int NC>
static int unique_rows_t(DgrArena &arena, const int32_t *keys, const int32_t *n_dev, int64_t n_cap,
                         int32_t *first_flag, int32_t *rank, int32_t *n_unique_dev,
                         int32_t **table_out, uint32_t *mask_out, hipStream_t stream) {
  uint32_t cap = dgr_next_pow2((uint64_t)(2 * n_cap > 64 ? 2 * n_cap : 64));
  int32_t *table;
  DGR_ALLOC(table, arena, int32_t, cap);
  DGR_HIP_CHECK(hipMemsetAsync(table, 0xff, (size_t)cap * sizeof(int32_t), stream));
  unique_insert<NC><<<grid_for(n_cap), 256, 0, stream>>>(keys, n_dev, n_cap, table, cap - 1);
  unique_flag<NC><<<grid_for(n_cap), 256, 0, stream>>>(keys, n_dev, n_cap, table, cap - 1, first_flag,
                                                      nullptr);
  DGR_LAUNCH_CHECK();
  DGR_CHECK(dgr_exclusive_scan_i32(arena, first_flag, rank, n_cap, n_unique_dev, stream));
  *table_out = table;
  *mask_out = cap - 1;
  return DGR_OK;
}

int dgr_unique_rows(DgrArena &arena, const int32_t *keys, int64_t n, int nc, int32_t *first_flag,
                    int32_t *rank, int32_t *n_unique_dev, int32_t **table_out, uint32_t *mask_out,
                    hipStream_t stream) {
  if (nc == 4)
    return unique_rows_t<4>(arena, keys, nullptr, n, first_flag, rank, n_unique_dev, table_out,
                            mask_out, stream);
  if (nc == 7)
    return unique_rows_t<7>(arena, keys, nullptr, n, first_flag, rank, n_unique_dev, table_out,
                            mask_out, stream);
  dgr_set_error("unsupported coordinate width %d", nc);
  return DGR_EINVAL;
}

// ------------------------------------------------------------------------------------------
// coordinate maps of one sparse tensor: ts = 1 (input, must be unique) and ts = 2,4,8
// ------------------------------------------------------------------------------------------
template <int NC>
static int build_coord_maps_t(DgrArena &arena, const int32_t *coords, int64_t N, DgrMapSet *ms,
                              hipStream_t stream) {
  // ts = 1: hash the caller's rows in place; a duplicate row raises the error flag.
  DgrCoordMap &c1 = ms->cm[0];
  c1.coords = const_cast<int32_t *>(coords);
  c1.n_cap = N;
  c1.ts = 1;
  DGR_ALLOC(c1.n_dev, arena, int32_t, 1);
  int32_t n32 = (int32_t)N;
  DGR_HIP_CHECK(hipMemcpyAsync(c1.n_dev, &n32, sizeof(int32_t), hipMemcpyHostToDevice, stream));
  {
    uint32_t cap = dgr_next_pow2((uint64_t)(2 * N > 64 ? 2 * N : 64));
    DGR_ALLOC(c1.table, arena, int32_t, cap);
    c1.table_mask = cap - 1;
    DGR_HIP_CHECK(hipMemsetAsync(c1.table, 0xff, (size_t)cap * sizeof(int32_t), stream));
    unique_insert<NC><<<grid_for(N), 256, 0, stream>>>(coords, nullptr, N, c1.table, cap - 1);
    unique_flag<NC><<<grid_for(N), 256, 0, stream>>>(coords, nullptr, N, c1.table, cap - 1, nullptr,
                                                    ms->overflow);
    DGR_LAUNCH_CHECK();
  }
  // ts = 2,4,8: unique(floor(c / ts) * ts), first-occurrence order
  for (int l = 1; l < 4; ++l) {
    DgrCoordMap &p = ms->cm[l - 1];
    DgrCoordMap &c = ms->cm[l];
    c.ts = p.ts * 2;
    c.n_cap = p.n_cap;
    int32_t *keys, *flag, *rank;
    DGR_ALLOC(keys, arena, int32_t, p.n_cap * NC);
    DGR_ALLOC(flag, arena, int32_t, p.n_cap);
    DGR_ALLOC(rank, arena, int32_t, p.n_cap);
    DGR_ALLOC(c.coords, arena, int32_t, c.n_cap * NC);
    DGR_ALLOC(c.n_dev, arena, int32_t, 1);
    stride_keys<NC><<<grid_for(p.n_cap), 256, 0, stream>>>(p.coords, p.n_dev, c.ts, keys);
    DGR_CHECK(unique_rows_t<NC>(arena, keys, p.n_dev, p.n_cap, flag, rank, c.n_dev, &c.table,
                                &c.table_mask, stream));
    unique_compact<NC><<<grid_for(p.n_cap), 256, 0, stream>>>(keys, p.n_dev, p.n_cap, flag, rank,
                                                             c.coords, nullptr);
    table_relabel<<<grid_for((int64_t)c.table_mask + 1), 256, 0, stream>>>(c.table, c.table_mask + 1,
                                                                           rank);
    DGR_LAUNCH_CHECK();
  }
  return DGR_OK;
}

// ------------------------------------------------------------------------------------------
// D = 6: group the rows of a coordinate map by their first-half key (batch, x0, y0, z0)
// ------------------------------------------------------------------------------------------
__global__ void half_keys_kernel(const int32_t *__restrict__ coords7, const int32_t *n_dev,
                                 int32_t *__restrict__ keys4) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *n_dev) return;
#pragma unroll
  for (int d = 0; d < 4; ++d) keys4[r * 4 + d] = coords7[r * 7 + d];
}

__global__ void bucket_count_kernel(const int32_t *__restrict__ keys4, const int32_t *n_dev,
                                    const int32_t *__restrict__ table, uint32_t mask,
                                    const int32_t *__restrict__ bkeys, int32_t *__restrict__ row_bucket,
                                    int32_t *counts) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *n_dev) return;
  int32_t q[4] = {keys4[r * 4], keys4[r * 4 + 1], keys4[r * 4 + 2], keys4[r * 4 + 3]};
  const int b = dgr_lookup<4>(table, mask, bkeys, q);
  row_bucket[r] = b;
  atomicAdd(&counts[b], 1);
}

// entry of a bucket: the row's SECOND half (x1, y1, z1) and the row itself, contiguous per bucket -- the kernel-map search
// scans a bucket's entries with sequential 16-byte reads instead of an index read + a dependent coordinate read per row.
// Coarse levels (`coords_new` given) are RENUMBERED on the way: the row's position in bucket order becomes its row number
// (dgr_build_half_buckets below): its coordinates move there, canon[new] = old, inv[old] = new; inside a bucket the
// rows keep their original order (`members`), so the numbering does not depend on the order the atomics were served in.
// Renumbered levels: the members of every bucket first (in whatever order the atomics hand out), so that the fill can
// RANK a row among its bucket's members by row number -- the new numbering is then the same in every run (pair lists,
// product rows and everything a LayerRun keeps of a coarse level are reproducible), at one short extra pass: a bucket
// has one member at the finest level and tens to hundreds at stride 8.
__global__ void bucket_members_kernel(const int32_t *__restrict__ row_bucket, const int32_t *n_dev,
                                      const int32_t *__restrict__ start, int32_t *cursor, int32_t *__restrict__ members) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *n_dev) return;
  const int b = row_bucket[r];
  members[start[b] + atomicAdd(&cursor[b], 1)] = (int32_t)r;
}

__global__ void bucket_fill_kernel(const int32_t *__restrict__ row_bucket, const int32_t *n_dev,
                                   const int32_t *__restrict__ start, int32_t *cursor,
                                   const int32_t *__restrict__ members,
                                   const int32_t *__restrict__ coords7, int4 *__restrict__ second,
                                   int32_t *__restrict__ coords_new, int32_t *__restrict__ canon, int32_t *__restrict__ inv) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *n_dev) return;
  const int b = row_bucket[r];
  const int32_t *c = coords7 + r * 7;
  int p;
  if (members) {   // deterministic: position = rank of the row number inside the bucket
    int rank = 0;
    for (int e = start[b], end = start[b + 1]; e < end; ++e) rank += members[e] < (int32_t)r;
    p = start[b] + rank;
  } else {
    p = start[b] + atomicAdd(&cursor[b], 1);   // not renumbered: the order inside a bucket is never seen
  }
  if (coords_new) {
#pragma unroll
    for (int d = 0; d < 7; ++d) coords_new[(int64_t)p * 7 + d] = c[d];
    canon[p] = (int32_t)r;
    inv[r] = p;
    second[p] = make_int4(c[4], c[5], c[6], p);
  } else {
    second[p] = make_int4(c[4], c[5], c[6], (int32_t)r);
  }
}

// ------------------------------------------------------------------------------------------
// D = 6: the rows of the coarse maps are numbered in the order of their first-half buckets (bucket_fill_kernel).  The
// kernel-map search walks the rows in that order (64 lanes = 64 rows of one bucket scan the same neighbour bucket in
// lock step); the placing pass then touches, per hit record, the row's mask / prefix / CSR data, the (row group,
// offset) cell and the pair's position -- whole cache lines each, scattered over the map when consecutive records belong
// to rows numbered in first-occurrence order (0.34 ms for the 3.57 M records of a batch's stride-8 map), neighbouring
// ones when row number = bucket position (0.25 ms).  Nothing outside the library sees a coarse map's row order (the
// getters translate through `canon`); results do not depend on it: every output row adds its pairs in ascending offset
// order whatever the rows are called.  This kernel re-labels the coordinate hash of a renumbered level.
// ------------------------------------------------------------------------------------------
struct RenumberJobs {
  const int32_t *inv[4];
  int32_t *table[4];
  uint32_t cap[4];
};
__global__ void renumber_table_kernel(RenumberJobs j) {
  const int l = blockIdx.y;
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= j.cap[l]) return;
  const int v = j.table[l][s];
  if (v >= 0) j.table[l][s] = j.inv[l][v];
}

// The first-half buckets of ALL levels of a 6-D sparse tensor, step by step together: the levels are independent once
// their coordinate maps exist, so the hash tables are cleared by one memset, the bucket counters by another, and the
// two scans per level (bucket ranks, bucket starts) run as two multi-array scans -- round 3 built level after level:
// 8 clears and 24 scan launches per forward.
// Levels >= `renumber_from` are numbered in bucket order on the way (see renumber_table_kernel below for why).
int dgr_build_half_buckets(DgrArena &arena, DgrCoordMap *cm, DgrHalfBuckets *hb, int levels, int renumber_from,
                           hipStream_t stream) {
  DGR_REQUIRE(levels >= 1 && levels <= 4 && renumber_from >= 1, "half buckets: %d levels", levels);
  int32_t *keys[4], *flag[4], *rank[4], *nb_dev[4], *row_bucket[4], *counts[4], *cursor[4];
  uint32_t cap[4];
  size_t table_words = 0, count_words = 0;
  for (int l = 0; l < levels; ++l) {
    const int64_t n = cm[l].n_cap;
    cap[l] = dgr_next_pow2((uint64_t)(2 * n > 64 ? 2 * n : 64));
    table_words += cap[l];
    count_words += 2 * (size_t)(n + 1);
  }
  int32_t *tables, *cnts;
  DGR_ALLOC(tables, arena, int32_t, table_words);
  DGR_ALLOC(cnts, arena, int32_t, count_words);
  DGR_HIP_CHECK(hipMemsetAsync(tables, 0xff, table_words * sizeof(int32_t), stream));
  DGR_HIP_CHECK(hipMemsetAsync(cnts, 0, count_words * sizeof(int32_t), stream));
  for (int l = 0; l < levels; ++l) {
    const int64_t n = cm[l].n_cap;
    hb[l].table = tables; tables += cap[l];
    hb[l].mask = cap[l] - 1;
    counts[l] = cnts; cursor[l] = cnts + (n + 1); cnts += 2 * (n + 1);
    DGR_ALLOC(keys[l], arena, int32_t, n * 4);
    DGR_ALLOC(flag[l], arena, int32_t, n);
    DGR_ALLOC(rank[l], arena, int32_t, n);
    DGR_ALLOC(nb_dev[l], arena, int32_t, 1);
    DGR_ALLOC(row_bucket[l], arena, int32_t, n);
    DGR_ALLOC(hb[l].bkeys, arena, int32_t, n * 4);
    DGR_ALLOC(hb[l].start, arena, int32_t, n + 1);
    DGR_ALLOC(hb[l].second, arena, int4, n);
  }
  // distinct first halves, first-occurrence order (unique_rows_t, all levels at once)
  for (int l = 0; l < levels; ++l) {
    const int64_t n = cm[l].n_cap;
    half_keys_kernel<<<grid_for(n), 256, 0, stream>>>(cm[l].coords, cm[l].n_dev, keys[l]);
    unique_insert<4><<<grid_for(n), 256, 0, stream>>>(keys[l], cm[l].n_dev, n, hb[l].table, hb[l].mask);
    unique_flag<4><<<grid_for(n), 256, 0, stream>>>(keys[l], cm[l].n_dev, n, hb[l].table, hb[l].mask, flag[l], nullptr);
  }
  DGR_LAUNCH_CHECK();
  {
    const int32_t *ins[4];
    int32_t *outs[4], *tots[4];
    int64_t ns[4];
    for (int l = 0; l < levels; ++l) { ins[l] = flag[l]; outs[l] = rank[l]; tots[l] = nb_dev[l]; ns[l] = cm[l].n_cap; }
    DGR_CHECK(dgr_exclusive_scan_multi(arena, levels, ins, outs, ns, tots, stream));
  }
  for (int l = 0; l < levels; ++l) {
    const int64_t n = cm[l].n_cap;
    unique_compact<4><<<grid_for(n), 256, 0, stream>>>(keys[l], cm[l].n_dev, n, flag[l], rank[l], hb[l].bkeys, nullptr);
    table_relabel<<<grid_for((int64_t)hb[l].mask + 1), 256, 0, stream>>>(hb[l].table, hb[l].mask + 1, rank[l]);
    bucket_count_kernel<<<grid_for(n), 256, 0, stream>>>(keys[l], cm[l].n_dev, hb[l].table, hb[l].mask, hb[l].bkeys, row_bucket[l],
                                                         counts[l]);
  }
  DGR_LAUNCH_CHECK();
  {
    const int32_t *ins[4];
    int32_t *outs[4], *tots[4];
    int64_t ns[4];
    for (int l = 0; l < levels; ++l) { ins[l] = counts[l]; outs[l] = hb[l].start; tots[l] = nullptr; ns[l] = cm[l].n_cap + 1; }
    DGR_CHECK(dgr_exclusive_scan_multi(arena, levels, ins, outs, ns, tots, stream));
  }
  RenumberJobs rj = {};
  int nrj = 0;
  int64_t max_cap = 0;
  for (int l = 0; l < levels; ++l) {
    const int64_t n = cm[l].n_cap;
    int32_t *coords_new = nullptr, *canon = nullptr, *inv = nullptr;
    if (l >= renumber_from) {
      DGR_ALLOC(coords_new, arena, int32_t, n * 7);
      DGR_ALLOC(canon, arena, int32_t, n);
      DGR_ALLOC(inv, arena, int32_t, n);
      rj.inv[nrj] = inv; rj.table[nrj] = cm[l].table; rj.cap[nrj] = cm[l].table_mask + 1;
      max_cap = std::max<int64_t>(max_cap, rj.cap[nrj]);
      ++nrj;
    }
    const int32_t *members = nullptr;
    if (coords_new) {   // (rank[l] is free again: the bucket ranks went into the hash table)
      bucket_members_kernel<<<grid_for(n), 256, 0, stream>>>(row_bucket[l], cm[l].n_dev, hb[l].start, cursor[l], rank[l]);
      members = rank[l];
    }
    bucket_fill_kernel<<<grid_for(n), 256, 0, stream>>>(row_bucket[l], cm[l].n_dev, hb[l].start, cursor[l], members,
                                                        cm[l].coords, hb[l].second, coords_new, canon, inv);
    if (coords_new) { cm[l].coords = coords_new; cm[l].canon = canon; }
    hb[l].built = true;
  }
  if (nrj) renumber_table_kernel<<<dim3((unsigned)grid_for(max_cap), nrj), 256, 0, stream>>>(rj);   // coordinate hash: old -> new rows
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

int dgr_build_coord_maps(DgrArena &arena, const int32_t *coords, int64_t N, DgrMapSet *ms,
                         hipStream_t stream) {
  if (ms->nc == 4) return build_coord_maps_t<4>(arena, coords, N, ms, stream);
  if (ms->nc == 7) return build_coord_maps_t<7>(arena, coords, N, ms, stream);
  dgr_set_error("unsupported dimension D=%d (3 and 6 are on the DGR path)", ms->D);
  return DGR_EINVAL;
}

// ------------------------------------------------------------------------------------------
// voxelisation (ME.utils.sparse_quantize + batched_coordinates + xyz[sel])
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void voxel_keys(const T *__restrict__ xyz, int64_t M, T inv_unused, T voxel, int32_t batch,
                           int32_t *__restrict__ keys) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= M) return;
  keys[r * 4] = batch;
#pragma unroll
  for (int d = 0; d < 3; ++d) keys[r * 4 + 1 + d] = (int32_t)floor(xyz[r * 3 + d] / voxel);
}

template <typename T>
__global__ void voxel_gather_xyz(const T *__restrict__ xyz, const int64_t *__restrict__ sel,
                                 const int32_t *n_dev, float *__restrict__ out) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *n_dev) return;
  int64_t s = sel[r];
#pragma unroll
  for (int d = 0; d < 3; ++d) out[r * 3 + d] = (float)xyz[s * 3 + d];
}

extern "C" int dgr_voxelize(dgr_ctx *ctx, const void *xyz, int is_f64, int64_t M, double voxel_size,
                            int32_t batch_index, int64_t *sel_out, int32_t *coords_out,
                            float *xyz_out, int64_t *n_out, dgr_stream stream_) {
  DGR_REQUIRE(ctx && xyz && sel_out && coords_out && n_out, "dgr_voxelize: NULL argument");
  DGR_REQUIRE(M > 0 && M < (1ll << 30), "dgr_voxelize: M=%lld out of range", (long long)M);
  DGR_REQUIRE(voxel_size > 0, "dgr_voxelize: voxel_size must be positive");
  hipStream_t stream = (hipStream_t)stream_;
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  DGR_CHECK(ctx->arena.reset());
  DgrArena &A = ctx->arena;
  int32_t *keys, *flag, *rank, *n_dev, *table;
  uint32_t mask;
  DGR_ALLOC(keys, A, int32_t, M * 4);
  DGR_ALLOC(flag, A, int32_t, M);
  DGR_ALLOC(rank, A, int32_t, M);
  DGR_ALLOC(n_dev, A, int32_t, 1);
  if (is_f64)
    voxel_keys<double><<<grid_for(M), 256, 0, stream>>>((const double *)xyz, M, 0.0, voxel_size,
                                                       batch_index, keys);
  else
    voxel_keys<float><<<grid_for(M), 256, 0, stream>>>((const float *)xyz, M, 0.f, (float)voxel_size,
                                                      batch_index, keys);
  DGR_CHECK(unique_rows_t<4>(A, keys, nullptr, M, flag, rank, n_dev, &table, &mask, stream));
  unique_compact<4><<<grid_for(M), 256, 0, stream>>>(keys, nullptr, M, flag, rank, coords_out, sel_out);
  if (xyz_out) {
    if (is_f64)
      voxel_gather_xyz<double><<<grid_for(M), 256, 0, stream>>>((const double *)xyz, sel_out, n_dev,
                                                               xyz_out);
    else
      voxel_gather_xyz<float><<<grid_for(M), 256, 0, stream>>>((const float *)xyz, sel_out, n_dev,
                                                              xyz_out);
  }
  DGR_LAUNCH_CHECK();
  int32_t n32 = 0;
  DGR_HIP_CHECK(hipMemcpyAsync(&n32, n_dev, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  DGR_HIP_CHECK(hipStreamSynchronize(stream));
  *n_out = n32;
  return DGR_OK;
}
