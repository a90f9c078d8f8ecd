// Kernel maps: for every output coordinate and every kernel offset delta_k, probe the input
// coordinate hash for (out + delta_k * ts_in).  Replaces MinkowskiEngine's kernel-map
// construction behind ME.MinkowskiConvolution(Transpose) (model/residual_block.py:15-80).
//
// Output layout (rule-major COO): pairs sorted by (k, out) -- pair_in[p], pair_out[p] for
// p in [rule_ptr[k], rule_ptr[k+1]) -- plus tile_ptr[k], the exclusive prefix of
// ceil(P_k / DGR_TILE_M) used by the sparse-conv kernel to enumerate MFMA tiles.
// Offset enumeration: first spatial dimension fastest -- kept in ONE place:
// `offset_of`.  Deterministic: two passes (count, fill) with block-level ranks, no atomics on
// the pair positions.
#include <string.h>

#include <algorithm>

#include "dgr_internal.h"
#include "hash.h"

constexpr int KM_THREADS = 256;

// delta for offset index k, in units of ts (first spatial dimension fastest)
template <int D>
__device__ __forceinline__ void offset_of(int k, int ks, int ts, int32_t *delta) {
  const int half = ks >> 1;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    delta[d] = ((k % ks) - half) * ts;
    k /= ks;
  }
}

template <int D>
__device__ __forceinline__ int probe(const int32_t *__restrict__ out_coords, int64_t o,
                                     const int32_t *delta, const int32_t *__restrict__ in_coords,
                                     const int32_t *__restrict__ in_table, uint32_t in_mask) {
  constexpr int NC = D + 1;
  int32_t q[NC];
  q[0] = out_coords[o * NC];
#pragma unroll
  for (int d = 0; d < D; ++d) q[1 + d] = out_coords[o * NC + 1 + d] + delta[d];
  return dgr_lookup<NC>(in_table, in_mask, in_coords, q);
}

// ---- pass 1, generic: grid = (row blocks, K).  Probes every (output row, offset); the hit (or -1)
// is cached in hits[k * n_cap + o] so that pass 2 never probes again, and the number of hits of
// offset k among the rows of block rb goes to block_counts[k * RB + rb].
template <int D>
__global__ void __launch_bounds__(KM_THREADS)
    kmap_search(const int32_t *__restrict__ out_coords, const int32_t *n_out_dev,
                const int32_t *__restrict__ in_coords, const int32_t *__restrict__ in_table,
                uint32_t in_mask, int ks, int ts_in, int RB, int64_t n_cap, int32_t *__restrict__ hits,
                int32_t *__restrict__ block_counts, int KW, uint32_t *mask_out, uint32_t *mask_in) {
  __shared__ int wave_cnt[KM_THREADS / 64];
  const int rb = blockIdx.x, k = blockIdx.y;
  const int n_out = *n_out_dev;
  if (rb * KM_THREADS >= n_out) {
    if (threadIdx.x == 0) block_counts[(int64_t)k * RB + rb] = 0;
    return;
  }
  int32_t delta[D];
  offset_of<D>(k, ks, ts_in, delta);
  const int64_t o = (int64_t)rb * KM_THREADS + threadIdx.x;
  int hit = -1;
  if (o < n_out) {
    hit = probe<D>(out_coords, o, delta, in_coords, in_table, in_mask);
    hits[(int64_t)k * n_cap + o] = hit;
    if (hit >= 0) {
      atomicOr(&mask_out[o * KW + (k >> 5)], 1u << (k & 31));
      if (mask_in) atomicOr(&mask_in[(int64_t)hit * KW + (k >> 5)], 1u << (k & 31));
    }
  }
  unsigned long long m = __ballot(hit >= 0);
  if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
#pragma unroll
    for (int w = 0; w < KM_THREADS / 64; ++w) s += wave_cnt[w];
    block_counts[(int64_t)k * RB + rb] = s;
  }
}

__device__ __forceinline__ int mask_rank(const uint32_t *__restrict__ m, int k) {
  int r = 0;
  const int w = k >> 5;
  for (int i = 0; i < w; ++i) r += __popc(m[i]);
  return r + __popc(m[w] & ((1u << (k & 31)) - 1u));
}

// wpre (optional): pairs of the row in the mask words before word i (a row can have all K = 729 offsets: 16 bits)
__global__ void mask_count_kernel(const uint32_t *__restrict__ mask, int KW, const int32_t *n_dev, int64_t n_cap,
                                  int32_t *__restrict__ cnt, unsigned short *__restrict__ wpre = nullptr) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_cap) return;
  int c = 0;
  if (r < *n_dev)
    for (int i = 0; i < KW; ++i) {
      if (wpre) wpre[r * KW + i] = (unsigned short)c;
      c += __popc(mask[r * KW + i]);
    }
  cnt[r] = c;
}
// the same for up to KM_JOBS6 bit matrices in ONE launch (blockIdx.y = matrix), the row's words requested together (the
// loop above alternates loads and stores: 23 dependent round trips per row; three launches of 30 us per 6-D forward)
constexpr int KM_MC_JOBS = 8;
struct MaskCountJobs {
  const uint32_t *mask[KM_MC_JOBS];
  const int32_t *n_dev[KM_MC_JOBS];
  long long n_cap[KM_MC_JOBS];
  int32_t *cnt[KM_MC_JOBS];
  unsigned short *wpre[KM_MC_JOBS];
};
template <int KW>
__global__ void mask_count_multi(MaskCountJobs J) {
  const int m = blockIdx.y;
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= J.n_cap[m]) return;
  int c = 0;
  if (r < *J.n_dev[m]) {
    uint32_t w[KW];
#pragma unroll
    for (int i = 0; i < KW; ++i) w[i] = J.mask[m][r * KW + i];
#pragma unroll
    for (int i = 0; i < KW; ++i) {
      J.wpre[m][r * KW + i] = (unsigned short)c;
      c += __popc(w[i]);
    }
  }
  J.cnt[m][r] = c;
}
// rank of offset k among the row's set offsets, with the per-word prefix of the row
__device__ __forceinline__ int mask_rank_pre(const uint32_t *__restrict__ m, const unsigned short *__restrict__ pre, int k) {
  return pre[k >> 5] + __popc(m[k >> 5] & ((1u << (k & 31)) - 1u));
}

// =========================== D = 6: bit-matrix pipeline =====================================
// A 6-D map has 729 offsets but only 2..45 neighbours per row (0.3..6 % density), so nothing of
// size [K, N] is ever materialised except ONE BIT per (row, offset):
//   bits     the search (pruned_search6) sets mask_out[o][k] (and mask_in[i][k] for maps that are
//            also used swapped).  Same-stride maps are symmetric -- (o, k) -> i  <=>  (i, K-1-k) -> o
//            -- so only offsets below the centre are searched and every hit sets both bits.
//            Every hit is also APPENDED as a record (offset, row, input row) to a hit list -- in no particular
//            order, into the searching wave's own region (HitList below: no global counter); the records replace a
//            re-probe of every set bit in the placing pass.
//   colmask  transposes the bit matrix per 64-row group (one ballot per offset), counts the pairs of every
//            (offset, 256-row block) cell -> exclusive scan = cell bases -- and of every row (CSR row pointers).
//   place    one thread per hit record: ranks the pair inside its row (CSR slot) and inside its cell (rule-major
//            position) with popcounts.  The positions depend on the bit matrix only, not on the order of the list.
// Result identical to the generic path: pairs sorted by (k, out), CSR slots in ascending k.
// All seven maps of a sparse tensor go through these phases TOGETHER (build_kernel_maps6): one clear, one multi-array
// scan, one finalisation and one tile-descriptor launch for all of them.

// hit record: offset k (10 bits) | output row (27 bits) | input row (27 bits)
__device__ __forceinline__ unsigned long long hit_pack(int k, int64_t o, int64_t in) {
  return ((unsigned long long)k << 54) | ((unsigned long long)o << 27) | (unsigned long long)in;
}
// The hit list needs no global counter (one that is bumped per hit, per wave-call or even per 128-slot chunk serialises
// on its one address at ~40 ns per atomic: measured 13 / 5.3 ms instead of 2.4 ms per batch): every wave of the search
// owns a fixed REGION of the list, hands slots out from a cursor in LDS and leaves its record count in
// wave_count[wave id].  Round 3 sized a region for the most records 64 threads can produce (3456 for the pruned
// symmetric search: 6 KB of arena per output row, of which a few per cent were ever written).  Now a region holds
// KM_REGION records -- several times what a wave of the benchmark's maps produces -- and a wave that runs out of slots
// marks itself (count -1): the bit matrix is complete either way, and the placing kernel REPLAYS the search of a marked
// wave and places its pairs directly (positions depend on the bit matrix only).  Arena per output row: 0.9 KB
// (symmetric maps) / 1.7 KB (strided maps).
constexpr int KM_REGION = 512;
constexpr int KM_JOBS6 = 8;   // kernel maps per multi-job launch of the 6-D builder (blockIdx.y = map)
struct HitList {
  unsigned long long *recs;   // [waves][region]
  int32_t *wave_count;        // [waves]; -1: the region overflowed, replay the wave
  int region;                 // records per wave
};
__device__ __forceinline__ void hit_begin(volatile int *cur) {
  if ((threadIdx.x & 63) == 0) cur[threadIdx.x >> 6] = 0;
}
// First of `n` (1 or 2) consecutive slots for this lane, or -1 once the wave's region is full.  Called from divergent
// code by the lanes that found a pair: the active lanes of the wave share one cursor update.
__device__ __forceinline__ int64_t hit_slots(int n, const HitList &h, volatile int *cur, int wave_id) {
  const unsigned long long act = __ballot(1), two = __ballot(n == 2);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  int base = 0;
  if (lane == __ffsll((long long)act) - 1) {
    base = cur[w];
    const int need = __popcll(act) + __popcll(two);
    if (base >= 0 && base + need <= h.region) {
      cur[w] = base + need;
    } else {
      cur[w] = -1;
      base = -1;
    }
  }
  base = __builtin_amdgcn_readfirstlane(base);   // the first ACTIVE lane is the one that moved the cursor
  if (base < 0) return -1;
  return (int64_t)wave_id * h.region + base + __popcll(act & below) + __popcll(two & below);
}
__device__ __forceinline__ void hit_end(const HitList &h, volatile int *cur, int wave_id) {
  if ((threadIdx.x & 63) == 0) h.wave_count[wave_id] = cur[threadIdx.x >> 6];
}

// pruned search: one thread per (output row, first-half offset).  A 6-D neighbour (ca + da, cb + db) can only exist
// among the rows whose first half equals ca + da: look that bucket up once (27 -- symmetric: 14 -- look-ups per row
// instead of 729 probes) and test the second half of its entries, which lie contiguously in `hb.second` (sequential
// 16-byte reads; round 3 read a row index and then, dependently, the row's coordinates per candidate -- at tensor
// stride 8, where a bucket holds tens to hundreds of rows, that made the generic 364-probe search the cheaper one:
// 449 us for the one stride-8 map of a batch).  `emit(k, o, in)` is called, from divergent code, by the lane that
// found input row `in` under offset k of output row o; the search kernel sets bits and records the hit, the placing
// kernel's replay of an overflowed wave places the pair.
struct PrunedArgs {
  const int32_t *out_coords, *n_out_dev;
  const int4 *out_order;   // the OUTPUT map's rows in the order of its own first-half buckets: (x1, y1, z1, row)
  int64_t n_cap;           // row capacity of the output map (thread t = ja * n_cap + position in out_order)
  DgrHalfBuckets hb;       // first-half buckets of the INPUT map
  int ts_in, symmetric;
};
// Thread t = (first-half offset ja, position p in the output map's bucket order): the 64 lanes of a wave are 64
// consecutive rows of (mostly) ONE first-half bucket under the SAME first-half offset, so they scan the same neighbour
// bucket in lock step -- every load of the loop below is one address for the whole wave -- and a wave's records are
// spread over 64 rows instead of all offsets of 4.6 rows (rows with hundreds of neighbours overflowed a region alone).
template <class Emit>
__device__ __forceinline__ void pruned_search6(const PrunedArgs &a, int64_t t, Emit &&emit) {
  const int NJ = a.symmetric ? 14 : 27;
  const int ts_in = a.ts_in;
  const int ja = (int)(t / a.n_cap);
  const int64_t p = t - (int64_t)ja * a.n_cap;
  int b = -1;
  int64_t o = 0;
  int c4 = 0, c5 = 0, c6 = 0;
  if (ja < NJ && p < *a.n_out_dev) {
    const int4 me = a.out_order[p];
    o = me.w; c4 = me.x; c5 = me.y; c6 = me.z;
    const int32_t *co = a.out_coords + o * 7;
    int32_t q[4];
    q[0] = co[0];
    q[1] = co[1] + ((ja % 3) - 1) * ts_in;
    q[2] = co[2] + (((ja / 3) % 3) - 1) * ts_in;
    q[3] = co[3] + ((ja / 9) - 1) * ts_in;
    b = dgr_lookup<4>(a.hb.table, a.hb.mask, a.hb.bkeys, q);
  }
  const int beg = b >= 0 ? a.hb.start[b] : 0, end = b >= 0 ? a.hb.start[b + 1] : 0;
  // four bucket entries per trip, fetched together
  for (int p0 = beg; p0 < end; p0 += 4) {
    int4 e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) e[u] = a.hb.second[min(p0 + u, end - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (p0 + u >= end) break;
      const int d4 = e[u].x - c4, d5 = e[u].y - c5, d6 = e[u].z - c6;
      // every component must be -ts, 0 or +ts (all coordinates of a level are multiples of ts)
      if (abs(d4) <= ts_in && abs(d5) <= ts_in && abs(d6) <= ts_in) {
        const int k = ja + 27 * ((d4 / ts_in + 1) + 3 * (d5 / ts_in + 1) + 9 * (d6 / ts_in + 1));
        if (a.symmetric && ja == 13 && k > 364) continue;  // ja == 13 mirrors onto itself: upper half found from the other row
        emit(k, o, e[u].w);
      }
    }
  }
}

// The searches of ALL maps of a sparse tensor in one launch (round 5; round 4 launched map after map: seven grids of
// 38-250 us each with their tails idle): blockIdx.y = map, blockIdx.x = the map's search block (surplus blocks of the
// smaller maps exit at once).
struct Bits6Jobs {
  PrunedArgs pa[KM_JOBS6];
  uint32_t *mask_out[KM_JOBS6], *mask_in[KM_JOBS6];
  HitList hl[KM_JOBS6];
  int blocks[KM_JOBS6];
};
__global__ void __launch_bounds__(KM_THREADS)
    kmap_bits_pruned6(Bits6Jobs J, int KW) {
  __shared__ int hit_cur[KM_THREADS / 64];
  const int jm = blockIdx.y;
  if ((int)blockIdx.x >= J.blocks[jm]) return;
  const PrunedArgs a = J.pa[jm];
  uint32_t *const mask_out = J.mask_out[jm], *const mask_in = J.mask_in[jm];
  const HitList hl = J.hl[jm];
  hit_begin(hit_cur);
  const int wid = (int)(blockIdx.x * (KM_THREADS / 64) + (threadIdx.x >> 6));
  // (no early return: every wave leaves its record count behind)
  pruned_search6(a, (int64_t)blockIdx.x * KM_THREADS + threadIdx.x, [&](int k, int64_t o, int in) {
    atomicOr(&mask_out[o * KW + (k >> 5)], 1u << (k & 31));
    const int km = 728 - k;
    if (a.symmetric) {
      if (km != k) atomicOr(&mask_out[(int64_t)in * KW + (km >> 5)], 1u << (km & 31));
    } else if (mask_in) {
      atomicOr(&mask_in[(int64_t)in * KW + (k >> 5)], 1u << (k & 31));
    }
    const int n = (a.symmetric && km != k) ? 2 : 1;
    const int64_t s = hit_slots(n, hl, hit_cur, wid);
    if (s >= 0) {
      hl.recs[s] = hit_pack(k, o, in);
      if (n == 2) hl.recs[s + 1] = hit_pack(km, in, o);
    }
  });
  hit_end(hl, hit_cur, wid);
}

// transposed bit matrix: cell[g * K + k] = {rows of 64-row group g that have offset k (one ballot, 2 words),
// rule-major position of the group's first pair of that offset, 0}; counts[k * RB + rb] = pairs of the
// (offset, 256-row block) cell.  A pair's position = cell base (exclusive scan of counts) + the pairs of the earlier
// groups of the same block (here) + its rank inside the group's ballot (place_pair).
constexpr int KM_KMAX = 736;
struct Colmask6Jobs {   // blockIdx.y = map, blockIdx.x = 256-row block of the map
  const uint32_t *mask_out[KM_JOBS6];
  const int32_t *n_out_dev[KM_JOBS6];
  int RB[KM_JOBS6];
  int4 *cell[KM_JOBS6];
  int32_t *counts[KM_JOBS6], *row_cnt[KM_JOBS6];
  unsigned short *wpre[KM_JOBS6];
};
__global__ void __launch_bounds__(KM_THREADS)
    kmap_colmask(Colmask6Jobs J, int K, int KW) {
  __shared__ unsigned long long bal[KM_THREADS / 64][KM_KMAX];
  const int jm = blockIdx.y;
  const int RB = J.RB[jm];
  if ((int)blockIdx.x >= RB) return;
  const uint32_t *__restrict__ mask_out = J.mask_out[jm];
  const int32_t *n_out_dev = J.n_out_dev[jm];
  int4 *__restrict__ cell = J.cell[jm];
  int32_t *__restrict__ counts = J.counts[jm], *__restrict__ row_cnt = J.row_cnt[jm];
  unsigned short *__restrict__ wpre = J.wpre[jm];
  const int rb = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_out = *n_out_dev;
  const int64_t o = (int64_t)rb * KM_THREADS + threadIdx.x;
  int row_pairs = 0;
  // the row's whole mask first (KW <= 24 words, all loads in flight together: the loop below alternates ballots and
  // stores, which kept the loads one memory round trip apart -- 23 dependent trips per wave)
  constexpr int KW_MAX = (KM_KMAX + 31) / 32;
  uint32_t words[KW_MAX];
#pragma unroll
  for (int w = 0; w < KW_MAX; ++w) words[w] = (w < KW && o < n_out) ? mask_out[o * KW + w] : 0u;
#pragma unroll
  for (int w = 0; w < KW_MAX; ++w) {
    if (w < KW) {   // (uniform)
      const uint32_t word = words[w];
      if (o < n_out) wpre[o * KW + w] = (unsigned short)row_pairs;
      row_pairs += __popc(word);
      unsigned long long mine = 0;
#pragma unroll
      for (int b = 0; b < 32; ++b) {
        const unsigned long long m = __ballot((word >> b) & 1u);
        if (lane == b) mine = m;
      }
      if (lane < 32 && 32 * w + lane < KM_KMAX) bal[wave][32 * w + lane] = mine;
    }
  }
  if (o < n_out) row_cnt[o] = row_pairs;   // (rows beyond the count: cleared together with the bit matrix)
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += KM_THREADS) {
    int run = 0;
#pragma unroll
    for (int g = 0; g < KM_THREADS / 64; ++g) {
      const unsigned long long m = bal[g][k];
      cell[((int64_t)rb * (KM_THREADS / 64) + g) * K + k] = make_int4((int)(m & 0xffffffffull), (int)(m >> 32), run, 0);
      run += __popcll(m);
    }
    counts[(int64_t)k * RB + rb] = run;
  }
}

// everything needed to put ONE pair (k, o) -> in into the map.  pair_out / pair_k may be NULL (maps whose consumers
// never read them): two random stores less per pair.
struct PlaceArgs {
  int K, KW, RB;
  const uint32_t *mask_out;
  const int32_t *out_ptr;
  const int4 *cell;
  const int32_t *base;
  int32_t *pair_in, *pair_out;
  uint16_t *pair_k;
  int32_t *out_pos;
  int64_t pair_cap;
  int32_t *overflow;
  const uint32_t *mask_in;
  const int32_t *in_ptr;
  int32_t *in_pos;
  const unsigned short *wpre_out, *wpre_in;
};
__device__ __forceinline__ void place_pair(const PlaceArgs &p, int k, int64_t o, int64_t in) {
  const int64_t slot = (int64_t)p.out_ptr[o] + mask_rank_pre(p.mask_out + o * p.KW, p.wpre_out + o * p.KW, k);
  const int64_t g = o >> 6;
  const int4 c = p.cell[g * p.K + k];   // ONE 16-byte read: column ballot + pairs of the block's earlier groups
  const unsigned long long cm = ((unsigned long long)(uint32_t)c.y << 32) | (uint32_t)c.x;
  const int64_t pos = (int64_t)p.base[(int64_t)k * p.RB + (g >> 2)] + c.z + __popcll(cm & ((1ull << (o & 63)) - 1ull));
  if (pos < p.pair_cap && slot < p.pair_cap) {
    p.pair_in[pos] = (int32_t)in;
    if (p.pair_out) p.pair_out[pos] = (int32_t)o;
    if (p.pair_k) p.pair_k[pos] = (uint16_t)k;
    p.out_pos[slot] = (int32_t)pos;
    if (p.mask_in) p.in_pos[p.in_ptr[in] + mask_rank_pre(p.mask_in + in * p.KW, p.wpre_in + in * p.KW, k)] = (int32_t)pos;
  } else {
    *p.overflow = 2;
  }
}

// one wave per region of the hit list: one lane per record places the pair; a wave whose region overflowed during the
// search (count -1) repeats that wave's search and places what it finds
struct Place6Jobs {   // blockIdx.y = map; the map's blocks stride over its hit-list regions
  HitList h[KM_JOBS6];
  PlaceArgs p[KM_JOBS6];
  PrunedArgs replay[KM_JOBS6];
  int n_waves[KM_JOBS6], blocks[KM_JOBS6];
};
__global__ void __launch_bounds__(KM_THREADS)
    kmap_place_hits(Place6Jobs J) {
  const int jm = blockIdx.y;
  const int nblk = J.blocks[jm];
  if ((int)blockIdx.x >= nblk) return;
  const HitList h = J.h[jm];
  const PlaceArgs p = J.p[jm];
  const PrunedArgs replay = J.replay[jm];
  const int n_waves = J.n_waves[jm];
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * (KM_THREADS / 64) + (threadIdx.x >> 6); r < n_waves; r += nblk * (KM_THREADS / 64)) {
    const int n = h.wave_count[r];
    if (n < 0) {
      pruned_search6(replay, (int64_t)r * 64 + lane, [&](int k, int64_t o, int in) {
        place_pair(p, k, o, in);
        if (replay.symmetric && 728 - k != k) place_pair(p, 728 - k, in, o);
      });
      continue;
    }
    for (int e = lane; e < n; e += 64) {
      const unsigned long long rec = h.recs[(int64_t)r * h.region + e];
      place_pair(p, (int)(rec >> 54), (int64_t)((rec >> 27) & 0x7ffffffull), (int64_t)(rec & 0x7ffffffull));
    }
  }
}

// ---- pass 2: grid = (row blocks, K / 8).  A block owns 256 output rows and 8 consecutive offsets:
// it reads the 8 cell counts, and for the non-empty cells (a few % in 6-D) ranks the cached hits
// inside the block (row order) and writes the pairs at cell_base + rank.  Sorted by (k, out), no
// atomics, no probing; 8x fewer (mostly empty) blocks than one block per cell.
constexpr int KM_KGROUP = 8;
__global__ void __launch_bounds__(KM_THREADS)
    kmap_fill(const int32_t *n_out_dev, int RB, int K, int64_t n_cap, const int32_t *__restrict__ hits,
              const int32_t *__restrict__ block_counts, const int32_t *__restrict__ block_base,
              int32_t *__restrict__ pair_in, int32_t *__restrict__ pair_out, int64_t pair_cap,
              int32_t *overflow, int KW, const uint32_t *__restrict__ mask_out,
              const int32_t *__restrict__ out_ptr, int32_t *__restrict__ out_pos,
              const uint32_t *__restrict__ mask_in, const int32_t *__restrict__ in_ptr,
              int32_t *__restrict__ in_pos, uint16_t *__restrict__ pair_k) {
  __shared__ int wave_cnt[2][KM_THREADS / 64];
  const int rb = blockIdx.x;
  const int n_out = *n_out_dev;
  if (rb * KM_THREADS >= n_out) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t o = (int64_t)rb * KM_THREADS + threadIdx.x;
  int parity = 0;
  for (int kk = 0; kk < KM_KGROUP; ++kk) {
    const int k = blockIdx.y * KM_KGROUP + kk;
    if (k >= K) break;
    const int64_t cell = (int64_t)k * RB + rb;
    if (block_counts[cell] == 0) continue;      // block-uniform
    int hit = -1;
    if (o < n_out) hit = hits[(int64_t)k * n_cap + o];
    const unsigned long long m = __ballot(hit >= 0);
    if (lane == 0) wave_cnt[parity][wave] = __popcll(m);
    __syncthreads();                            // double-buffered counts: one barrier per live cell
    if (hit >= 0) {
      int base = block_base[cell];
      for (int w = 0; w < wave; ++w) base += wave_cnt[parity][w];
      const int64_t pos = base + __popcll(m & ((1ull << lane) - 1ull));
      if (pos < pair_cap) {
        pair_in[pos] = hit;
        if (pair_out) pair_out[pos] = (int32_t)o;
        if (pair_k) pair_k[pos] = (uint16_t)k;
        // CSR slot = row start + number of this row's offsets below k (ascending-k order per row)
        out_pos[out_ptr[o] + mask_rank(mask_out + o * KW, k)] = (int32_t)pos;
        if (mask_in) in_pos[in_ptr[hit] + mask_rank(mask_in + (int64_t)hit * KW, k)] = (int32_t)pos;
      } else {
        *overflow = 2;
      }
    }
    parity ^= 1;
  }
}

// Finalisation of up to KM_MAXJOBS kernel maps in ONE launch each (blockIdx.x / blockIdx.y = map): the seven 6-D maps of
// a forward reach this point together (build_kernel_maps6).
constexpr int KM_MAXJOBS = 8;
struct FinalizeJobs {
  const int32_t *base[KM_MAXJOBS], *total[KM_MAXJOBS];
  int32_t *rule_ptr[KM_MAXJOBS], *tile_ptr[KM_MAXJOBS];
  int4 *desc[KM_MAXJOBS];
  long long tile_cap[KM_MAXJOBS];
  int K[KM_MAXJOBS], RB[KM_MAXJOBS];
};
// rule_ptr[k] = block_base[k * RB], rule_ptr[K] = total; tile_ptr = exclusive scan of ceil(P_k / DGR_TILE_M).
// One block per map, K <= 1024.
__global__ void __launch_bounds__(1024) kmap_finalize(FinalizeJobs j) {
  __shared__ int s[1024];
  const int m = blockIdx.x, K = j.K[m], RB = j.RB[m];
  const int32_t *block_base = j.base[m];
  const int k = threadIdx.x;
  int start = 0, end = 0;
  if (k < K) {
    start = block_base[(int64_t)k * RB];
    end = (k + 1 < K) ? block_base[(int64_t)(k + 1) * RB] : *j.total[m];
    j.rule_ptr[m][k] = start;
    if (k == K - 1) j.rule_ptr[m][K] = end;
  }
  const int tiles = (end - start + DGR_TILE_M - 1) / DGR_TILE_M;
  s[k] = tiles;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = (k >= d) ? s[k - d] : 0;
    __syncthreads();
    s[k] += v;
    __syncthreads();
  }
  if (k < K) j.tile_ptr[m][k] = s[k] - tiles;
  if (k == K - 1) j.tile_ptr[m][K] = s[k];
}

// one thread per tile: (k, first pair, pair count) so that the conv kernels fetch a tile with ONE
// 16-byte load instead of a 10-step binary search over tile_ptr on their critical path
__device__ __forceinline__ void tile_desc_one(const int32_t *__restrict__ tile_ptr, const int32_t *__restrict__ rule_ptr, int K,
                                              int4 *__restrict__ desc, int64_t tile_cap, int64_t t, int tile_m) {
  if (t >= tile_cap || t >= tile_ptr[K]) return;
  int lo = 0, hi = K;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tile_ptr[mid] <= t) lo = mid; else hi = mid;
  }
  const int pstart = rule_ptr[lo] + ((int)t - tile_ptr[lo]) * tile_m;
  desc[t] = make_int4(lo, pstart, min(tile_m, rule_ptr[lo + 1] - pstart), 0);
}
__global__ void tile_desc_kernel(FinalizeJobs j) {
  const int m = blockIdx.y;
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  tile_desc_one(j.tile_ptr[m], j.rule_ptr[m], j.K[m], j.desc[m], j.tile_cap[m], t, DGR_TILE_M);
}

// the parts of a kernel map that outlive the build
static int alloc_kernel_map(DgrArena &arena, int K, int64_t n_cap, int64_t n_in_cap, int max_pairs_per_row, bool need_in_csr,
                            bool want_pair_out, bool want_pair_k, DgrKernelMap *km) {
  DGR_REQUIRE(K <= 1024, "kernel volume %d > 1024 not supported", K);
  km->K = K;
  const int64_t per_row = K < max_pairs_per_row ? K : max_pairs_per_row;
  km->pair_cap = per_row * n_cap;
  DGR_REQUIRE(km->pair_cap < (1ll << 31), "kernel map too large (%lld pairs)", (long long)km->pair_cap);
  DGR_ALLOC(km->rule_ptr, arena, int32_t, K + 1);
  DGR_ALLOC(km->tile_ptr, arena, int32_t, K + 1);
  DGR_ALLOC(km->pair_in, arena, int32_t, km->pair_cap);
  km->pair_out = nullptr;
  if (want_pair_out) DGR_ALLOC(km->pair_out, arena, int32_t, km->pair_cap);
  km->tile_cap = km->pair_cap / DGR_TILE_M + K;
  DGR_ALLOC(km->tile_desc, arena, int4, km->tile_cap);
  DGR_ALLOC(km->out_ptr, arena, int32_t, n_cap + 1);
  DGR_ALLOC(km->out_pos, arena, int32_t, km->pair_cap);
  km->pair_k = nullptr;
  if (want_pair_k) DGR_ALLOC(km->pair_k, arena, uint16_t, km->pair_cap);
  if (need_in_csr) {
    DGR_ALLOC(km->in_ptr, arena, int32_t, n_in_cap + 1);
    DGR_ALLOC(km->in_pos, arena, int32_t, km->pair_cap);
  }
  return DGR_OK;
}

static void finalize_job(FinalizeJobs &fj, int m, const int32_t *base, const int32_t *total, int K, int RB, DgrKernelMap *km) {
  fj.base[m] = base; fj.total[m] = total; fj.K[m] = K; fj.RB[m] = RB;
  fj.rule_ptr[m] = km->rule_ptr; fj.tile_ptr[m] = km->tile_ptr;
  fj.desc[m] = km->tile_desc; fj.tile_cap[m] = km->tile_cap;
}

// D = 3 (rule-major maps of 3-D nets that do not run on neighbour tables; the stand-alone maps object): one map per call
static int build_kernel_map3(DgrArena &arena, const DgrCoordMap &in, const DgrCoordMap &out, int ks, int max_pairs_per_row,
                             bool need_in_csr, bool want_pair_out, bool want_pair_k, DgrKernelMap *km,
                             int32_t *overflow, hipStream_t stream) {
  constexpr int D = 3;
  const int K = ks * ks * ks;
  const int64_t n_cap = out.n_cap, n_in_cap = in.n_cap;
  const int RB = (int)dgr_ceil_div(n_cap, KM_THREADS);
  DGR_CHECK(alloc_kernel_map(arena, K, n_cap, n_in_cap, max_pairs_per_row, need_in_csr, want_pair_out, want_pair_k, km));
  const int KW = (K + 31) / 32;
  // transients (released after the last pass): per-(offset, block) counts, row bitmasks and the dense hit cache [K, n_cap]
  DgrArena::Mark mk = arena.mark();
  int32_t *counts, *base, *total;
  DGR_ALLOC(counts, arena, int32_t, (int64_t)K * RB);
  DGR_ALLOC(base, arena, int32_t, (int64_t)K * RB);
  DGR_ALLOC(total, arena, int32_t, 1);
  uint32_t *mask_out, *mask_in = nullptr;
  int32_t *cnt_out, *cnt_in = nullptr;
  {
    const size_t w_out = (size_t)(n_cap + 1) * KW, w_in = need_in_csr ? (size_t)(n_in_cap + 1) * KW : 0;
    DGR_ALLOC(mask_out, arena, uint32_t, w_out + w_in);
    if (need_in_csr) mask_in = mask_out + w_out;
    DGR_HIP_CHECK(hipMemsetAsync(mask_out, 0, (w_out + w_in) * sizeof(uint32_t), stream));
  }
  DGR_ALLOC(cnt_out, arena, int32_t, n_cap + 1);
  if (need_in_csr) DGR_ALLOC(cnt_in, arena, int32_t, n_in_cap + 1);
  int32_t *hits;
  DGR_ALLOC(hits, arena, int32_t, (int64_t)K * n_cap);
  kmap_search<D><<<dim3(RB, K), KM_THREADS, 0, stream>>>(out.coords, out.n_dev, in.coords, in.table, in.table_mask, ks,
                                                         in.ts, RB, n_cap, hits, counts, KW, mask_out, mask_in);
  DGR_LAUNCH_CHECK();
  // per-row pair counts -> CSR row pointers (out rows; in rows for maps used swapped)
  mask_count_kernel<<<(int)dgr_ceil_div(n_cap + 1, 256), 256, 0, stream>>>(mask_out, KW, out.n_dev, n_cap + 1, cnt_out);
  if (need_in_csr)
    mask_count_kernel<<<(int)dgr_ceil_div(n_in_cap + 1, 256), 256, 0, stream>>>(mask_in, KW, in.n_dev, n_in_cap + 1, cnt_in);
  {
    const int32_t *ins[3] = {cnt_out, counts, cnt_in};
    int32_t *outs[3] = {km->out_ptr, base, km->in_ptr}, *tots[3] = {nullptr, total, nullptr};
    const int64_t ns[3] = {n_cap + 1, (int64_t)K * RB, n_in_cap + 1};
    DGR_CHECK(dgr_exclusive_scan_multi(arena, need_in_csr ? 3 : 2, ins, outs, ns, tots, stream));
  }
  FinalizeJobs fj = {};
  finalize_job(fj, 0, base, total, K, RB, km);
  kmap_finalize<<<1, 1024, 0, stream>>>(fj);
  tile_desc_kernel<<<dim3((unsigned)dgr_ceil_div(km->tile_cap, 256), 1), 256, 0, stream>>>(fj);
  kmap_fill<<<dim3(RB, (K + KM_KGROUP - 1) / KM_KGROUP), KM_THREADS, 0, stream>>>(
      out.n_dev, RB, K, n_cap, hits, counts, base, km->pair_in, km->pair_out, km->pair_cap, overflow, KW, mask_out,
      km->out_ptr, km->out_pos, mask_in, km->in_ptr, km->in_pos, km->pair_k);
  DGR_LAUNCH_CHECK();
  arena.rewind(mk);
  km->built = true;
  return DGR_OK;
}

// D = 6: ALL kernel maps of a sparse tensor (four same-stride maps, three strided ones) are built phase by phase
// together -- one clear for all bit matrices, the searches and transposing passes back to back, ONE multi-array scan
// (17 arrays), one finalisation launch, one tile-descriptor launch, then the placing passes.  Round 3 built map after
// map: 21 scan launches, 7 finalisations, 7 descriptor launches and 7 clears per forward, ~5 us each and nothing else
// running next to them.
struct Kmap6Job {
  const DgrCoordMap *in, *out;
  const DgrHalfBuckets *hb;     // first-half buckets of `in`
  const DgrHalfBuckets *hb_out; // ... of `out` (the order the search walks the output rows in)
  bool need_in_csr, want_pair_out, want_pair_k;
  DgrKernelMap *km;
};
static int build_kernel_maps6(DgrArena &arena, const Kmap6Job *jobs, int nj, int max_pairs_per_row, int32_t *overflow,
                              hipStream_t stream) {
  constexpr int K = 729, KW = (K + 31) / 32;
  DGR_REQUIRE(nj >= 1 && nj <= KM_MAXJOBS, "6-D kernel maps: %d jobs", nj);
  struct Tr {   // transients of one job
    int RB, symmetric;
    int64_t n_cap, n_in_cap, hit_waves;
    int32_t *counts, *base, *total, *cnt_out, *cnt_in;
    uint32_t *mask_out, *mask_in;
    int4 *cell;
    HitList hl;
    unsigned short *wpre_out, *wpre_in;
  } t[KM_MAXJOBS];
  for (int m = 0; m < nj; ++m) {
    const Kmap6Job &J = jobs[m];
    t[m].n_cap = J.out->n_cap; t[m].n_in_cap = J.in->n_cap;
    DGR_REQUIRE(t[m].n_cap < (1ll << 27) && t[m].n_in_cap < (1ll << 27), "6-D kernel maps: more than 2^27 rows");
    DGR_CHECK(alloc_kernel_map(arena, K, t[m].n_cap, t[m].n_in_cap, max_pairs_per_row, J.need_in_csr, J.want_pair_out,
                               J.want_pair_k, J.km));
  }
  DgrArena::Mark mk = arena.mark();
  // the bit matrices (out rows; in rows for maps used swapped) and the row counts of ALL jobs: one allocation, one clear
  {
    size_t words = 0;
    for (int m = 0; m < nj; ++m)
      words += (size_t)(t[m].n_cap + 1) * KW + (jobs[m].need_in_csr ? (size_t)(t[m].n_in_cap + 1) * KW : 0) + (size_t)(t[m].n_cap + 1);
    uint32_t *pool;
    DGR_ALLOC(pool, arena, uint32_t, words);
    DGR_HIP_CHECK(hipMemsetAsync(pool, 0, words * sizeof(uint32_t), stream));
    for (int m = 0; m < nj; ++m) {
      t[m].mask_out = pool; pool += (size_t)(t[m].n_cap + 1) * KW;
      t[m].mask_in = nullptr;
      if (jobs[m].need_in_csr) { t[m].mask_in = pool; pool += (size_t)(t[m].n_in_cap + 1) * KW; }
      t[m].cnt_out = reinterpret_cast<int32_t *>(pool); pool += (size_t)(t[m].n_cap + 1);
    }
  }
  for (int m = 0; m < nj; ++m) {
    const Kmap6Job &J = jobs[m];
    Tr &r = t[m];
    r.RB = (int)dgr_ceil_div(r.n_cap, KM_THREADS);
    // same-stride maps (in and out are the SAME coordinate set) are symmetric: search half the offsets
    r.symmetric = (J.in->coords == J.out->coords && !J.need_in_csr) ? 1 : 0;
    DGR_REQUIRE(J.hb && J.hb->built && J.hb_out && J.hb_out->built, "6-D kernel map: first-half buckets missing");
    DGR_ALLOC(r.counts, arena, int32_t, (int64_t)K * r.RB);
    DGR_ALLOC(r.base, arena, int32_t, (int64_t)K * r.RB);
    DGR_ALLOC(r.total, arena, int32_t, 1);
    r.cnt_in = nullptr;
    if (J.need_in_csr) DGR_ALLOC(r.cnt_in, arena, int32_t, r.n_in_cap + 1);
    DGR_ALLOC(r.cell, arena, int4, (int64_t)r.RB * (KM_THREADS / 64) * K);
    // hit list: one region per wave of the search (thread = (first-half offset, row in bucket order)); a wave whose
    // region overflows is replayed by the placing kernel
    const int64_t search_blocks = dgr_ceil_div(r.n_cap * (r.symmetric ? 14 : 27), KM_THREADS);
    r.hit_waves = search_blocks * (KM_THREADS / 64);
    // (a wave = 64 rows under one first-half offset: a dozen records at the fine levels and in the strided maps, ~180 in
    // the stride-8 map with its 39 neighbours per row, several times that in its dense corners)
    r.hl.region = (J.in->ts >= 8 && r.symmetric) ? 2 * KM_REGION : KM_REGION / 2;
    DGR_REQUIRE(r.hit_waves < (1ll << 31), "6-D kernel map: too many search waves");
    DGR_ALLOC(r.hl.recs, arena, unsigned long long, r.hit_waves * r.hl.region);
    DGR_ALLOC(r.hl.wave_count, arena, int32_t, r.hit_waves);
    DGR_ALLOC(r.wpre_out, arena, unsigned short, (r.n_cap + 1) * KW);
    r.wpre_in = nullptr;
    if (J.need_in_csr) DGR_ALLOC(r.wpre_in, arena, unsigned short, (r.n_in_cap + 1) * KW);
  }
  // ---- search (bits + hit records) of all maps in one launch, then the transposing / counting passes in one launch
  static_assert(KM_JOBS6 >= KM_MAXJOBS, "multi-job launch tables");
  {
    Bits6Jobs bj = {};
    Colmask6Jobs cj = {};
    int max_sb = 0, max_rb = 0;
    for (int m = 0; m < nj; ++m) {
      const Kmap6Job &J = jobs[m];
      Tr &r = t[m];
      bj.pa[m] = PrunedArgs{J.out->coords, J.out->n_dev, J.hb_out->second, r.n_cap, *J.hb, J.in->ts, r.symmetric};
      bj.mask_out[m] = r.mask_out; bj.mask_in[m] = r.mask_in; bj.hl[m] = r.hl;
      bj.blocks[m] = (int)(r.hit_waves / (KM_THREADS / 64));
      max_sb = std::max(max_sb, bj.blocks[m]);
      cj.mask_out[m] = r.mask_out; cj.n_out_dev[m] = J.out->n_dev; cj.RB[m] = r.RB; cj.cell[m] = r.cell;
      cj.counts[m] = r.counts; cj.row_cnt[m] = r.cnt_out; cj.wpre[m] = r.wpre_out;
      max_rb = std::max(max_rb, r.RB);
    }
    kmap_bits_pruned6<<<dim3((unsigned)max_sb, (unsigned)nj), KM_THREADS, 0, stream>>>(bj, KW);
    kmap_colmask<<<dim3((unsigned)max_rb, (unsigned)nj), KM_THREADS, 0, stream>>>(cj, K, KW);
    {
      MaskCountJobs mj = {};
      int nm = 0;
      long long max_rows = 0;
      for (int m = 0; m < nj; ++m)
        if (jobs[m].need_in_csr) {
          mj.mask[nm] = t[m].mask_in; mj.n_dev[nm] = jobs[m].in->n_dev; mj.n_cap[nm] = t[m].n_in_cap + 1;
          mj.cnt[nm] = t[m].cnt_in; mj.wpre[nm] = t[m].wpre_in;
          max_rows = std::max(max_rows, mj.n_cap[nm]);
          ++nm;
        }
      static_assert(KM_MC_JOBS >= KM_MAXJOBS, "mask-count job table");
      if (nm) mask_count_multi<KW><<<dim3((unsigned)dgr_ceil_div(max_rows, 256), (unsigned)nm), 256, 0, stream>>>(mj);
    }
  }
  DGR_LAUNCH_CHECK();
  // ---- every scan of every map: CSR row pointers (out rows; in rows for maps used swapped) and the cell bases
  {
    const int32_t *ins[DGR_SCAN_MAX];
    int32_t *outs[DGR_SCAN_MAX], *tots[DGR_SCAN_MAX];
    int64_t ns[DGR_SCAN_MAX];
    int c = 0;
    for (int m = 0; m < nj; ++m) {
      ins[c] = t[m].cnt_out; outs[c] = jobs[m].km->out_ptr; tots[c] = nullptr; ns[c++] = t[m].n_cap + 1;
      ins[c] = t[m].counts; outs[c] = t[m].base; tots[c] = t[m].total; ns[c++] = (int64_t)K * t[m].RB;
      if (jobs[m].need_in_csr) { ins[c] = t[m].cnt_in; outs[c] = jobs[m].km->in_ptr; tots[c] = nullptr; ns[c++] = t[m].n_in_cap + 1; }
    }
    DGR_CHECK(dgr_exclusive_scan_multi(arena, c, ins, outs, ns, tots, stream));
  }
  FinalizeJobs fj = {};
  int64_t max_tiles = 0;
  for (int m = 0; m < nj; ++m) {
    finalize_job(fj, m, t[m].base, t[m].total, K, t[m].RB, jobs[m].km);
    max_tiles = std::max<int64_t>(max_tiles, jobs[m].km->tile_cap);
  }
  kmap_finalize<<<nj, 1024, 0, stream>>>(fj);
  tile_desc_kernel<<<dim3((unsigned)dgr_ceil_div(max_tiles, 256), nj), 256, 0, stream>>>(fj);
  // ---- place: all maps in one launch
  {
    Place6Jobs pj = {};
    int max_pb = 0;
    for (int m = 0; m < nj; ++m) {
      const Kmap6Job &J = jobs[m];
      Tr &r = t[m];
      DgrKernelMap *km = J.km;
      pj.p[m] = PlaceArgs{K, KW, r.RB, r.mask_out, km->out_ptr, r.cell, r.base, km->pair_in, km->pair_out, km->pair_k, km->out_pos,
                          km->pair_cap, overflow, r.mask_in, km->in_ptr, km->in_pos, r.wpre_out, r.wpre_in};
      pj.replay[m] = PrunedArgs{J.out->coords, J.out->n_dev, J.hb_out->second, r.n_cap, *J.hb, J.in->ts, r.symmetric};
      pj.h[m] = r.hl;
      pj.n_waves[m] = (int)r.hit_waves;
      pj.blocks[m] = (int)std::min<int64_t>(dgr_ceil_div(r.hit_waves, KM_THREADS / 64), 16384);
      max_pb = std::max(max_pb, pj.blocks[m]);
      km->built = true;
    }
    kmap_place_hits<<<dim3((unsigned)max_pb, (unsigned)nj), KM_THREADS, 0, stream>>>(pj);
  }
  DGR_LAUNCH_CHECK();
  arena.rewind(mk);
  return DGR_OK;
}

// =========================== D = 3: dense neighbour tables =====================================
// One launch per (input map, output map) pair: a thread owns an output row and one z-slab of the 3^3
// offsets (9 probes issued together).  SIGN = +1: q = o + delta_k * ts (same-stride and strided convs,
// out = the output map's rows); SIGN = -1: q = f - delta_k * ts (transposed convs: fine output row f,
// coarse input c with f = c + delta_k * ts_fine).  Rows beyond the device-side count get -1.
// All ten tables of a 3-D sparse tensor in ONE launch (round 5; round 4: ten launches of 11-130 us with idle tails):
// blockIdx.z = table, blockIdx.y = z-slab, blockIdx.x = row block of the table (surplus blocks of the smaller tables exit).
constexpr int NBR_JOBS = 10;
struct NbrJobs {
  const int32_t *out_coords[NBR_JOBS], *n_out_dev[NBR_JOBS], *in_coords[NBR_JOBS], *in_table[NBR_JOBS];
  uint32_t in_mask[NBR_JOBS];
  int ts[NBR_JOBS];        // signed: +ts for q = o + delta ts, -ts for q = f - delta ts (transposed convs)
  long long n_pad[NBR_JOBS];
  int32_t *nbr[NBR_JOBS];
};
__global__ void __launch_bounds__(KM_THREADS) nbr_search3(NbrJobs J) {
  const int jm = blockIdx.z;
  const int64_t n_pad = J.n_pad[jm];
  const int64_t o = (int64_t)blockIdx.x * KM_THREADS + threadIdx.x;
  if (o >= n_pad) return;
  int32_t *__restrict__ nbr = J.nbr[jm];
  const int kz = blockIdx.y;
  if (o >= *J.n_out_dev[jm]) {
#pragma unroll
    for (int u = 0; u < 9; ++u) nbr[(int64_t)(9 * kz + u) * n_pad + o] = -1;
    return;
  }
  const int sts = J.ts[jm];
  const int4 c = *reinterpret_cast<const int4 *>(J.out_coords[jm] + o * 4);
  int32_t q[9][4];
#pragma unroll
  for (int u = 0; u < 9; ++u) {   // offset index k = x + 3 y + 9 z (first spatial dimension fastest: offset_of)
    q[u][0] = c.x;
    q[u][1] = c.y + ((u % 3) - 1) * sts;
    q[u][2] = c.z + ((u / 3) - 1) * sts;
    q[u][3] = c.w + (kz - 1) * sts;
  }
  int hit[9];
  dgr_lookup_many<4, 9>(J.in_table[jm], J.in_mask[jm], J.in_coords[jm], q, hit);
#pragma unroll
  for (int u = 0; u < 9; ++u) nbr[(int64_t)(9 * kz + u) * n_pad + o] = hit[u];
}

// ---- the rows of a (fine) map grouped by the parity class of their coordinates (DgrNbrTable::perm; conv_up.hip): a
// stable counting sort in two small launches for the three fine levels together -- per 256-row block the eight class
// counts, then every block adds up the counts of the blocks before it and places its rows.  (One atomic per wave and
// class on eight global counters instead: +0.3 ms per forward, the counters serialise at ~40 ns per atomic.)
struct ClsJobs {
  const int32_t *coords[3], *n_dev[3];
  int ts[3], nb[3];
  int32_t *blk_cnt[3], *perm[3], *cls_count[3];
  long long cls_cap[3];
};
__device__ __forceinline__ int parity_class(const int32_t *coords, int64_t o, int ts) {
  const int4 c = *reinterpret_cast<const int4 *>(coords + o * 4);
  return ((c.y / ts) & 1) | (((c.z / ts) & 1) << 1) | (((c.w / ts) & 1) << 2);
}
__global__ void __launch_bounds__(KM_THREADS) cls_count_kernel(ClsJobs J) {
  __shared__ int wcnt[KM_THREADS / 64][8];
  const int l = blockIdx.y, b = blockIdx.x;
  if (b >= J.nb[l]) return;
  const int64_t o = (int64_t)b * KM_THREADS + threadIdx.x;
  const int cls = o < *J.n_dev[l] ? parity_class(J.coords[l], o, J.ts[l]) : 8;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const unsigned long long m = __ballot(cls == c);
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6][c] = __popcll(m);
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    int n = 0;
    for (int w = 0; w < KM_THREADS / 64; ++w) n += wcnt[w][threadIdx.x];
    J.blk_cnt[l][b * 8 + threadIdx.x] = n;
  }
}
__global__ void __launch_bounds__(KM_THREADS) cls_fill_kernel(ClsJobs J) {
  __shared__ int part[KM_THREADS / 64][8], before[8], wcnt[KM_THREADS / 64][8];
  const int l = blockIdx.y, b = blockIdx.x;
  if (b >= J.nb[l]) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // rows of every class in the blocks before this one
  int acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int bb = threadIdx.x; bb < b; bb += KM_THREADS) {
    const int4 lo = *reinterpret_cast<const int4 *>(J.blk_cnt[l] + bb * 8), hi = *reinterpret_cast<const int4 *>(J.blk_cnt[l] + bb * 8 + 4);
    acc[0] += lo.x; acc[1] += lo.y; acc[2] += lo.z; acc[3] += lo.w;
    acc[4] += hi.x; acc[5] += hi.y; acc[6] += hi.z; acc[7] += hi.w;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    int v = acc[c];
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    if (lane == 0) part[wave][c] = v;
  }
  const int64_t o = (int64_t)b * KM_THREADS + threadIdx.x;
  const int cls = o < *J.n_dev[l] ? parity_class(J.coords[l], o, J.ts[l]) : 8;
  int rank = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const unsigned long long m = __ballot(cls == c);
    if (lane == 0) wcnt[wave][c] = __popcll(m);
    if (cls == c) rank = __popcll(m & ((1ull << lane) - 1ull));
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    int n = 0;
    for (int w = 0; w < KM_THREADS / 64; ++w) n += part[w][threadIdx.x];
    before[threadIdx.x] = n;
    if (b == J.nb[l] - 1) {   // the last block knows every class's total
      for (int w = 0; w < KM_THREADS / 64; ++w) n += wcnt[w][threadIdx.x];
      J.cls_count[l][threadIdx.x] = n;
    }
  }
  __syncthreads();
  if (cls < 8) {
    int pos = before[cls] + rank;
    for (int w = 0; w < wave; ++w) pos += wcnt[w][cls];
    J.perm[l][(int64_t)cls * J.cls_cap[l] + pos] = (int32_t)o;
  }
}

// queue one table (allocation here, the search in nbr_tables_launch)
static int build_nbr_table(DgrArena &arena, const DgrCoordMap &in, const DgrCoordMap &out, int ts, int sign,
                           DgrNbrTable *t, NbrJobs *jobs, int *nj, int32_t *cls_count = nullptr) {
  t->K = 27;
  t->n_pad = dgr_ceil_div(out.n_cap, DGR_OS_ROWS) * DGR_OS_ROWS;
  DGR_ALLOC(t->nbr, arena, int32_t, t->n_pad * 27);
  t->perm = nullptr; t->cls_count = nullptr; t->cls_cap = 0;
  if (cls_count) {   // (the row list and the counts are written by cls_fill_kernel)
    t->cls_cap = t->n_pad;
    t->cls_count = cls_count;
    DGR_ALLOC(t->perm, arena, int32_t, 8 * t->cls_cap);
  }
  DGR_REQUIRE(*nj < NBR_JOBS, "neighbour tables: more than %d per launch", NBR_JOBS);
  const int m = (*nj)++;
  jobs->out_coords[m] = out.coords; jobs->n_out_dev[m] = out.n_dev; jobs->in_coords[m] = in.coords;
  jobs->in_table[m] = in.table; jobs->in_mask[m] = in.table_mask; jobs->ts[m] = sign > 0 ? ts : -ts;
  jobs->n_pad[m] = t->n_pad; jobs->nbr[m] = t->nbr;
  t->built = true;
  return DGR_OK;
}
static int nbr_tables_launch(const NbrJobs &jobs, int nj, hipStream_t stream) {
  long long n_max = 0;
  for (int m = 0; m < nj; ++m) n_max = std::max(n_max, jobs.n_pad[m]);
  if (nj == 0) return DGR_OK;
  nbr_search3<<<dim3((unsigned)dgr_ceil_div(n_max, KM_THREADS), 3, (unsigned)nj), KM_THREADS, 0, stream>>>(jobs);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

__global__ void nbr_count_kernel(const int32_t *__restrict__ nbr, int64_t n_pad, const int32_t *n_out_dev,
                                 unsigned long long *__restrict__ counts) {
  const int k = blockIdx.y;
  const int n = *n_out_dev;
  int c = 0;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (int64_t)gridDim.x * blockDim.x)
    c += nbr[(int64_t)k * n_pad + o] >= 0;
  for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(&counts[k], (unsigned long long)c);
}

int dgr_nbr_counts(const DgrNbrTable &t, const int32_t *n_out_dev, int64_t counts[27]) {
  unsigned long long *dev;
  DGR_HIP_CHECK(hipMalloc((void **)&dev, 27 * sizeof(unsigned long long)));
  DGR_HIP_CHECK(hipMemset(dev, 0, 27 * sizeof(unsigned long long)));
  nbr_count_kernel<<<dim3(64, 27), 256>>>(t.nbr, t.n_pad, n_out_dev, dev);
  unsigned long long host[27];
  hipError_t e = hipMemcpy(host, dev, sizeof(host), hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  DGR_HIP_CHECK(e);
  for (int k = 0; k < 27; ++k) counts[k] = (int64_t)host[k];
  return DGR_OK;
}

int dgr_build_coord_maps(DgrArena &arena, const int32_t *coords, int64_t N, DgrMapSet *ms,
                         hipStream_t stream);
int dgr_build_half_buckets(DgrArena &arena, DgrCoordMap *cm, DgrHalfBuckets *hb, int levels, int renumber_from, hipStream_t stream);

int dgr_build_maps(DgrArena &arena, const int32_t *coords, int64_t N, int D, int conv1_ks,
                   DgrMapSet *ms, hipStream_t stream, bool skip_conv1_map, bool lean, bool nbr_tables) {
  DGR_REQUIRE(D == 3 || D == 6, "D=%d not supported (the DGR path uses D=3 and D=6)", D);
  DGR_REQUIRE(conv1_ks % 2 == 1 && conv1_ks >= 1, "conv1 kernel size must be odd");
  DGR_REQUIRE(N > 0 && N < (1ll << 30), "N=%lld out of range", (long long)N);
  ms->D = D;
  ms->nc = D + 1;
  ms->conv1_ks = conv1_ks;
  if (!ms->overflow) {  // callers may pre-set a flag word that outlives arena rewinds
    DGR_ALLOC(ms->overflow, arena, int32_t, 1);
    DGR_HIP_CHECK(hipMemsetAsync(ms->overflow, 0, sizeof(int32_t), stream));
  }
  DGR_CHECK(dgr_build_coord_maps(arena, coords, N, ms, stream));
  if (nbr_tables) {
    // D = 3 network forward: ten dense neighbour tables, ONE launch for all of them -- no pair lists, no CSR, no scans
    DGR_REQUIRE(D == 3, "neighbour tables are a D = 3 structure");
    ms->use_nbr = true;
    NbrJobs nbj = {};
    int n_nbj = 0;
    int32_t *cls;   // parity-class row counts of the three fine maps the transposed convs write
    DGR_ALLOC(cls, arena, int32_t, 3 * 8);
    for (int l = 0; l < 4; ++l) DGR_CHECK(build_nbr_table(arena, ms->cm[l], ms->cm[l], ms->cm[l].ts, +1, &ms->nsame[l], &nbj, &n_nbj));
    for (int l = 0; l < 3; ++l) {
      DGR_CHECK(build_nbr_table(arena, ms->cm[l], ms->cm[l + 1], ms->cm[l].ts, +1, &ms->ndown[l], &nbj, &n_nbj));
      DGR_CHECK(build_nbr_table(arena, ms->cm[l + 1], ms->cm[l], ms->cm[l].ts, -1, &ms->nup[l], &nbj, &n_nbj, cls + 8 * l));
    }
    DGR_CHECK(nbr_tables_launch(nbj, n_nbj, stream));
    {
      ClsJobs cj = {};
      int max_nb = 0;
      for (int l = 0; l < 3; ++l) {
        cj.coords[l] = ms->cm[l].coords; cj.n_dev[l] = ms->cm[l].n_dev; cj.ts[l] = ms->cm[l].ts;
        cj.nb[l] = (int)dgr_ceil_div(ms->cm[l].n_cap, KM_THREADS);
        DGR_ALLOC(cj.blk_cnt[l], arena, int32_t, (int64_t)cj.nb[l] * 8);
        cj.perm[l] = ms->nup[l].perm; cj.cls_count[l] = ms->nup[l].cls_count; cj.cls_cap[l] = ms->nup[l].cls_cap;
        max_nb = std::max(max_nb, cj.nb[l]);
      }
      cls_count_kernel<<<dim3((unsigned)max_nb, 3), KM_THREADS, 0, stream>>>(cj);
      cls_fill_kernel<<<dim3((unsigned)max_nb, 3), KM_THREADS, 0, stream>>>(cj);
      DGR_LAUNCH_CHECK();
    }
    if (lean) {   // the network forward needs nothing else (conv1 runs fused with its neighbour search)
      if (conv1_ks != 3 && !skip_conv1_map)
        DGR_CHECK(build_kernel_map3(arena, ms->cm[0], ms->cm[0], conv1_ks, 1024, false, false, true, &ms->conv1, ms->overflow,
                                    stream));
      return DGR_OK;
    }
  }
  // capacity per output row: exact (K) in 3-D; in 6-D 3^6 = 729 offsets but measured mean
  // occupancy is 2..40 neighbours -- reserve 160 per row and raise the overflow flag beyond.
  const int cap_row = (D == 3) ? 1024 : 160;
  // `lean` (the network forward): pair_out is only read by the transposed convs (strided maps used swapped),
  // pair_k only by the small-Cin conv1 (same[0] / the conv1 map); the stand-alone maps object keeps everything
  if (D == 6) {
    DGR_REQUIRE(conv1_ks == 3, "6-D kernel maps support kernel size 3 only (got %d)", conv1_ks);
    // first-half buckets of every level: the pruned search of all seven maps (since round 4 also at tensor stride 8)
    // (the coarse maps are numbered in bucket order on the way; level 0 keeps the caller's row order)
    DGR_CHECK(dgr_build_half_buckets(arena, ms->cm, ms->hb, 4, 1, stream));
    Kmap6Job jobs[7];
    for (int l = 0; l < 4; ++l) jobs[l] = {&ms->cm[l], &ms->cm[l], &ms->hb[l], &ms->hb[l], false, !lean, !lean || l == 0, &ms->same[l]};
    // strided maps are also used swapped by the transposed convs: the in-major CSR too
    for (int l = 0; l < 3; ++l) jobs[4 + l] = {&ms->cm[l], &ms->cm[l + 1], &ms->hb[l], &ms->hb[l + 1], true, true, !lean, &ms->down[l]};
    DGR_CHECK(build_kernel_maps6(arena, jobs, 7, cap_row, ms->overflow, stream));
    ms->conv1 = ms->same[0];
    return DGR_OK;
  }
  auto build = [&](const DgrCoordMap &in, const DgrCoordMap &out, int ks, bool rev, bool want_k, DgrKernelMap *km) -> int {
    const bool want_out = !lean || rev, want_pk = !lean || want_k;
    return build_kernel_map3(arena, in, out, ks, cap_row, rev, want_out, want_pk, km, ms->overflow, stream);
  };
  for (int l = 0; l < 4; ++l) DGR_CHECK(build(ms->cm[l], ms->cm[l], 3, false, l == 0, &ms->same[l]));
  if (conv1_ks == 3)
    ms->conv1 = ms->same[0];
  else if (!skip_conv1_map)
    DGR_CHECK(build(ms->cm[0], ms->cm[0], conv1_ks, false, true, &ms->conv1));
  // strided maps are also used swapped by the transposed convs: build the in-major CSR too
  for (int l = 0; l < 3; ++l) DGR_CHECK(build(ms->cm[l], ms->cm[l + 1], 3, true, false, &ms->down[l]));
  return DGR_OK;
}

// ------------------------------------------------------------------------------------------
// stand-alone maps object (inspection / parity tests)
// ------------------------------------------------------------------------------------------
struct dgr_maps {
  dgr_ctx *ctx;
  DgrArena arena;  // private arena so the maps survive other calls on the ctx
  DgrMapSet ms;
  int32_t *coords_copy = nullptr;
  hipStream_t stream;
};

extern "C" int dgr_maps_create(dgr_ctx *ctx, const int32_t *coords, int64_t N, int D,
                               int conv1_kernel_size, dgr_maps **out, dgr_stream stream_) {
  DGR_REQUIRE(ctx && coords && out, "dgr_maps_create: NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  dgr_maps *m = new dgr_maps();
  m->ctx = ctx;
  m->stream = stream;
  int rc = DGR_OK;
  do {
    m->coords_copy = m->arena.get<int32_t>((size_t)N * (D + 1));
    if (!m->coords_copy) { rc = DGR_ENOMEM; break; }
    if (hipMemcpyAsync(m->coords_copy, coords, (size_t)N * (D + 1) * sizeof(int32_t),
                       hipMemcpyDeviceToDevice, stream) != hipSuccess) { rc = DGR_EHIP; break; }
    rc = dgr_build_maps(m->arena, m->coords_copy, N, D, conv1_kernel_size, &m->ms, stream, false, false, D == 3);
    if (rc != DGR_OK) break;
    int32_t flag = 0;
    if (hipMemcpyAsync(&flag, m->ms.overflow, sizeof(int32_t), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) { rc = DGR_EHIP; dgr_set_error("maps sync failed"); break; }
    if (flag == 1) { dgr_set_error("duplicate coordinates in the sparse tensor input"); rc = DGR_EINVAL; break; }
    if (flag == 2) { dgr_set_error("kernel-map capacity exceeded"); rc = DGR_ENOMEM; break; }
  } while (0);
  if (rc != DGR_OK) {
    m->arena.release();
    delete m;
    return rc;
  }
  *out = m;
  return DGR_OK;
}

extern "C" void dgr_maps_destroy(dgr_maps *m) {
  if (!m) return;
  (void)hipDeviceSynchronize();
  m->arena.release();
  delete m;
}

static int level_of(int ts) { return ts == 1 ? 0 : ts == 2 ? 1 : ts == 4 ? 2 : ts == 8 ? 3 : -1; }

extern "C" int dgr_maps_get_coords(dgr_maps *m, int ts, int32_t *host_out, int64_t capacity,
                                   int64_t *n) {
  DGR_REQUIRE(m && n, "NULL argument");
  int l = level_of(ts);
  DGR_REQUIRE(l >= 0, "tensor stride %d not in {1,2,4,8}", ts);
  int32_t n32 = 0;
  DGR_HIP_CHECK(hipMemcpy(&n32, m->ms.cm[l].n_dev, sizeof(int32_t), hipMemcpyDeviceToHost));
  *n = n32;
  if (host_out) {
    DGR_REQUIRE(capacity >= (int64_t)n32 * m->ms.nc, "host buffer too small");
    const int nc = m->ms.nc;
    if (!m->ms.cm[l].canon) {
      DGR_HIP_CHECK(hipMemcpy(host_out, m->ms.cm[l].coords, (size_t)n32 * nc * sizeof(int32_t), hipMemcpyDeviceToHost));
    } else {   // rows back in first-occurrence order
      std::vector<int32_t> tmp((size_t)n32 * nc), canon(n32);
      DGR_HIP_CHECK(hipMemcpy(tmp.data(), m->ms.cm[l].coords, tmp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
      DGR_HIP_CHECK(hipMemcpy(canon.data(), m->ms.cm[l].canon, canon.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
      for (int32_t p = 0; p < n32; ++p) memcpy(host_out + (size_t)canon[p] * nc, tmp.data() + (size_t)p * nc, nc * sizeof(int32_t));
    }
  }
  return DGR_OK;
}

extern "C" int dgr_maps_get_kernel_map(dgr_maps *m, int kind, int ts, int32_t *rule_ptr,
                                       int64_t rule_cap, int32_t *pair_in, int32_t *pair_out,
                                       int64_t pair_cap, int64_t *K, int64_t *P) {
  DGR_REQUIRE(m && K && P, "NULL argument");
  int l = level_of(ts);
  DGR_REQUIRE(l >= 0, "tensor stride %d not in {1,2,4,8}", ts);
  if (kind >= 3 && kind <= 5) {
    // D = 3 neighbour tables (kind 3: same stride, 4: ts -> 2 ts, 5: transposed 2 ts -> ts) as (k, in, out)
    // triplets sorted by (k, out) -- the layout the rule-major getter returns
    const DgrNbrTable *t = kind == 3 ? &m->ms.nsame[l] : (l < 3 ? (kind == 4 ? &m->ms.ndown[l] : &m->ms.nup[l]) : nullptr);
    DGR_REQUIRE(t && t->built, "no such neighbour table (kind %d, ts %d)", kind, ts);
    const DgrCoordMap &om = kind == 4 ? m->ms.cm[l + 1] : m->ms.cm[l];
    int32_t n_out = 0;
    DGR_HIP_CHECK(hipMemcpy(&n_out, om.n_dev, sizeof(int32_t), hipMemcpyDeviceToHost));
    std::vector<int32_t> tab((size_t)t->n_pad * 27);
    DGR_HIP_CHECK(hipMemcpy(tab.data(), t->nbr, tab.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    *K = 27;
    int64_t total = 0;
    for (int k = 0; k < 27; ++k) {
      if (rule_ptr && k < rule_cap) rule_ptr[k] = (int32_t)total;
      for (int o = 0; o < n_out; ++o) {
        const int32_t v = tab[(size_t)k * t->n_pad + o];
        if (v < 0) continue;
        if (pair_in && pair_out) {
          DGR_REQUIRE(total < pair_cap, "pair buffers too small");
          pair_in[total] = v;
          pair_out[total] = o;
        }
        ++total;
      }
    }
    if (rule_ptr) {
      DGR_REQUIRE(rule_cap >= 28, "rule_ptr buffer too small");
      rule_ptr[27] = (int32_t)total;
    }
    for (int64_t o = n_out; o < t->n_pad; ++o)
      for (int k = 0; k < 27; ++k) DGR_REQUIRE(tab[(size_t)k * t->n_pad + o] == -1, "neighbour table padding is not -1");
    *P = total;
    return DGR_OK;
  }
  const DgrKernelMap *km = nullptr;
  if (kind == 0) km = &m->ms.same[l];
  else if (kind == 1) km = &m->ms.conv1;
  else if (kind == 2 && l < 3) km = &m->ms.down[l];
  DGR_REQUIRE(km && km->built, "no such kernel map (kind %d, ts %d)", kind, ts);
  *K = km->K;
  int32_t total = 0;
  DGR_HIP_CHECK(hipMemcpy(&total, km->rule_ptr + km->K, sizeof(int32_t), hipMemcpyDeviceToHost));
  *P = total;
  if (rule_ptr) {
    DGR_REQUIRE(rule_cap >= km->K + 1, "rule_ptr buffer too small");
    DGR_HIP_CHECK(hipMemcpy(rule_ptr, km->rule_ptr, (size_t)(km->K + 1) * sizeof(int32_t),
                            hipMemcpyDeviceToHost));
  }
  if (pair_in && pair_out) {
    DGR_REQUIRE(pair_cap >= total, "pair buffers too small");
    DGR_HIP_CHECK(hipMemcpy(pair_in, km->pair_in, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost));
    DGR_HIP_CHECK(hipMemcpy(pair_out, km->pair_out, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost));
    // rows in first-occurrence numbering (coarse 6-D maps are numbered by bucket inside the library), pairs of a rule
    // sorted by output row in THAT numbering: the layout the getter documents
    const DgrCoordMap &cin = m->ms.cm[l], &cout = m->ms.cm[kind == 2 ? l + 1 : l];
    if (cin.canon || cout.canon) {
      auto fetch = [&](const DgrCoordMap &c, std::vector<int32_t> &v) -> int {
        if (!c.canon) return DGR_OK;
        int32_t n = 0;
        DGR_HIP_CHECK(hipMemcpy(&n, c.n_dev, sizeof(int32_t), hipMemcpyDeviceToHost));
        v.resize(n);
        DGR_HIP_CHECK(hipMemcpy(v.data(), c.canon, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
        return DGR_OK;
      };
      std::vector<int32_t> ci, co, rp(km->K + 1);
      DGR_CHECK(fetch(cin, ci));
      DGR_CHECK(fetch(cout, co));
      DGR_HIP_CHECK(hipMemcpy(rp.data(), km->rule_ptr, rp.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
      std::vector<std::pair<int32_t, int32_t>> tmp;
      for (int k = 0; k < km->K; ++k) {
        tmp.clear();
        for (int32_t q = rp[k]; q < rp[k + 1]; ++q)
          tmp.push_back({cout.canon ? co[pair_out[q]] : pair_out[q], cin.canon ? ci[pair_in[q]] : pair_in[q]});
        std::sort(tmp.begin(), tmp.end());
        for (size_t u = 0; u < tmp.size(); ++u) { pair_out[rp[k] + u] = tmp[u].first; pair_in[rp[k] + u] = tmp[u].second; }
      }
    }
  }
  return DGR_OK;
}
