// Device-side coordinate hashing shared by coordmap.hip and kmap.hip.
// COO coordinates are int32 rows [batch, x_0 .. x_{D-1}] (NC = 1 + D ints).  The hash table is
// open addressing with linear probing; entries are row indices (-1 = empty) into the coordinate
// array that owns the keys, so a probe is: 4-byte table read (L2-resident: 2 x N entries) and,
// only on an occupied slot, an NC-int row compare.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DGR_EMPTY (-1)

__device__ __forceinline__ uint32_t dgr_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

template <int NC>
__device__ __forceinline__ uint32_t dgr_hash_row(const int32_t *c) {
  uint32_t h = 0x9747b28cu;
#pragma unroll
  for (int d = 0; d < NC; ++d) {
    uint32_t k = (uint32_t)c[d];
    k *= 0xcc9e2d51u;
    k = dgr_rotl(k, 15);
    k *= 0x1b873593u;
    h ^= k;
    h = dgr_rotl(h, 13);
    h = h * 5u + 0xe6546b64u;
  }
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

template <int NC>
__device__ __forceinline__ bool dgr_rows_equal(const int32_t *a, const int32_t *b) {
  bool eq = true;
#pragma unroll
  for (int d = 0; d < NC; ++d) eq &= (a[d] == b[d]);
  return eq;
}

// Row index of `q` in (coords, table) or -1.
template <int NC>
__device__ __forceinline__ int dgr_lookup(const int32_t *__restrict__ table, uint32_t mask,
                                          const int32_t *__restrict__ coords, const int32_t *q) {
  uint32_t slot = dgr_hash_row<NC>(q) & mask;
  while (true) {
    int v = table[slot];
    if (v == DGR_EMPTY) return -1;
    if (dgr_rows_equal<NC>(coords + (int64_t)v * NC, q)) return v;
    slot = (slot + 1) & mask;
  }
}

// U independent look-ups at once: the U table reads are issued together, then the key rows of the occupied
// slots, so a thread pays two memory latencies for U probes instead of 2 U (collisions fall back to the
// sequential probe loop, continuing behind the first slot).
template <int NC, int U>
__device__ __forceinline__ void dgr_lookup_many(const int32_t *__restrict__ table, uint32_t mask,
                                                const int32_t *__restrict__ coords, const int32_t (*q)[NC],
                                                int *found) {
  uint32_t slot[U];
  int v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    slot[u] = dgr_hash_row<NC>(q[u]) & mask;
    v[u] = table[slot[u]];
  }
  bool eq[U];
#pragma unroll
  for (int u = 0; u < U; ++u) eq[u] = dgr_rows_equal<NC>(coords + (int64_t)max(v[u], 0) * NC, q[u]);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (v[u] == DGR_EMPTY) {
      found[u] = -1;
    } else if (eq[u]) {
      found[u] = v[u];
    } else {   // collision: keep probing
      uint32_t s2 = (slot[u] + 1) & mask;
      int r = -1;
      while (true) {
        const int w = table[s2];
        if (w == DGR_EMPTY) break;
        if (dgr_rows_equal<NC>(coords + (int64_t)w * NC, q[u])) { r = w; break; }
        s2 = (s2 + 1) & mask;
      }
      found[u] = r;
    }
  }
}
