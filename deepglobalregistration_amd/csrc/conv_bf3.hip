// Rule-major sparse convolution, phase 1, for the WIDE layers (Cout >= 128: the 6-D block3/block4/conv4*
// layers that hold 95 % of the FLOPs) with f32-equivalent arithmetic on the bf16 matrix pipe.
//
// gfx950 has no reduced-precision fast path for f32 inputs (no xf32), and v_mfma_f32_32x32x2_f32 runs at the
// f32 VECTOR rate (157 TFLOP/s), 1/16 of the bf16 MFMA rate.  An f32 number is EXACTLY the sum of three bf16
// numbers (x = h + m + l: 8 + 8 + 8 = 24 significant bits, split by truncation), so
//     a b = (ah + am + al)(bh + bm + bl) = ah bh + (ah bm + am bh) + (ah bl + al bh + am bm) + O(2^-24 |a b|):
// six v_mfma_f32_32x32x16_bf16 products (each bf16 x bf16 product is exact in f32, accumulation in f32) replace
// eight v_mfma_f32_32x32x2_f32 per 16 input channels: 6 x 32 = 192 instead of 8 x 64 = 512 matrix cycles, with
// the error of the dropped terms (2^-24 relative) at the level of ONE f32 rounding.  Measured on MI355X
// (tools/microbench/bf16x3_tile.hip, 256 -> 256 tile against an f64 reference): max error / max|y| = 5.7e-7
// vs 4.3e-7 for the exact-f32 MFMA chain; the parity tests hold the same 1e-4 bound as for the f32 kernel.
// The sum order is fixed (per pair, per 16-channel block: the small terms first), so results stay bit-reproducible.
//
// Structure = sparse_conv_mfma_v2 (conv.hip): persistent XCD-aware grid over rule-major 64-pair tiles, gather
// requested two phases ahead, one barrier per phase, product rows out as 16-byte stores.  Differences:
//   * landing splits every gathered f32 into its three bf16 pieces (2 AND, 2 SUB, packs) and writes three bf16
//     planes [64 rows][64 channels] per phase buffer (the ReLU of the producer is applied before the split);
//   * weights are pre-split at load time (net.hip) into three fragment arrays in v_mfma_f32_32x32x16_bf16
//     A-operand order: WB[piece][k][s][nb][lane] = 8 bf16 = W[k][16 s + 8 (lane >> 5) + e][32 nb + (lane & 31)].
//
// NP = 2 (default): the same on the f16 pipe with TWO pieces.  An f32 x scaled by a power of two s into f16 range
// splits as s x = h + m + d with h = rn16(s x), m = rn16(s x - h), |d| <= 2^-22 |s x|: 22 of the 24 significant
// bits, every bit for most operands.  Three v_mfma_f32_32x32x16_f16 (wm.ah, wh.am, wh.ah; the dropped wm.am is
// <= 2^-22 of the product) replace the six bf16 products: half the matrix cycles again.  The scales are exact:
// one power of two per INPUT ROW (row_scale[], from dgr_row_scale: the row's largest |x| lands in [2^14, 2^15),
// so nothing overflows f16 and small channels keep their bits down to 2^-38 of the row's maximum) and one per
// layer for the weights; the product row is multiplied by the two inverse powers of two on the way out.
// Measured (tools/microbench/bf3_check.hip, f64 reference): see DESIGN.md "Numerics of the split-operand convs".
#include <stdlib.h>

#include "dgr_internal.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct ConvBf3Args {
  const float *in;
  float *y;               // [pairs, cout] per-pair product rows
  float *y_scratch;       // a few rows nobody reads (fixed-count stores of the wave-specialised kernel)
  const uint4 *wb;        // three pieces, each [K][CP/16][cout/32][64] uint4
  const int32_t *pair_in, *tile_ptr;
  const int4 *tile_desc;
  int64_t piece_stride;   // uint4 per piece
  int in_ld, in_relu, cin, cout, K;
  const float *row_scale; // NP = 2: power-of-two scale per input row
  float w_unscale;        // NP = 2: inverse of the layer's weight scale
};

// s x = h + m (+ <= 2^-22 |s x|): two f16 pieces by round-to-nearest; sx is a power of two
__device__ __forceinline__ void dgr_split2(float x, float sx, _Float16 &h, _Float16 &m) {
  const float xs = x * sx;
  h = (_Float16)xs;
  m = (_Float16)(xs - (float)h);
}
__device__ __forceinline__ float dgr_inv_pow2(float s) {   // 1 / s for a normal power of two
  return __builtin_bit_cast(float, 0x7f000000u - __builtin_bit_cast(uint32_t, s));
}

// x = h + m + l exactly, each piece a bf16 value held in the upper half of a 32-bit word
__device__ __forceinline__ void dgr_split3(float x, uint32_t &h, uint32_t &m, uint32_t &l) {
  h = __builtin_bit_cast(uint32_t, x) & 0xffff0000u;
  const float r1 = x - __builtin_bit_cast(float, h);
  m = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
  l = __builtin_bit_cast(uint32_t, r1 - __builtin_bit_cast(float, m));   // <= 8 significant bits left: exact
}

template <int CP, int MB, int NB, int NP>
__global__ void __launch_bounds__(256, 2) sparse_conv_bf16x3(ConvBf3Args a) {
  constexpr int THREADS = 256, WN = 4;
  constexpr int TM = 32 * MB;
  static_assert(TM == DGR_TILE_M, "tile height must match the kernel-map tiling");
  constexpr int NBLK = NB * WN;               // 32-column blocks of the output
  constexpr int CK = 64;                      // channels per phase
  constexpr int PPT = CP / CK;                // phases per tile
  static_assert(CP % CK == 0, "phase width must divide Cin");
  constexpr int C4K = CK / 4;                 // 16-byte f32 pieces per row per phase
  constexpr int NCH = TM * C4K / THREADS;     // pieces per thread per phase (4)
  constexpr int LDP = CK + 8;                 // bf16 elements per plane row (16 bytes of padding)
  constexpr int SK = CK / 16;                 // k-steps (16 channels) per phase
  constexpr int S = CP / 16;                  // k-steps per tile
  constexpr int PLANE = TM * LDP;             // bf16 elements per plane
  __shared__ __attribute__((aligned(16))) unsigned short Ps[2][NP][PLANE];
  __shared__ int idxbuf[4][TM];
  __shared__ float scalebuf[NP == 2 ? 4 : 1][TM];   // NP = 2: the tiles' row scales, published one phase after the indices

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.tile_ptr[a.K];
  const int per = (T + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  const int t_end = min(T, (xcd + 1) * per);
  const int nj = gridDim.x >> 3;
  const int t_first = xcd * per + (blockIdx.x >> 3);
  if (t_first >= t_end) return;
  const int n_my = (t_end - t_first + nj - 1) / nj;  // tiles of this block: t_first + i * nj
  const int NQ = n_my * PPT;

  auto locate = [&](int i, int &k, int &pstart, int &count) {
    const int4 d = a.tile_desc[t_first + i * nj];
    k = d.x; pstart = d.y; count = d.z;
  };
  auto load_idx = [&](int i) -> int {
    if (i >= n_my || tid >= TM) return -1;
    int k, pstart, count;
    locate(i, k, pstart, count);
    return tid < count ? a.pair_in[pstart + tid] : -1;
  };
  // Gathered rows travel through TWO register sets (phase p uses set p & 1): a phase's rows are requested three
  // phases before they are multiplied and landed one phase before -- two whole phases for the memory system
  // (measured: with one phase of lead the kernel ran at the speed of the gather latency under load, ~7 us)
  f32x4 G0[NCH], G1[NCH];
  uint32_t ok0 = 0, ok1 = 0;
  auto gather_piece = [&](int q, int i, f32x4 *G, uint32_t &g_ok) {   // requests only (see conv.hip)
    const int *idx = idxbuf[(q / PPT) & 3];
    const int ch = tid + i * THREADS;
    const int r = ch / C4K, c = (q % PPT) * CK + (ch % C4K) * 4;
    const int row = idx[r];
#ifdef DGR_BF3_ABL_NOGATHER   // timing ablations for tools/layer_bench.py (outputs are garbage)
    G[i] = f32x4{(float)row, (float)c, 1.f, 2.f};
#else
    G[i] = *reinterpret_cast<const f32x4 *>(a.in + (int64_t)max(row, 0) * a.in_ld + min(c, a.cin - 4));
#endif
    g_ok = (g_ok & ~(1u << i)) | ((row >= 0 && c < a.cin) ? (1u << i) : 0u);
  };
  const int relu_lo = a.in_relu ? 0 : (int)0x80000000;
  auto land_piece = [&](int q, int i, const f32x4 *G, uint32_t g_ok) {     // registers -> three bf16 planes of buffer q & 1
#ifdef DGR_BF3_ABL_NOLAND
    if (G[i].x != 123.456f) return;
#endif
    unsigned short *dst = &Ps[q & 1][0][0];
    const int ch = tid + i * THREADS;
    const bool ok = (g_ok >> i) & 1u;
    const i32x4 gi = __builtin_bit_cast(i32x4, G[i]);   // (bit_cast of a single vector ELEMENT reads element 0)
    const int o = (ch / C4K) * LDP + (ch % C4K) * 4;
    if constexpr (NP == 2) {
      const float sx = scalebuf[(q / PPT) & 3][ch / C4K];
      _Float16 hh[4], mm[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int xb = max(gi[u], relu_lo);
        xb = ok ? xb : 0;
        dgr_split2(__builtin_bit_cast(float, xb), sx, hh[u], mm[u]);
      }
      *reinterpret_cast<u32x2 *>(dst + o) = u32x2{__builtin_bit_cast(uint32_t, f16x2{hh[0], hh[1]}),
                                                  __builtin_bit_cast(uint32_t, f16x2{hh[2], hh[3]})};
      *reinterpret_cast<u32x2 *>(dst + PLANE + o) = u32x2{__builtin_bit_cast(uint32_t, f16x2{mm[0], mm[1]}),
                                                          __builtin_bit_cast(uint32_t, f16x2{mm[2], mm[3]})};
      return;
    }
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int xb = max(gi[u], relu_lo);   // pending ReLU as one integer max
      xb = ok ? xb : 0;
      dgr_split3(__builtin_bit_cast(float, xb), h[u], m[u], l[u]);
    }
    *reinterpret_cast<u32x2 *>(dst + o) = u32x2{(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
    *reinterpret_cast<u32x2 *>(dst + PLANE + o) = u32x2{(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]};
    if constexpr (NP == 3)
      *reinterpret_cast<u32x2 *>(dst + 2 * PLANE + o) = u32x2{(l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u)};
  };
  // weight operands of k-step s (16 input channels) of rule k: 3 pieces x NB column blocks, 16 bytes per lane each
  auto wload = [&](int k, int s, uint4 (*w)[NP]) {
#ifdef DGR_BF3_ABL_BONCE
    const uint4 *p = a.wb + ((int64_t)(0 * S + (s & 1)) * NBLK + wn * NB) * 64 + lane;   // L1-resident: no L2 weight stream
#else
    const uint4 *p = a.wb + ((int64_t)(k * S + s) * NBLK + wn * NB) * 64 + lane;
#endif
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) w[j][pc] = p[(int64_t)pc * a.piece_stride + j * 64];
  };

  // ---- prologue
  {
    const int i0 = load_idx(0), i1 = load_idx(1), i2 = load_idx(2), i3 = load_idx(3);
    if (tid < TM) { idxbuf[0][tid] = i0; idxbuf[1][tid] = i1; idxbuf[2][tid] = i2; idxbuf[3][tid] = i3; }
    if constexpr (NP == 2) {
      if (tid < TM) {
        const float s0 = a.row_scale[max(i0, 0)], s1 = a.row_scale[max(i1, 0)], s2 = a.row_scale[max(i2, 0)],
                    s3 = a.row_scale[max(i3, 0)];
        scalebuf[0][tid] = s0; scalebuf[1][tid] = s1; scalebuf[2][tid] = s2; scalebuf[3][tid] = s3;
      }
    }
  }
  int next_pub = 4;   // the ring holds the tiles of phases q + 1 .. q + 3; tile (q + 4) / PPT is published one phase ahead
  int idx_reg = load_idx(4);
  float sc_reg = 1.f;   // NP = 2: scale of the row published LAST phase, on its way to scalebuf
  int sc_slot = -1;
  int k, pstart, count;
  locate(0, k, pstart, count);
  int kn = k, pn = 0, cn = 0;   // the next tile's descriptor, fetched at the start of the current one
  if (n_my > 1) locate(1, kn, pn, cn);
  // k-step ring: the weights of k-step g + WD - 1 are requested at step g.  With two pieces a k-step is 12 MFMAs
  // (384 matrix cycles), less than the L2 latency under load, so the ring is 4 deep there
#ifndef DGR_BF3_WD
#define DGR_BF3_WD (NP == 2 ? 4 : 2)
#endif
  constexpr int WD = DGR_BF3_WD;
  static_assert(WD == 2 || WD == 4, "ring depth");
  uint4 w[WD][NB][NP];
#pragma unroll
  for (int g = 0; g < WD - 1; ++g) wload(k, g, w[g]);
  __syncthreads();
  // the index ring holds tiles 0..3: phases 0, 1, 2 can be requested now
#pragma unroll
  for (int i = 0; i < NCH; ++i) gather_piece(0, i, G0, ok0);
#pragma unroll
  for (int i = 0; i < NCH; ++i) land_piece(0, i, G0, ok0);
#pragma unroll
  for (int i = 0; i < NCH; ++i) gather_piece(min(1, NQ - 1), i, G1, ok1);
#pragma unroll
  for (int i = 0; i < NCH; ++i) gather_piece(min(2, NQ - 1), i, G0, ok0);
  __syncthreads();

  f32x16 acc[MB][NB];
  static_assert(NCH == SK, "one gather piece per k-step");
  // one phase; G / g_ok = the register set of phase q + 1 (landed here, then re-requested for phase q + 3)
  auto phase = [&](int q, f32x4 *G, uint32_t &g_ok) {
    const int h = q % PPT;
    if constexpr (NP == 2) {   // the scale requested when its index was published has had a whole phase
      if (sc_slot >= 0 && tid < TM) scalebuf[sc_slot][tid] = sc_reg;
      sc_slot = -1;
    }
    if ((q + 4) / PPT >= next_pub && next_pub < n_my) {
      if (tid < TM) idxbuf[next_pub & 3][tid] = idx_reg;
      if constexpr (NP == 2) {
        if (tid < TM) sc_reg = a.row_scale[max(idx_reg, 0)];
        sc_slot = next_pub & 3;
      }
      ++next_pub;
      idx_reg = load_idx(next_pub);
    }
    if (h == 0) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    const unsigned short *pl = &Ps[q & 1][0][0] + (lane & 31) * LDP + 8 * (lane >> 5);
    const int s0 = h * SK;
    const int q_req = min(q + 3, NQ - 1);   // clamped: the tail re-requests the last phase, landed where nobody reads
#pragma unroll
    for (int s = 0; s < SK; ++s) {
      // the next k-step's operands (of this tile, or the first of the next tile at the tile's last step)
      if (s0 + s + WD - 1 < S) {
        wload(k, s0 + s + WD - 1, w[(s + WD - 1) % WD]);
      } else if (q / PPT + 1 < n_my) {
        wload(kn, s0 + s + WD - 1 - S, w[(s + WD - 1) % WD]);
      }
      __builtin_amdgcn_sched_barrier(0);   // pin the prefetch ahead of the MFMA block (see conv.hip)
      uint4 ah[MB], am[MB], al[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const unsigned short *p = pl + i * 32 * LDP + s * 16;
        ah[i] = *reinterpret_cast<const uint4 *>(p);
        am[i] = *reinterpret_cast<const uint4 *>(p + PLANE);
        if constexpr (NP == 3) al[i] = *reinterpret_cast<const uint4 *>(p + 2 * PLANE);
      }
      uint4 (*wc)[NP] = w[s % WD];
      // six products per accumulator, small terms first; the (i, j) loops are innermost so that consecutive
      // MFMAs write different accumulators
#define DGR_BF3_TERM(WP, AX)                                                                                        \
  _Pragma("unroll") for (int i = 0; i < MB; ++i) _Pragma("unroll") for (int j = 0; j < NB; ++j)                     \
      acc[i][j] = NP == 3 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wc[j][WP]),                     \
                                                                   __builtin_bit_cast(bf16x8, AX[i]), acc[i][j], 0, 0, 0)     \
                          : __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wc[j][WP]),                       \
                                                                  __builtin_bit_cast(f16x8, AX[i]), acc[i][j], 0, 0, 0);
#ifdef DGR_BF3_ABL_NOMFMA
      if (ah[0].x == 0x12345678u)
#endif
      {
      if constexpr (NP == 3) {
        DGR_BF3_TERM(2, ah)   // wl . ah
        DGR_BF3_TERM(0, al)   // wh . al
        DGR_BF3_TERM(1, am)   // wm . am
      }
      DGR_BF3_TERM(1, ah)   // wm . ah
      DGR_BF3_TERM(0, am)   // wh . am
      DGR_BF3_TERM(0, ah)   // wh . ah
      }
#undef DGR_BF3_TERM
      // piece s of the NEXT phase: split + land (requested one phase ago), then request piece s of the phase after
      land_piece(q + 1, s, G, g_ok);
      gather_piece(q_req, s, G, g_ok);
    }
    if (h == PPT - 1) {
      const int pst = pstart, cnt = count;
      k = kn; pstart = pn; count = cn;
      if (q / PPT + 2 < n_my) locate(q / PPT + 2, kn, pn, cn);
      // product rows: D column (pair) = lane & 31, D row (channel) = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int r = 32 * i + (lane & 31);
        if (r < cnt) {
          float f = 1.f;
          if constexpr (NP == 2) f = dgr_inv_pow2(scalebuf[(q / PPT) & 3][r]) * a.w_unscale;
#ifdef DGR_BF3_ABL_STORE0
          float *dst = a.y + (int64_t)(r + 64 * (blockIdx.x & 1023)) * a.cout;   // L2-resident product rows
#else
          float *dst = a.y + (int64_t)(pst + r) * a.cout;
#endif
#pragma unroll
          for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int col = 32 * (wn * NB + j) + 8 * g + 4 * (lane >> 5);
              f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
              if constexpr (NP == 2) v *= f;
              *reinterpret_cast<f32x4 *>(dst + col) = v;
            }
        }
      }
    }
    __syncthreads();
  };
  for (int q = 0; q < NQ; q += 2) {   // register sets alternate statically
    phase(q, G1, ok1);
    if (q + 1 < NQ) phase(q + 1, G0, ok0);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Wave-specialised variant of the two-piece kernel (the default for the wide layers).
//
// Why: s_waitcnt vmcnt counts loads AND stores, in order.  In sparse_conv_bf16x3 every wave issues the gathers
// (HBM / MALL latency, requested phases ahead), its weight fragments (L2 latency, requested a few k-steps ahead) and
// the product-row stores; waiting for a weight fragment therefore also waits for every OLDER operation: for gathers
// issued only ~4 k-steps earlier (their long prefetch distance never materialises) and, after a tile's epilogue,
// for its 16 stores.  With the matrix work halved by the two-piece split the kernel ran at the speed of those
// latencies (ablations on the 256 -> 256 block4 layers: all MFMAs removed 2.19 -> 2.02 ms, stores removed -> 1.46 ms).
// Here each latency domain lives in waves of its own, with their own counters:
//   * waves 0..3 (compute): weight ring (L2) -> MFMAs on the landed f16 planes -> raw accumulators into an LDS
//     stage.  Their only memory operations are the weight loads, four per k-step, so every wait count is static.
//   * waves 4..7 (producers): index / scale rings; gathers FOUR phases in flight through four register sets; split
//     into the two f16 planes of a three-deep LDS ring (a plane buffer is complete one whole phase before it is
//     multiplied, so the compute waves prefetch their first operands across the phase barrier); and the finished
//     tile's product rows from the LDS stage to HBM as whole rows (scaled back by the two inverse powers of two),
//     spread over the phases of the next tile -- a store is many phases old before any wait reaches it.
// One workgroup (512 threads, 1 compute + 1 producer wave per SIMD) per CU, persistent over an XCD-aware share of
// the rule-major tiles as before.  Same sums in the same order as sparse_conv_bf16x3<.., 2>: bit-identical output.
template <int CP, int NB>
__global__ void __launch_bounds__(512, 1) sparse_conv_f16x2_ws(ConvBf3Args a, const int4 *__restrict__ tdesc) {
  constexpr int NP = 2, MB = 2, WN = 4, PW = 4, TM = 64, CK = 64, C4K = CK / 4, LDP = CK + 8, SK = CK / 16;
  static_assert(TM == DGR_TILE_M && CP % CK == 0, "shape");
  constexpr int PPT = CP / CK, S = CP / 16, PLANE = TM * LDP, NBLK = NB * WN, COUT = 32 * NBLK;
  constexpr int NBUF = 3;      // plane buffers: multiplied | complete, prefetchable | being landed
  constexpr int RING = 16;     // tiles in the index / scale rings
  constexpr int AHEAD = 8;     // tile t + AHEAD is published at the first phase of tile t
  constexpr int NSET = 4;      // gather register sets = phases in flight
  constexpr int LEAD = NSET + 2;   // a phase is requested LEAD phases before it is multiplied, landed 2 before
  constexpr int PTH = 64 * PW; // producer threads
  constexpr int NCH = TM * C4K / PTH;   // 16-byte pieces per producer thread per phase (4)
  constexpr int WD = 4;        // weight ring: k-step g + 3 is requested at step g
  constexpr int LDS_ST = COUT + 4;      // stage row stride (floats)
  constexpr int NSTG = PPT == 1 ? 2 : 1;   // the stage is read during the next tile's first phases: two when a tile is one phase
  constexpr int NSP = PPT > 1 ? PPT - 1 : 1;            // phases of the next tile over which a tile's rows go out
  constexpr int ROWS_SP = (TM + NSP - 1) / NSP;         // rows per such phase
  constexpr int CPR = COUT / 4;                         // 16-byte pieces per product row
  constexpr int RPP = PTH / CPR;                        // rows per pass of the producer threads
  __shared__ __attribute__((aligned(16))) unsigned short Ps[NBUF][NP][PLANE];
  __shared__ __attribute__((aligned(16))) float stage[NSTG][TM][LDS_ST];
  __shared__ int idxbuf[RING][TM];
  __shared__ float scalebuf[RING][TM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.tile_ptr[a.K];
  const int per = (T + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  const int t_end = min(T, (xcd + 1) * per);
  const int nj = gridDim.x >> 3;
  const int t_first = xcd * per + (blockIdx.x >> 3);
  if (t_first >= t_end) return;
  const int n_my = (t_end - t_first + nj - 1) / nj;  // tiles of this block: t_first + i * nj
  const int NQ = n_my * PPT;
  // (k, first pair, count) of the block's i-th tile; tdesc = a.tile_desc as a restrict-qualified kernel argument, so
  // that the uniform read becomes a scalar load (its own counter) instead of a vector load that drains vmcnt
  auto desc = [&](int i) -> int4 { return tdesc[t_first + min(i, n_my - 1) * nj]; };

  if (wave >= WN) {
    // =========================================================== producers
    const int ptid = tid - 64 * WN;
    const bool pub = wave == WN;   // wave 4, lane = tile row: keeps the rings filled
    int idx_reg = -1;
    bool idx_ok = false;
    float sc_reg = 1.f;
    int4 d_reg = make_int4(0, 0, 0, 0);
#ifdef DGR_WS_STATIC_STORES
    {
      const int4 d8 = desc(AHEAD);
      idx_reg = a.pair_in[d8.y + min(lane, max(d8.z - 1, 0))];
      idx_ok = AHEAD < n_my && lane < d8.z;
      d_reg = desc(AHEAD + 1);
      const int4 d7 = desc(AHEAD - 1);
      const int v7 = (AHEAD - 1 < n_my && lane < d7.z) ? a.pair_in[d7.y + lane] : -1;
      sc_reg = a.row_scale[max(v7, 0)];
    }
#endif
    if (pub) {
      int last = -1;
      for (int j = 0; j < AHEAD; ++j) {
        const int4 d = desc(j);
        const int v = (j < n_my && lane < d.z) ? a.pair_in[d.y + lane] : -1;
        idxbuf[j][lane] = v;
        if (j < AHEAD - 1) scalebuf[j][lane] = a.row_scale[max(v, 0)];
        last = v;
      }
      sc_reg = a.row_scale[max(last, 0)];                  // tile AHEAD - 1, published by the first event
      const int4 d = desc(AHEAD);
      idx_reg = (AHEAD < n_my && lane < d.z) ? a.pair_in[d.y + lane] : -1;   // tile AHEAD
      d_reg = desc(AHEAD + 1);
    }
    __syncthreads();   // P1: rings hold tiles 0 .. AHEAD - 1
    f32x4 G[NSET][NCH];
    uint32_t okm[NSET] = {0, 0, 0, 0};
    const int relu_lo = a.in_relu ? 0 : (int)0x80000000;
    auto gather = [&](int p, f32x4 *Gs, uint32_t &ok) {   // requests only
      p = min(p, NQ - 1);
      const int *idx = idxbuf[(p / PPT) & (RING - 1)];
      const int cbase = (p % PPT) * CK;
      ok = 0;
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int ch = ptid + i * PTH;
        const int row = idx[ch / C4K];
#ifdef DGR_WS_ABL_NOGATHER
        Gs[i] = f32x4{(float)row, (float)cbase, 1.f, 2.f};
#else
        Gs[i] = *reinterpret_cast<const f32x4 *>(a.in + (int64_t)max(row, 0) * a.in_ld + cbase + (ch % C4K) * 4);
#endif
        ok |= row >= 0 ? (1u << i) : 0u;
      }
    };
    auto land = [&](int p, const f32x4 *Gs, uint32_t ok) {   // registers -> the two f16 planes of buffer p % 3
      unsigned short *dst = &Ps[p % NBUF][0][0];
      const float *sc = scalebuf[(min(p, NQ - 1) / PPT) & (RING - 1)];
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
#ifdef DGR_WS_ABL_NOLAND
        if (Gs[i].x != 123.456f) continue;
#endif
        const int ch = ptid + i * PTH;
        const float sx = sc[ch / C4K];
        const i32x4 gi = __builtin_bit_cast(i32x4, Gs[i]);   // (bit_cast of a single vector ELEMENT reads element 0)
        const bool good = (ok >> i) & 1u;
        _Float16 hh[4], mm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int xb = max(gi[u], relu_lo);   // pending ReLU as one integer max
          xb = good ? xb : 0;
          dgr_split2(__builtin_bit_cast(float, xb), sx, hh[u], mm[u]);
        }
        const int o = (ch / C4K) * LDP + (ch % C4K) * 4;
        *reinterpret_cast<u32x2 *>(dst + o) = u32x2{__builtin_bit_cast(uint32_t, f16x2{hh[0], hh[1]}),
                                                    __builtin_bit_cast(uint32_t, f16x2{hh[2], hh[3]})};
        *reinterpret_cast<u32x2 *>(dst + PLANE + o) = u32x2{__builtin_bit_cast(uint32_t, f16x2{mm[0], mm[1]}),
                                                            __builtin_bit_cast(uint32_t, f16x2{mm[2], mm[3]})};
      }
    };
    // rows [r0, r1) of finished tile t: LDS stage -> HBM, whole rows, scaled back
    auto store_rows = [&](int t, int r0, int r1) {
      const int4 d = desc(t);
      const float *sc = scalebuf[t & (RING - 1)];
      const float(*st)[LDS_ST] = stage[NSTG == 2 ? (t & 1) : 0];
      r1 = min(r1, d.z);
      const int c4 = (ptid % CPR) * 4;
#ifndef DGR_WS_ABL_NOSTORE
      for (int r = r0 + ptid / CPR; r < r1; r += RPP) {
        f32x4 v = *reinterpret_cast<const f32x4 *>(&st[r][c4]);
        v *= dgr_inv_pow2(sc[r]) * a.w_unscale;
        // streaming store: a product row is read exactly once, by reduce_rows (non-temporal load), and should not
        // push the gathered input rows out of L2 / MALL (measured: conv -2 %, the reduction behind it -8 %; with the
        // 16-byte pieces of round 1 the same hint cost 10 % because they stopped merging in L2)
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(a.y + (int64_t)(d.y + r) * a.cout + c4));
      }
#endif
    };
    // the same for rows [h ROWS_SP, (h + 1) ROWS_SP) with a FIXED number of store instructions (rows past the tile's
    // end go to a scratch row): the compiler's vmcnt bookkeeping stays exact across the phase instead of falling back
    // to the conservative count that a loop of unknown trip count forces on every later wait of this wave
    auto store_rows_static = [&](int t, int h) {
      const int4 d = desc(max(t, 0));
      const float *sc = scalebuf[t & (RING - 1)];
      const float(*st)[LDS_ST] = stage[NSTG == 2 ? (t & 1) : 0];
      const int c4 = (ptid % CPR) * 4;
      const int r1 = t < 0 ? 0 : min((h + 1) * ROWS_SP, d.z);
      constexpr int NPASS = (ROWS_SP + RPP - 1) / RPP;
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const int r = h * ROWS_SP + ptid / CPR + i * RPP;
        const bool ok = r < r1;
        const int rr = min(r, TM - 1);
        f32x4 v = *reinterpret_cast<const f32x4 *>(&st[rr][c4]);
        v *= dgr_inv_pow2(sc[rr]) * a.w_unscale;
        float *dst = ok ? a.y + (int64_t)(d.y + r) * a.cout + c4 : a.y_scratch + (ptid / CPR) * a.cout + c4;
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(dst));
      }
    };
#pragma unroll
    for (int j = 0; j < NSET; ++j) gather(j, G[j], okm[j]);
    land(0, G[0], okm[0]); gather(NSET, G[0], okm[0]);
    land(1, G[1], okm[1]); gather(NSET + 1, G[1], okm[1]);
    __syncthreads();   // P2: phases 0 and 1 are landed
    // phase q: land phase q + 2 (requested four phases ago), request phase q + LEAD into the freed set
    auto pphase = [&](int q, f32x4 *Gs, uint32_t &ok) {
      const int t = q / PPT, h = q % PPT;
#ifdef DGR_WS_STATIC_STORES
      if (h == 0) {   // every producer wave loads (static op counts); only the publishing wave writes the rings
        const int idx_pub = idx_ok ? idx_reg : -1;   // (the select happens here, a tile after the load)
        if (pub) {
          idxbuf[(t + AHEAD) & (RING - 1)][lane] = idx_pub;
          scalebuf[(t + AHEAD - 1) & (RING - 1)][lane] = sc_reg;
        }
        sc_reg = a.row_scale[max(idx_pub, 0)];
        idx_reg = a.pair_in[d_reg.y + min(lane, max(d_reg.z - 1, 0))];
        idx_ok = t + AHEAD + 1 < n_my && lane < d_reg.z;
        d_reg = desc(t + AHEAD + 2);
      }
#else
      if (pub && h == 0) {   // values loaded at the previous event have had at least a phase
        idxbuf[(t + AHEAD) & (RING - 1)][lane] = idx_reg;
        scalebuf[(t + AHEAD - 1) & (RING - 1)][lane] = sc_reg;
        sc_reg = a.row_scale[max(idx_reg, 0)];
        idx_reg = (t + AHEAD + 1 < n_my && lane < d_reg.z) ? a.pair_in[d_reg.y + lane] : -1;
        d_reg = desc(t + AHEAD + 2);
      }
#endif
      land(q + 2, Gs, ok);
      gather(q + LEAD, Gs, ok);
#ifdef DGR_WS_STATIC_STORES
      if (h < NSP) store_rows_static(t - 1, h);   // t = 0: every row goes to the scratch row
#else
      if (t > 0 && h < NSP) store_rows(t - 1, h * ROWS_SP, (h + 1) * ROWS_SP);
#endif
      __syncthreads();
    };
    for (int q = 0; q < NQ; q += NSET) {   // set of phase p = p % 4: static register indexing
      pphase(q, G[2], okm[2]);
      if (q + 1 < NQ) pphase(q + 1, G[3], okm[3]);
      if (q + 2 < NQ) pphase(q + 2, G[0], okm[0]);
      if (q + 3 < NQ) pphase(q + 3, G[1], okm[1]);
    }
    store_rows(n_my - 1, 0, TM);   // the last tile's rows (its stage is complete since the last barrier)
    return;
  }

  // ============================================================= compute waves
  const int wn = wave;
  int4 dc = desc(0), dn = desc(1);   // this tile's and the next tile's descriptor
  uint4 w[WD][NB][NP];
  auto wload = [&](int k, int s, uint4 (*ws)[NP]) {
    const uint4 *p = a.wb + ((int64_t)(k * S + s) * NBLK + wn * NB) * 64 + lane;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) ws[j][pc] = p[(int64_t)pc * a.piece_stride + j * 64];
  };
#pragma unroll
  for (int g = 0; g < WD - 1; ++g) wload(dc.x, g, w[g]);
  __syncthreads();   // P1
  __syncthreads();   // P2
  uint4 op[2][MB][NP];   // operands of the current / next k-step
  const int lofs = (lane & 31) * LDP + 8 * (lane >> 5);
  auto oload = [&](int buf, int s, uint4 (*o)[NP]) {
    const unsigned short *p = &Ps[buf][0][0] + lofs + s * 16;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      o[i][0] = *reinterpret_cast<const uint4 *>(p + i * 32 * LDP);
      o[i][1] = *reinterpret_cast<const uint4 *>(p + i * 32 * LDP + PLANE);
    }
  };
  oload(0, 0, op[0]);
  f32x16 acc[MB][NB];
  int buf = 0;
  for (int q = 0; q < NQ; ++q) {
    const int h = q % PPT;
    if (h == 0) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    const int nbuf = buf == NBUF - 1 ? 0 : buf + 1;
    const int s0 = h * SK;
#pragma unroll
    for (int s = 0; s < SK; ++s) {
      {   // unconditional (past the last tile: the last tile's descriptor again), so the wait counts are static
        const int sg = s0 + s + WD - 1;
        wload(sg < S ? dc.x : dn.x, sg < S ? sg : sg - S, w[(s + WD - 1) % WD]);
      }
      // next k-step's operands; the next phase's buffer has been complete since the last barrier
      if (s + 1 < SK) oload(buf, s + 1, op[(s + 1) & 1]);
      else oload(nbuf, 0, op[0]);
      __builtin_amdgcn_sched_barrier(0);   // pin the prefetches ahead of the MFMA block
      uint4 (*wc)[NP] = w[s % WD];
      uint4 (*oc)[NP] = op[s & 1];
#ifdef DGR_WS_ABL_NOMFMA
      if (oc[0][0].x == 0x12345678u)
#endif
      {
#define DGR_WS_TERM(WP, OP)                                                                                           \
  _Pragma("unroll") for (int i = 0; i < MB; ++i) _Pragma("unroll") for (int j = 0; j < NB; ++j)                       \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wc[j][WP]),                         \
                                                         __builtin_bit_cast(f16x8, oc[i][OP]), acc[i][j], 0, 0, 0);
      DGR_WS_TERM(1, 0)   // wm . ah
      DGR_WS_TERM(0, 1)   // wh . am
      DGR_WS_TERM(0, 0)   // wh . ah
#undef DGR_WS_TERM
      }
    }
    if (h == PPT - 1) {
      // raw accumulators -> LDS stage: D column (pair) = lane & 31, D row (channel) = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
      float(*st)[LDS_ST] = stage[NSTG == 2 ? ((q / PPT) & 1) : 0];
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        float *dst = &st[32 * i + (lane & 31)][0];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int col = 32 * (wn * NB + j) + 8 * g + 4 * (lane >> 5);
            *reinterpret_cast<f32x4 *>(dst + col) =
                f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          }
      }
      dc = dn;
      dn = desc(q / PPT + 2);
    }
    buf = nbuf;
    __syncthreads();
  }
}

template <int CP, int NB>
static int launch_ws(const ConvBf3Args &ka, int64_t tile_bound, int num_cus, hipStream_t stream) {
  int64_t grid = num_cus;
  if (tile_bound < grid) grid = tile_bound;
  grid = (grid + 7) / 8 * 8;
  if (grid < 8) grid = 8;
  sparse_conv_f16x2_ws<CP, NB><<<(int)grid, 512, 0, stream>>>(ka, ka.tile_desc);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

template <int CP, int MB, int NB, int NP>
static int launch_bf3(const ConvBf3Args &ka, int64_t tile_bound, int num_cus, hipStream_t stream) {
  static int per_cu = 0;
  if (per_cu == 0) {
    int n = 0;
    DGR_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sparse_conv_bf16x3<CP, MB, NB, NP>, 256, 0));
    per_cu = n < 1 ? 1 : (n > 8 ? 8 : n);
  }
  int64_t grid = (int64_t)num_cus * per_cu;
  if (tile_bound < grid) grid = tile_bound;
  grid = (grid + 7) / 8 * 8;
  if (grid < 8) grid = 8;
  sparse_conv_bf16x3<CP, MB, NB, NP><<<(int)grid, 256, 0, stream>>>(ka);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

// LPR = lanes per row (16 bytes each; wider rows loop): largest |x| (after the pending ReLU) -> the power of two that
// moves it into [2^14, 2^15).  A wave covers 64 / LPR rows at a time.
template <int LPR>
__global__ void __launch_bounds__(256) row_scale_kernel(const float *__restrict__ in, int in_ld, int cin, int relu,
                                                        const int32_t *__restrict__ n_dev, float *__restrict__ out) {
  constexpr int RPW = 64 / LPR;   // rows per wave
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR, l = lane % LPR;
  const int n = *n_dev;
  const int relu_lo = relu ? 0 : (int)0x80000000;
  const int64_t stride = (int64_t)gridDim.x * 4 * RPW;
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW; r0 < n; r0 += stride) {
    const int64_t r = r0 + sub;
    uint32_t mx = 0;
    if (r < n) {
      const float *row = in + r * in_ld;
      for (int c = l * 4; c < cin; c += LPR * 4) {
        const i32x4 v = __builtin_bit_cast(i32x4, *reinterpret_cast<const f32x4 *>(row + c));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int vi = v[u];
          mx = max(mx, (uint32_t)max(vi, relu_lo) & 0x7fffffffu);   // |x| as an integer: monotone in the magnitude
        }
      }
    }
#pragma unroll
    for (int d = LPR / 2; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if (l == 0 && r < n) {
      // biased exponent clamped to [20, 240]: scale = 2^(14 - (e - 127)); zero / denormal rows get a finite scale
      const uint32_t e = min(max(mx >> 23, 20u), 240u);
      out[r] = mx == 0 ? 1.f : __builtin_bit_cast(float, (268u - e) << 23);
    }
  }
}

int dgr_row_scale(const float *in, int in_ld, int cin, int relu, const int32_t *n_dev, int64_t n_cap, float *out,
                  hipStream_t stream) {
  DGR_REQUIRE((cin & 3) == 0 && (in_ld & 3) == 0 && cin >= 4, "row scale: channel count and row stride must be multiples of 4");
  const int lpr = cin >= 256 ? 64 : cin >= 128 ? 32 : cin >= 64 ? 16 : cin >= 32 ? 8 : 4;   // power of two: shuffle tree
  int64_t grid = dgr_ceil_div(n_cap, 4 * (64 / lpr));
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
  switch (lpr) {
    case 64: row_scale_kernel<64><<<(int)grid, 256, 0, stream>>>(in, in_ld, cin, relu, n_dev, out); break;
    case 32: row_scale_kernel<32><<<(int)grid, 256, 0, stream>>>(in, in_ld, cin, relu, n_dev, out); break;
    case 16: row_scale_kernel<16><<<(int)grid, 256, 0, stream>>>(in, in_ld, cin, relu, n_dev, out); break;
    case 8: row_scale_kernel<8><<<(int)grid, 256, 0, stream>>>(in, in_ld, cin, relu, n_dev, out); break;
    default: row_scale_kernel<4><<<(int)grid, 256, 0, stream>>>(in, in_ld, cin, relu, n_dev, out); break;
  }
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

bool dgr_conv_bf3_supported(int cin_pad, int cin, int cout) {
  return cin == cin_pad && (cin == 64 || cin == 128 || cin == 256) && (cout == 128 || cout == 256);
}

int dgr_conv_bf3_launch(const DgrConvLaunch &a, const void *wb, int64_t piece_stride, int pieces, float w_unscale,
                        const float *row_scale, int num_cus, hipStream_t stream, const char **kernel_name) {
  DGR_REQUIRE(a.pair_in && wb && dgr_conv_bf3_supported(a.cin_pad, a.cin, a.cout), "bf16x3 conv: unsupported layer");
  DGR_REQUIRE(pieces == 3 || (pieces == 2 && row_scale), "split-operand conv: 2 pieces need the input's row scales");
  DGR_REQUIRE((a.in_ld & 3) == 0, "bf16x3 conv: input row stride must be a multiple of 4");
  ConvBf3Args ka;
  ka.in = a.in; ka.y = a.y; ka.y_scratch = a.y_scratch; ka.wb = static_cast<const uint4 *>(wb); ka.piece_stride = piece_stride;
  ka.pair_in = a.pair_in; ka.tile_ptr = a.tile_ptr; ka.tile_desc = a.tile_desc;
  ka.in_ld = a.in_ld; ka.in_relu = a.in_relu; ka.cin = a.cin; ka.cout = a.cout; ka.K = a.K;
  ka.row_scale = row_scale; ka.w_unscale = w_unscale;
  const int64_t tile_bound = a.tile_bound > 0 ? a.tile_bound : (int64_t)num_cus * 4;
  static const bool ws = getenv("DGR_BF3_NOWS") == nullptr;   // wave-specialised two-piece kernel (default)
#define DGR_BF3(CPV, NBV)                                                              \
  do {                                                                                 \
    if (pieces == 2 && ws) {                                                           \
      if (kernel_name) *kernel_name = "sparse_conv_f16x2_ws<" #CPV ", " #NBV ">";      \
      return launch_ws<CPV, NBV>(ka, tile_bound, num_cus, stream);                     \
    }                                                                                  \
    if (pieces == 2) {                                                                 \
      if (kernel_name) *kernel_name = "sparse_conv_bf16x3<" #CPV ", 2, " #NBV ", 2>";  \
      return launch_bf3<CPV, 2, NBV, 2>(ka, tile_bound, num_cus, stream);              \
    }                                                                                  \
    if (kernel_name) *kernel_name = "sparse_conv_bf16x3<" #CPV ", 2, " #NBV ", 3>";    \
    return launch_bf3<CPV, 2, NBV, 3>(ka, tile_bound, num_cus, stream);                \
  } while (0)
  if (a.cout == 128) {
    if (a.cin == 64) DGR_BF3(64, 1);
    if (a.cin == 128) DGR_BF3(128, 1);
    DGR_BF3(256, 1);
  }
  if (a.cin == 128) DGR_BF3(128, 2);
  DGR_BF3(256, 2);
#undef DGR_BF3
}
