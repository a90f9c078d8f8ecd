// Sparse convolution forward for gfx950: rule-major gather -> LDS tile -> f32 MFMA -> per-pair rows,
// then a deterministic segmented reduction per output row.
// Replaces the ME.MinkowskiConvolution / MinkowskiConvolutionTranspose forward invoked by every
// conv of ResUNetBN2C (model/resunet.py:598-649, model/residual_block.py:15-80).
//
// Since round 2 this file holds the exact-f32 MFMA kernel of the layers the split-operand kernels do not cover
// (k = 1 convs, C_out = 64 in the 6-D net, odd widths; all wide layers with DGR_EXACT_F32=1), the reduction pass
// shared with conv_wide.hip, and the FCGF conv1 kernels; the wide 6-D layers run in conv_wide.hip and every K = 27
// conv of the 3-D net in conv_os.hip.
//
// Phase 1 (sparse_conv_mfma_v2): the kernel map lists pairs (in, out) per kernel offset k ("rule").  A
// tile is <= 64 pairs of ONE rule: the 64 gathered input rows [64 x Cin] are staged in LDS once and
// multiplied with the rule's dense [Cin x Cout] slice on the matrix cores
// (v_mfma_f32_32x32x2_f32: exact f32, bitwise an fma chain); the product rows go to Y[pair, Cout]
// with plain coalesced stores (32 consecutive lanes write 128 consecutive bytes).  The grid is
// persistent and XCD-aware: each of the 8 XCDs walks a contiguous range of tiles, i.e. a contiguous
// range of rules, so a rule's weight slice is pulled into ONE L2.
// Phase 2 (reduce_rows): every output row sums its Y rows in ascending-k order (output-major CSR
// from kmap.hip), starting from the folded batch-norm shift (+ residual), and is written once.
// Measured alternative that this replaces: scatter with global_atomic_add_f32 is capped at ~333 G
// lanes/s chip-wide (tools/microbench/atomic_xcd.hip, independent of XCD locality) and serialises
// with the MFMA phase; plain stores + one streaming pass are faster AND bit-reproducible.
//
// Weight layout (prepared once in net.hip): W[k][s][nb][lane][c], s = Cin_pad/8 K-steps,
// nb = Cout_pad/32 column blocks, value = W_folded[k][8 s + 4 (lane>>5) + c][32 nb + (lane&31)],
// so a wave fetches the B operands of 4 MFMAs with one coalesced 16-byte load per lane.
#include <stdlib.h>

#include <map>

#include "dgr_internal.h"
#include <algorithm>
#include "hash.h"
#include "split.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ConvKArgs {
  const float *in;
  float *out;          // identity maps: written directly (acc + shift); otherwise unused
  float *y;            // [pairs, y_ld] per-pair product rows
  const float *shift;  // identity maps: folded bias / BN shift (may be null)
  const float *w;
  const int32_t *pair_in, *pair_out, *tile_ptr, *rule_ptr, *n_rows_dev;
  const int4 *tile_desc;
  int in_ld, out_ld, in_relu, y_ld;
  int cin, cin_pad, cout, K;
  int l2_normalize;    // identity maps, Cout <= 32 (one 32-channel block per row): x / (|x|_2 + 1e-8) on the way out
};

// Product rows are written once and read once.  Measured on the 256-wide layers (3.7 GB of product rows per
// launch): reading them with non-temporal loads in the reduce pass is 19 % faster (no L2 allocation for data
// that is never touched again); non-temporal STORES in the MFMA kernel cost it 10 % (the 16-byte pieces of a
// row no longer merge in L2) and narrow layers, whose product rows fit the L2 / MALL, prefer plain loads.
template <bool STREAM>
__device__ __forceinline__ f32x4 dgr_y_load(const float *p) {
  if (STREAM) return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
  return *reinterpret_cast<const f32x4 *>(p);
}
// ------------------------------------------------------------------------------------------
// v2: compile-time Cin, software-pipelined in PHASES of <= 128 input channels.
//   * LDS holds two A buffers [64 rows x CK channels]; phase q multiplies out of buffer q&1 while
//     the rows of phase q+1 (requested one phase earlier, held in <= 8 float4 registers per thread)
//     are dropped into the other buffer and the rows of phase q+2 are requested -> ONE barrier per
//     phase, and every gather has a whole MFMA phase to land.
//   * vmcnt retires in order and store acknowledgements are slow, so the first three B-operand
//     loads of a tile are issued before the previous tile's stores; B runs a 4-deep register ring.
//   * MFMA operands are swapped (D = W^T . In^T): a lane owns one pair and 4 x 4 consecutive output
//     channels, so product rows leave as 16-byte stores.
//   * row indices travel through a 4-slot LDS ring, loaded two tiles ahead.
// ------------------------------------------------------------------------------------------
#define DGR_WIDE_CK 128     // phase width (input channels) of the widest configuration
#define DGR_WIDE_WAVES 2    // waves per SIMD the widest configuration is compiled for (64 / 3 measured within +-1 %)
constexpr int conv_phase_width(int cp, int acc_blocks) {
  const int cap = acc_blocks >= 4 ? DGR_WIDE_CK : 128;
  return cp > cap ? cap : cp;
}

template <int CP, int WM, int WN, int MB, int NB, bool VEC>
__global__ void __launch_bounds__(64 * WM * WN, (MB * NB >= 4 ? DGR_WIDE_WAVES : 1)) sparse_conv_mfma_v2(ConvKArgs a) {
  constexpr int THREADS = 64 * WM * WN;
  constexpr int TM = 32 * MB * WM;
  static_assert(TM == DGR_TILE_M, "tile height must match the kernel-map tiling");
  constexpr int NBLK = NB * WN;
  constexpr int CK = conv_phase_width(CP, MB * NB);  // channels per phase
  constexpr int PPT = CP / CK;                // phases per tile
  static_assert(CP % CK == 0, "phase width must divide Cin");
  constexpr int C4K = CK / 4;                 // 16-byte pieces per row per phase
  constexpr int NCH = TM * C4K / THREADS;     // pieces per thread per phase
  static_assert(TM * C4K % THREADS == 0, "gather pieces must divide evenly");
  constexpr int LDA = CK + 4;
  constexpr int SK = CK / 8;                  // K-steps per phase
  constexpr int S = CP / 8;                   // K-steps per tile
  constexpr int RING = 4;
  static_assert(SK == 1 || SK % RING == 0, "phase length must be a multiple of the B ring");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *As = lds;                                                 // [2][TM][LDA]
  int *idxbuf = reinterpret_cast<int *>(lds + 2 * TM * LDA);       // [4][TM]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const bool identity = (a.pair_in == nullptr);
  const int n_rows = identity ? *a.n_rows_dev : 0;
  const int T = identity ? (n_rows + TM - 1) / TM : a.tile_ptr[a.K];
  const int per = (T + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  const int t_end = min(T, (xcd + 1) * per);
  const int nj = gridDim.x >> 3;
  const int t_first = xcd * per + (blockIdx.x >> 3);
  if (t_first >= t_end) return;
  const int n_my = (t_end - t_first + nj - 1) / nj;  // tiles of this block: t_first + i * nj
  const int NQ = n_my * PPT;                         // phases of this block

  auto locate = [&](int i, int &k, int &pstart, int &count) {
    const int tt = t_first + i * nj;
    if (identity) {
      k = 0;
      pstart = tt * TM;
      count = min(TM, n_rows - pstart);
    } else {
      const int4 d = a.tile_desc[tt];  // one scalar 16-byte load (kmap.hip: tile_desc_kernel)
      k = d.x;
      pstart = d.y;
      count = d.z;
    }
  };
  auto load_idx = [&](int i) -> int {  // input row of tile-row `tid` of the block's i-th tile, or -1
    if (i >= n_my || tid >= TM) return -1;
    int k, pstart, count;
    locate(i, k, pstart, count);
    if (tid >= count) return -1;
    return identity ? pstart + tid : a.pair_in[pstart + tid];
  };
  f32x4 G[NCH];
  uint32_t g_ok = 0;  // bit i: piece i of the requested phase is a real (row, channel) piece
  // Requests only: NOTHING here may consume a loaded value (a consumer right behind the load costs an
  // s_waitcnt vmcnt(0) per piece -- the gather then runs serialised, and behind every B operand in
  // flight).  Invalid pieces are loaded from a clamped address and zeroed in land(); the ReLU-on-read
  // is applied in land() too, one phase later, when the data has long arrived.
  auto gather = [&](int q) {  // request the rows of phase q (tile q / PPT, channels (q % PPT) * CK ..)
    const int *idx = idxbuf + ((q / PPT) & 3) * TM;
    const int cbase = (q % PPT) * CK;
    g_ok = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = tid + i * THREADS;
      const int r = ch / C4K, c = cbase + (ch % C4K) * 4;
      int row = idx[r];
      if (VEC) {
        const int rr = max(row, 0);
        const int cc = min(c, a.cin - 4);
        G[i] = *reinterpret_cast<const f32x4 *>(a.in + (int64_t)rr * a.in_ld + cc);
        g_ok |= (row >= 0 && c < a.cin) ? (1u << i) : 0u;
      } else {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row >= 0) {
          const float *src = a.in + (int64_t)row * a.in_ld + c;
          if (c + 0 < a.cin) v.x = src[0];
          if (c + 1 < a.cin) v.y = src[1];
          if (c + 2 < a.cin) v.z = src[2];
          if (c + 3 < a.cin) v.w = src[3];
        }
        G[i] = v;
        g_ok |= 1u << i;
      }
    }
  };
  auto land = [&](int q) {  // registers -> A buffer q & 1
    float *dst = As + (q & 1) * TM * LDA;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = tid + i * THREADS;
      f32x4 v = G[i];
      if (a.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      const bool ok = (g_ok >> i) & 1u;
      v.x = ok ? v.x : 0.f;
      v.y = ok ? v.y : 0.f;
      v.z = ok ? v.z : 0.f;
      v.w = ok ? v.w : 0.f;
      *reinterpret_cast<f32x4 *>(dst + (ch / C4K) * LDA + (ch % C4K) * 4) = v;
    }
  };

  // ---- prologue: indices of tiles 0..2 into the LDS ring, tile 3's into a register; phase 0 landed,
  //      phase 1 in flight; first B operands of tile 0 requested
  {
    const int i0 = load_idx(0), i1 = load_idx(1), i2 = load_idx(2);
    if (tid < TM) { idxbuf[tid] = i0; idxbuf[TM + tid] = i1; idxbuf[2 * TM + tid] = i2; }
  }
  int next_pub = 3;
  int idx_reg = load_idx(3);
  int k, pstart, count;
  locate(0, k, pstart, count);
  f32x4 b[RING][NB];
  {
    const f32x4 *w0 = reinterpret_cast<const f32x4 *>(a.w) + ((int64_t)k * S * NBLK + wn * NB) * 64 + lane;
#pragma unroll
    for (int r = 0; r < RING - 1; ++r)
      if (r < S) {
#pragma unroll
        for (int j = 0; j < NB; ++j) b[r][j] = w0[((int64_t)r * NBLK + j) * 64];
      }
  }
  __syncthreads();
  gather(0);
  land(0);
  if (1 < NQ) gather(1);
  __syncthreads();

  f32x16 acc[MB][NB];
  for (int q = 0; q < NQ; ++q) {
    const int h = q % PPT;
    if (q + 1 < NQ) land(q + 1);      // requested one phase ago
    if (q + 2 < NQ) gather(q + 2);    // a whole phase to arrive
    if ((q + 3) / PPT >= next_pub && next_pub < n_my) {  // indices for the gather of the phase after next
      if (tid < TM) idxbuf[(next_pub & 3) * TM + tid] = idx_reg;
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
    // ---- MFMA over this phase's channels
    const f32x4 *wk = reinterpret_cast<const f32x4 *>(a.w) + ((int64_t)k * S * NBLK + wn * NB) * 64 + lane;
    const float *arow = As + (q & 1) * TM * LDA + (32 * wm * MB + (lane & 31)) * LDA + 4 * (lane >> 5);
    const int s0 = h * SK;
#pragma unroll
    for (int s = 0; s < SK; ++s) {
      if (s0 + s + RING - 1 < S) {
#pragma unroll
        for (int j = 0; j < NB; ++j) b[(s + RING - 1) % RING][j] = wk[((int64_t)(s0 + s + RING - 1) * NBLK + j) * 64];
      }
      // pin the prefetch HERE: without it the scheduler sinks each load next to its first use
      // (load-to-use distance 0, full L2 latency exposed on every K-step; seen in the ISA)
      __builtin_amdgcn_sched_barrier(0);
      f32x4 av[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i) av[i] = *reinterpret_cast<const f32x4 *>(arow + i * 32 * LDA + s * 8);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[s % RING][j][c], av[i][c], acc[i][j], 0, 0, 0);
    }
    if (h == PPT - 1) {
      // ---- tile finished: request the next tile's first B operands BEFORE this tile's stores
      const int pst = pstart, cnt = count;
      const int i_next = q / PPT + 1;
      if (i_next < n_my) {
        locate(i_next, k, pstart, count);
        const f32x4 *w1 = reinterpret_cast<const f32x4 *>(a.w) + ((int64_t)k * S * NBLK + wn * NB) * 64 + lane;
#pragma unroll
        for (int r = 0; r < RING - 1; ++r)
          if (r < S) {
#pragma unroll
            for (int j = 0; j < NB; ++j) b[r][j] = w1[((int64_t)r * NBLK + j) * 64];
          }
      }
      // product rows: D column (pair) = lane & 31, D row (channel) = (e&3) + 8 (e>>2) + 4 (lane>>5)
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int r = 32 * (wm * MB + i) + (lane & 31);
        if (identity && a.l2_normalize) {
          // the feature row of the `final` conv leaves unit-norm (model/resunet.py:643-647): a row's <= 32 channels sit
          // in lanes l and l ^ 32 of the one wave column -- shift first, one cross-lane add, one division per value
          float ss = 0.f;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int col = 8 * g + 4 * (lane >> 5) + u;
              float v = acc[i][0][4 * g + u] + ((a.shift && col < a.cout) ? a.shift[col] : 0.f);
              if (col >= a.cout) v = 0.f;
              acc[i][0][4 * g + u] = v;
              ss += v * v;
            }
          ss += __shfl_xor(ss, 32, 64);
          const float den = sqrtf(ss) + 1e-8f;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][0][e] = acc[i][0][e] / den;
        }
        if (r < cnt) {
          float *dst = identity ? a.out + (int64_t)(pst + r) * a.out_ld : a.y + (int64_t)(pst + r) * a.y_ld;
#pragma unroll
          for (int j = 0; j < NB; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int col = 32 * (wn * NB + j) + 8 * g + 4 * (lane >> 5);
              f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
              if (identity) {
                if (col + 3 < a.cout && (a.out_ld & 3) == 0) {
                  if (a.shift && !a.l2_normalize) v += *reinterpret_cast<const f32x4 *>(a.shift + col);
                  *reinterpret_cast<f32x4 *>(dst + col) = v;
                } else {
#pragma unroll
                  for (int u = 0; u < 4; ++u)
                    if (col + u < a.cout) dst[col + u] = v[u] + ((a.shift && !a.l2_normalize) ? a.shift[col + u] : 0.f);
                }
              } else if (col < a.cout) {
                *reinterpret_cast<f32x4 *>(dst + col) = v;  // y_ld = cout is a multiple of 32 here
              }
            }
          }
        }
      }
    }
    __syncthreads();  // buffer q&1 is free again; buffer (q+1)&1 and the index ring are visible
  }
}

template <int CP, int WM, int WN, int MB, int NB, bool VEC>
static int launch_v2(const ConvKArgs &ka, int64_t tile_bound, int num_cus, hipStream_t stream) {
  constexpr int THREADS = 64 * WM * WN;
  constexpr int CKL = conv_phase_width(CP, MB * NB);
  const size_t lds_bytes = (size_t)2 * DGR_TILE_M * (CKL + 4) * sizeof(float) + 4 * DGR_TILE_M * sizeof(int);
  static int per_cu = 0;
  if (per_cu == 0) {
    DGR_HIP_CHECK(hipFuncSetAttribute((const void *)sparse_conv_mfma_v2<CP, WM, WN, MB, NB, VEC>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    int n = 0;
    DGR_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sparse_conv_mfma_v2<CP, WM, WN, MB, NB, VEC>,
                                                               THREADS, lds_bytes));
    per_cu = n < 1 ? 1 : (n > 2048 / THREADS ? 2048 / THREADS : n);
  }
  int64_t grid = (int64_t)num_cus * per_cu;
  if (tile_bound < grid) grid = tile_bound;
  grid = (grid + 7) / 8 * 8;
  if (grid < 8) grid = 8;
  sparse_conv_mfma_v2<CP, WM, WN, MB, NB, VEC><<<(int)grid, THREADS, lds_bytes, stream>>>(ka);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

// ------------------------------------------------------------------------------------------
// Kernel-volume-1 convs (conv1_tr, final: MinkowskiConvolution with kernel_size 1 = a dense [N, Cin] x [Cin, Cout]
// product over the rows of ONE map, model/resunet.py:565-596): a streaming kernel.  No kernel map, no index ring, no
// LDS: a wave takes 32 consecutive rows, every lane requests its 16-byte pieces of them straight into the B-operand
// registers of v_mfma_f32_32x32x2_f32 (lane = row (lane & 31), input channels 8 j + 4 (lane >> 5) .. + 3 -- the layout
// the rule-major kernel builds in LDS), all CP / 8 requests of a tile in flight at once; the layer's weights stay in
// registers for all the tiles of the wave.  Same operands in the same order as sparse_conv_mfma_v2 on an identity map:
// bit-identical results, HBM-bound instead of latency-bound (195 k rows, 96 -> 64: 51 -> ~25 us).
// ------------------------------------------------------------------------------------------
template <int CP, int NBLK>
__global__ void __launch_bounds__(256) identity_conv_kernel(ConvKArgs a) {
  constexpr int S = CP / 8;
  const int lane = threadIdx.x & 63;
  const int n_rows = *a.n_rows_dev;
  const int n_tiles = (n_rows + 31) >> 5;
  const int wave_id = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), n_waves = (int)((gridDim.x * blockDim.x) >> 6);
  f32x4 w[S][NBLK];
  {
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(a.w) + lane;
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int j = 0; j < NBLK; ++j) w[s][j] = wp[(s * NBLK + j) * 64];
  }
  const int relu_lo = a.in_relu ? 0 : (int)0x80000000;   // pending ReLU of the producer as one integer max per value
  auto request = [&](int t, f32x4 (&x)[S]) {
    const int64_t row = min((int64_t)t * 32 + (lane & 31), (int64_t)n_rows - 1);   // rows past the end: the last row again
    const float *src = a.in + row * a.in_ld + 4 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < S; ++s) x[s] = *reinterpret_cast<const f32x4 *>(src + 8 * s);
  };
  f32x4 xa[S], xb[S];
  auto tile = [&](int t, f32x4 (&x)[S]) {
    f32x16 acc[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      i32x4 v = __builtin_bit_cast(i32x4, x[s]);
      v.x = max(v.x, relu_lo); v.y = max(v.y, relu_lo); v.z = max(v.z, relu_lo); v.w = max(v.w, relu_lo);
      const f32x4 xv = __builtin_bit_cast(f32x4, v);
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < NBLK; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s][j][c], xv[c], acc[j], 0, 0, 0);
    }
    // D column (row of the tile) = lane & 31, D row (channel) = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    const int64_t row = (int64_t)t * 32 + (lane & 31);
    if (a.l2_normalize) {
      // the feature row of the `final` conv leaves unit-norm (model/resunet.py:643-647): a row's <= 32 channels sit in
      // lanes l and l ^ 32 -- shift first, one cross-lane add, one division per value
      float ss = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int col = 8 * g + 4 * (lane >> 5) + u;
          float v = acc[0][4 * g + u] + ((a.shift && col < a.cout) ? a.shift[col] : 0.f);
          if (col >= a.cout) v = 0.f;
          acc[0][4 * g + u] = v;
          ss += v * v;
        }
      ss += __shfl_xor(ss, 32, 64);
      const float den = sqrtf(ss) + 1e-8f;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[0][e] = acc[0][e] / den;
    }
    if (row < n_rows) {
      float *dst = a.out + row * a.out_ld;
#pragma unroll
      for (int j = 0; j < NBLK; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = 32 * j + 8 * g + 4 * (lane >> 5);
          f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
          if (col + 3 < a.cout && (a.out_ld & 3) == 0) {
            if (a.shift && !a.l2_normalize) v += *reinterpret_cast<const f32x4 *>(a.shift + col);
            *reinterpret_cast<f32x4 *>(dst + col) = v;
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (col + u < a.cout) dst[col + u] = v[u] + ((a.shift && !a.l2_normalize) ? a.shift[col + u] : 0.f);
          }
        }
    }
  };
  // two tiles in flight: the next tile's rows are requested before this tile is multiplied
  int t = wave_id;
  if (t < n_tiles) request(t, xa);
  for (; t < n_tiles; t += 2 * n_waves) {
    const int t1 = t + n_waves;
    if (t1 < n_tiles) request(t1, xb);
    tile(t, xa);
    if (t1 < n_tiles) {
      if (t1 + n_waves < n_tiles) request(t1 + n_waves, xa);
      tile(t1, xb);
    }
  }
}

template <int CP, int NBLK>
static int launch_identity(const ConvKArgs &ka, int64_t rows_cap, int num_cus, hipStream_t stream) {
  // enough waves to keep every CU's memory pipe full, at least two tiles each where the tensor allows it
  int64_t blocks = std::min<int64_t>((int64_t)num_cus * 4, std::max<int64_t>(1, dgr_ceil_div(rows_cap, 32 * 4 * 2)));
  identity_conv_kernel<CP, NBLK><<<(int)blocks, 256, 0, stream>>>(ka);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

int dgr_conv_launch(const DgrConvLaunch &a, int num_cus, hipStream_t stream, const char **kernel_name) {
  ConvKArgs ka;
  ka.in = a.in; ka.out = a.out; ka.w = a.w; ka.y = a.y; ka.shift = a.shift; ka.y_ld = a.cout;
  ka.pair_in = a.pair_in; ka.pair_out = a.pair_out; ka.tile_ptr = a.tile_ptr; ka.rule_ptr = a.rule_ptr;
  ka.n_rows_dev = a.n_rows_dev; ka.tile_desc = a.tile_desc;
  ka.in_ld = a.in_ld; ka.out_ld = a.out_ld; ka.in_relu = a.in_relu;
  ka.cin = a.cin; ka.cin_pad = a.cin_pad; ka.cout = a.cout; ka.K = a.K;
  ka.l2_normalize = a.l2_normalize;
  DGR_REQUIRE(!a.l2_normalize || (a.pair_in == nullptr && a.cout <= 32 && a.cout_pad == 32),
              "fused l2 normalisation needs an identity map and Cout <= 32 (got %d)", a.cout);
  DGR_REQUIRE(a.cin_pad % 8 == 0 && a.cin_pad >= a.cin && a.cin_pad <= 256, "bad cin_pad %d", a.cin_pad);
  const int64_t tile_bound = a.tile_bound > 0 ? a.tile_bound : (int64_t)num_cus * 4;
  // specialised (compile-time Cin, pipelined) instantiations for every layer shape of ResUNetBN2C
  const bool vec = ((a.cin | a.in_ld) & 3) == 0;
  if (a.pair_in == nullptr && vec && a.cin == a.cin_pad && a.tile_bound > 0) {   // kernel-volume-1 conv: streaming kernel
#define DGR_ID(CPV, NBV)                                                                          \
  if (a.cin_pad == CPV && a.cout_pad == 32 * NBV) {                                               \
    if (kernel_name) *kernel_name = "identity_conv_kernel<" #CPV ", " #NBV ">";                   \
    return launch_identity<CPV, NBV>(ka, a.tile_bound * DGR_TILE_M, num_cus, stream);             \
  }
    DGR_ID(64, 1) DGR_ID(64, 2) DGR_ID(96, 2)
#undef DGR_ID
  }
#define DGR_V2(CPV, WMV, WNV, MBV, NBV)                                                                        \
  if (a.cin_pad == CPV) {                                                                                       \
    if (kernel_name)                                                                                            \
      *kernel_name = vec ? "sparse_conv_mfma_v2<" #CPV ", " #WMV ", " #WNV ", " #MBV ", " #NBV ", true>"        \
                         : "sparse_conv_mfma_v2<" #CPV ", " #WMV ", " #WNV ", " #MBV ", " #NBV ", false>";      \
    return vec ? launch_v2<CPV, WMV, WNV, MBV, NBV, true>(ka, tile_bound, num_cus, stream)                     \
               : launch_v2<CPV, WMV, WNV, MBV, NBV, false>(ka, tile_bound, num_cus, stream);                   \
  }
  switch (a.cout_pad) {
    case 32: DGR_V2(8, 2, 1, 1, 1) DGR_V2(32, 2, 1, 1, 1) DGR_V2(64, 2, 1, 1, 1) break;
    case 64: DGR_V2(32, 2, 2, 1, 1) DGR_V2(64, 2, 2, 1, 1) DGR_V2(96, 2, 2, 1, 1) DGR_V2(128, 2, 2, 1, 1)
             DGR_V2(256, 2, 2, 1, 1) break;
    case 128: DGR_V2(64, 1, 4, 2, 1) DGR_V2(128, 1, 4, 2, 1) DGR_V2(256, 1, 4, 2, 1) break;
    case 256: DGR_V2(128, 1, 4, 2, 2) DGR_V2(256, 1, 4, 2, 2) break;
    default: break;
  }
#undef DGR_V2
  dgr_set_error("sparse conv: no kernel instantiation for Cin (padded) %d -> Cout (padded) %d", a.cin_pad, a.cout_pad);
  return DGR_EINVAL;
}

// ------------------------------------------------------------------------------------------
// phase 2: out[o, :] = shift (+ residual[o, :]) + sum over the row's pairs (ascending k) of Y[pos, :]
// LPR lanes cooperate on one output row (one float4 column group each); rows are independent.
// ------------------------------------------------------------------------------------------
template <int LPR, bool SPLIT>
__global__ void __launch_bounds__(256)
    reduce_rows_kernel(const float *__restrict__ y, int y_ld, const int32_t *__restrict__ ptr,
                       const int32_t *__restrict__ pos, const int32_t *n_dev, float *__restrict__ out, int out_ld,
                       const float *__restrict__ shift, const float *__restrict__ res, int res_ld, int res_relu,
                       unsigned char *__restrict__ planes, float *__restrict__ scale_out, int out_relu) {
  constexpr int ROWS = 256 / LPR;
  const int n = *n_dev;
  const int sub = threadIdx.x / LPR, c = (threadIdx.x % LPR) * 4;
  for (int64_t row = (int64_t)blockIdx.x * ROWS + sub; row < n; row += (int64_t)gridDim.x * ROWS) {
    f32x4 acc = shift ? *reinterpret_cast<const f32x4 *>(shift + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (res) {
      f32x4 r = *reinterpret_cast<const f32x4 *>(res + row * res_ld + c);
      if (res_relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
      acc += r;
    }
    int j = ptr[row];
    const int end = ptr[row + 1];
    // eight independent product-row loads in flight (their eight slot indices are fetched together first);
    // the additions stay in ascending-k order
    for (; j + 8 <= end; j += 8) {
      int p[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] = pos[j + u];
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = dgr_y_load<LPR == 64>(y + (int64_t)p[u] * y_ld + c);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; j + 4 <= end; j += 4) {
      const int p0 = pos[j], p1 = pos[j + 1], p2 = pos[j + 2], p3 = pos[j + 3];
      const f32x4 v0 = dgr_y_load<LPR == 64>(y + (int64_t)p0 * y_ld + c);
      const f32x4 v1 = dgr_y_load<LPR == 64>(y + (int64_t)p1 * y_ld + c);
      const f32x4 v2 = dgr_y_load<LPR == 64>(y + (int64_t)p2 * y_ld + c);
      const f32x4 v3 = dgr_y_load<LPR == 64>(y + (int64_t)p3 * y_ld + c);
      acc += v0; acc += v1; acc += v2; acc += v3;
    }
    for (; j < end; ++j) acc += dgr_y_load<LPR == 64>(y + (int64_t)pos[j] * y_ld + c);
    if (!SPLIT || out) *reinterpret_cast<f32x4 *>(out + row * out_ld + c) = acc;   // (split-only tensors: out == nullptr)
    if constexpr (SPLIT) {
      // the row as the wide-layer kernel gathers it (conv_wide.hip): scale from the row's largest |x| after the
      // consumers' pending ReLU (the LPR lanes of a row are consecutive lanes of one wave), two f16 planes
      const int lo = out_relu ? 0 : (int)0x80000000;
      const i32x4 ab = __builtin_bit_cast(i32x4, acc);   // (bit_cast of a single vector ELEMENT reads element 0)
      int xb[4];
      uint32_t mx = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xb[u] = max(ab[u], lo);   // pending ReLU as one integer max
        mx = max(mx, (uint32_t)xb[u] & 0x7fffffffu);
      }
#pragma unroll
      for (int d = LPR / 2; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
      const float sx = dgr_row_scale_of(mx);
      f16x4 h, m;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        _Float16 hh, mm;
        dgr_split2(__builtin_bit_cast(float, xb[u]), sx, hh, mm);
        h[u] = hh; m[u] = mm;
      }
      // lanes 2 j and 2 j + 1 hold channels 8 j .. 8 j + 3 and 8 j + 4 .. 8 j + 7: they swap one piece, so that the even
      // lane stores 16 bytes of h and the odd lane 16 bytes of m (one 16-byte store per lane instead of two 8-byte ones)
      const bool odd = threadIdx.x & 1;
      const u32x2 hb = __builtin_bit_cast(u32x2, h), mb = __builtin_bit_cast(u32x2, m);
      const u32x2 give = odd ? hb : mb;
      u32x2 got;
      got.x = (uint32_t)__shfl_xor((int)give.x, 1, 64);
      got.y = (uint32_t)__shfl_xor((int)give.y, 1, 64);
      const u32x4 st = odd ? u32x4{got.x, got.y, mb.x, mb.y} : u32x4{hb.x, hb.y, got.x, got.y};
      unsigned char *dst = planes + row * (LPR * 16);
      *reinterpret_cast<u32x4 *>(dst + dgr_split_row_offset(c & ~7, odd ? 1 : 0)) = st;
      if (c == 0) scale_out[row] = sx;
    }
  }
}

int dgr_reduce_rows(const float *y, int cout, const int32_t *ptr, const int32_t *pos, const int32_t *n_dev,
                    int64_t n_cap, float *out, int out_ld, const float *shift, const float *res, int res_ld,
                    int res_relu, hipStream_t stream, const DgrSplitRows *split, int out_relu) {
  DGR_REQUIRE((out_ld & 3) == 0 && (res == nullptr || (res_ld & 3) == 0), "reduce_rows: row strides must be x4");
  DGR_REQUIRE(out || split, "reduce_rows: no output");
  DGR_REQUIRE(!split || (split->planes && split->scale && split->channels == cout && cout % 64 == 0),
              "reduce_rows: split rows need planes, scales and the layer's width (a multiple of 64)");
  const int lpr = cout / 4;
  const int64_t want = std::max<int64_t>(1, dgr_ceil_div(n_cap, 256 / lpr));
  // grid-stride kernels: at most four resident rounds, and a whole number of them
#define DGR_RR1(L, SP)                                                                                         \
  do {                                                                                                         \
    static int resident = 0;                                                                                   \
    if (resident == 0) {                                                                                       \
      int per_cu = 0, dev = 0, cus = 0;                                                                        \
      DGR_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reduce_rows_kernel<L, SP>, 256, 0)); \
      DGR_HIP_CHECK(hipGetDevice(&dev));                                                                       \
      DGR_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));                  \
      resident = (per_cu < 1 ? 1 : per_cu) * (cus < 1 ? 1 : cus);                                              \
    }                                                                                                          \
    const int64_t blocks = want > resident ? (int64_t)resident * std::min<int64_t>(4, want / resident) : want; \
    reduce_rows_kernel<L, SP><<<(int)blocks, 256, 0, stream>>>(y, cout, ptr, pos, n_dev, out, out_ld, shift,   \
                                                               res, res_ld, res_relu, split ? split->planes : nullptr, \
                                                               split ? split->scale : nullptr, out_relu);      \
  } while (0)
#define DGR_RR(L)                       \
  do {                                  \
    if (split) DGR_RR1(L, true);        \
    else DGR_RR1(L, false);             \
  } while (0)
  switch (cout) {
    case 32: DGR_RR1(8, false); break;
    case 64: DGR_RR(16); break;
    case 128: DGR_RR(32); break;
    case 256: DGR_RR(64); break;
    default:
      dgr_set_error("reduce_rows: unsupported width %d", cout);
      return DGR_EINVAL;
  }
#undef DGR_RR1
#undef DGR_RR
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

// ------------------------------------------------------------------------------------------
// conv1 (Cin <= 8 -> 32): output stationary.  FCGF's first layer has 343 offsets, ~74 neighbours per
// voxel and ONE input channel: as a rule-major GEMM it would write and re-read 260 MB of product rows
// for 0.13 GFLOP.  Here 32 lanes own one output voxel (one channel each), walk its pairs in
// ascending-k order (the same order as the reduction pass, bit-compatible) and write the row once.
// Weights are read in place from the MFMA-tiled layout: W[k][ci][co] sits at
// ((k * 64 + 32 * (ci / 4) + co) * 4 + ci % 4).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    conv_small_cin_kernel(const float *__restrict__ in, int in_ld, int in_relu, int cin,
                          const float *__restrict__ w, const float *__restrict__ shift,
                          const int32_t *__restrict__ out_ptr, const int32_t *__restrict__ out_pos,
                          const int32_t *__restrict__ pair_in, const uint16_t *__restrict__ pair_k,
                          const int32_t *n_out_dev, float *__restrict__ out, int out_ld) {
  const int n = *n_out_dev;
  const int co = threadIdx.x & 31;
  for (int64_t o = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5); o < n; o += (int64_t)gridDim.x * 8) {
    float acc = shift ? shift[co] : 0.f;
    const int end = out_ptr[o + 1];
    int j = out_ptr[o];
    // eight pairs per round: the three dependent index loads (slot -> pair -> input row) of all eight
    // are in flight together; the additions stay in ascending-k order
    for (; j + 8 <= end; j += 8) {
      int pos[8], row[8], kk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) pos[u] = out_pos[j + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) { row[u] = pair_in[pos[u]]; kk[u] = pair_k[pos[u]]; }
      float y[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float *wk = w + (int64_t)kk[u] * 256;
        float t = 0.f;
        for (int ci = 0; ci < cin; ++ci) {
          float x = in[(int64_t)row[u] * in_ld + ci];
          if (in_relu) x = fmaxf(x, 0.f);
          t = fmaf(x, wk[(32 * (ci >> 2) + co) * 4 + (ci & 3)], t);
        }
        y[u] = t;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += y[u];
    }
    for (; j < end; ++j) {
      const int pos = out_pos[j];
      const int row = pair_in[pos];
      const float *wk = w + (int64_t)pair_k[pos] * 256;
      float y = 0.f;
      for (int ci = 0; ci < cin; ++ci) {
        float x = in[(int64_t)row * in_ld + ci];
        if (in_relu) x = fmaxf(x, 0.f);
        y = fmaf(x, wk[(32 * (ci >> 2) + co) * 4 + (ci & 3)], y);   // same k-ordered fma chain as the MFMA
      }
      acc += y;
    }
    out[o * out_ld + co] = acc;
  }
}

// The same layer with ONE THREAD per output voxel (all 32 output channels in its registers).  The 6-D inlier net's conv1
// has 1.8 pairs per row (the centre offset and, for half the rows, one or two neighbours): with 32 lanes per voxel the
// kernel is 55 k waves that each walk a chain of four dependent loads per pair (row pointer -> slot -> pair -> input
// row / weights), 81 us for 110 k rows; with a thread per voxel it is 1.7 k waves, all resident at once, and the chains
// of 64 voxels run side by side.  Same fma chain per pair, pairs added in ascending-k order: bit-identical results.
template <int CIN>
__global__ void __launch_bounds__(256)
    conv_small_cin_row_kernel(const float *__restrict__ in, int in_ld, int in_relu, const float *__restrict__ w,
                              const float *__restrict__ shift, const int32_t *__restrict__ out_ptr,
                              const int32_t *__restrict__ out_pos, const int32_t *__restrict__ pair_in,
                              const uint16_t *__restrict__ pair_k, const int32_t *n_out_dev, float *__restrict__ out,
                              int out_ld) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= *n_out_dev) return;
  float acc[32];
#pragma unroll
  for (int co = 0; co < 32; ++co) acc[co] = shift ? shift[co] : 0.f;
  const int end = out_ptr[o + 1];
  for (int j = out_ptr[o]; j < end; ++j) {
    const int pos = out_pos[j];
    const int row = pair_in[pos];
    const f32x4 *wk = reinterpret_cast<const f32x4 *>(w + (int64_t)pair_k[pos] * 256);
    float x[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      x[ci] = in[(int64_t)row * in_ld + ci];
      if (in_relu) x[ci] = fmaxf(x[ci], 0.f);
    }
#pragma unroll
    for (int co = 0; co < 32; ++co) {
      // W[k][ci][co] sits at ((k * 64 + 32 * (ci / 4) + co) * 4 + ci % 4): two 16-byte loads per output channel
      const f32x4 w0 = wk[co];
      float t = 0.f;
#pragma unroll
      for (int ci = 0; ci < (CIN < 4 ? CIN : 4); ++ci) t = fmaf(x[ci], w0[ci], t);
      if constexpr (CIN > 4) {
        const f32x4 w1 = wk[32 + co];
#pragma unroll
        for (int ci = 4; ci < CIN; ++ci) t = fmaf(x[ci], w1[ci - 4], t);
      }
      acc[co] += t;
    }
  }
  f32x4 *dst = reinterpret_cast<f32x4 *>(out + o * out_ld);
#pragma unroll
  for (int q = 0; q < 8; ++q) dst[q] = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
}

// Cin = 6 (the 'coords' features of the inlier net): a QUAD of lanes per output voxel, lane q of the quad computing output
// channels 8 q .. 8 q + 7.  What the thread-per-voxel kernel above spends its time on is fetching W[k] -- 64 16-byte
// loads per pair and thread, every lane of an instruction at another offset's kilobyte: 64 cache lines per
// instruction through the CU's address path.  Here W[k] is stored quad-major (wq[k][i][q][4], i = 2 ci + half: the four
// lanes of a quad read 64 CONTIGUOUS bytes per instruction, one request), 12 loads per pair and lane.  The fma chain of
// a (pair, output channel) and the order of the pairs are those of the kernel above: bit-identical results.
__global__ void __launch_bounds__(256)
    conv_cin6_quad_kernel(const float *__restrict__ in, int in_ld, int in_relu, const float *__restrict__ wq,
                          const float *__restrict__ shift, const int32_t *__restrict__ out_ptr,
                          const int32_t *__restrict__ out_pos, const int32_t *__restrict__ pair_in,
                          const uint16_t *__restrict__ pair_k, const int32_t *n_out_dev, float *__restrict__ out,
                          int out_ld) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t o = t >> 2;
  const int q = (int)(t & 3);
  if (o >= *n_out_dev) return;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = shift ? shift[8 * q + e] : 0.f;
  const int end = out_ptr[o + 1];
  for (int j = out_ptr[o]; j < end; ++j) {
    const int pos = out_pos[j];
    const int row = pair_in[pos];
    const f32x4 *wk = reinterpret_cast<const f32x4 *>(wq + (int64_t)pair_k[pos] * 192) + q;
    float x[6];
#pragma unroll
    for (int ci = 0; ci < 6; ++ci) {
      x[ci] = in[(int64_t)row * in_ld + ci];
      if (in_relu) x[ci] = fmaxf(x[ci], 0.f);
    }
    float tt[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) tt[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {   // ci = i / 2 ascending: per output channel the chain x0 w0, x1 w1, ... of the kernel above
      const f32x4 wv = wk[4 * i];
#pragma unroll
      for (int e = 0; e < 4; ++e) tt[4 * (i & 1) + e] = fmaf(x[i >> 1], wv[e], tt[4 * (i & 1) + e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += tt[e];
  }
  f32x4 *dst = reinterpret_cast<f32x4 *>(out + o * out_ld + 8 * q);
  dst[0] = f32x4{acc[0], acc[1], acc[2], acc[3]};
  dst[1] = f32x4{acc[4], acc[5], acc[6], acc[7]};
}

int dgr_conv_small_cin(const float *in, int in_ld, int in_relu, int cin, const float *w_tiled, const float *w_quad,
                       const float *shift, const DgrKernelMap &km, const int32_t *n_out_dev, int64_t n_out_cap, float *out,
                       int out_ld, hipStream_t stream) {
  DGR_REQUIRE(cin >= 1 && cin <= 8, "small-Cin conv: cin=%d", cin);
  if ((out_ld & 3) == 0 && cin == 6 && w_quad) {
    conv_cin6_quad_kernel<<<(int)dgr_ceil_div(n_out_cap * 4, 256), 256, 0, stream>>>(
        in, in_ld, in_relu, w_quad, shift, km.out_ptr, km.out_pos, km.pair_in, km.pair_k, n_out_dev, out, out_ld);
    DGR_LAUNCH_CHECK();
    return DGR_OK;
  }
  if ((out_ld & 3) == 0 && (cin == 6 || cin == 1)) {   // the inlier net's input widths ('coords' / 'ones' features)
    const int rb = (int)dgr_ceil_div(n_out_cap, 256);
    if (cin == 6)
      conv_small_cin_row_kernel<6><<<rb, 256, 0, stream>>>(in, in_ld, in_relu, w_tiled, shift, km.out_ptr, km.out_pos, km.pair_in,
                                                         km.pair_k, n_out_dev, out, out_ld);
    else
      conv_small_cin_row_kernel<1><<<rb, 256, 0, stream>>>(in, in_ld, in_relu, w_tiled, shift, km.out_ptr, km.out_pos, km.pair_in,
                                                         km.pair_k, n_out_dev, out, out_ld);
    DGR_LAUNCH_CHECK();
    return DGR_OK;
  }
  int64_t blocks = dgr_ceil_div(n_out_cap, 8);
  if (blocks > 16384) blocks = 16384;
  conv_small_cin_kernel<<<(int)blocks, 256, 0, stream>>>(in, in_ld, in_relu, cin, w_tiled, shift, km.out_ptr,
                                                          km.out_pos, km.pair_in, km.pair_k, n_out_dev, out, out_ld);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

// ------------------------------------------------------------------------------------------
// conv1 of the FCGF net fused with its neighbour search (3-D, ks^3 = 343 offsets for 3DMatch, Cin <= 8
// -> 32): the ks^3 kernel map is used by this ONE layer, so building it (343 N probes, a dense hit
// table, ranking, CSR) only to gather ~74 single-channel neighbours per voxel is pure overhead.  Here
// a wave owns one output voxel at a time: its 64 lanes probe 64 offsets at once; the hits are then
// replayed in ascending-k order (ballot + readlane) with lane = output channel accumulating
// x[hit] * W[k][ci][co] -- the same k-ordered sum as the map-based path, bit for bit.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
    conv1_probe_kernel(const int32_t *__restrict__ coords, const int32_t *n_dev, const int32_t *__restrict__ table,
                       uint32_t mask, int ks, const float *__restrict__ in, int in_ld, int cin,
                       const float *__restrict__ w, const float *__restrict__ shift, float *__restrict__ out,
                       int out_ld, int32_t *pair_count, const int32_t *skip_flag, uint32_t *__restrict__ out_amax) {
  if (skip_flag && *skip_flag) return;   // the dense-grid kernel did the layer
  const int n = *n_dev;
  const int lane = threadIdx.x & 63;
  const int co = lane & 31;
  const int K = ks * ks * ks, half = ks >> 1;
  const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int pairs = 0;
  for (int64_t o = wave_id; o < n; o += n_waves) {
    const int32_t b = coords[o * 4], x = coords[o * 4 + 1], y = coords[o * 4 + 2], z = coords[o * 4 + 3];
    float acc = shift ? shift[co] : 0.f;
    for (int k0 = 0; k0 < K; k0 += 64) {
      const int k = k0 + lane;
      int hit = -1;
      if (k < K) {
        // offset enumeration: first spatial dimension fastest (same convention as kmap.hip: offset_of)
        int32_t q[4] = {b, x + (k % ks) - half, y + ((k / ks) % ks) - half, z + (k / (ks * ks)) - half};
        hit = dgr_lookup<4>(table, mask, coords, q);
      }
      // every lane fetches the input channels of ITS hit now (lane parallel); the serial replay below
      // then only moves registers (readlane) and reads weights
      // (unconditional clamped loads: a per-lane guard would make the compiler branch and wait per load)
      float xv[8];
      const int64_t hrow = (int64_t)max(hit, 0) * in_ld;
#pragma unroll
      for (int ci = 0; ci < 8; ++ci) {
        xv[ci] = 0.f;
        if (ci < cin) xv[ci] = in[hrow + ci];   // cin is wave-uniform
      }
      unsigned long long live = __ballot(hit >= 0);
      pairs += __popcll(live);
      while (live) {
        const int src = __ffsll((long long)live) - 1;
        live &= live - 1;
        float t = 0.f;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
          if (ci < cin) {
            const float xs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xv[ci]), src));
            const float wv = w[(int64_t)(k0 + src) * 256 + (32 * (ci >> 2) + co) * 4 + (ci & 3)];
            t = fmaf(xs, wv, t);   // same k-ordered fma chain as the MFMA path
          }
        }
        acc += t;
      }
    }
    if (lane < 32) out[o * out_ld + co] = acc;
    if (out_amax) {   // (both half-waves hold the same 32 channels)
      uint32_t mx = __builtin_bit_cast(uint32_t, acc) & 0x7fffffffu;
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
      if (lane == 0) atomicMax(out_amax + o, mx);
    }
  }
  if (lane == 0 && pair_count) atomicAdd(pair_count, pairs);
}

// ------------------------------------------------------------------------------------------
// Dense-grid variant of the fused conv1.  A 3-D fragment is compact: the bounding boxes of the batch
// elements (padded by ks/2) hold a few million cells, so the voxel -> row map fits a plain int32 grid
// and a neighbour probe is ONE load of ks consecutive cells per (ky, kz) row -- no hashing, no probe
// chains, 49 independent 28-byte reads per voxel instead of 343 dependent table + key reads.
//   bbox    per-batch-element min / max (block-reduced, 6 atomics per block)
//   layout  grid dims, per-element bases, total cell count; `ok` = fits the cell budget and every batch
//           index is in [0, 64) -- otherwise the hash-probe kernel (launched behind) does the layer
//   clear / fill
//   conv    a wave takes TWO voxels at a time.  Phase A (all lanes, one voxel after the other): lane =
//           (ky, kz) row reads its ks cells; ballots rank the hits in ascending k; (k, row) lists go to
//           LDS.  Phase B: half-wave = voxel, lane = output channel walks its list -- the loads of a step
//           do not depend on the running sum, only the adds are serial -- same k-ordered sum as the
//           map-based path, bit for bit.
// ------------------------------------------------------------------------------------------
constexpr int GRID_MAXB = 64;
struct DgrGridMeta {
  int32_t ok, bad;
  long long total;
  int32_t mn[GRID_MAXB][3], mx[GRID_MAXB][3], dim[GRID_MAXB][3];
  long long base[GRID_MAXB];
};

__global__ void grid_init_kernel(DgrGridMeta *m, int32_t *pair_count) {
  const int b = threadIdx.x;
  if (b == 0) { m->ok = 0; m->bad = 0; m->total = 0; if (pair_count) *pair_count = 0; }
  if (b < GRID_MAXB) {
    for (int d = 0; d < 3; ++d) { m->mn[b][d] = INT32_MAX; m->mx[b][d] = INT32_MIN; m->dim[b][d] = 0; }
    m->base[b] = 0;
  }
}

__global__ void __launch_bounds__(256)
    grid_bbox_kernel(const int32_t *__restrict__ coords, const int32_t *n_dev, DgrGridMeta *m) {
  __shared__ int32_t red[6][256 / 64];
  __shared__ int32_t b0s, mixed;
  const int n = *n_dev;
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if ((int64_t)blockIdx.x * 256 >= n) return;
  const bool live = o < n;
  const int4 c = live ? *reinterpret_cast<const int4 *>(coords + o * 4) : make_int4(0, 0, 0, 0);
  if (threadIdx.x == 0) { b0s = c.x; mixed = 0; }
  __syncthreads();
  if (live && c.x != b0s) mixed = 1;
  if (live && (c.x < 0 || c.x >= GRID_MAXB)) m->bad = 1;
  __syncthreads();
  // min / max of the lanes selected by `mine` over the wave (the others carry the identities)
  auto wave_minmax = [&](bool mine, int32_t (&v)[6]) {
    v[0] = mine ? c.y : INT32_MAX; v[1] = mine ? c.z : INT32_MAX; v[2] = mine ? c.w : INT32_MAX;
    v[3] = mine ? c.y : INT32_MIN; v[4] = mine ? c.z : INT32_MIN; v[5] = mine ? c.w : INT32_MIN;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        v[d] = min(v[d], __shfl_xor(v[d], s, 64));
        v[3 + d] = max(v[3 + d], __shfl_xor(v[3 + d], s, 64));
      }
    }
  };
  int32_t v[6];
  if (mixed) {
    // the block straddles batch elements (a handful of blocks per call): one reduction per wave and batch element
    // present in it, six atomics each.  (Per-thread atomics here -- 1536 on the same few addresses per such block, at
    // ~40 ns each chip-wide -- made this kernel 50 us of the 290-us conv1 layer.)
    const bool ok = live && c.x >= 0 && c.x < GRID_MAXB;
    unsigned long long todo = __ballot(ok);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int b = __shfl(c.x, leader, 64);
      const bool mine = ok && c.x == b;
      wave_minmax(mine, v);
      if ((int)(threadIdx.x & 63) == leader) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { atomicMin(&m->mn[b][d], v[d]); atomicMax(&m->mx[b][d], v[3 + d]); }
      }
      todo &= ~__ballot(mine);
    }
    return;
  }
  wave_minmax(live, v);
  if ((threadIdx.x & 63) == 0)
    for (int d = 0; d < 6; ++d) red[d][threadIdx.x >> 6] = v[d];
  __syncthreads();
  if (threadIdx.x < 6) {
    const int d = threadIdx.x;
    int32_t r = red[d][0];
    for (int w = 1; w < 4; ++w) r = d < 3 ? min(r, red[d][w]) : max(r, red[d][w]);
    const int b = b0s;
    if (b >= 0 && b < GRID_MAXB) {
      if (d < 3) atomicMin(&m->mn[b][d], r); else atomicMax(&m->mx[b][d - 3], r);
    }
  }
}

// one wave: lane b sizes batch element b, a shuffle scan yields the bases (launched with 64 threads = GRID_MAXB)
__global__ void grid_layout_kernel(DgrGridMeta *m, int pad, long long cap) {
  static_assert(GRID_MAXB == 64, "one lane per batch element");
  const int b = threadIdx.x;
  long long vol = 0;
  if (m->mx[b][0] >= m->mn[b][0]) {   // the element has rows
    vol = 1;
    for (int d = 0; d < 3; ++d) {
      const long long e = (long long)m->mx[b][d] - m->mn[b][d] + 1 + 2 * pad;
      m->dim[b][d] = (int32_t)(e < (1ll << 30) ? e : (1ll << 30));
      vol = (vol <= cap && e <= cap) ? vol * e : cap + 1;
    }
  }
  long long incl = vol;   // inclusive scan (every term <= cap + 1: no overflow over 64 lanes)
#pragma unroll
  for (int s = 1; s < 64; s <<= 1) {
    const long long up = __shfl_up(incl, s, 64);
    if (b >= s) incl += up;
  }
  m->base[b] = incl - vol;
  const long long total = __shfl(incl, 63, 64);
  if (b == 0) {
    m->total = total <= cap ? total : cap + 1;
    m->ok = (!m->bad && total <= cap) ? 1 : 0;
  }
}

__global__ void grid_clear_kernel(const DgrGridMeta *m, int32_t *__restrict__ cells) {
  if (!m->ok) return;
  const long long total = m->total;
  for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < total; i += (long long)gridDim.x * blockDim.x * 4)
    *reinterpret_cast<int4 *>(cells + i) = make_int4(-1, -1, -1, -1);   // capacity is padded to a multiple of 4
}

__device__ __forceinline__ long long grid_cell(const DgrGridMeta *m, int b, int x, int y, int z, int pad) {
  const long long dx = m->dim[b][0], dy = m->dim[b][1];
  return m->base[b] + ((long long)(z - m->mn[b][2] + pad) * dy + (y - m->mn[b][1] + pad)) * dx + (x - m->mn[b][0] + pad);
}

// cells[cell of row o] = o -- or, for a one-channel input (`values`), the row's input value itself: the MFMA kernel then
// needs no second, dependent load per neighbour.  Empty cells hold -1 = 0xffffffff either way (as a float: a NaN
// payload no arithmetic produces; an input feature with exactly these bits would read as "no voxel").
constexpr uint32_t GRID_EMPTY = 0xffffffffu;
__global__ void grid_fill_kernel(const int32_t *__restrict__ coords, const int32_t *n_dev, const DgrGridMeta *m, int pad,
                                 int32_t *__restrict__ cells, const float *__restrict__ values, int values_ld) {
  if (!m->ok) return;
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= *n_dev) return;
  const int4 c = *reinterpret_cast<const int4 *>(coords + o * 4);
  cells[grid_cell(m, c.x, c.y, c.z, c.w, pad)] = values ? __builtin_bit_cast(int32_t, values[o * values_ld]) : (int32_t)o;
}

constexpr int C1_MAXK = 343;   // ks <= 7
__global__ void __launch_bounds__(256)
    conv1_grid_kernel(const int32_t *__restrict__ coords, const int32_t *n_dev, const DgrGridMeta *__restrict__ m,
                      const int32_t *__restrict__ cells, int ks, const float *__restrict__ in, int in_ld, int cin,
                      const float *__restrict__ w, const float *__restrict__ shift, float *__restrict__ out,
                      int out_ld, int32_t *pair_count, uint32_t *__restrict__ out_amax) {
  __shared__ int2 lists[256 / 64][2][C1_MAXK + 1];
  if (!m->ok) return;
  const int n = *n_dev;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int half = ks >> 1, ks2 = ks * ks;
  const int ky = lane % ks, kz = lane / ks;
  const int vsel = lane >> 5, co = lane & 31;
  const unsigned long long below = (1ull << lane) - 1ull;
  const int64_t wave_id = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  int pairs = 0;
  for (int64_t o0 = wave_id * 2; o0 < n; o0 += n_waves * 2) {
    int nh[2] = {0, 0};
    // ---- phase A: hit lists of the two voxels, ascending k
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int64_t o = o0 + v;
      if (o >= n) break;   // wave-uniform
      const int4 c = *reinterpret_cast<const int4 *>(coords + o * 4);
      int hit[7];
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) hit[kx] = -1;
      if (lane < ks2) {
        const long long c0 = grid_cell(m, c.x, c.y - half, c.z + ky - half, c.w + kz - half, half);
#pragma unroll
        for (int kx = 0; kx < 7; ++kx)
          if (kx < ks) hit[kx] = cells[c0 + kx];
      }
      unsigned long long mk[7];
      int before = 0;
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) {
        mk[kx] = __ballot(hit[kx] >= 0);
        before += __popcll(mk[kx] & below);
        nh[v] += __popcll(mk[kx]);
      }
      int r = before;
#pragma unroll
      for (int kx = 0; kx < 7; ++kx)
        if (hit[kx] >= 0) lists[wv][v][r++] = make_int2(kx + ks * lane, hit[kx]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- phase B: half-wave = voxel, lane = output channel
    pairs += nh[0] + nh[1];
    const int my_n = vsel ? nh[1] : nh[0];
    const int trips = max(nh[0], nh[1]);
    float acc = shift ? shift[co] : 0.f;
    // eight list entries per trip: all loads of a trip (per input channel) are issued together -- nothing
    // between them consumes a loaded value -- and only the adds form a serial chain
    for (int r0 = 0; r0 < trips; r0 += 8) {
      int2 e[8];
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        e[u] = (r0 + u < my_n) ? lists[wv][vsel][r0 + u] : make_int2(0, 0);
        t[u] = 0.f;
      }
      for (int ci = 0; ci < cin; ++ci) {   // wave-uniform trip count (1 for FCGF)
        float xs[8], ws[8];
        const int wofs = (32 * (ci >> 2) + co) * 4 + (ci & 3);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          xs[u] = in[(int64_t)e[u].y * in_ld + ci];
          ws[u] = w[(int64_t)e[u].x * 256 + wofs];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = fmaf(xs[u], ws[u], t[u]);   // same k-ordered fma chain as the MFMA path
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = (r0 + u < my_n) ? acc + t[u] : acc;
    }
    const int64_t o = o0 + vsel;
    if (o < n) out[o * out_ld + co] = acc;
    if (out_amax) {   // a voxel's 32 channels are the 32 lanes of a half-wave
      uint32_t mx = o < n ? (__builtin_bit_cast(uint32_t, acc) & 0x7fffffffu) : 0u;
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
      if (co == 0 && o < n) atomicMax(out_amax + o, mx);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();   // the lists are rewritten by the next pair of voxels
  }
  if (lane == 0 && pair_count) atomicAdd(pair_count, pairs);
}

// ------------------------------------------------------------------------------------------
// The same layer on the matrix cores, for ONE input channel (FCGF: feats = ones [N, 1]): per voxel the layer is a
// 343-term (ks = 7) dot product per output channel, i.e. out[16 voxels x 32 channels] = G[16 x ks^3] . W[ks^3 x 32]
// with G[v][k] = x[neighbour k of v] or 0 -- a GEMM whose left operand is gathered from the dense grid.  A wave owns
// 16 voxels and walks the ks z-slabs of the kernel: lanes = (voxel, y-row) items read their ks consecutive cells of the
// VALUE grid (grid_fill_kernel: the cell holds the voxel's input value, so a neighbour costs no second, dependent load;
// 28 contiguous bytes per item = two vector loads) into a [16 x ks^2] slab of G in LDS, double-buffered per wave (no
// block barrier), and ceil(ks^2 / 4) x 2 v_mfma_f32_16x16x4_f32 multiply it with the slab's weights.  Zeros of G
// contribute exactly 0 to the fma chain, so a channel's value is the ascending-k chain over the hits, as in the scalar
// kernels (one rounding per term instead of two).
// Weights: wt[kz][s][lane] = {W[k][lane & 15], W[k][16 + (lane & 15)]} with k = kz ks^2 + 4 s + (lane >> 4) (zero past
// the slab): the A operands of step s as ONE coalesced 8-byte load per lane (net.hip); operands are swapped
// (D = W^T G^T): a lane ends with one voxel and 4 consecutive channels per 16-channel block.
// Round 3 kept row indices in the grid: per item 7 index loads + 7 dependent value loads, every lane on its own cache
// line -- 196 scattered load instructions per 16 voxels, 213 us for 195 k voxels, bound by the address path.
// ------------------------------------------------------------------------------------------
template <int KS>
__global__ void __launch_bounds__(256)
    conv1_grid_mfma(const int32_t *__restrict__ coords, const int32_t *n_dev, const DgrGridMeta *__restrict__ m,
                    const int32_t *__restrict__ cells, const float *__restrict__ wt, const float *__restrict__ shift,
                    float *__restrict__ out, int out_ld, int32_t *pair_count, uint32_t *__restrict__ out_amax) {
  constexpr int KS2 = KS * KS, HALF = KS / 2;
  constexpr int RS = (KS2 + 3) / 4 * 4;          // slab length padded to MFMA k-steps
  constexpr int LDG = RS + 1;
  constexpr int ITEMS = 16 * KS;                 // (voxel, y-row) items per slab
  constexpr int ROUNDS = (ITEMS + 63) / 64;
  __shared__ float G[4][2][16][LDG];
  __shared__ int4 vc[4][16];
  __shared__ int pair_sum;
  if (!m->ok) return;
  const int n = *n_dev;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t v0 = ((int64_t)blockIdx.x * 4 + wv) * 16;
  if (threadIdx.x == 0) pair_sum = 0;
  __syncthreads();
  int pairs = 0;
  if (v0 < n) {
    if (lane < 16) vc[wv][lane] = v0 + lane < n ? *reinterpret_cast<const int4 *>(coords + (v0 + lane) * 4) : make_int4(-1, 0, 0, 0);
    // the padding columns stay zero for the whole kernel
    for (int e = lane; e < 2 * 16 * (LDG - KS2); e += 64) {
      const int b = e / (16 * (LDG - KS2)), r = (e / (LDG - KS2)) % 16, c = KS2 + e % (LDG - KS2);
      G[wv][b][r][c] = 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t cell[ROUNDS][KS];
    // request the grid cells of slab kz for this lane's items (dead items read cell 0 of the grid: no branch)
    auto cells_of = [&](int kz) {
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const int it = lane + 64 * r;
        const int v = it / KS, ky = it % KS;
        const int4 c = vc[wv][min(v, 15)];
        const bool live = it < ITEMS && c.x >= 0;
        const long long c0 = live ? grid_cell(m, c.x, c.y - HALF, c.z + ky - HALF, c.w + kz - HALF, HALF) : 0;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) cell[r][kx] = (uint32_t)cells[c0 + kx];
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) cell[r][kx] = live ? cell[r][kx] : GRID_EMPTY;
      }
    };
    // the landed values -> slab buffer b
    auto fill = [&](int b) {
#pragma unroll
      for (int r = 0; r < ROUNDS; ++r) {
        const int it = lane + 64 * r;
        const int v = it / KS, ky = it % KS;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const bool hit = cell[r][kx] != GRID_EMPTY;
          pairs += hit;
          if (it < ITEMS) G[wv][b][v][kx + KS * ky] = hit ? __builtin_bit_cast(float, cell[r][kx]) : 0.f;
        }
      }
    };
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    cells_of(0);
    fill(0);
    for (int kz = 0; kz < KS; ++kz) {
      if (kz + 1 < KS) cells_of(kz + 1);          // in flight while this slab is multiplied
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float *g = &G[wv][kz & 1][lane & 15][lane >> 4];
      const f32x2 *w = reinterpret_cast<const f32x2 *>(wt) + (int64_t)kz * (RS / 4) * 64 + lane;
#pragma unroll
      for (int s = 0; s < RS / 4; ++s) {
        const float b = g[4 * s];
        const f32x2 a = w[s * 64];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b, acc[1], 0, 0, 0);
      }
      if (kz + 1 < KS) fill((kz + 1) & 1);
    }
    const int64_t o = v0 + (lane & 15);
    uint32_t mx = 0;
    if (o < n) {
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) {
        f32x4 v = acc[jb];
        if (shift) v += *reinterpret_cast<const f32x4 *>(shift + 16 * jb + 4 * (lane >> 4));
        *reinterpret_cast<f32x4 *>(out + o * out_ld + 16 * jb + 4 * (lane >> 4)) = v;
        const i32x4 b = __builtin_bit_cast(i32x4, v);
        mx = max(mx, max(max((uint32_t)b.x & 0x7fffffffu, (uint32_t)b.y & 0x7fffffffu), max((uint32_t)b.z & 0x7fffffffu, (uint32_t)b.w & 0x7fffffffu)));
      }
    }
    if (out_amax) {   // the row's largest |x| for its split-operand consumers (conv_os.hip / conv_dense.hip): a voxel's
                      // channels sit in the four lanes (lane & 15) + 16 q
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, 16, 64));
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, 32, 64));
      if (lane < 16 && o < n) atomicMax(out_amax + o, mx);
    }
  }
  if (pair_count) {   // (statistics: one atomic per workgroup)
    for (int d = 32; d > 0; d >>= 1) pairs += __shfl_down(pairs, d, 64);
    if (lane == 0 && pairs) atomicAdd(&pair_sum, pairs);
    __syncthreads();
    if (threadIdx.x == 0 && pair_sum) atomicAdd(pair_count, pair_sum);
  }
}

int dgr_conv1_probe(DgrArena &arena, const DgrCoordMap &cm, int ks, const float *in, int in_ld, int cin,
                    const float *w_tiled, const float *shift, float *out, int out_ld, int32_t *pair_count,
                    hipStream_t stream, const float *w_compact, const char **kernel_name, uint32_t *out_amax) {
  DGR_REQUIRE(cin >= 1 && cin <= 8 && ks % 2 == 1, "conv1 probe: cin=%d ks=%d", cin, ks);
  if (pair_count && ks > 7) DGR_HIP_CHECK(hipMemsetAsync(pair_count, 0, sizeof(int32_t), stream));   // (else: grid_init_kernel)
  const int32_t *grid_done = nullptr;
  if (ks <= 7) {
    // dense grid: up to 64 M cells (256 MB) of transient arena memory
    static const long long cap = 64ll << 20;
    DgrArena::Mark mk = arena.mark();
    DgrGridMeta *meta;
    int32_t *cells;
    DGR_ALLOC(meta, arena, DgrGridMeta, 1);
    DGR_ALLOC(cells, arena, int32_t, cap + 4);
    grid_init_kernel<<<1, 64, 0, stream>>>(meta, pair_count);
    grid_bbox_kernel<<<(int)dgr_ceil_div(cm.n_cap, 256), 256, 0, stream>>>(cm.coords, cm.n_dev, meta);
    grid_layout_kernel<<<1, 64, 0, stream>>>(meta, ks >> 1, cap);
    grid_clear_kernel<<<2048, 256, 0, stream>>>(meta, cells);
    // one input channel + MFMA-tiled weights: the grid holds the input VALUES and conv1_grid_mfma does the layer
    const bool mfma = cin == 1 && w_compact && (out_ld & 3) == 0 && (ks == 3 || ks == 5 || ks == 7);
    grid_fill_kernel<<<(int)dgr_ceil_div(cm.n_cap, 256), 256, 0, stream>>>(cm.coords, cm.n_dev, meta, ks >> 1, cells,
                                                                           mfma ? in : nullptr, in_ld);
    // grid-stride kernel: a whole number of resident rounds (a ragged last round costs up to 10 %)
    static int resident = 0;
    if (resident == 0) {
      int per_cu = 0, dev = 0, cus = 0;
      DGR_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv1_grid_kernel, 256, 0));
      DGR_HIP_CHECK(hipGetDevice(&dev));
      DGR_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      resident = (per_cu < 1 ? 1 : per_cu) * (cus < 1 ? 1 : cus);
    }
    if (mfma) {
      const int wgs = (int)dgr_ceil_div(cm.n_cap, 64);
      if (ks == 7) conv1_grid_mfma<7><<<wgs, 256, 0, stream>>>(cm.coords, cm.n_dev, meta, cells, w_compact, shift, out, out_ld, pair_count, out_amax);
      else if (ks == 5) conv1_grid_mfma<5><<<wgs, 256, 0, stream>>>(cm.coords, cm.n_dev, meta, cells, w_compact, shift, out, out_ld, pair_count, out_amax);
      else conv1_grid_mfma<3><<<wgs, 256, 0, stream>>>(cm.coords, cm.n_dev, meta, cells, w_compact, shift, out, out_ld, pair_count, out_amax);
      if (kernel_name) *kernel_name = ks == 7 ? "conv1_grid_mfma<7>" : ks == 5 ? "conv1_grid_mfma<5>" : "conv1_grid_mfma<3>";
    } else {
    int64_t blocks = dgr_ceil_div(cm.n_cap, 8);
    if (blocks > resident) blocks = (int64_t)resident * std::min<int64_t>(4, blocks / resident);
    conv1_grid_kernel<<<(int)blocks, 256, 0, stream>>>(cm.coords, cm.n_dev, meta, cells, ks, in, in_ld, cin, w_tiled, shift,
                                                       out, out_ld, pair_count, out_amax);
    if (kernel_name) *kernel_name = "conv1_grid_kernel";
    }
    DGR_LAUNCH_CHECK();
    grid_done = &meta->ok;
    arena.rewind(mk);   // stream order keeps the grid alive until the kernels above are done
  }
  {
    // (grid-stride; it returns at once when the dense-grid kernel did the layer: a modest grid keeps that case cheap)
    int64_t blocks = dgr_ceil_div(cm.n_cap, 4);
    if (blocks > 4096) blocks = 4096;
    conv1_probe_kernel<<<(int)blocks, 256, 0, stream>>>(cm.coords, cm.n_dev, cm.table, cm.table_mask, ks, in, in_ld, cin,
                                                        w_tiled, shift, out, out_ld, pair_count, grid_done, out_amax);
  }
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

// F / (||F||_2 + 1e-8) per row, model/resunet.py:643-647.  One row per thread (C <= 64).
__global__ void l2_normalize_kernel(const float *__restrict__ in, int in_ld, float *__restrict__ out,
                                    int out_ld, int c, int relu, const int32_t *n_dev) {
  const int64_t n = *n_dev;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < c; ++j) {
      float v = in[r * in_ld + j];
      if (relu) v = fmaxf(v, 0.f);
      s += v * v;
    }
    const float den = sqrtf(s) + 1e-8f;
    for (int j = 0; j < c; ++j) {
      float v = in[r * in_ld + j];
      if (relu) v = fmaxf(v, 0.f);
      out[r * out_ld + j] = v / den;
    }
  }
}

int dgr_l2_normalize_rows(const float *in, int in_ld, float *out, int out_ld, int c, int relu,
                          const int32_t *n_dev, int64_t n_cap, hipStream_t stream) {
  int64_t blocks = dgr_ceil_div(n_cap, 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  l2_normalize_kernel<<<(int)blocks, 256, 0, stream>>>(in, in_ld, out, out_ld, c, relu, n_dev);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}
