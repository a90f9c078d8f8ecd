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
#include "dgr_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct ConvBf3Args {
  const float *in;
  float *y;               // [pairs, cout] per-pair product rows
  const uint4 *wb;        // three pieces, each [K][CP/16][cout/32][64] uint4
  const int32_t *pair_in, *tile_ptr;
  const int4 *tile_desc;
  int64_t piece_stride;   // uint4 per piece
  int in_ld, in_relu, cin, cout, K;
};

// x = h + m + l exactly, each piece a bf16 value held in the upper half of a 32-bit word
__device__ __forceinline__ void dgr_split3(float x, uint32_t &h, uint32_t &m, uint32_t &l) {
  h = __builtin_bit_cast(uint32_t, x) & 0xffff0000u;
  const float r1 = x - __builtin_bit_cast(float, h);
  m = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
  l = __builtin_bit_cast(uint32_t, r1 - __builtin_bit_cast(float, m));   // <= 8 significant bits left: exact
}

template <int CP, int MB, int NB>
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
  __shared__ __attribute__((aligned(16))) unsigned short Ps[2][3][PLANE];
  __shared__ int idxbuf[4][TM];

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
    unsigned short *dst = &Ps[q & 1][0][0];
    const int ch = tid + i * THREADS;
    const bool ok = (g_ok >> i) & 1u;
    uint32_t h[4], m[4], l[4];
    const i32x4 gi = __builtin_bit_cast(i32x4, G[i]);   // (bit_cast of a single vector ELEMENT reads element 0)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int xb = max(gi[u], relu_lo);   // pending ReLU as one integer max
      xb = ok ? xb : 0;
      dgr_split3(__builtin_bit_cast(float, xb), h[u], m[u], l[u]);
    }
    const int o = (ch / C4K) * LDP + (ch % C4K) * 4;
    *reinterpret_cast<u32x2 *>(dst + o) = u32x2{(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
    *reinterpret_cast<u32x2 *>(dst + PLANE + o) = u32x2{(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]};
    *reinterpret_cast<u32x2 *>(dst + 2 * PLANE + o) = u32x2{(l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u)};
  };
  // weight operands of k-step s (16 input channels) of rule k: 3 pieces x NB column blocks, 16 bytes per lane each
  auto wload = [&](int k, int s, uint4 (*w)[3]) {
#ifdef DGR_BF3_ABL_BONCE
    const uint4 *p = a.wb + ((int64_t)(0 * S + (s & 1)) * NBLK + wn * NB) * 64 + lane;   // L1-resident: no L2 weight stream
#else
    const uint4 *p = a.wb + ((int64_t)(k * S + s) * NBLK + wn * NB) * 64 + lane;
#endif
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int pc = 0; pc < 3; ++pc) w[j][pc] = p[(int64_t)pc * a.piece_stride + j * 64];
  };

  // ---- prologue
  {
    const int i0 = load_idx(0), i1 = load_idx(1), i2 = load_idx(2), i3 = load_idx(3);
    if (tid < TM) { idxbuf[0][tid] = i0; idxbuf[1][tid] = i1; idxbuf[2][tid] = i2; idxbuf[3][tid] = i3; }
  }
  int next_pub = 4;   // the ring holds the tiles of phases q + 1 .. q + 3; tile (q + 4) / PPT is published one phase ahead
  int idx_reg = load_idx(4);
  int k, pstart, count;
  locate(0, k, pstart, count);
  uint4 w[2][NB][3];            // k-step ring of depth 2
  wload(k, 0, w[0]);
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
    if ((q + 4) / PPT >= next_pub && next_pub < n_my) {
      if (tid < TM) idxbuf[next_pub & 3][tid] = idx_reg;
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
      if (s0 + s + 1 < S) {
        wload(k, s0 + s + 1, w[(s + 1) & 1]);
      } else if (q / PPT + 1 < n_my) {
        int k2, p2, c2;
        locate(q / PPT + 1, k2, p2, c2);
        wload(k2, 0, w[(s + 1) & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);   // pin the prefetch ahead of the MFMA block (see conv.hip)
      bf16x8 ah[MB], am[MB], al[MB];
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const unsigned short *p = pl + i * 32 * LDP + s * 16;
        ah[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(p));
        am[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(p + PLANE));
        al[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(p + 2 * PLANE));
      }
      uint4 (*wc)[3] = w[s & 1];
      // six products per accumulator, small terms first; the (i, j) loops are innermost so that consecutive
      // MFMAs write different accumulators
#define DGR_BF3_TERM(WP, AX)                                                                                        \
  _Pragma("unroll") for (int i = 0; i < MB; ++i) _Pragma("unroll") for (int j = 0; j < NB; ++j)                     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wc[j][WP]), AX[i], acc[i][j], 0, 0, 0);
      DGR_BF3_TERM(2, ah)   // wl . ah
      DGR_BF3_TERM(0, al)   // wh . al
      DGR_BF3_TERM(1, am)   // wm . am
      DGR_BF3_TERM(1, ah)   // wm . ah
      DGR_BF3_TERM(0, am)   // wh . am
      DGR_BF3_TERM(0, ah)   // wh . ah
#undef DGR_BF3_TERM
      // piece s of the NEXT phase: split + land (requested one phase ago), then request piece s of the phase after
      land_piece(q + 1, s, G, g_ok);
      gather_piece(q_req, s, G, g_ok);
    }
    if (h == PPT - 1) {
      const int pst = pstart, cnt = count;
      if (q / PPT + 1 < n_my) locate(q / PPT + 1, k, pstart, count);
      // product rows: D column (pair) = lane & 31, D row (channel) = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const int r = 32 * i + (lane & 31);
        if (r < cnt) {
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
              const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
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

template <int CP, int MB, int NB>
static int launch_bf3(const ConvBf3Args &ka, int64_t tile_bound, int num_cus, hipStream_t stream) {
  static int per_cu = 0;
  if (per_cu == 0) {
    int n = 0;
    DGR_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sparse_conv_bf16x3<CP, MB, NB>, 256, 0));
    per_cu = n < 1 ? 1 : (n > 8 ? 8 : n);
  }
  int64_t grid = (int64_t)num_cus * per_cu;
  if (tile_bound < grid) grid = tile_bound;
  grid = (grid + 7) / 8 * 8;
  if (grid < 8) grid = 8;
  sparse_conv_bf16x3<CP, MB, NB><<<(int)grid, 256, 0, stream>>>(ka);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

bool dgr_conv_bf3_supported(int cin_pad, int cin, int cout) {
  return cin == cin_pad && (cin == 64 || cin == 128 || cin == 256) && (cout == 128 || cout == 256);
}

int dgr_conv_bf3_launch(const DgrConvLaunch &a, const void *wb, int64_t piece_stride, int num_cus, hipStream_t stream,
                        const char **kernel_name) {
  DGR_REQUIRE(a.pair_in && wb && dgr_conv_bf3_supported(a.cin_pad, a.cin, a.cout), "bf16x3 conv: unsupported layer");
  DGR_REQUIRE((a.in_ld & 3) == 0, "bf16x3 conv: input row stride must be a multiple of 4");
  ConvBf3Args ka;
  ka.in = a.in; ka.y = a.y; ka.wb = static_cast<const uint4 *>(wb); ka.piece_stride = piece_stride;
  ka.pair_in = a.pair_in; ka.tile_ptr = a.tile_ptr; ka.tile_desc = a.tile_desc;
  ka.in_ld = a.in_ld; ka.in_relu = a.in_relu; ka.cin = a.cin; ka.cout = a.cout; ka.K = a.K;
  const int64_t tile_bound = a.tile_bound > 0 ? a.tile_bound : (int64_t)num_cus * 4;
#define DGR_BF3(CPV, NBV)                                                              \
  do {                                                                                 \
    if (kernel_name) *kernel_name = "sparse_conv_bf16x3<" #CPV ", 2, " #NBV ">";       \
    return launch_bf3<CPV, 2, NBV>(ka, tile_bound, num_cus, stream);                   \
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
