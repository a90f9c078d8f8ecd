// Output-stationary fused sparse convolution for the 3-D (FCGF) net on gfx950.
// Replaces, for D = 3, the ME.MinkowskiConvolution / MinkowskiConvolutionTranspose forward of every
// K = 27 conv of ResUNetBN2C (model/resunet.py:598-649, model/residual_block.py:15-80) -- same
// arithmetic as the rule-major path (per pair a product row, summed per output row in ascending offset
// order on top of the folded batch-norm shift and the residual), but with NO product rows in HBM and NO
// reduction pass:
//
//   * a workgroup owns MB (64 | 32 | 16 by level) consecutive output rows and a slice of CS output channels;
//     the rows' accumulators [MB x CS] live in LDS for the whole layer and are written once at the end;
//   * the only kernel-map structure is the dense neighbour table nbr[27][n_pad] (kmap.hip): per offset k
//     the block's entries are ballot-compacted into a list of (input row, local output row);
//   * 16-row groups -- an offset with c pairs yields ceil(c / 16) -- are packed into tiles of TM / 16 groups;
//     the gathered input rows go through a double-buffered LDS tile (requested one phase ahead, landed
//     one phase later: one barrier per phase) and are multiplied with W[k]; the map's fill (47 % of a
//     64-row block per offset on 3DMatch-shaped clouds) does not turn into idle matrix cycles; operands
//     are swapped (D = W^T In^T) so that a lane ends up with one pair and 4 consecutive channels and the
//     accumulation into the LDS row is one 16-byte read-modify-write;
//   * channel ranges are exclusive per wave (and the waves that share a channel range work on groups of
//     ONE offset, whose output rows are distinct), so the LDS accumulation needs no atomics and no extra
//     barrier, and the sum order is ascending k: results do not depend on scheduling (bit-reproducible).
//
// Arithmetic (PM): 2 = every f32 operand as two f16 pieces under exact power-of-two row / layer scales, three
// v_mfma_f32_16x16x32_f16 per 32 input channels (default; error analysis in conv_wide.hip, measured against f64 in
// tests/test_gpu_split_f64.py); 0 = v_mfma_f32_16x16x4_f32 on the f32 operands (DGR_EXACT_F32=1).
//
// Weight layouts (net.hip, per layer): split pieces WB[piece][k][s][jb][lane] = 8 halves =
// W_folded[k][32 s + 8 (lane >> 4) + e][16 jb + (lane & 15)]; f32: W16[k][g][jb][lane][c] =
// W_folded[k][16 g + 4 (lane >> 4) + c][16 jb + (lane & 15)] -- one coalesced 16-byte load per lane either way.
#include <stdlib.h>

#include <type_traits>

#include "dgr_internal.h"
#include "split.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#ifdef DGR_OS_STAGE_CLK
// tools/os_stage_clk.py (build with EXTRA=-DDGR_OS_STAGE_CLK): per-workgroup wall-clock sums of the kernel's stages --
// [0] list compaction, [1] accumulator init + group list, [2] the phases, [3] epilogue, [4] workgroups, [5] phases
__device__ unsigned long long dgr_os_stage_clk[8];
#define DGR_OS_CLK(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&dgr_os_stage_clk[i], t_ - t_prev); t_prev = t_; } } while (0)
extern "C" int dgr_debug_os_stage_clk(unsigned long long *out8, int reset) {
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(dgr_os_stage_clk), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(dgr_os_stage_clk), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#else
#define DGR_OS_CLK(i) do { } while (0)
#endif
struct ConvOsArgs {
  const float *in;
  float *out;
  const float *w16, *shift, *res;
  const uint4 *wb3;        // split weights: two f16 pieces, each [27][CP/32][cout/16][64] x 16 bytes
  int64_t piece_stride;    // 16-byte units per piece
  const int32_t *nbr, *n_out_dev;
  int64_t n_pad;
  int in_ld, in_relu, out_ld, out_relu, res_ld, res_relu;
  int cin, cout, nb16;   // nb16 = cout / 16
  const uint32_t *row_amax;  // PM = 2: bits of every input row's largest |x| (after the pending ReLU), left behind by
                             // the row's producer; the row's power-of-two scale is dgr_row_scale_of of it
  float w_unscale;           // PM = 2: inverse of the layer's weight scale
  uint32_t *out_amax, *out_amax2;   // (nullable) the same for the rows this layer writes: atomicMax per row
};

// CP = input channels (multiple of 32), CS = output-channel slice of a workgroup (32 | 64), MB = output rows
// per workgroup (16 | 32 | 64), CK = input channels per pipeline phase (32 | 64), TM = pair slots per tile (32 | 64)
// PM = pieces per operand: 0 = exact-f32 MFMA, 2 = f16 x 2 with exact power-of-two row / layer scales (three
// v_mfma_f32_16x16x32_f16 per 32 input channels)
// GW = groups of a tile per wave: GW = TM / 16 -> CS / 16 waves, each walks all groups of the tile (the coarse levels,
// where an offset rarely fills more than one group); GW = 1 -> (TM / 16) x (CS / 16) waves, one group each: twice
// the waves on the same LDS for the two finest levels, whose phases are chains of dependent LDS round trips
template <int CP, int CS, int MB, int CK, int TM, int PM, int GW>
__global__ void __launch_bounds__(CS * 4 * (TM / 16 / GW)) sparse_conv_os(ConvOsArgs a) {
  constexpr bool BF3 = PM != 0;
  static_assert(PM == 0 || PM == 2, "arithmetic mode");
  constexpr int NP = PM == 0 ? 1 : 2;
  constexpr int GP = TM / 16;            // 16-row groups per tile
  constexpr int NCW = CS / 16;           // 16-channel blocks of the slice
  constexpr int NWR = GP / GW;           // wave rows: each owns GW consecutive groups of every tile
  static_assert(GP % GW == 0, "groups per wave must divide the groups per tile");
  constexpr int NW = NCW * NWR;
  constexpr int THREADS = 64 * NW;
  constexpr int KV = 27;
  constexpr int PPT = CP / CK;           // phases per tile
  constexpr int LDA = CK + 4, LDC = CS + 4;
  constexpr int C4K = CK / 4;
  constexpr int NCH = (TM * C4K + THREADS - 1) / THREADS;   // 16-byte gather pieces per thread per phase
  constexpr bool PIECE_GUARD = TM * C4K % THREADS != 0;      // (Cin = 32 into a 64-channel slice: half the threads have none)
  constexpr int G = CK / 16;                // MFMA groups (4 MFMAs, 16 input channels) per phase
  constexpr int GT = CP / 16;               // groups per tile
  constexpr int KPW = (KV + NW - 1) / NW;   // offsets compacted per wave
  constexpr int NGMAX = KV * ((MB / 16 + GP - 1) / GP * GP) + GP;
  static_assert(CP % CK == 0, "shape");
  constexpr int LDP = CK + 8;                 // split planes: f16 elements per plane row
  constexpr int PLANE = TM * LDP;             // bf16 elements per plane
  constexpr int G32 = CK / 32;                // split planes: 32-channel k-steps per phase
  constexpr int ABYTES = BF3 ? 2 * NP * PLANE * 2 : 2 * TM * LDA * 4;
  __shared__ __attribute__((aligned(16))) char abuf[ABYTES];   // f32: As[2][TM][LDA]; split: planes [2][2][TM][LDP] f16
  float (*As)[TM][LDA] = reinterpret_cast<float (*)[TM][LDA]>(abuf);
  unsigned short *Ps = reinterpret_cast<unsigned short *>(abuf);
  __shared__ __attribute__((aligned(16))) float acc_s[MB][LDC];
  __shared__ int in_idx[KV][MB];
  __shared__ float in_scale[PM == 2 ? KV : 1][MB];   // PM = 2: scale of the listed input rows
  __shared__ unsigned char out_loc[KV][MB];
  __shared__ int cnt[KV];
  __shared__ int grp[NGMAX + GP];   // k | first list entry << 8 | entries << 16, ascending k
  __shared__ int n_grp;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wave % NCW;   // this wave's 16-channel block
  const int rb0 = (wave / NCW) * GW;   // ... and its first group of every tile
  const int n_out = *a.n_out_dev;
  const int nblocks = (n_out + MB - 1) / MB;
  // XCD-aware block order: workgroup b runs on XCD b % 8; give every XCD a contiguous range of row blocks
  const int per = (nblocks + 7) >> 3;
  const int j = blockIdx.x >> 3;
  const int blk = (blockIdx.x & 7) * per + j;
  if (j >= per || blk >= nblocks) return;
  const int slice = blockIdx.y;
  const int64_t row0 = (int64_t)blk * MB;
#ifdef DGR_OS_STAGE_CLK
  unsigned long long t_prev = __builtin_amdgcn_s_memrealtime();
#endif

  if constexpr (PM == 2)
    for (int e = tid; e < KV * MB; e += THREADS) (&in_scale[0][0])[e] = 1.f;   // slots past a list's end: a finite scale
  __syncthreads();
  // ---- 1. per offset: compact the block's neighbour-table column into (input row, local output row)
  {
    int v[KPW];
#pragma unroll
    for (int u = 0; u < KPW; ++u) {
      const int k = wave + u * NW;
      v[u] = (k < KV && lane < MB) ? a.nbr[(int64_t)k * a.n_pad + row0 + lane] : -1;
    }
#pragma unroll
    for (int u = 0; u < KPW; ++u) {
      const int k = wave + u * NW;
      if (k < KV) {
        const unsigned long long m = __ballot(v[u] >= 0);
        if (v[u] >= 0) {
          const int pos = __popcll(m & ((1ull << lane) - 1ull));
          in_idx[k][pos] = v[u];
          out_loc[k][pos] = (unsigned char)lane;
          if constexpr (PM == 2) in_scale[k][pos] = dgr_row_scale_of(a.row_amax[v[u]]);
        }
        if (lane == 0) cnt[k] = __popcll(m);
      }
    }
  }
  DGR_OS_CLK(0);
  // ---- 2. accumulators start from the folded batch-norm shift (+ residual)
  for (int e = tid; e < MB * (CS / 4); e += THREADS) {
    const int r = e / (CS / 4), c = (e % (CS / 4)) * 4;
    f32x4 v = a.shift ? *reinterpret_cast<const f32x4 *>(a.shift + slice * CS + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.res && row0 + r < n_out) {
      f32x4 x = *reinterpret_cast<const f32x4 *>(a.res + (row0 + r) * a.res_ld + slice * CS + c);
      if (a.res_relu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
      v += x;
    }
    *reinterpret_cast<f32x4 *>(&acc_s[r][c]) = v;
  }
  __syncthreads();
  // ---- 3. 16-row groups in ascending offset order: an offset with c pairs yields ceil(c / 16) groups; four
  //         groups (of possibly different offsets) make one 64-slot tile
  // A tile never mixes offsets (an offset's group count is padded to a multiple of GP with empty groups): the GP
  // waves that share a 16-channel block work on the groups of ONE offset, whose output rows are distinct, so
  // their LDS read-modify-writes never meet
  if (wave == 0) {
    const int c = lane < KV ? cnt[lane] : 0;
    const int ng = NWR > 1 ? (((c + 15) >> 4) + GP - 1) / GP * GP : ((c + 15) >> 4);
    int x = ng;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
      const int y = __shfl_up(x, d, 64);
      if (lane >= d) x += y;
    }
    const int first = x - ng;
    for (int u = 0; u < ng; ++u) grp[first + u] = lane | ((16 * u) << 8) | (max(0, min(16, c - 16 * u)) << 16);
    if (lane == KV - 1) {
      n_grp = x;
      for (int u = 0; u < GP; ++u) grp[x + u] = 0;   // padding groups: offset 0, no entries
    }
  }
  __syncthreads();
  const int NG = n_grp;
  const int NT = (NG + GP - 1) / GP;
  const int NQ = NT * PPT;
  DGR_OS_CLK(1);

  f32x4 Gr[NCH];
  // Requests only -- nothing here consumes a loaded value (see conv.hip).  A slot beyond its group's entries
  // reads row 0: its product is never accumulated (MFMA rows are independent), so nothing is zeroed.
  auto gather = [&](int q) {
    const int t = q / PPT, cbase = (q % PPT) * CK;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = PIECE_GUARD ? min(tid + i * THREADS, TM * C4K - 1) : tid + i * THREADS;
      const int r = ch / C4K, c = cbase + (ch % C4K) * 4;
      const int info = grp[GP * t + (r >> 4)];
      const int row = in_idx[info & 255][((info >> 8) & 255) + (r & 15)];
      const uint32_t off = (uint32_t)(((r & 15) < (info >> 16)) ? row : 0) * (uint32_t)a.in_ld + (uint32_t)c;
      Gr[i] = *reinterpret_cast<const f32x4 *>(a.in + off);
    }
  };
  // pending ReLU of the producer as ONE integer max per value (negative floats are negative integers; no
  // canonicalising second instruction as with fmaxf on freshly loaded data)
  const int relu_lo = a.in_relu ? 0 : (int)0x80000000;
  auto land = [&](int q) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int ch = tid + i * THREADS;
      if (PIECE_GUARD && ch >= TM * C4K) continue;
      i32x4 v = __builtin_bit_cast(i32x4, Gr[i]);
      v.x = max(v.x, relu_lo); v.y = max(v.y, relu_lo); v.z = max(v.z, relu_lo); v.w = max(v.w, relu_lo);
      if constexpr (!BF3) {
        *reinterpret_cast<i32x4 *>(&As[q & 1][0][0] + (ch / C4K) * LDA + (ch % C4K) * 4) = v;
      } else {
        // s x = h + m (+ <= 2^-22): two f16 planes; slots past their group's entries carry row 0 under a
        // finite foreign scale (in_scale is initialised to 1) and are never accumulated; the landing past the last
        // phase (into the buffer nobody reads) looks the last tile's groups up again
        const int r = ch / C4K;
        const int info = grp[GP * (min(q, NQ - 1) / PPT) + (r >> 4)];
        const float sx = in_scale[info & 255][min(((info >> 8) & 255) + (r & 15), MB - 1)];
        _Float16 hh[4], mm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int vi = v[u];
          const float xs = __builtin_bit_cast(float, vi) * sx;
          hh[u] = (_Float16)xs;
          mm[u] = (_Float16)(xs - (float)hh[u]);
        }
        unsigned short *dst = Ps + (q & 1) * NP * PLANE + r * LDP + (ch % C4K) * 4;
        *reinterpret_cast<u32x2 *>(dst) = u32x2{__builtin_bit_cast(uint32_t, f16x2{hh[0], hh[1]}),
                                                __builtin_bit_cast(uint32_t, f16x2{hh[2], hh[3]})};
        *reinterpret_cast<u32x2 *>(dst + PLANE) = u32x2{__builtin_bit_cast(uint32_t, f16x2{mm[0], mm[1]}),
                                                        __builtin_bit_cast(uint32_t, f16x2{mm[2], mm[3]})};
      }
    }
  };
  // A operands (weights) of this wave's group in phase q: coalesced 16-byte loads per lane, straight
  // from L2 (a layer's 27 slices are at most 7 MB and shared by every workgroup)
  const int jb = slice * NCW + cw;
  constexpr int WREGS = BF3 ? NP * G32 : G;    // 16-byte operand registers per group
  struct WSet { uint4 v[WREGS]; };
  auto wstep = [&](int qq, int u, WSet &w) {   // weights of this wave's u-th group in phase qq (clamped)
    const int q = min(qq, NQ - 1);
    const int k = __builtin_amdgcn_readfirstlane(grp[GP * (q / PPT) + rb0 + u]) & 255;
    if constexpr (!BF3) {
      const uint4 *p = reinterpret_cast<const uint4 *>(a.w16) + (int64_t)jb * 64 + lane + (int64_t)(k * GT + (q % PPT) * G) * a.nb16 * 64;
#pragma unroll
      for (int g = 0; g < G; ++g) w.v[g] = p[(int64_t)g * a.nb16 * 64];
    } else {
      const uint4 *p = a.wb3 + (int64_t)jb * 64 + lane + (int64_t)(k * (CP / 32) + (q % PPT) * G32) * a.nb16 * 64;
#pragma unroll
      for (int g = 0; g < G32; ++g)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) w.v[NP * g + pc] = p[(int64_t)pc * a.piece_stride + (int64_t)g * a.nb16 * 64];
    }
  };

  f32x4 acc[GW];
  WSet w[GW];   // weights of this wave's groups; each set is re-requested for the NEXT phase right after its use
  // the wave's u-th 16-row group on its accumulator
  auto mfma_group = [&](int u, int buf) {
    const int rb = rb0 + u;
    if constexpr (!BF3) {
      const float *arow = &As[buf][lane & 15][4 * (lane >> 4)];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const f32x4 av = *reinterpret_cast<const f32x4 *>(arow + rb * 16 * LDA + g * 16);
        const f32x4 wv = __builtin_bit_cast(f32x4, w[u].v[g]);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c], av[c], acc[u], 0, 0, 0);
      }
    } else {
      const unsigned short *prow = Ps + buf * NP * PLANE + (rb * 16 + (lane & 15)) * LDP + 8 * (lane >> 4);
#pragma unroll
      for (int g = 0; g < G32; ++g) {
        const f16x8 ah = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(prow + 32 * g));
        const f16x8 am = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(prow + PLANE + 32 * g));
        const f16x8 wh = __builtin_bit_cast(f16x8, w[u].v[2 * g]), wm = __builtin_bit_cast(f16x8, w[u].v[2 * g + 1]);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm, ah, acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, am, acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah, acc[u], 0, 0, 0);
      }
    }
  };
  if (NQ > 0) {
    gather(0);
#pragma unroll
    for (int u = 0; u < GW; ++u) wstep(0, u, w[u]);
    land(0);
    gather(NQ > 1 ? 1 : 0);
  }
  __syncthreads();

  for (int q = 0; q < NQ; ++q) {
    const int t = q / PPT, h = q % PPT;
    // unconditional (clamped) requests keep the loop body free of branches around the memory operations: the
    // last iterations re-request the last phase and land it in the buffer nobody reads any more
    land(q + 1);                         // requested one phase ago
    gather(min(q + 2, NQ - 1));          // a whole phase to arrive
    if (PPT == 1 || h == 0) {
#pragma unroll
      for (int u = 0; u < GW; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // a group's weights were requested a whole phase ago (right after their previous use): the other groups'
    // MFMAs, the barrier and the next landing cover the L2 latency
#pragma unroll
    for (int u = 0; u < GW; ++u) {
      const int n_rows = __builtin_amdgcn_readfirstlane(grp[GP * t + rb0 + u]) >> 16;   // wave-uniform
      __builtin_amdgcn_sched_barrier(0);
      if (u == 0 || n_rows > 0) mfma_group(u, q & 1);
      wstep(q + 1, u, w[u]);
    }
    if (PPT == 1 || h == PPT - 1) {
      // tile done: lane = pair (lane & 15) of its group, channels 16 cw + 4 (lane >> 4) .. +3 of the slice
#pragma unroll
      for (int u = 0; u < GW; ++u) {
        const int info = grp[GP * t + rb0 + u];
        if ((lane & 15) < (info >> 16)) {
          const int e = ((info >> 8) & 255) + (lane & 15);
          float *p = &acc_s[out_loc[info & 255][e]][16 * cw + 4 * (lane >> 4)];
          f32x4 v = *reinterpret_cast<f32x4 *>(p);
          if constexpr (PM == 2) {
            const float f = __builtin_bit_cast(float, 0x7f000000u - __builtin_bit_cast(uint32_t, in_scale[info & 255][e])) * a.w_unscale;
            v += acc[u] * f;
          } else {
            v += acc[u];
          }
          *reinterpret_cast<f32x4 *>(p) = v;
        }
      }
    }
    __syncthreads();   // tile buffer q & 1 is free again; buffer (q + 1) & 1 is complete
  }
  DGR_OS_CLK(2);
#ifdef DGR_OS_STAGE_CLK
  if (threadIdx.x == 0) { atomicAdd(&dgr_os_stage_clk[4], 1ull); atomicAdd(&dgr_os_stage_clk[5], (unsigned long long)NQ); }
#endif
  // ---- 5. write the block's rows once (ReLU applied here when the tensor carries one: consumers that
  //         re-apply it see an idempotent max)
  //         ... and leave the rows' largest |x| behind for the split-operand consumers of this tensor (one integer
  //         atomicMax per row and channel slice; the CS / 4 lanes of a row are consecutive lanes of one wave)
  const float out_lo = a.out_relu ? 0.f : -__builtin_inff();
  static_assert(MB * (CS / 4) % THREADS == 0 || THREADS % (CS / 4) == 0, "row lanes stay inside a wave");
  for (int e0 = 0; e0 < MB * (CS / 4); e0 += THREADS) {
    const int e = e0 + tid;
    const bool live = e < MB * (CS / 4);
    const int r = live ? e / (CS / 4) : 0, c = (e % (CS / 4)) * 4;
    uint32_t mx = 0;
    if (live && row0 + r < n_out) {
      f32x4 v = *reinterpret_cast<const f32x4 *>(&acc_s[r][c]);
      v.x = fmaxf(v.x, out_lo); v.y = fmaxf(v.y, out_lo); v.z = fmaxf(v.z, out_lo); v.w = fmaxf(v.w, out_lo);
      *reinterpret_cast<f32x4 *>(a.out + (row0 + r) * a.out_ld + slice * CS + c) = v;
      const i32x4 b = __builtin_bit_cast(i32x4, v);
      mx = max(max((uint32_t)b.x & 0x7fffffffu, (uint32_t)b.y & 0x7fffffffu), max((uint32_t)b.z & 0x7fffffffu, (uint32_t)b.w & 0x7fffffffu));
    }
    if (a.out_amax || a.out_amax2) {   // (kernel-uniform)
#pragma unroll
      for (int d = CS / 8; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
      if (live && (e % (CS / 4)) == 0 && row0 + r < n_out) {
        if (a.out_amax) atomicMax(a.out_amax + row0 + r, mx);
        if (a.out_amax2) atomicMax(a.out_amax2 + row0 + r, mx);
      }
    }
  }
  DGR_OS_CLK(3);
}

template <int CP, int CS, int MB, int CK, int TM>
static int launch_os(const ConvOsArgs &ka, int64_t n_out_cap, hipStream_t stream) {
  // one group per wave where it measured faster (8 clouds of 27 k voxels: 32 -> 32 at level 0 147 -> 124 us,
  // 64 -> 64 at levels 0 / 1 108 -> 99 us); with Cin >= 128 (several phases per tile) or fewer rows per block the
  // extra phases of the padded group list cost more than the added waves hide (136 -> 178 us at 256 -> 64)
  constexpr int GW = (MB >= 64 && CP <= 64) ? 1 : TM / 16;
  int64_t blocks = dgr_ceil_div(n_out_cap, MB);
  blocks = (blocks + 7) / 8 * 8;
  dim3 grid((unsigned)blocks, (unsigned)(ka.cout / CS));
  constexpr int threads = CS * 4 * (TM / 16 / GW);
  if (ka.wb3 && ka.row_amax)
    sparse_conv_os<CP, CS, MB, CK, TM, 2, GW><<<grid, threads, 0, stream>>>(ka);
  else
    sparse_conv_os<CP, CS, MB, CK, TM, 0, GW><<<grid, threads, 0, stream>>>(ka);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

int dgr_conv_os_launch(const DgrConvOsLaunch &a, hipStream_t stream, const char **kernel_name) {
  if (a.dense) return dgr_conv_dense_launch(a, stream, kernel_name);
  if (a.up) return dgr_conv_up_launch(a, stream, kernel_name);
  DGR_REQUIRE(a.nbr && a.nbr->built && a.nbr->K == 27, "output-stationary conv: no neighbour table");
  DGR_REQUIRE((a.cin & 3) == 0 && (a.in_ld & 3) == 0 && (a.out_ld & 3) == 0 && (a.res == nullptr || (a.res_ld & 3) == 0),
              "output-stationary conv: channel counts and row strides must be multiples of 4");
  DGR_REQUIRE(a.cout % 32 == 0 && a.cin == a.cin_pad, "output-stationary conv: Cin = %d, Cout = %d must be multiples of 32",
              a.cin, a.cout);
  DGR_REQUIRE(a.n_in_cap > 0 && a.n_in_cap * (int64_t)a.in_ld < (1ll << 32) / 4 * 4,
              "output-stationary conv: input tensor beyond 32-bit element offsets");
  ConvOsArgs ka;
  static const bool os_f32 = getenv("DGR_EXACT_F32") != nullptr;
  ka.in = a.in; ka.out = a.out; ka.w16 = a.w16; ka.shift = a.shift; ka.res = a.res;
  ka.wb3 = os_f32 ? nullptr : static_cast<const uint4 *>(a.wb3);
  ka.piece_stride = a.piece_stride;
  ka.row_amax = ka.wb3 ? a.row_amax : nullptr;
  ka.w_unscale = a.w_unscale;
  ka.out_amax = a.out_amax; ka.out_amax2 = a.out_amax2;
  DGR_REQUIRE(!ka.wb3 || ka.row_amax, "output-stationary conv: the split weights need the input rows' maxima");
  ka.nbr = a.nbr->nbr; ka.n_out_dev = a.n_out_dev; ka.n_pad = a.nbr->n_pad;
  ka.in_ld = a.in_ld; ka.in_relu = a.in_relu; ka.out_ld = a.out_ld; ka.out_relu = a.out_relu;
  ka.res_ld = a.res_ld; ka.res_relu = a.res_relu;
  ka.cin = a.cin; ka.cout = a.cout; ka.nb16 = a.cout / 16;
#define DGR_OS_TM 32      // pair slots per tile (64: more idle group slots, 2.60 -> 2.82 ms FCGF conv time)
#define DGR_OS_MBBIG 64   // output rows per workgroup at the two finest levels
#define DGR_STR2(x) #x
#define DGR_STR(x) DGR_STR2(x)
#define DGR_OS(CPV, CSV, MBV, CKV)                                                                      \
  do {                                                                                                  \
    constexpr int tm = (MBV) < DGR_OS_TM ? ((MBV) < 32 ? 32 : (MBV)) : DGR_OS_TM;                       \
    if (kernel_name) *kernel_name = ka.row_amax ? "sparse_conv_os<" #CPV ", " #CSV ", " DGR_STR(MBV) ", " DGR_STR(CKV) ", f16x2>"  \
                                                 : "sparse_conv_os<" #CPV ", " #CSV ", " DGR_STR(MBV) ", " DGR_STR(CKV) ", f32>"; \
    return launch_os<CPV, CSV, MBV, CKV, tm>(ka, a.n_out_cap, stream);                                  \
  } while (0)
  // rows per workgroup by level: the coarse levels have few rows (1/3, 1/12, 1/60 of the input on
  // 3DMatch-shaped clouds) and need smaller blocks to fill 256 CUs
  const int mb = a.rows_per_block;
  const bool narrow = a.cout == 32;   // one 32-channel slice; wider layers: 64-channel slices over grid.y
#define DGR_OS_MB(CPV, CKV)                                                 \
  do {                                                                      \
    if (narrow) { DGR_OS(CPV, 32, DGR_OS_MBBIG, CKV); }                     \
    else if (mb == 64) { DGR_OS(CPV, 64, DGR_OS_MBBIG, CKV); }              \
    else if (mb == 32) { DGR_OS(CPV, 64, 32, CKV); }                        \
    else { DGR_OS(CPV, 64, 16, CKV); }                                      \
  } while (0)
  // input channels per pipeline phase: 64, or 128 where the caller asks for it (a.phase_channels; Cin >= 128: half the
  // phases -- a barrier and a chain of dependent LDS / L2 round trips each -- per tile)
  const bool ck128 = a.phase_channels == 128;
  switch (a.cin_pad) {
    case 32: DGR_OS_MB(32, 32);
    case 64: DGR_OS_MB(64, 64);
    case 128: if (ck128) DGR_OS_MB(128, 128); else DGR_OS_MB(128, 64);
    case 256: if (ck128) DGR_OS_MB(256, 128); else DGR_OS_MB(256, 64);
    default: break;
  }
#undef DGR_OS_MB
#undef DGR_OS
  dgr_set_error("output-stationary conv: no instantiation for Cin (padded) %d", a.cin_pad);
  return DGR_EINVAL;
}
