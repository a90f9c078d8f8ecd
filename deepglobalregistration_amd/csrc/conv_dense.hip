// Dense-tile fused sparse convolution for the same-stride K = 27 layers of the 3-D (FCGF) net with C_in, C_out <= 64
// on gfx950.  Same role and the same arithmetic as the output-stationary kernel of conv_os.hip (the
// ME.MinkowskiConvolution forward of model/resunet.py:598-649 / model/residual_block.py:15-80 with the folded
// batch-norm shift, the residual and the pending ReLUs; per (output row, offset) one product row, added in ascending
// offset order), but WITHOUT pair lists:
//
//   * a wave owns RG x 16 consecutive output rows and ALL output channels; a lane owns one row of each 16-row group
//     (lane & 15) and the 8-channel chunk (lane >> 4) of every 32-channel k-step -- exactly the B operand of
//     v_mfma_f32_16x16x32_f16 -- so the gathered neighbour row goes from global memory straight into MFMA operand
//     registers (split into its two f16 pieces on the way): no LDS tile, no compaction, no slot lists;
//   * a missing neighbour is a zero operand: 27 dense tiles per row group.  On 3DMatch-shaped clouds 45 % (stride 1)
//     to 60 % (stride 2, 4) of the table is filled, so the matrix pipe does 1.7 .. 2.2 x the useful work -- on a pipe the
//     list-based kernel keeps 8 - 16 % busy, in exchange for its three dependent LDS look-ups per slot, its LDS
//     read-modify-write accumulation and its per-group weight re-loads (32 KB of weight fragments per 8 KB of
//     gathered rows through the CU's vector-memory path);
//   * the offset's weights (both pieces, <= 16 KB) are staged ONCE per workgroup and offset in LDS (double-buffered,
//     one barrier per offset) and read by every wave as MFMA A operands: each 16-byte fragment serves RG row groups;
//   * accumulators stay in registers for the whole layer: per offset a zero-initialised tile `tmp`, folded as
//     total += tmp * 2^-e(row, k) / weight scale -- the same two roundings per (row, offset) as conv_os.hip, in the
//     same ascending-k order on top of shift + residual: results are bit-identical to that kernel's and do not
//     depend on scheduling.
//
// Weight layout: the split pieces of conv_os.hip, WB[piece][k][s][jb][lane] = 8 halves =
// W_folded[k][32 s + 8 (lane >> 4) + e][16 jb + (lane & 15)] (net.hip).
#include "dgr_internal.h"
#include "split.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // (native vector: HIP's uint4 struct copies as memcpy and stays in scratch)

struct ConvDenseArgs {
  const float *in;
  float *out;
  const float *shift, *res;
  const u32x4 *wb;         // split weights: two f16 pieces, each [27][CIN/32][COUT/16][64] x 16 bytes
  int64_t piece_stride;    // 16-byte units per piece
  const int32_t *nbr, *n_out_dev;
  int64_t n_pad;
  int in_ld, in_relu, out_ld, out_relu, res_ld, res_relu;
  const float *row_scale;  // power-of-two scale per input row (dgr_row_scale)
  float w_unscale;         // inverse of the layer's weight scale
};

// CIN, COUT in {32, 64}; RG = 16-row groups per wave; WAVES per workgroup
template <int CIN, int COUT, int RG, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 2) sparse_conv_dense_f16x2(ConvDenseArgs a) {
  constexpr int KV = 27;
  constexpr int S = CIN / 32, NCB = COUT / 16;
  constexpr int THREADS = 64 * WAVES;
  constexpr int MB = WAVES * RG * 16;            // output rows per workgroup
  constexpr int WP = S * NCB * 64;               // 16-byte units per piece and offset
  constexpr int WU = 2 * WP;                     // ... per offset
  constexpr int WPT = (WU + THREADS - 1) / THREADS;
  static_assert(WU % THREADS == 0 || WU < THREADS, "weight staging shape");
  __shared__ u32x4 wlds[2][WU];
  __shared__ int nbr_s[KV][MB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_out = *a.n_out_dev;
  const int nblocks = (n_out + MB - 1) / MB;
  // XCD-aware block order: workgroup b runs on XCD b % 8; give every XCD a contiguous range of row blocks
  const int per = (nblocks + 7) >> 3;
  const int j = blockIdx.x >> 3;
  const int blk = (blockIdx.x & 7) * per + j;
  if (j >= per || blk >= nblocks) return;
  const int64_t row0 = (int64_t)blk * MB;

  // the offset's weights: global -> registers (one offset ahead) -> LDS buffer (k & 1)
  struct WRegs { u32x4 v[WPT]; };
  auto wload = [&](int k, WRegs &w) {
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
      const int c = min(tid + u * THREADS, WU - 1);
      const int p = c / WP, i = c - p * WP;
      w.v[u] = a.wb[(int64_t)p * a.piece_stride + (int64_t)k * WP + i];
    }
  };
  auto wstore = [&](int buf, const WRegs &w) {
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
      const int c = tid + u * THREADS;
      if (WU % THREADS == 0 || c < WU) wlds[buf][c] = w.v[u];
    }
  };
  WRegs wr;
  wload(0, wr);
  for (int e = tid; e < KV * MB; e += THREADS) {
    const int k = e / MB, r = e - k * MB;
    nbr_s[k][r] = (row0 + r < n_out) ? a.nbr[(int64_t)k * a.n_pad + row0 + r] : -1;
  }
  // accumulators start from the folded batch-norm shift (+ residual): lane = row (lane & 15) of each group,
  // channels 16 cb + 4 (lane >> 4) .. + 3
  const int lr = lane & 15, lq = lane >> 4;
  f32x4 total[RG][NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const f32x4 sh = a.shift ? *reinterpret_cast<const f32x4 *>(a.shift + 16 * cb + 4 * lq) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      f32x4 v = sh;
      const int64_t row = row0 + (wave * RG + rg) * 16 + lr;
      if (a.res && row < n_out) {
        f32x4 x = *reinterpret_cast<const f32x4 *>(a.res + row * a.res_ld + 16 * cb + 4 * lq);
        if (a.res_relu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
        v += x;
      }
      total[rg][cb] = v;
    }
  }
  __syncthreads();
  wstore(0, wr);
  wload(1, wr);

  // gathered rows of the NEXT offset: requested here, consumed (split into pieces) at the top of the next iteration
  f32x4 raw[RG][S][2];
  float sc[RG];
  int nv[RG];
  auto gather = [&](int k) {
    // unconditional requests (a missing neighbour reads row 0 under scale 0 -> a zero operand): no branches in the loop
    // body, so that the requests stay where they are written (conv_os.hip, same reason)
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const int n = nbr_s[k][(wave * RG + rg) * 16 + lr];
      const int ne = max(n, 0);
#ifdef DGR_DENSE_ABL_NOGATHER   // timing ablations (outputs are garbage): tools/ab_fcgf.py with DGR_HIP_LIB
      const float sv = 1.f;
#pragma unroll
      for (int s = 0; s < S; ++s) raw[rg][s][0] = raw[rg][s][1] = f32x4{(float)ne, 1.f, 2.f, 3.f};
#else
      const float *p = a.in + (int64_t)ne * a.in_ld + 8 * lq;
      const float sv = a.row_scale[ne];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        raw[rg][s][0] = *reinterpret_cast<const f32x4 *>(p + 32 * s);
        raw[rg][s][1] = *reinterpret_cast<const f32x4 *>(p + 32 * s + 4);
      }
#endif
      sc[rg] = sv;   // (selected against `nv` when it is consumed: nothing here may wait for the request)
      nv[rg] = n;
    }
  };
  gather(0);
  const int relu_lo = a.in_relu ? 0 : (int)0x80000000;   // pending ReLU of the producer as one integer max per value
  __syncthreads();

#pragma unroll 1
  for (int k = 0; k < KV; ++k) {
    // ---- operands of this offset: s x = h + m, two f16 pieces (dgr_split2), in MFMA B layout
    f16x8 bh[RG][S], bm[RG][S];
    float fold[RG];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const float sx = nv[rg] >= 0 ? sc[rg] : 0.f;
      fold[rg] = sx != 0.f ? dgr_inv_pow2(sx) * a.w_unscale : 0.f;
#pragma unroll
      for (int s = 0; s < S; ++s) {
#ifdef DGR_DENSE_ABL_NOCONV
        bh[rg][s] = __builtin_bit_cast(f16x8, raw[rg][s][0]);
        bm[rg][s] = __builtin_bit_cast(f16x8, raw[rg][s][1]);
        continue;
#endif
        float x[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const i32x4 v = __builtin_bit_cast(i32x4, raw[rg][s][h]);
#pragma unroll
          for (int u = 0; u < 4; ++u) x[4 * h + u] = __builtin_bit_cast(float, max(v[u], relu_lo));
        }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
          const f32x2 xs = f32x2{x[u], x[u + 1]} * sx;
          const f16x2 hh = __builtin_convertvector(xs, f16x2);
          const f16x2 mm = __builtin_convertvector(xs - __builtin_convertvector(hh, f32x2), f16x2);
          bh[rg][s][u] = hh[0]; bh[rg][s][u + 1] = hh[1];
          bm[rg][s][u] = mm[0]; bm[rg][s][u + 1] = mm[1];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // operands complete before the raw registers are re-requested (else the scheduler
                                         // keeps both generations alive and a register-pair false dependency makes the
                                         // conversions wait for the NEW requests)
    // ---- next offset: weights into the other LDS buffer (its readers finished before the last barrier), rows and
    //      the offset after next's weights requested; everything arrives during this offset's MFMAs
#ifndef DGR_DENSE_ABL_NOW
    wstore((k + 1) & 1, wr);   // (after the last offset: into the buffer nobody reads any more)
#endif
    gather(min(k + 1, KV - 1));
#ifndef DGR_DENSE_ABL_NOW
    wload(min(k + 2, KV - 1), wr);
#endif
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks these requests below the MFMAs: no time in flight)
    // ---- this offset's tile: tmp = W[k]^T x (zero for missing neighbours)
    f32x4 tmp[RG][NCB];
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) tmp[rg][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4 *wl = wlds[k & 1];
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const f16x8 wh = __builtin_bit_cast(f16x8, wl[(s * NCB + cb) * 64 + lane]);
        const f16x8 wm = __builtin_bit_cast(f16x8, wl[WP + (s * NCB + cb) * 64 + lane]);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) {
#ifdef DGR_DENSE_ABL_NOMFMA
          if (wh[0] != (_Float16)123.f) continue;
#endif
          tmp[rg][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm, bh[rg][s], tmp[rg][cb], 0, 0, 0);
          tmp[rg][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bm[rg][s], tmp[rg][cb], 0, 0, 0);
          tmp[rg][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh[rg][s], tmp[rg][cb], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int rg = 0; rg < RG; ++rg)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) total[rg][cb] += tmp[rg][cb] * fold[rg];
#ifndef DGR_DENSE_ABL_NOBAR
    __syncthreads();   // buffer k & 1 is free again; buffer (k + 1) & 1 is complete
#endif
  }

  // ---- the rows are written once (ReLU applied here when the tensor carries one)
  const float out_lo = a.out_relu ? 0.f : -__builtin_inff();
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    const int64_t row = row0 + (wave * RG + rg) * 16 + lr;
    if (row < n_out) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        f32x4 v = total[rg][cb];
        v.x = fmaxf(v.x, out_lo); v.y = fmaxf(v.y, out_lo); v.z = fmaxf(v.z, out_lo); v.w = fmaxf(v.w, out_lo);
        *reinterpret_cast<f32x4 *>(a.out + row * a.out_ld + 16 * cb + 4 * lq) = v;
      }
    }
  }
}

bool dgr_conv_dense_supported(int cin, int cin_pad, int cout) {
  return cin == cin_pad && (cin == 32 || cin == 64) && (cout == 32 || cout == 64);
}

template <int CIN, int COUT>
static int launch_dense(const ConvDenseArgs &ka, int64_t n_out_cap, hipStream_t stream) {
#ifndef DGR_DENSE_RG
#define DGR_DENSE_RG 2
#endif
#ifndef DGR_DENSE_WAVES
#define DGR_DENSE_WAVES 4
#endif
  constexpr int RG = DGR_DENSE_RG, WAVES = DGR_DENSE_WAVES, MB = WAVES * RG * 16;
  int64_t blocks = dgr_ceil_div(n_out_cap, MB);
  blocks = (blocks + 7) / 8 * 8;
  sparse_conv_dense_f16x2<CIN, COUT, RG, WAVES><<<(unsigned)blocks, 64 * WAVES, 0, stream>>>(ka);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

int dgr_conv_dense_launch(const DgrConvOsLaunch &a, hipStream_t stream, const char **kernel_name) {
  DGR_REQUIRE(a.nbr && a.nbr->built && a.nbr->K == 27, "dense-tile conv: no neighbour table");
  DGR_REQUIRE(dgr_conv_dense_supported(a.cin, a.cin_pad, a.cout), "dense-tile conv: Cin = %d, Cout = %d not built", a.cin, a.cout);
  DGR_REQUIRE(a.wb3 && a.row_scale, "dense-tile conv: needs the split weights and the input's row scales");
  DGR_REQUIRE((a.in_ld & 3) == 0 && (a.out_ld & 3) == 0 && (a.res == nullptr || (a.res_ld & 3) == 0),
              "dense-tile conv: row strides must be multiples of 4");
  ConvDenseArgs ka;
  ka.in = a.in; ka.out = a.out; ka.shift = a.shift; ka.res = a.res;
  ka.wb = static_cast<const u32x4 *>(a.wb3); ka.piece_stride = a.piece_stride;
  ka.nbr = a.nbr->nbr; ka.n_out_dev = a.n_out_dev; ka.n_pad = a.nbr->n_pad;
  ka.in_ld = a.in_ld; ka.in_relu = a.in_relu; ka.out_ld = a.out_ld; ka.out_relu = a.out_relu;
  ka.res_ld = a.res_ld; ka.res_relu = a.res_relu;
  ka.row_scale = a.row_scale; ka.w_unscale = a.w_unscale;
#define DGR_DENSE(CI, CO)                                                                   \
  if (a.cin == CI && a.cout == CO) {                                                        \
    if (kernel_name) *kernel_name = "sparse_conv_dense_f16x2<" #CI ", " #CO ">";            \
    return launch_dense<CI, CO>(ka, a.n_out_cap, stream);                                   \
  }
  DGR_DENSE(32, 32) DGR_DENSE(32, 64) DGR_DENSE(64, 32) DGR_DENSE(64, 64)
#undef DGR_DENSE
  return DGR_EINVAL;
}
