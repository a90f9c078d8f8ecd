// Dense-tile fused sparse convolution for the same-stride K = 27 layers of the 3-D (FCGF) net with C_in, C_out <= 64
// on gfx950.  Same role and the same arithmetic as the output-stationary kernel of conv_os.hip (the
// ME.MinkowskiConvolution forward of model/resunet.py:598-649 / model/residual_block.py:15-80 with the folded
// batch-norm shift, the residual and the pending ReLUs; per (output row, offset) one product row, added in ascending
// offset order), but WITHOUT pair lists:
//
//   * a wave owns RG x 16 consecutive output rows and ALL output channels.  Per offset it gathers the rows' neighbours
//     in a QUAD-COALESCED request layout (lane 4 r + c reads 16 bytes of row r: a quad of lanes = 64 contiguous bytes of
//     one row, one request of the vector-memory path; requesting in MFMA operand order, lane = 16 chunk + row, makes
//     every lane of a quad touch another row: 38 instead of 65 B/ns/CU when L2-resident), through bounds-checked buffer
//     loads (a missing neighbour is an offset past the tensor: zeros without a memory access, no branch), splits them
//     into the two f16 pieces in registers and brings them into the B-operand order of v_mfma_f32_16x16x32_f16 (lane =
//     row (lane & 15), 8-channel chunk (lane >> 4)) with ds_bpermute (the LDS crossbar, no LDS memory): no LDS tile, no
//     compaction, no slot lists;
//   * a missing neighbour is a zero operand: 27 dense tiles per row group.  On 3DMatch-shaped clouds 45 % (stride 1)
//     to 60 % (stride 2, 4) of the table is filled, so the matrix pipe does 1.7 .. 2.2 x the useful work -- on a pipe the
//     list-based kernel keeps 8 - 16 % busy, in exchange for its three dependent LDS look-ups per slot, its LDS
//     read-modify-write accumulation and its per-group weight re-loads (32 KB of weight fragments per 8 KB of
//     gathered rows through the CU's vector-memory path);
//   * the offset's weights (both pieces, <= 16 KB) stream from L2 through a register ring per wave, in this kernel's
//     operand order (pre-permuted at load time, net.hip): no LDS stage, no barrier in the offset loop;
//   * accumulators stay in registers for the whole layer: per offset a zero-initialised tile `tmp`, folded as
//     total += tmp * 2^-e(row, k) / weight scale -- the same two roundings per (row, offset) as conv_os.hip, in the
//     same ascending-k order on top of shift + residual.  Results do not depend on scheduling; they agree with the
//     list-based kernel to a few f32 ulps of the tensor's scale, NOT bitwise: the quad-coalesced gather hands the
//     channels of a 32-channel step to the matrix unit in another order (4 lq .. + 3 and 16 + 4 lq .. + 3 per lane),
//     so the f32 sums inside one MFMA round differently (tests/test_gpu_dense_conv.py: 2e-6 of the unit-norm output).
//   * input rows carry their largest |x| (bits, written by their producer's epilogue: out_amax below and in
//     conv_os.hip / conv.hip) -- the row's power-of-two scale is dgr_row_scale_of of it; no separate scale pass.
//
// What bounded it until round 4 (the round-3 reading "the gather rate of random rows" was
// wrong: row order and occupancy change nothing): the LDS traffic of the loop -- every wave re-read the offset's 16 KB of
// staged weights -- and the barrier per offset.  Now the weights stream from L2 per wave (below): +10 %, the same bits.
//
// Weight layout: the split pieces of conv_os.hip, WB[piece][k][s][jb][lane] = 8 halves =
// W_folded[k][32 s + 8 (lane >> 4) + e][16 jb + (lane & 15)], re-ordered per fragment group for the gather's channel
// order: lane (col, lq) holds half (lq & 1) of the natural fragments of lanes (col, lq >> 1) and (col, (lq >> 1) + 2) (net.hip).
#include "dgr_internal.h"
#include "split.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // (native vector: HIP's uint4 struct copies as memcpy and stays in scratch)

#ifndef DGR_DENSE_RG
#define DGR_DENSE_RG 2      // 16-row groups per wave with 64 input or output channels ...
#endif
#ifndef DGR_DENSE_RG32
#define DGR_DENSE_RG32 2    // ... and at 32 -> 32
#endif
#ifndef DGR_DENSE_WD
#define DGR_DENSE_WD 8      // weight ring: at most this many steps in flight
#endif
struct ConvDenseArgs {
  const float *in;
  float *out;
  const float *shift, *res;
  const u32x4 *wb;         // split weights: two f16 pieces, each [27][CIN/32][COUT/16][64] x 16 bytes
  int64_t piece_stride;    // 16-byte units per piece
  const int32_t *nbr, *n_out_dev;
  int64_t n_pad;
  int in_ld, in_relu, out_ld, out_relu, res_ld, res_relu;
  const uint32_t *row_amax;   // bits of every input row's largest |x| after the pending ReLU (from the row's producer)
  uint32_t *out_amax, *out_amax2;   // (nullable) the same for the rows written here: atomicMax per row
  float w_unscale;         // inverse of the layer's weight scale
  uint32_t in_bytes;       // size of the input tensor (row capacity x row stride): bound of the buffer loads
  // Rows as READY-MADE OPERANDS (round 6).  in_ds: the input tensor written by its producer as "dense split rows" -- per
  // row 4 CIN bytes = [h plane: CIN halves][m plane: CIN halves], the two f16 pieces of scale(row) * max(x, 0 if ReLU),
  // every 32-channel group in THIS kernel's gather order (16-byte chunk c = channels 32 s + 4 c .. + 3 and 32 s + 16 + 4 c
  // .. + 3): the offset loop then holds no conversion at all (it was 4/5 of the kernel's vector instructions, more issue
  // cycles than its MFMAs).  out_ds: the rows written here in that form for the next dense-tile layer (the middle tensor
  // of a residual block: `out` may then be null).  Same dgr_row_scale_of / dgr_split2 arithmetic on the same values as the
  // consumer-side split: bit-identical results (tests/test_gpu_dense_conv.py).
  const unsigned char *in_ds;
  unsigned char *out_ds;
};

// CIN, COUT in {32, 64}; RG = 16-row groups per wave; WAVES per workgroup; PS: the input comes as dense split rows
template <int CIN, int COUT, int RG, int WAVES, bool PS>
__global__ void __launch_bounds__(64 * WAVES, 2) sparse_conv_dense_f16x2(ConvDenseArgs a) {
  constexpr int KV = 27;
  constexpr int S = CIN / 32, NCB = COUT / 16;
  constexpr int THREADS = 64 * WAVES;
  constexpr int MB = WAVES * RG * 16;            // output rows per workgroup
  constexpr int NST = S * NCB;                   // (32-channel step, 16-column block) steps per offset
  constexpr int WD = NST >= 8 ? DGR_DENSE_WD : NST >= 4 ? 4 : 2;   // weight ring: the fragments of WD steps in flight per wave
  static_assert(NST % WD == 0, "the weight ring turns a whole number of times per offset (static register indexing)");
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

  // the offset's weights never touch LDS: every wave streams the fragments of its (offset, 32-channel step, 16-column
  // block) sequence -- contiguous in memory, pre-permuted at load time into this kernel's operand order (net.hip) --
  // through a register ring WD steps deep.  (Staged per workgroup and offset in LDS, as until round 4, they cost a
  // barrier per offset and two thirds of the kernel's LDS reads, and those reads, not the MFMAs, bounded it:
  // DESIGN.md section 8.)
  u32x4 rh[WD], rm[WD];
  const u32x4 *wph = a.wb + lane, *wpm = a.wb + a.piece_stride + lane;
  // (consumption order of an offset's NST fragments: column block outer, 32-channel step inner -- one accumulator tile
  // per row group alive at a time; in memory the fragments lie step-major)
  auto wreq = [&](int g, int i) {
    const int gc = min(g, KV * NST - 1);
    const int k = gc / NST, jj = gc - k * NST;
    const int gg = k * NST + (jj % S) * NCB + jj / S;
    rh[i] = wph[(int64_t)gg * 64];
    rm[i] = wpm[(int64_t)gg * 64];
  };
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

  // gathered rows of the NEXT offset: requested here, consumed (split into pieces) at the top of the next iteration
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      PS ? (void *)const_cast<unsigned char *>(a.in_ds) : (void *)const_cast<float *>(a.in), 0, a.in_bytes, 0x00020000);
  // the gathered rows of one offset (raw f32, requested DEPTH offsets ahead), their row scales and table entries
  struct RowSet { f32x4 raw[RG][S][2]; uint32_t mx[RG]; int nv[RG]; };
  // Request layout: lane L = 4 r + c reads 16 bytes of row r of the group so that every QUAD of lanes reads 64
  // contiguous bytes of one row -- one request of the vector-memory path per quad.  (Requesting straight in MFMA
  // operand layout, lane = 16 chunk + row, makes every lane of a quad touch a different row: four requests per quad,
  // measured 64 cycles per instruction and the kernel bounded by nothing else.)  The operand layout is restored after
  // the split by ds_bpermute (the LDS crossbar, no LDS memory).
  auto gather = [&](int k, RowSet &g) {
    // unconditional requests: no branches in the loop body, so that the requests stay where they are written
    // (conv_os.hip, same reason)
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const int n = nbr_s[k][(wave * RG + rg) * 16 + (lane >> 2)];
      const int ne = max(n, 0);
      // buffer loads: a missing neighbour gets an offset past the tensor -- the bounds check returns zeros without a
      // memory access, and the request stays branch-free
      const uint32_t off = n >= 0 ? (uint32_t)n * (uint32_t)((PS ? CIN : a.in_ld) * 4) + 16u * (lane & 3) : a.in_bytes;
      const uint32_t mv = a.row_amax[ne];
#pragma unroll
      for (int s = 0; s < S; ++s) {
        if constexpr (PS) {   // chunk (lane & 3) of the 32-channel group's h plane and of its m plane
          g.raw[rg][s][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off + 64 * s, 0, 0));
          g.raw[rg][s][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off + 2 * CIN + 64 * s, 0, 0));
        } else {
          g.raw[rg][s][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off + 128 * s, 0, 0));
          g.raw[rg][s][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off + 128 * s + 64, 0, 0));
        }
      }
      g.mx[rg] = mv;   // (turned into the scale, and selected against `nv`, when it is consumed: nothing here may wait for the request)
      g.nv[rg] = n;
    }
  };
  // offsets the row requests run ahead (one register set each).  Measured on the 195 k-row 64 -> 64 layers: depth 2
  // (230 VGPRs) 202 us, depth 1 (176 VGPRs) 205 us -- the gather is bounded by the memory system's rate for random
  // 256-byte rows beyond the L2 (tools/microbench/vmem_bw.hip), not by latency
  constexpr int DEPTH = 1;
  static_assert(DEPTH == 1 || DEPTH == 2, "prefetch depth");
  RowSet gA, gB;
  gather(0, gA);
  if (DEPTH == 2) gather(1, gB);
#pragma unroll
  for (int i = 0; i < WD; ++i) wreq(i, i);
  const int relu_lo = a.in_relu ? 0 : (int)0x80000000;   // pending ReLU of the producer as one integer max per value

  auto body = [&](int k, RowSet &g) {
    // ---- operands of this offset: s x = h + m, two f16 pieces (dgr_split2), in MFMA B layout
    f16x8 bh[RG][S], bm[RG][S];
    float fold[RG];
    // lane T = 16 lq + lr of the operand layout takes from lane 4 lr + lq of the request layout
    const int from = 4 * (4 * (lane & 15) + (lane >> 4));
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const float sx = g.nv[rg] >= 0 ? dgr_row_scale_of(g.mx[rg]) : 0.f;
      const float fl = sx != 0.f ? dgr_inv_pow2(sx) * a.w_unscale : 0.f;
      fold[rg] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(16 * (lane & 15), __builtin_bit_cast(int, fl)));
#pragma unroll
      for (int s = 0; s < S; ++s) {
        i32x4 hw, mw;   // four dwords = eight halves each: elements 0..3 from the first request, 4..7 from the second
        if constexpr (PS) {
          hw = __builtin_bit_cast(i32x4, g.raw[rg][s][0]);
          mw = __builtin_bit_cast(i32x4, g.raw[rg][s][1]);
        } else
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const i32x4 v = __builtin_bit_cast(i32x4, g.raw[rg][s][h]);
#pragma unroll
          for (int u = 0; u < 4; u += 2) {
            const f32x2 xs = f32x2{__builtin_bit_cast(float, max(v[u], relu_lo)), __builtin_bit_cast(float, max(v[u + 1], relu_lo))} * sx;
            const f16x2 hh = __builtin_convertvector(xs, f16x2);
            const f16x2 mm = __builtin_convertvector(xs - __builtin_convertvector(hh, f32x2), f16x2);
            hw[2 * h + u / 2] = __builtin_bit_cast(int, hh);
            mw[2 * h + u / 2] = __builtin_bit_cast(int, mm);
          }
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          hw[d] = __builtin_amdgcn_ds_bpermute(from, hw[d]);
          mw[d] = __builtin_amdgcn_ds_bpermute(from, mw[d]);
        }
        bh[rg][s] = __builtin_bit_cast(f16x8, hw);
        bm[rg][s] = __builtin_bit_cast(f16x8, mw);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // operands complete before the raw registers are re-requested (else the scheduler
                                         // keeps both generations alive and a register-pair false dependency makes the
                                         // conversions wait for the NEW requests)
    // ---- next offset's rows requested; they arrive during this offset's MFMAs (the weight ring is refilled step by
    //      step below: in the in-order memory counter every ring slot is older than the rows requested after it)
    gather(min(k + DEPTH, KV - 1), g);   // refills the set just consumed
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks these requests below the MFMAs: no time in flight)
    // ---- this offset's tiles, one 16-column block after the other: tmp = W[k]^T x (zero for missing neighbours), folded
    //      into the running total one block late -- nothing waits for the matrix pipe.  Per 32-channel step: its MFMAs,
    //      then the two requests that refill the ring slot they read (the same fragment of the NEXT offset: a whole
    //      offset in flight).  The scheduling barriers pin that order: left alone, the scheduler collects all sixteen
    //      requests of an offset at the top of the loop, a few hundred cycles before the first MFMA that needs them.
    f32x4 prev[RG];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      f32x4 tmp[RG];
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) tmp[rg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int j = cb * S + s;
        const f16x8 wh = __builtin_bit_cast(f16x8, rh[j % WD]);
        const f16x8 wm = __builtin_bit_cast(f16x8, rm[j % WD]);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) tmp[rg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm, bh[rg][s], tmp[rg], 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) tmp[rg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bm[rg][s], tmp[rg], 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) tmp[rg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh[rg][s], tmp[rg], 0, 0, 0);
        wreq(k * NST + j + WD, j % WD);   // the ring slot just consumed
        if (s == 0 && cb > 0) {
#pragma unroll
          for (int rg = 0; rg < RG; ++rg) total[rg][cb - 1] += prev[rg] * fold[rg];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int rg = 0; rg < RG; ++rg) prev[rg] = tmp[rg];
    }
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) total[rg][NCB - 1] += prev[rg] * fold[rg];
  };
  if (DEPTH == 1) {
#pragma unroll 1
    for (int k = 0; k < KV; ++k) body(k, gA);
  } else {
#pragma unroll 1
    for (int k = 0; k < KV - 1; k += 2) {
      body(k, gA);
      body(k + 1, gB);
    }
    body(KV - 1, gA);
  }

  // ---- the rows are written once (ReLU applied here when the tensor carries one)
  //      ... and their largest |x| is left behind for the split-operand consumers of this tensor: a row's channels sit
  //      in the four lanes lr + 16 lq of this wave
  const float out_lo = a.out_relu ? 0.f : -__builtin_inff();
  const int out_relu_lo = a.out_relu ? 0 : (int)0x80000000;
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    const int64_t row = row0 + (wave * RG + rg) * 16 + lr;
    uint32_t mx = 0;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      f32x4 v = total[rg][cb];
      v.x = fmaxf(v.x, out_lo); v.y = fmaxf(v.y, out_lo); v.z = fmaxf(v.z, out_lo); v.w = fmaxf(v.w, out_lo);
      total[rg][cb] = v;
      if (row < n_out) {
        if (a.out) *reinterpret_cast<f32x4 *>(a.out + row * a.out_ld + 16 * cb + 4 * lq) = v;
        const i32x4 b = __builtin_bit_cast(i32x4, v);
        mx = max(mx, max(max((uint32_t)b.x & 0x7fffffffu, (uint32_t)b.y & 0x7fffffffu), max((uint32_t)b.z & 0x7fffffffu, (uint32_t)b.w & 0x7fffffffu)));
      }
    }
    if (a.out_amax || a.out_amax2 || a.out_ds) {   // (kernel-uniform)
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, 16, 64));
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, 32, 64));
      if (lq == 0 && row < n_out) {
        if (a.out_amax) atomicMax(a.out_amax + row, mx);
        if (a.out_amax2) atomicMax(a.out_amax2 + row, mx);
      }
    }
    if (a.out_ds && row < n_out) {
      // the row once more as the next dense-tile layer's operands: exactly what that layer's own split makes of the f32
      // row (the row's scale from the same maximum, the pending ReLU as the same integer max, dgr_split2's roundings);
      // lane (lr, lq) holds chunk lq of every 32-channel group: channels 4 lq .. + 3 of column blocks 2 s and 2 s + 1
      const float sx = dgr_row_scale_of(mx);
      unsigned char *dst = a.out_ds + row * (int64_t)(4 * COUT) + 16 * lq;
#pragma unroll
      for (int s = 0; s < COUT / 32; ++s) {
        u32x4 hw, mw;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const i32x4 v = __builtin_bit_cast(i32x4, total[rg][2 * s + h]);
#pragma unroll
          for (int u = 0; u < 4; u += 2) {
            const f32x2 xs = f32x2{__builtin_bit_cast(float, max(v[u], out_relu_lo)), __builtin_bit_cast(float, max(v[u + 1], out_relu_lo))} * sx;
            const f16x2 hh = __builtin_convertvector(xs, f16x2);
            const f16x2 mm = __builtin_convertvector(xs - __builtin_convertvector(hh, f32x2), f16x2);
            hw[2 * h + u / 2] = __builtin_bit_cast(uint32_t, hh);
            mw[2 * h + u / 2] = __builtin_bit_cast(uint32_t, mm);
          }
        }
        *reinterpret_cast<u32x4 *>(dst + 64 * s) = hw;
        *reinterpret_cast<u32x4 *>(dst + 2 * COUT + 64 * s) = mw;
      }
    }
  }
}

bool dgr_conv_dense_supported(int cin, int cin_pad, int cout) {
  return cin == cin_pad && (cin == 32 || cin == 64) && (cout == 32 || cout == 64);
}

template <int CIN, int COUT>
static int launch_dense(const ConvDenseArgs &ka, int64_t n_out_cap, hipStream_t stream) {
  // (8 waves per workgroup / 16 rows per wave measured 2.56 / 2.50 ms FCGF conv time against 2.51 with this shape)
  constexpr int RG = (CIN == 32 && COUT == 32) ? DGR_DENSE_RG32 : DGR_DENSE_RG, WAVES = 4, MB = WAVES * RG * 16;
  int64_t blocks = dgr_ceil_div(n_out_cap, MB);
  blocks = (blocks + 7) / 8 * 8;
  if (ka.in_ds)
    sparse_conv_dense_f16x2<CIN, COUT, RG, WAVES, true><<<(unsigned)blocks, 64 * WAVES, 0, stream>>>(ka);
  else
    sparse_conv_dense_f16x2<CIN, COUT, RG, WAVES, false><<<(unsigned)blocks, 64 * WAVES, 0, stream>>>(ka);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

int dgr_conv_dense_launch(const DgrConvOsLaunch &a, hipStream_t stream, const char **kernel_name) {
  DGR_REQUIRE(a.nbr && a.nbr->built && a.nbr->K == 27, "dense-tile conv: no neighbour table");
  DGR_REQUIRE(dgr_conv_dense_supported(a.cin, a.cin_pad, a.cout), "dense-tile conv: Cin = %d, Cout = %d not built", a.cin, a.cout);
  DGR_REQUIRE(a.wbd && a.row_amax, "dense-tile conv: needs the split weights in its operand order and the input rows' maxima");
  DGR_REQUIRE((a.in_ld & 3) == 0 && (a.out_ld & 3) == 0 && (a.res == nullptr || (a.res_ld & 3) == 0),
              "dense-tile conv: row strides must be multiples of 4");
  ConvDenseArgs ka;
  ka.in = a.in; ka.out = a.out; ka.shift = a.shift; ka.res = a.res;
  ka.wb = static_cast<const u32x4 *>(a.wbd); ka.piece_stride = a.piece_stride;
  ka.nbr = a.nbr->nbr; ka.n_out_dev = a.n_out_dev; ka.n_pad = a.nbr->n_pad;
  ka.in_ld = a.in_ld; ka.in_relu = a.in_relu; ka.out_ld = a.out_ld; ka.out_relu = a.out_relu;
  ka.res_ld = a.res_ld; ka.res_relu = a.res_relu;
  ka.row_amax = a.row_amax; ka.w_unscale = a.w_unscale;
  ka.out_amax = a.out_amax; ka.out_amax2 = a.out_amax2;
  ka.in_ds = a.in_dsplit; ka.out_ds = a.out_dsplit;
  DGR_REQUIRE(a.out || a.out_dsplit, "dense-tile conv: no output");
  DGR_REQUIRE(a.n_in_cap > 0 && a.n_in_cap * (int64_t)a.in_ld * 4 < (1ll << 31), "dense-tile conv: input tensor beyond 2 GB");
  ka.in_bytes = (uint32_t)(a.n_in_cap * (int64_t)(a.in_dsplit ? a.cin : a.in_ld) * 4);
#define DGR_DENSE(CI, CO)                                                                   \
  if (a.cin == CI && a.cout == CO) {                                                        \
    if (kernel_name) *kernel_name = a.in_dsplit ? "sparse_conv_dense_f16x2<" #CI ", " #CO ", ps>" : "sparse_conv_dense_f16x2<" #CI ", " #CO ">"; \
    return launch_dense<CI, CO>(ka, a.n_out_cap, stream);                                   \
  }
  DGR_DENSE(32, 32) DGR_DENSE(32, 64) DGR_DENSE(64, 32) DGR_DENSE(64, 64)
#undef DGR_DENSE
  return DGR_EINVAL;
}
