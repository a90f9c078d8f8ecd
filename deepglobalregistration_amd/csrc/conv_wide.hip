// Rule-major sparse convolution, phase 1, for the WIDE layers of the 6-D net (Cout >= 128, Cin in {64, 128, 256}: block3,
// block4, conv3/4, conv4_tr, block4_tr -- 95 % of all FLOPs) with f32-level results on the f16 matrix pipe.
// Replaces the per-rule gather -> GEMM -> scatter of MinkowskiConvolution(Transpose).forward for these layers
// (model/residual_block.py:31-38,56-72; model/resunet.py:461-566).
//
// Arithmetic.  gfx950 has no fast path for f32 matrix operands (v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate,
// 157 TFLOP/s; f16 MFMA at 2.5 PFLOP/s).  An f32 x scaled by a power of two s into f16 range splits as
//     s x = h + m + d,   h = rn16(s x),  m = rn16(s x - h),  |d| <= 2^-22 |s x|
// (22 of the 24 significant bits; every bit for most operands), and a product keeps  wh xh + wh xm + wm xh  (the dropped
// wm xm is <= 2^-22 of the product): THREE v_mfma_f32_32x32x16_f16 per MAC, accumulated in f32.  The scales are exact:
// one power of two per input ROW (its largest |x| lands in [2^14, 2^15): nothing overflows f16, and a channel 2^-17
// below its row's maximum still keeps all 22 bits) and one per layer for the weights (pre-split at load time, net.hip);
// the product row is multiplied by the two inverse powers of two on the way out.  tests/test_gpu_split_f64.py holds the
// kernel to an f64 reference next to the exact-f32 MFMA kernel.
//
// Input = SPLIT ROWS.  The split is a property of the input row, not of the (row, offset) pair: the layer that produces
// a tensor (reduce_rows, conv.hip) writes, next to the f32 row, the row scale and the two f16 planes with the consumer's
// pending ReLU applied -- the same 4 bytes per element -- in the layout this kernel gathers:
//     row r of a C-channel tensor = C / 64 blocks of 256 bytes: [h of channels 64 b .. 64 b + 63][m of the same]
// so that a (row, 64-channel phase) is ONE contiguous 256-byte piece.  (Round 2 gathered f32 rows and re-did the
// scale / convert / subtract for every pair, ~34 times per row: 227 M vector instructions per launch next to 43 M MFMAs,
// and the four gather waves of a workgroup were as busy as the matrix pipe.)
//
// Structure.  A tile is <= 64 pairs of ONE kernel offset (rule-major order, kmap.hip); a persistent 512-thread workgroup
// per CU walks an XCD-aware share of the tiles (block b runs on XCD b % 8; an XCD gets a contiguous range of offsets, so
// a weight slice lives in one L2).  Work is split by latency domain, because s_waitcnt vmcnt retires loads AND stores
// in order -- a wave that gathers, loads weights and stores waits for all three whenever it waits for one:
//   * waves 0..3 (compute): their only memory operations are weight fragments -- a register ring WD k-steps deep (2;
//     4 for the 64-channel-output shape, whose k-steps are three MFMAs long), NB x 2 16-byte loads per k-step,
//     unconditional, every wait count static -- then 3 MB NB v_mfma_f32_32x32x16_f16 per k-step on the landed planes
//     (the next k-step's operands are prefetched from LDS, across the phase barrier too).  Two accumulator sets
//     alternate: while tile t + 1 is multiplied, the finished tile t leaves its registers for an LDS stage, a few
//     16-byte pieces per k-step, in the shadow of the MFMAs.  Wave w computes rows 32 MB (w / WN) .. and output
//     channels 32 NB (w % WN) ..: WM = 1 (four column waves, 64 rows each) for C_out >= 128, WM = 2 (two row halves x
//     two column waves) for C_out = 64.
//   * waves 4, 5 (requesters): gather the split rows with plain 16-byte loads, NSET = 4 64-channel phases in flight
//     through four register sets, into a ring of NBUF = 3 LDS phase buffers; they also keep the index ring (eight tiles
//     ahead) and the scale ring.  (LDS-DMA gathers were measured here: ~200 cycles of issue per 1-KB piece, and one wave
//     mixing DMA with ordinary loads drains vmcnt(0).)
//   * waves 6, 7 (storers): the staged product rows to HBM as whole rows, non-temporal, RPH rows per phase so that the
//     store stream is even; scaled back by the row's and the layer's inverse powers of two on the way.
// One raw s_waitcnt lgkmcnt(0) + s_barrier per phase (all eight waves); 130 KB of LDS at C_out = 256.
// What bounds it: the three streams through the CU's vector-memory path -- 4 bytes of weights per
// MAC-column (L2), the gathered rows (beyond L2) and the product rows -- not the matrix pipe (MFMA busy 0.43).
//
// LDS phase buffer: [64 rows][16 slots of 16 bytes]; logical chunk c of a row (c = 8 piece + channel / 8) sits in slot
// c ^ (row & 15): a 16-lane group of ds_read_b128 (rows distinct mod 16, same chunk) touches 16 distinct slots of the
// 256-byte bank row -- conflict-free (measured: 0 bank conflicts) -- and a requester's load instruction (4 rows x 16
// slots) still reads whole 256-byte pieces.
#include <stdlib.h>

#include <type_traits>

#include "dgr_internal.h"
#include "split.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvWideArgs {
  const unsigned char *planes;   // split input rows, 4 CP bytes per row (layout above)
  const float *row_scale;        // power-of-two scale per input row
  float *y;                      // [pairs, cout] per-pair product rows
  const uint4 *wb;               // two pieces, each [K][CP/16][cout/32][64] uint4 (v_mfma_f32_32x32x16_f16 A order)
  int64_t piece_stride;          // uint4 per piece
  const int32_t *pair_in, *tile_ptr;
  int cout, K;
  float w_unscale;               // inverse of the layer's weight scale
  unsigned long long *clk;       // profiling only (else null): {earliest start, latest end} of the kernel's waves on the
                                 // 100-MHz wall clock -- the kernel's execution span as rocprofv3 --kernel-trace reports it,
                                 // measurable while other streams share the GPU (a HIP-event span then includes the wait
                                 // for compute units)
};

#define DGR_LDS_PTR(off) ((__attribute__((address_space(3))) void *)(lds + (off)))
#define DGR_GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

template <int N>
__device__ __forceinline__ void dgr_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void dgr_phase_barrier() {   // LDS traffic of this wave done, then the workgroup barrier
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// product-row store passes of phase h of a tile (of PPT): the tile's NPASS passes spread over its first NSP phases
__host__ __device__ constexpr int dgr_wide_passes(int h, int npass, int nsp) {
  return h < nsp ? npass / nsp + (h < npass % nsp ? 1 : 0) : 0;
}
__host__ __device__ constexpr int dgr_wide_pass_base(int h, int npass, int nsp) {
  int b = 0;
  for (int i = 0; i < h; ++i) b += dgr_wide_passes(i, npass, nsp);
  return b;
}

template <int CP, int NB, int WM>
__global__ void __launch_bounds__(512, 1) sparse_conv_wide_f16x2(ConvWideArgs a, const int4 *__restrict__ tdesc) {
  constexpr int NCW = 4;                 // compute waves
  constexpr int NP = 2, MB = 2 / WM, WN = NCW / WM, PW = 4, TM = 64, CK = 64, SK = CK / 16;
  static_assert(WM == 1 || WM == 2, "row split of the compute waves");
  static_assert(TM == DGR_TILE_M && CP % CK == 0, "shape");
  constexpr int PPT = CP / CK, S = CP / 16, NBLK = NB * WN, COUT = 32 * NBLK;
  static_assert(PPT == 1 || PPT == 2 || PPT == 4, "a round of NSET phases is a whole number of tiles");
  constexpr int ROWB = 4 * CP;           // bytes per split row
  constexpr int NBUF = 3;                // phase buffers in LDS: multiplied | complete, prefetchable | being landed
  constexpr int NSET = 4;                // register sets of a requesting wave = phases in flight from memory
  constexpr int LEAD = NSET + 2;         // a phase is requested LEAD phases before it is multiplied, landed 2 before
  constexpr int BUFB = TM * 256;         // bytes per phase buffer
  constexpr int RING = 16;               // tiles in the index / scale rings
  constexpr int AH_I = 8, AH_S = 4;      // tile t + AH_I's indices and tile t + AH_S's scales are requested at tile t
  static_assert(RING >= 2 * AH_I && AH_I > LEAD && AH_S >= 3 && AH_S < AH_I, "ring distances");
  // weight ring: k-step g + WD - 1 is requested at step g.  Two steps for the 12-MFMA k-steps of the C_out = 256 shapes
  // (four: same speed in tools/microbench/wide_check, five spilled registers); the C_out <= 128 shapes have six or three
  // MFMAs per k-step -- 200 / 100 cycles, a fraction of an L2 round trip -- so four steps of weights stay in flight
  constexpr int WD = NB >= 2 ? 2 : 4;
  static_assert(SK % WD == 0, "the weight ring turns a whole number of times per phase (static register indexing)");
  constexpr int LDS_ST = COUT + 4;       // stage row stride (floats)
  constexpr int NSTG = PPT == 1 ? 2 : 1; // one-phase tiles: a tile is staged while the previous one is still going out
  constexpr int HB = PPT / 2;            // phase of the next tile in which rows 32 .. 63 of a finished tile are staged
  constexpr int RPH = TM / PPT;          // product rows that leave per phase
  constexpr int PD = 2;                  // mover waves that request; the other PW - PD store
  constexpr int PTH = 64 * (PW - PD);    // storing threads
  constexpr int CPR = COUT / 4;          // 16-byte pieces per product row
  constexpr int RPP = PTH / CPR;         // rows per store pass of the storing threads
  constexpr int NPS = RPH / RPP;         // store passes per phase
  static_assert(RPH % RPP == 0, "store passes");
  constexpr int OFF_STAGE = NBUF * BUFB;
  constexpr int OFF_IDX = OFF_STAGE + NSTG * TM * LDS_ST * 4;
  constexpr int OFF_SC = OFF_IDX + RING * TM * 4;
  constexpr int LDS_BYTES = OFF_SC + RING * TM * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (a.clk && tid == 0) atomicMin(a.clk, (unsigned long long)wall_clock64());
  const int T = a.tile_ptr[a.K];
  const int per = (T + 7) >> 3;
  const int xcd = blockIdx.x & 7;
  const int t_end = min(T, (xcd + 1) * per);
  const int nj = gridDim.x >> 3;
  const int t_first = xcd * per + (blockIdx.x >> 3);
  if (t_first >= t_end) return;
  const int n_my = (t_end - t_first + nj - 1) / nj;  // tiles of this block: t_first + i * nj
  // The product rows of tile t leave during tiles t + 1 and t + 2 (below): every role walks n_my + 2 tiles of phases
  const int n_loop = n_my + 2;
  // (k, first pair, count) of the block's i-th tile, clamped to its last one; tdesc = the tile descriptors as a
  // restrict-qualified kernel argument, so that the uniform read is a scalar load (its own counter)
  auto desc = [&](int i) -> int4 { return tdesc[t_first + min(max(i, 0), n_my - 1) * nj]; };
  auto stage_of = [&](int t) -> float * {
    return reinterpret_cast<float *>(lds + OFF_STAGE) + (NSTG == 2 ? (t & 1) : 0) * TM * LDS_ST;
  };
  // The conveyor of a finished tile t (accumulators stay in their registers, the next tile uses the other set):
  //   staging (compute waves, between the MFMAs of tile t + 1): rows 0 .. 31 in phase 0, rows 32 .. 63 in phase HB;
  //   leaving (storing waves): rows [(h - 1) RPH, h RPH) in phase h = 1 .. PPT - 1 of tile t + 1, the last RPH rows in
  //   phase 0 of tile t + 2 -- RPH rows every phase, so the stores flow evenly (they, i.e. the HBM write rate, bound
  //   this kernel together with the other two streams).  PPT = 1: both halves are staged in the one phase of tile t + 1
  //   and leave in the one phase of tile t + 2, through two stages.

  if (wave >= NCW) {
    // ================================================================ movers
    // waves 4, 5 request, waves 6, 7 store: vmcnt retires in order, so a wave that did both would see its
    // loads "land" only when every older store has been acknowledged
    const int pw = wave - NCW;
    if (pw < PD) {
      // ------------------------------------------------------------ requesters
      // Plain loads into registers (NSET phases in flight), then 16-byte LDS writes: the lane-linear destination of a
      // load instruction (4 rows x 16 slots) carries the bank swizzle through the per-lane SOURCE chunk.
      // This wave's vmcnt queue holds nothing but its own unconditional loads: the compiler's counted waits are exact.
      constexpr int RW = TM / PD, NI = RW / 4;   // rows / load instructions per wave per phase
      f32x4 G[NSET][NI];
      const unsigned char *rb[NI];
      auto row_bases = [&](int i) {   // tile i: this lane's source rows, swizzled chunk included
        const int *idx = reinterpret_cast<const int *>(lds + OFF_IDX + (i & (RING - 1)) * TM * 4);
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) {
          const int r = RW * pw + 4 * jj + (lane >> 4);
          rb[jj] = a.planes + (int64_t)idx[r] * ROWB + 16 * ((lane & 15) ^ (r & 15));
        }
      };
      auto request = [&](int p, f32x4 *Gs) {   // phase p (tile p / PPT; past the last tile: the last tile again)
        if (p % PPT == 0) row_bases(p / PPT);
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) Gs[jj] = *reinterpret_cast<const f32x4 *>(rb[jj] + (p % PPT) * 256);
      };
      auto land = [&](int p, const f32x4 *Gs) {   // registers -> phase buffer p % NBUF
        unsigned char *dst = lds + (p % NBUF) * BUFB + (NI * pw) * 1024 + lane * 16;
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) *reinterpret_cast<f32x4 *>(dst + jj * 1024) = Gs[jj];
      };
      // index / scale rings: tile u's input rows are loaded at tile u - AH_I and published at tile u - AH_I + 1, its row
      // scales (through the published indices) loaded at tile u - AH_S and published at u - AH_S + 1 -- a load has a
      // whole tile to land.  Both requesting waves keep the (identical) rings, so each reads what it wrote itself; the
      // storing waves see the scales through the phase barriers.
      int *ring_i = reinterpret_cast<int *>(lds + OFF_IDX);
      float *ring_s = reinterpret_cast<float *>(lds + OFF_SC);
      auto load_idx = [&](int i) -> int {   // rows past the tile's end: its last pair again (a valid row)
        const int4 d = desc(i);
        return a.pair_in[d.y + min(lane, d.z - 1)];
      };
#pragma unroll
      for (int i = 0; i < AH_I - 1; ++i) ring_i[i * TM + lane] = load_idx(i);
#pragma unroll
      for (int i = 0; i < AH_S - 1; ++i) ring_s[i * TM + lane] = a.row_scale[ring_i[i * TM + lane]];
      int idx_reg = load_idx(AH_I - 1);
      float sc_reg = a.row_scale[ring_i[(AH_S - 1) * TM + lane]];
      auto rings = [&](int t) {   // first phase of tile t: publish tile t + AH - 1, request tile t + AH
        ring_i[((t + AH_I - 1) & (RING - 1)) * TM + lane] = idx_reg;
        ring_s[((t + AH_S - 1) & (RING - 1)) * TM + lane] = sc_reg;
        idx_reg = load_idx(t + AH_I);
        sc_reg = a.row_scale[ring_i[((t + AH_S) & (RING - 1)) * TM + lane]];
      };
      dgr_phase_barrier();   // P0
#pragma unroll
      for (int j = 0; j < NSET; ++j) request(j, G[j]);
      land(0, G[0]); request(NSET, G[0]);
      land(1, G[1]); request(NSET + 1, G[1]);
      dgr_phase_barrier();   // P1: phases 0 and 1 are in LDS
      // ---- phase q: land phase q + 2 (requested four phases ago), request phase q + LEAD into the freed set
      auto phase = [&](int q, f32x4 *Gs) {
        if (q % PPT == 0) rings(q / PPT);
        land(q + 2, Gs);
        request(q + LEAD, Gs);
        dgr_phase_barrier();
      };
      for (int q = 0; q < n_loop * PPT; q += NSET) {   // set of phase p = p % NSET: static register indexing
        phase(q, G[2]);
        if (q + 1 < n_loop * PPT) phase(q + 1, G[3]);
        if (q + 2 < n_loop * PPT) phase(q + 2, G[0]);
        if (q + 3 < n_loop * PPT) phase(q + 3, G[1]);
      }
      return;
    }
    // -------------------------------------------------------------- storers
    const int ptid = tid - 64 * (NCW + PD);
    // rows [r0, r0 + RPH) of finished tile u: LDS stage -> HBM, whole rows, scaled back.  Streaming stores: a product
    // row is read exactly once, by reduce_rows.
    auto store_rows = [&](int u, int r0) {
      if (u < 0 || u >= n_my) return;
      const int4 d = desc(u);
      const float *sc = reinterpret_cast<const float *>(lds + OFF_SC + (u & (RING - 1)) * TM * 4);
      const float *st = stage_of(u);
      const int c4 = (ptid % CPR) * 4;
      f32x4 v[NPS];
#pragma unroll
      for (int i = 0; i < NPS; ++i) {
        const int r = r0 + i * RPP + ptid / CPR;
        v[i] = *reinterpret_cast<const f32x4 *>(st + r * LDS_ST + c4);
        v[i] *= dgr_inv_pow2(sc[r]) * a.w_unscale;
      }
#pragma unroll
      for (int i = 0; i < NPS; ++i) {
        const int r = r0 + i * RPP + ptid / CPR;
        if (r < d.z) __builtin_nontemporal_store(v[i], reinterpret_cast<f32x4 *>(a.y + (int64_t)(d.y + r) * COUT + c4));
      }
    };
    dgr_phase_barrier();   // P0
    dgr_phase_barrier();   // P1
    auto sphase = [&](int t, auto hc) {
      constexpr int h = decltype(hc)::value;
      if constexpr (h == 0) store_rows(t - 2, (PPT - 1) * RPH);
      else store_rows(t - 1, (h - 1) * RPH);
      dgr_phase_barrier();
    };
    for (int t = 0; t < n_loop; ++t) {
      sphase(t, std::integral_constant<int, 0>());
      if constexpr (PPT > 1) sphase(t, std::integral_constant<int, 1>());
      if constexpr (PPT > 2) {
        sphase(t, std::integral_constant<int, 2>());
        sphase(t, std::integral_constant<int, 3>());
      }
    }
    if (a.clk && lane == 0) {   // the storing waves issue the kernel's last memory operations
      dgr_wait_vmcnt<0>();
      atomicMax(a.clk + 1, (unsigned long long)wall_clock64());
    }
    return;
  }

  // ================================================================== compute waves
  const int wn = wave % WN, wm = wave / WN;   // output-channel block / row half of this wave
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
  for (int g = 0; g < WD - 1; ++g) wload(dc.x, g % S, w[g]);
  dgr_phase_barrier();   // P0
  dgr_phase_barrier();   // P1
  // operands of k-step s out of phase buffer b: row = 32 (MB wm + i) + (lane & 31), chunk = 2 s + (lane >> 5) of piece h
  // (+ 8: m)
  uint4 op[2][MB][NP];
  // byte offset of this lane's h chunk of k-step 0 in its first row; k-step s: ^ (32 s) (chunk 2 s + (lane >> 5) =
  // 2 s ^ (lane >> 5)); the m chunk: ^ 128.  (The row term is a multiple of 256: it does not meet the xor bits.)
  const int lofs0 = (32 * MB * wm + (lane & 31)) * 256 + 16 * ((lane >> 5) ^ (lane & 15));
  auto oload = [&](int b, int s, uint4 (*o)[NP]) {
    const unsigned char *p = lds + b * BUFB + (lofs0 ^ (32 * s));
    const unsigned char *pm = lds + b * BUFB + (lofs0 ^ (32 * s) ^ 128);
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      o[i][0] = *reinterpret_cast<const uint4 *>(p + i * 32 * 256);
      o[i][1] = *reinterpret_cast<const uint4 *>(pm + i * 32 * 256);
    }
  };
  oload(0, 0, op[0]);
  int buf = 0;
  // raw accumulators of this wave's row group i -> LDS stage, NPC 16-byte pieces per k-step (of the group's 4 NB):
  // D column (pair) = lane & 31, D row (channel) = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  constexpr int NPC = (4 * NB + SK - 1) / SK;
  auto stage_pieces = [&](float *st, const f32x16 (&ac)[MB][NB], int i, int s) {
    float *dst = st + (32 * (MB * wm + i) + (lane & 31)) * LDS_ST;
#pragma unroll
    for (int u = 0; u < NPC; ++u) {
      const int wq = s * NPC + u, j = wq / 4, g = wq % 4;
      if (wq < 4 * NB) {
        const int col = 32 * (wn * NB + j) + 8 * g + 4 * (lane >> 5);
        *reinterpret_cast<f32x4 *>(dst + col) = f32x4{ac[i][j][4 * g], ac[i][j][4 * g + 1], ac[i][j][4 * g + 2], ac[i][j][4 * g + 3]};
      }
    }
  };
  // One tile of phases: multiply tile t into `ac` (if it exists), stage the finished tile t - 1 out of `ad` on the way
  auto ctile = [&](int t, f32x16 (&ac)[MB][NB], const f32x16 (&ad)[MB][NB]) {
    // (the two tiles past the last one multiply stale buffers into accumulators nobody stages: an `if` around the
    // weight loads would cost the exact wait counts of every tile)
    const bool stage_prev = t >= 1 && t <= n_my;
    float *st = stage_of(t - 1);
#pragma unroll
    for (int h = 0; h < PPT; ++h) {
      if (h == 0) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) ac[i][j][e] = 0.f;
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
        // pin the prefetches ahead of the MFMA block (left to the compiler, or interleaved one per MFMA gap with
        // sched_group_barrier, the loads are waited for early: 2.36 / 2.19 ms against 1.95 ms, tools/microbench/wide_check)
        __builtin_amdgcn_sched_barrier(0);
        uint4 (*wc)[NP] = w[s % WD];
        uint4 (*oc)[NP] = op[s & 1];
#define DGR_WIDE_TERM(WP, OP)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < MB; ++i) _Pragma("unroll") for (int j = 0; j < NB; ++j)                       \
      ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wc[j][WP]),                          \
                                                        __builtin_bit_cast(f16x8, oc[i][OP]), ac[i][j], 0, 0, 0);
        DGR_WIDE_TERM(1, 0)   // wm . xh
        DGR_WIDE_TERM(0, 1)   // wh . xm
        DGR_WIDE_TERM(0, 0)   // wh . xh
#undef DGR_WIDE_TERM
        // the finished tile's accumulators leave for the stage a few pieces per k-step, in the shadow of the MFMAs
        // (MB = 2: row group 0 in phase 0, row group 1 in phase HB; MB = 1: the wave's one row group in phase 0)
        if (stage_prev) {
          if (h == 0) stage_pieces(st, ad, 0, s);
          if constexpr (MB > 1) {
            if (h == HB) stage_pieces(st, ad, 1, s);
          }
        }
      }
      if (h == PPT - 1) {
        dc = dn;
        dn = desc(t + 2);
      }
      buf = nbuf;
      dgr_phase_barrier();
    }
  };
  f32x16 accA[MB][NB], accB[MB][NB];
  for (int t = 0; t < n_loop; t += 2) {   // the two accumulator sets alternate statically
    ctile(t, accA, accB);
    if (t + 1 < n_loop) ctile(t + 1, accB, accA);
  }
  if (a.clk && lane == 0) atomicMax(a.clk + 1, (unsigned long long)wall_clock64());
}

template <int CP, int NB, int WM>
static int launch_wide(const ConvWideArgs &ka, const int4 *tile_desc, int64_t tile_bound, int num_cus, hipStream_t stream) {
  int64_t grid = num_cus;
  if (tile_bound < grid) grid = tile_bound;
  grid = (grid + 7) / 8 * 8;
  if (grid < 8) grid = 8;
  sparse_conv_wide_f16x2<CP, NB, WM><<<(int)grid, 512, 0, stream>>>(ka, tile_desc);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

bool dgr_conv_wide_supported(int cin_pad, int cin, int cout) {
  if (cin != cin_pad) return false;
  return (cout == 64 && cin == 64) || (cout == 128 && (cin == 64 || cin == 128 || cin == 256)) ||
         (cout == 256 && (cin == 128 || cin == 256));
}

int dgr_conv_wide_launch(const DgrConvLaunch &a, const DgrSplitRows &in, const void *wb, int64_t piece_stride,
                         float w_unscale, int num_cus, hipStream_t stream, const char **kernel_name) {
  DGR_REQUIRE(a.pair_in && wb && in.planes && in.scale && dgr_conv_wide_supported(a.cin_pad, a.cin, a.cout),
              "wide conv: unsupported layer (Cin %d, Cout %d) or no split input rows", a.cin, a.cout);
  DGR_REQUIRE(in.channels == a.cin, "wide conv: the split rows have %d channels, the layer %d", in.channels, a.cin);
  DGR_REQUIRE(a.y, "wide conv: no product-row buffer");
  ConvWideArgs ka;
  ka.planes = in.planes; ka.row_scale = in.scale; ka.y = a.y;
  ka.wb = static_cast<const uint4 *>(wb); ka.piece_stride = piece_stride;
  ka.pair_in = a.pair_in; ka.tile_ptr = a.tile_ptr; ka.cout = a.cout; ka.K = a.K; ka.w_unscale = w_unscale;
  ka.clk = a.clk;
  const int64_t tile_bound = a.tile_bound > 0 ? a.tile_bound : (int64_t)num_cus * 4;
#define DGR_WIDE(CPV, NBV, WMV)                                                                 \
  do {                                                                                          \
    if (kernel_name) *kernel_name = "sparse_conv_wide_f16x2<" #CPV ", " #NBV ", " #WMV ">";     \
    return launch_wide<CPV, NBV, WMV>(ka, a.tile_desc, tile_bound, num_cus, stream);            \
  } while (0)
  if (a.cout == 64 && a.cin == 64) DGR_WIDE(64, 1, 2);
  if (a.cout == 128 && a.cin == 64) DGR_WIDE(64, 1, 1);
  if (a.cout == 128 && a.cin == 128) DGR_WIDE(128, 1, 1);
  if (a.cout == 128 && a.cin == 256) DGR_WIDE(256, 1, 1);
  if (a.cout == 256 && a.cin == 128) DGR_WIDE(128, 2, 1);
  if (a.cout == 256 && a.cin == 256) DGR_WIDE(256, 2, 1);
#undef DGR_WIDE
  dgr_set_error("wide conv: no kernel for Cin %d, Cout %d", a.cin, a.cout);
  return DGR_EINVAL;
}

// ---------------------------------------------------------------------------------------------------------------
// Row maxima / split rows of an EXISTING f32 tensor.  On the network path the producers leave both behind themselves
// (reduce_rows in conv.hip; the epilogues of conv_os.hip / conv_dense.hip / the conv1 kernels); these serve the
// single-layer debug entry point and inputs that did not come out of one of those kernels.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// LPR = lanes per row (16 bytes each; wider rows loop): largest |x| (after the pending ReLU) -> the power of two that
// moves it into [2^14, 2^15); SPLIT: also the row's two f16 planes.  A wave covers 64 / LPR rows at a time.
template <int LPR, bool SPLIT>
__global__ void __launch_bounds__(256) row_scale_kernel(const float *__restrict__ in, int in_ld, int cin, int relu,
                                                        const int32_t *__restrict__ n_dev, float *__restrict__ out,
                                                        unsigned char *__restrict__ planes, uint32_t *__restrict__ amax_out) {
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
        for (int u = 0; u < 4; ++u) mx = max(mx, (uint32_t)max(v[u], relu_lo) & 0x7fffffffu);   // |x| as an integer
      }
    }
#pragma unroll
    for (int d = LPR / 2; d >= 1; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    const float sx = dgr_row_scale_of(mx);
    if (l == 0 && r < n) {
      if (out) out[r] = sx;
      if (amax_out) amax_out[r] = mx;
    }
    if constexpr (SPLIT) {
      if (r < n) {
        const float *row = in + r * in_ld;
        unsigned char *dst = planes + r * 4 * cin;
        for (int c = l * 4; c < cin; c += LPR * 4) {
          const i32x4 v = __builtin_bit_cast(i32x4, *reinterpret_cast<const f32x4 *>(row + c));
          f16x4 h, m;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            _Float16 hh, mm;
            dgr_split2(__builtin_bit_cast(float, max(v[u], relu_lo)), sx, hh, mm);   // pending ReLU as an integer max
            h[u] = hh; m[u] = mm;
          }
          *reinterpret_cast<f16x4 *>(dst + dgr_split_row_offset(c, 0)) = h;
          *reinterpret_cast<f16x4 *>(dst + dgr_split_row_offset(c, 1)) = m;
        }
      }
    }
  }
}

static int row_scale_launch(const float *in, int in_ld, int cin, int relu, const int32_t *n_dev, int64_t n_cap, float *out,
                            unsigned char *planes, uint32_t *amax_out, hipStream_t stream) {
  DGR_REQUIRE((cin & 3) == 0 && (in_ld & 3) == 0 && cin >= 4, "row scale: channel count and row stride must be multiples of 4");
  const int lpr = cin >= 256 ? 64 : cin >= 128 ? 32 : cin >= 64 ? 16 : cin >= 32 ? 8 : 4;   // power of two: shuffle tree
  int64_t grid = dgr_ceil_div(n_cap, 4 * (64 / lpr));
  if (grid > 4096) grid = 4096;
  if (grid < 1) grid = 1;
#define DGR_RS(L)                                                                                                   \
  do {                                                                                                              \
    if (planes) row_scale_kernel<L, true><<<(int)grid, 256, 0, stream>>>(in, in_ld, cin, relu, n_dev, out, planes, amax_out); \
    else row_scale_kernel<L, false><<<(int)grid, 256, 0, stream>>>(in, in_ld, cin, relu, n_dev, out, nullptr, amax_out);      \
  } while (0)
  switch (lpr) {
    case 64: DGR_RS(64); break;
    case 32: DGR_RS(32); break;
    case 16: DGR_RS(16); break;
    case 8: DGR_RS(8); break;
    default: DGR_RS(4); break;
  }
#undef DGR_RS
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

int dgr_row_amax(const float *in, int in_ld, int cin, int relu, const int32_t *n_dev, int64_t n_cap, uint32_t *out,
                 hipStream_t stream) {
  return row_scale_launch(in, in_ld, cin, relu, n_dev, n_cap, nullptr, nullptr, out, stream);
}

int dgr_split_rows(const float *in, int in_ld, int relu, const int32_t *n_dev, int64_t n_cap, const DgrSplitRows &out,
                   hipStream_t stream) {
  DGR_REQUIRE(out.planes && out.scale && out.channels % 64 == 0, "split rows: need planes, scales and a multiple of 64 channels");
  return row_scale_launch(in, in_ld, out.channels, relu, n_dev, n_cap, out.scale, out.planes, nullptr, stream);
}
