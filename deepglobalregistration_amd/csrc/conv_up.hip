// Transposed 3-D convolutions (kernel 3, stride 2: conv4_tr / conv3_tr / conv2_tr of ResUNetBN2C, model/resunet.py:620-640,
// ME.MinkowskiConvolutionTranspose through model/residual_block.py:47-80) on gfx950, by PARITY CLASS of the output rows.
//
// A fine output row f meets the coarse input row c under offset delta iff c = f - delta ts lies on the coarse lattice
// (SURVEY.md A6): component d of delta is 0 where coordinate d of f is an even multiple of the fine stride ts and +-1
// where it is odd.  So a row can use only 2^(odd dims) of the 27 offsets -- 3.4 on average, the neighbour table of a
// transposed conv is 11 % filled -- and WHICH offsets is decided by the row's parity class alone.  The list-based
// output-stationary kernel (conv_os.hip) walks all 27 offsets of every 64-row block for 16-slot groups that are
// 7 / 16 full, one barrier-separated phase chain per group (1.5 - 2.5 us per phase whatever it holds:
// profiles/r06_os_stage_clk.txt); here
//
//   * the neighbour search (kmap.hip, nbr_search3) leaves the output rows grouped by class (DgrNbrTable::perm);
//   * a workgroup takes MB rows of ONE class and walks only that class's offsets, in ascending k (the summation order
//     of every other kernel of this layer family), as DENSE tiles: the dense-tile kernel's structure (conv_dense.hip:
//     quad-coalesced gather of the neighbours' f32 rows through bounds-checked buffer loads, split into the two f16
//     pieces in registers, ds_bpermute into MFMA operand order, weights through a per-wave register ring straight from
//     L2, accumulators in registers, no LDS tile, no lists, no barrier in the loop) with the input channels taken 64 at
//     a time (Cin = 128 | 256) and the output channels in slices of 64 (blockIdx.y);
//   * per (row, offset) the same arithmetic as conv_dense.hip / conv_os.hip: tmp = sum over the 32-channel steps of
//     w_m x_h + w_h x_m + w_h x_h (f32 accumulate in the MFMA), total += tmp * 2^-e(row) / weight scale, on top of the
//     folded batch-norm shift.  Results agree with the list-based kernel to a few f32 ulps of the tensor's scale (the
//     dense gather's channel order inside a 32-channel step: conv_dense.hip), tests/test_gpu_dense_conv.py.
//
// Weights: the layer's split pieces in the dense-tile kernel's operand order (net.hip, w16d),
// [piece][k][Cin / 32][Cout / 16][lane] x 16 bytes.
#include "dgr_internal.h"
#include "split.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ConvUpArgs {
  const float *in;
  float *out;
  const float *shift;
  const u32x4 *wb;
  int64_t piece_stride;      // 16-byte units per piece
  const int32_t *nbr;
  int64_t n_pad;
  const int32_t *perm, *cls_count;
  int64_t cls_cap;
  int in_ld, in_relu, out_ld, out_relu;
  int nb16;                  // Cout / 16 of the whole layer
  const uint32_t *row_amax;
  uint32_t *out_amax, *out_amax2;
  float w_unscale;
  uint32_t in_bytes;
};

// CIN in {128, 256}; a workgroup writes a 64-channel slice of MB = WAVES x RG x 16 rows of one parity class
template <int CIN, int RG, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 2) sparse_conv_up_f16x2(ConvUpArgs a) {
  constexpr int NCB = 4;                 // 16-column blocks of the slice
  constexpr int NCH = CIN / 64;          // 64-channel chunks of an input row
  constexpr int S_T = CIN / 32;          // 32-channel steps of the whole row (weight layout)
  constexpr int THREADS = 64 * WAVES;
  constexpr int MB = WAVES * RG * 16;
  constexpr int NST = 2 * NCB;           // (32-channel step, 16-column block) steps per (offset, chunk)
  constexpr int WD = NST;                // weight ring: one (offset, chunk) in flight per wave
  __shared__ int nbr_s[8][MB];
  __shared__ int rows_s[MB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slice = blockIdx.y;
  // ---- which class, which of its row blocks (classes one after the other, ceil(rows / MB) workgroups each)
  int cls = -1, lb = 0, n_cls = 0;
  {
    int b = blockIdx.x;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int n = a.cls_count[c];
      const int nb = (n + MB - 1) / MB;
      if (cls < 0 && b < nb) { cls = c; lb = b; n_cls = n; }
      b -= nb;
    }
  }
  if (cls < 0) return;
  // ---- the class's offsets in ascending k = dx + 3 dy + 9 dz (+ 13): component d is +-1 where bit d is set, else 0;
  //      five bits each in one 64-bit word (dynamic indexing without LDS or scratch)
  unsigned long long kpack = 0;
  int NL = 0;
  for (int k = 0; k < 27; ++k) {
    const int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
    if ((dx != 0) == ((cls & 1) != 0) && (dy != 0) == ((cls & 2) != 0) && (dz != 0) == ((cls & 4) != 0)) {
      kpack |= (unsigned long long)k << (5 * NL);
      ++NL;
    }
  }
  auto k_of = [&](int li) { return (int)((kpack >> (5 * li)) & 31ull); };
  const int NIT = NL * NCH;

  for (int i = tid; i < MB; i += THREADS) {
    const int64_t idx = (int64_t)lb * MB + i;
    rows_s[i] = idx < n_cls ? a.perm[(int64_t)cls * a.cls_cap + idx] : -1;
  }
  __syncthreads();
  for (int e = tid; e < NL * MB; e += THREADS) {
    const int li = e / MB, i = e - li * MB;
    const int r = rows_s[i];
    nbr_s[li][i] = r >= 0 ? a.nbr[(int64_t)k_of(li) * a.n_pad + r] : -1;
  }

  u32x4 rh[WD], rm[WD];
  const u32x4 *wph = a.wb + lane, *wpm = a.wb + a.piece_stride + lane;
  // ring position g = it * NST + jj; consumption order inside an iteration: column block outer, step inner
  auto wreq = [&](int g, int i) {
    const int gc = min(g, NIT * NST - 1);
    const int it = gc / NST, jj = gc - it * NST;
    const int li = it / NCH, ch = it - li * NCH;
    const int64_t frag = ((int64_t)k_of(li) * S_T + 2 * ch + (jj & 1)) * a.nb16 + slice * NCB + (jj >> 1);
    rh[i] = wph[frag * 64];
    rm[i] = wpm[frag * 64];
  };
  const int lr = lane & 15, lq = lane >> 4;
  f32x4 total[RG][NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const f32x4 sh = a.shift ? *reinterpret_cast<const f32x4 *>(a.shift + 64 * slice + 16 * cb + 4 * lq) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) total[rg][cb] = sh;
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in), 0, a.in_bytes, 0x00020000);
  struct RowSet { f32x4 raw[RG][2][2]; uint32_t mx[RG]; int nv[RG]; };
  // quad-coalesced request layout (conv_dense.hip): lane 4 r + c reads 16 bytes of row r
  auto gather = [&](int it, RowSet &g) {
    const int li = it / NCH, ch = it - li * NCH;
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const int n = nbr_s[li][(wave * RG + rg) * 16 + (lane >> 2)];
      const int ne = max(n, 0);
      const uint32_t off = n >= 0 ? (uint32_t)n * (uint32_t)(a.in_ld * 4) + 256u * ch + 16u * (lane & 3) : a.in_bytes;
      const uint32_t mv = a.row_amax[ne];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        g.raw[rg][s][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off + 128 * s, 0, 0));
        g.raw[rg][s][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off + 128 * s + 64, 0, 0));
      }
      g.mx[rg] = mv;
      g.nv[rg] = n;
    }
  };
  RowSet gA;
  gather(0, gA);
#pragma unroll
  for (int i = 0; i < WD; ++i) wreq(i, i);
  const int relu_lo = a.in_relu ? 0 : (int)0x80000000;

  f32x4 tmp[RG][NCB];
#pragma unroll 1
  for (int it = 0; it < NIT; ++it) {
    const int ch = it % NCH;
    // ---- operands of this (offset, chunk): s x = h + m, two f16 pieces (dgr_split2), in MFMA B layout
    f16x8 bh[RG][2], bm[RG][2];
    float fold[RG];
    const int from = 4 * (4 * (lane & 15) + (lane >> 4));
#pragma unroll
    for (int rg = 0; rg < RG; ++rg) {
      const float sx = gA.nv[rg] >= 0 ? dgr_row_scale_of(gA.mx[rg]) : 0.f;
      const float fl = sx != 0.f ? dgr_inv_pow2(sx) * a.w_unscale : 0.f;
      fold[rg] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(16 * (lane & 15), __builtin_bit_cast(int, fl)));
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        i32x4 hw, mw;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const i32x4 v = __builtin_bit_cast(i32x4, gA.raw[rg][s][h]);
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
    __builtin_amdgcn_sched_barrier(0);
    gather(min(it + 1, NIT - 1), gA);
    __builtin_amdgcn_sched_barrier(0);
    if (ch == 0) {
#pragma unroll
      for (int rg = 0; rg < RG; ++rg)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) tmp[rg][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int j = cb * 2 + s;
        const f16x8 wh = __builtin_bit_cast(f16x8, rh[j]);
        const f16x8 wm = __builtin_bit_cast(f16x8, rm[j]);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) tmp[rg][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm, bh[rg][s], tmp[rg][cb], 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) tmp[rg][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bm[rg][s], tmp[rg][cb], 0, 0, 0);
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) tmp[rg][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, bh[rg][s], tmp[rg][cb], 0, 0, 0);
        wreq((it + 1) * NST + j, j);   // the ring slot just consumed: the same step of the next (offset, chunk)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (ch == NCH - 1) {
#pragma unroll
      for (int rg = 0; rg < RG; ++rg)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) total[rg][cb] += tmp[rg][cb] * fold[rg];
    }
  }

  // ---- the rows are written once, through the class's row list
  const float out_lo = a.out_relu ? 0.f : -__builtin_inff();
#pragma unroll
  for (int rg = 0; rg < RG; ++rg) {
    const int64_t row = rows_s[(wave * RG + rg) * 16 + lr];
    uint32_t mx = 0;
    if (row >= 0) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        f32x4 v = total[rg][cb];
        v.x = fmaxf(v.x, out_lo); v.y = fmaxf(v.y, out_lo); v.z = fmaxf(v.z, out_lo); v.w = fmaxf(v.w, out_lo);
        *reinterpret_cast<f32x4 *>(a.out + row * a.out_ld + 64 * slice + 16 * cb + 4 * lq) = v;
        const i32x4 b = __builtin_bit_cast(i32x4, v);
        mx = max(mx, max(max((uint32_t)b.x & 0x7fffffffu, (uint32_t)b.y & 0x7fffffffu), max((uint32_t)b.z & 0x7fffffffu, (uint32_t)b.w & 0x7fffffffu)));
      }
    }
    if (a.out_amax || a.out_amax2) {   // (kernel-uniform)
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, 16, 64));
      mx = max(mx, (uint32_t)__shfl_xor((int)mx, 32, 64));
      if (lq == 0 && row >= 0) {
        if (a.out_amax) atomicMax(a.out_amax + row, mx);
        if (a.out_amax2) atomicMax(a.out_amax2 + row, mx);
      }
    }
  }
}

bool dgr_conv_up_supported(int cin, int cin_pad, int cout) {
  return cin == cin_pad && (cin == 128 || cin == 256) && cout % 64 == 0 && cout >= 64;
}

int dgr_conv_up_launch(const DgrConvOsLaunch &a, hipStream_t stream, const char **kernel_name) {
  DGR_REQUIRE(a.nbr && a.nbr->built && a.nbr->K == 27 && a.nbr->perm && a.nbr->cls_count,
              "parity-class transposed conv: no neighbour table with a class row list");
  DGR_REQUIRE(dgr_conv_up_supported(a.cin, a.cin_pad, a.cout), "parity-class transposed conv: Cin = %d, Cout = %d not built", a.cin, a.cout);
  DGR_REQUIRE(a.wbd && a.row_amax, "parity-class transposed conv: needs the split weights in the dense-tile operand order and the input rows' maxima");
  DGR_REQUIRE(a.res == nullptr, "parity-class transposed conv: no residual input");
  DGR_REQUIRE((a.in_ld & 3) == 0 && (a.out_ld & 3) == 0, "parity-class transposed conv: row strides must be multiples of 4");
  DGR_REQUIRE(a.n_in_cap > 0 && a.n_in_cap * (int64_t)a.in_ld * 4 < (1ll << 31), "parity-class transposed conv: input tensor beyond 2 GB");
  ConvUpArgs ka;
  ka.in = a.in; ka.out = a.out; ka.shift = a.shift;
  ka.wb = static_cast<const u32x4 *>(a.wbd); ka.piece_stride = a.piece_stride;
  ka.nbr = a.nbr->nbr; ka.n_pad = a.nbr->n_pad;
  ka.perm = a.nbr->perm; ka.cls_count = a.nbr->cls_count; ka.cls_cap = a.nbr->cls_cap;
  ka.in_ld = a.in_ld; ka.in_relu = a.in_relu; ka.out_ld = a.out_ld; ka.out_relu = a.out_relu;
  ka.nb16 = a.cout / 16;
  ka.row_amax = a.row_amax; ka.out_amax = a.out_amax; ka.out_amax2 = a.out_amax2;
  ka.w_unscale = a.w_unscale;
  ka.in_bytes = (uint32_t)(a.n_in_cap * (int64_t)a.in_ld * 4);
  // (64-row workgroups at the coarse levels measured no better: 63 against 60-70 us for conv4_tr, tools/r06_runs/run24.sh)
  constexpr int RG = 2, WAVES = 4, MB = WAVES * RG * 16;
  // (eight classes, each rounded up to whole workgroups)
  const dim3 grid((unsigned)(dgr_ceil_div(a.n_out_cap, MB) + 8), (unsigned)(a.cout / 64));
  if (a.cin == 128) {
    if (kernel_name) *kernel_name = "sparse_conv_up_f16x2<128>";
    sparse_conv_up_f16x2<128, RG, WAVES><<<grid, 64 * WAVES, 0, stream>>>(ka);
  } else {
    if (kernel_name) *kernel_name = "sparse_conv_up_f16x2<256>";
    sparse_conv_up_f16x2<256, RG, WAVES><<<grid, 64 * WAVES, 0, stream>>>(ka);
  }
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}
