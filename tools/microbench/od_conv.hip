// PROTOTYPE (round 2, compile-checked only -- it has NOT run on a GPU yet).
//
// "Dense-tile" variant of the 3-D output-stationary convolution for the two finest levels (Cin <= 64):
// a workgroup owns 64 consecutive output rows x 64 output channels and walks the 27 offsets; for every
// offset the 64 rows' neighbours (a missing neighbour is a row of zeros) are gathered into two f16 planes
// and multiplied as full 32 x 32 MFMA tiles; the per-offset result is folded into register accumulators
// with the row's inverse power-of-two scale.  Compared with sparse_conv_os (conv_os.hip) there are no
// compaction lists, no group bookkeeping and no LDS read-modify-write -- at the price of 2.1 x the
// matrix work (47 % fill), on a pipe that kernel keeps 8 % busy.  Same arithmetic as the product kernels:
// two f16 pieces per f32 operand under exact power-of-two row / layer scales, three products per MAC,
// sums per output element in ascending offset order on top of shift (+ residual).
//
// The harness builds a random surface-like voxel set, its 27-offset neighbour table, random features
// (rows spread over six orders of magnitude) and weights; runs this kernel and, on the same input,
// sparse_conv_os through dgr_conv_os_launch; checks both against an f64 host reference on a sample of
// rows and prints the average time of each.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -DOD_CIN=64 -o od_conv od_conv.hip && ./od_conv
#include "../../deepglobalregistration_amd/csrc/conv_os.hip"
#include "../../deepglobalregistration_amd/csrc/conv_bf3.hip"   // dgr_row_scale

#include <string.h>

#include <unordered_map>
#include <vector>

void dgr_set_error(const char *fmt, ...) { (void)fmt; }

#ifndef OD_CIN
#define OD_CIN 64
#endif

struct OdArgs {
  const float *in;
  float *out;
  const uint4 *wb;         // two f16 pieces, each [27][CP/16][cout/32][64] x 16 bytes (32x32x16 A-fragment order)
  int64_t piece_stride;
  const float *shift, *res;
  const float *row_scale;
  const int32_t *nbr;      // [27][n_pad]
  int64_t n_pad;
  int n_out, in_ld, in_relu, out_ld, out_relu, res_ld, res_relu, cout;
  float w_unscale;
};

template <int CP>
__global__ void __launch_bounds__(256) sparse_conv_od(OdArgs a) {
  constexpr int KV = 27, MB = 64, C4 = CP / 4, LDP = CP + 8, S = CP / 16, PLANE = MB * LDP;
  constexpr int NCH = MB * C4 / 256;   // 16-byte gather pieces per thread per offset (4 | 2)
  static_assert(MB * C4 % 256 == 0, "shape");
  __shared__ __attribute__((aligned(16))) unsigned short Ps[2][2][PLANE];
  __shared__ int nb_s[KV][MB];
  __shared__ float sc_s[KV][MB];
  __shared__ int kl[KV + 1];   // the offsets with at least one neighbour in this block, ascending; kl[KV] = how many

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rh = wave >> 1, ch = wave & 1;   // 32-row half, 32-column half of the 64 x 64 output tile
  const int slice = blockIdx.y;
  const int nblocks = (a.n_out + MB - 1) / MB;
  const int per = (nblocks + 7) >> 3;        // XCD-aware order as in sparse_conv_os
  const int j = blockIdx.x >> 3;
  const int blk = (blockIdx.x & 7) * per + j;
  if (j >= per || blk >= nblocks) return;
  const int64_t row0 = (int64_t)blk * MB;

  // ---- neighbour entries and their row scales into LDS; which offsets are present
  for (int e = tid; e < KV * MB; e += 256) {
    const int k = e / MB, r = e % MB;
    const int v = row0 + r < a.n_out ? a.nbr[(int64_t)k * a.n_pad + row0 + r] : -1;
    nb_s[k][r] = v;
    sc_s[k][r] = v >= 0 ? a.row_scale[v] : 1.f;
  }
  __syncthreads();
  if (wave == 0) {
    int present = 0;
    if (lane < KV) {
      for (int r = 0; r < MB; ++r) present |= nb_s[lane][r] >= 0;
    }
    const unsigned long long m = __ballot(present != 0);
    if (present) kl[__popcll(m & ((1ull << lane) - 1ull))] = lane;
    if (lane == 0) kl[KV] = __popcll(m);
  }
  __syncthreads();
  const int NQ = kl[KV];

  // ---- accumulators start from the folded batch-norm shift (+ residual)
  // D layout of v_mfma_f32_32x32x16: lane -> row n = lane & 31, channels m = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
  const int my_row = 32 * rh + (lane & 31);
  const int col0 = slice * 64 + 32 * ch + 4 * (lane >> 5);
  float total[16];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 v = a.shift ? *reinterpret_cast<const f32x4 *>(a.shift + col0 + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.res && row0 + my_row < a.n_out) {
      f32x4 x = *reinterpret_cast<const f32x4 *>(a.res + (row0 + my_row) * a.res_ld + col0 + 8 * g);
      if (a.res_relu) { x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f); }
      v += x;
    }
    total[4 * g] = v.x; total[4 * g + 1] = v.y; total[4 * g + 2] = v.z; total[4 * g + 3] = v.w;
  }

  f32x4 G[NCH];
  uint32_t okm = 0;
  const int relu_lo = a.in_relu ? 0 : (int)0x80000000;
  auto gather = [&](int q) {   // requests only
    const int k = kl[min(q, NQ - 1)];
    okm = 0;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int pc = tid + i * 256;
      const int row = nb_s[k][pc / C4];
      G[i] = *reinterpret_cast<const f32x4 *>(a.in + (int64_t)max(row, 0) * a.in_ld + (pc % C4) * 4);
      okm |= row >= 0 ? (1u << i) : 0u;
    }
  };
  auto land = [&](int q) {     // registers -> the two f16 planes of buffer q & 1 (missing neighbours: zeros)
    const int k = kl[min(q, NQ - 1)];
    unsigned short *dst = &Ps[q & 1][0][0];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int pc = tid + i * 256;
      const float sx = sc_s[k][pc / C4];
      const i32x4 gi = __builtin_bit_cast(i32x4, G[i]);
      const bool good = (okm >> i) & 1u;
      _Float16 hh[4], mm[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int xb = max(gi[u], relu_lo);
        xb = good ? xb : 0;
        dgr_split2(__builtin_bit_cast(float, xb), sx, hh[u], mm[u]);
      }
      const int o = (pc / C4) * LDP + (pc % C4) * 4;
      *reinterpret_cast<u32x2 *>(dst + o) = u32x2{__builtin_bit_cast(uint32_t, f16x2{hh[0], hh[1]}),
                                                  __builtin_bit_cast(uint32_t, f16x2{hh[2], hh[3]})};
      *reinterpret_cast<u32x2 *>(dst + PLANE + o) = u32x2{__builtin_bit_cast(uint32_t, f16x2{mm[0], mm[1]}),
                                                          __builtin_bit_cast(uint32_t, f16x2{mm[2], mm[3]})};
    }
  };
  // weight fragments of offset kl[q]: S k-steps x 2 pieces, 16 bytes per lane each, for this wave's 32 columns
  const int nblk = a.cout / 32, nb = slice * 2 + ch;
  struct WSet { uint4 v[S][2]; };
  auto wload = [&](int q, WSet &w) {
    const int k = kl[min(q, NQ - 1)];
    const uint4 *p = a.wb + ((int64_t)(k * S) * nblk + nb) * 64 + lane;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      w.v[s][0] = p[(int64_t)s * nblk * 64];
      w.v[s][1] = p[(int64_t)s * nblk * 64 + a.piece_stride];
    }
  };

  WSet w[2];
  if (NQ > 0) {
    gather(0);
    wload(0, w[0]);
    land(0);
    gather(1);
  }
  __syncthreads();
  const int lofs = my_row * LDP + 8 * (lane >> 5);
  auto phase = [&](int q, WSet &wc, WSet &wn) {
    land(q + 1);                    // requested one phase ago
    gather(q + 2);                  // a whole phase to arrive
    wload(q + 1, wn);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 tmp;
#pragma unroll
    for (int e = 0; e < 16; ++e) tmp[e] = 0.f;
    const unsigned short *pl = &Ps[q & 1][0][0] + lofs;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const f16x8 ah = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(pl + s * 16));
      const f16x8 am = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4 *>(pl + PLANE + s * 16));
      const f16x8 wh = __builtin_bit_cast(f16x8, wc.v[s][0]), wm = __builtin_bit_cast(f16x8, wc.v[s][1]);
      tmp = __builtin_amdgcn_mfma_f32_32x32x16_f16(wm, ah, tmp, 0, 0, 0);
      tmp = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, am, tmp, 0, 0, 0);
      tmp = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, ah, tmp, 0, 0, 0);
    }
    const float f = dgr_inv_pow2(sc_s[kl[q]][my_row]) * a.w_unscale;   // rows without this neighbour: tmp = 0 exactly
#pragma unroll
    for (int e = 0; e < 16; ++e) total[e] = fmaf(tmp[e], f, total[e]);
    __syncthreads();
  };
  for (int q = 0; q < NQ; q += 2) {   // the two weight sets alternate statically
    phase(q, w[0], w[1]);
    if (q + 1 < NQ) phase(q + 1, w[1], w[0]);
  }
  // ---- the block's rows, written once
  if (row0 + my_row < a.n_out) {
    const float lo = a.out_relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = {fmaxf(total[4 * g], lo), fmaxf(total[4 * g + 1], lo), fmaxf(total[4 * g + 2], lo), fmaxf(total[4 * g + 3], lo)};
      *reinterpret_cast<f32x4 *>(a.out + (row0 + my_row) * a.out_ld + col0 + 8 * g) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------ harness
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static uint16_t f16_bits(float x) { _Float16 h = (_Float16)x; uint16_t b; memcpy(&b, &h, 2); return b; }

int main() {
  const int cin = OD_CIN, cout = 64, K = 27;
  // surface-like voxel set: a few noisy planes in a 160^3 box
  std::vector<int> cx, cy, cz;
  std::unordered_map<uint64_t, int> idx;
  srand(5);
  auto key = [](int x, int y, int z) { return ((uint64_t)(x + 512) << 40) | ((uint64_t)(y + 512) << 20) | (uint64_t)(z + 512); };
  for (int p = 0; p < 6; ++p) {
    const float nx = rand() / (float)RAND_MAX - 0.5f, ny = rand() / (float)RAND_MAX - 0.5f, off = 40.f + 80.f * rand() / (float)RAND_MAX;
    for (int u = 0; u < 160; ++u)
      for (int v = 0; v < 160; ++v) {
        const int x = u, y = v, z = (int)(off + nx * u + ny * v + (rand() % 3 - 1) * (rand() % 2));
        if (z < 0 || z >= 160) continue;
        if (rand() % 100 < 15) continue;   // holes
        const uint64_t kk = key(x, y, z);
        if (idx.emplace(kk, (int)cx.size()).second) { cx.push_back(x); cy.push_back(y); cz.push_back(z); }
      }
  }
  const int N = (int)cx.size();
  const int64_t n_pad = (N + 63) / 64 * 64;
  std::vector<int32_t> nbr((size_t)K * n_pad, -1);
  int64_t pairs = 0;
  for (int k = 0; k < K; ++k) {
    const int dx = k % 3 - 1, dy = (k / 3) % 3 - 1, dz = k / 9 - 1;
    for (int i = 0; i < N; ++i) {
      auto it = idx.find(key(cx[i] + dx, cy[i] + dy, cz[i] + dz));
      if (it != idx.end()) { nbr[(size_t)k * n_pad + i] = it->second; ++pairs; }
    }
  }
  printf("N = %d voxels, %lld pairs (%.1f per row), Cin = %d, Cout = %d\n", N, (long long)pairs, (double)pairs / N, cin, cout);
  std::vector<float> in((size_t)N * cin), W((size_t)K * cin * cout), shift(cout), res((size_t)N * cout);
  for (int r = 0; r < N; ++r) {
    const float mag = powf(10.f, (float)(r % 7) - 3.f);
    for (int c = 0; c < cin; ++c) in[(size_t)r * cin + c] = (rand() / (float)RAND_MAX * 4.f - 2.f) * mag;
  }
  for (auto &v : W) v = (rand() / (float)RAND_MAX * 2.f - 1.f) * 0.1f;
  for (auto &v : shift) v = rand() / (float)RAND_MAX - 0.5f;
  for (auto &v : res) v = rand() / (float)RAND_MAX - 0.5f;
  float wmax = 0.f;
  for (auto v : W) wmax = fmaxf(wmax, fabsf(v));
  int we = 0; (void)frexpf(wmax, &we);
  const float w_scale = ldexpf(1.f, 15 - we), w_unscale = ldexpf(1.f, we - 15);
  // weights: (a) 32x32x16 fragment order for the prototype, (b) 16x16x32 fragment order for sparse_conv_os
  const int S16 = cin / 16, NB32 = cout / 32, S32 = cin / 32, NB16 = cout / 16;
  const int64_t piece_a = (int64_t)K * S16 * NB32 * 64, piece_b = (int64_t)K * S32 * NB16 * 64;
  std::vector<uint16_t> wa((size_t)2 * piece_a * 8), wbv((size_t)2 * piece_b * 8);
  for (int k = 0; k < K; ++k) {
    const float *src = W.data() + (size_t)k * cin * cout;
    for (int s = 0; s < S16; ++s) for (int nb = 0; nb < NB32; ++nb) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e) {
      const float xs = src[(size_t)(16 * s + 8 * (lane >> 5) + e) * cout + 32 * nb + (lane & 31)] * w_scale;
      const size_t o = ((((size_t)k * S16 + s) * NB32 + nb) * 64 + lane) * 8 + e;
      wa[o] = f16_bits(xs); wa[(size_t)piece_a * 8 + o] = f16_bits(xs - (float)(_Float16)xs);
    }
    for (int s = 0; s < S32; ++s) for (int jb = 0; jb < NB16; ++jb) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e) {
      const float xs = src[(size_t)(32 * s + 8 * (lane >> 4) + e) * cout + 16 * jb + (lane & 15)] * w_scale;
      const size_t o = ((((size_t)k * S32 + s) * NB16 + jb) * 64 + lane) * 8 + e;
      wbv[o] = f16_bits(xs); wbv[(size_t)piece_b * 8 + o] = f16_bits(xs - (float)(_Float16)xs);
    }
  }
  float *din, *dout, *dout2, *dshift, *dres, *drs; void *dwa, *dwb; int32_t *dnbr, *dn;
  CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dout, (size_t)N * cout * 4)); CK(hipMalloc(&dout2, (size_t)N * cout * 4));
  CK(hipMalloc(&dshift, cout * 4)); CK(hipMalloc(&dres, res.size() * 4)); CK(hipMalloc(&drs, (size_t)N * 4));
  CK(hipMalloc(&dwa, wa.size() * 2)); CK(hipMalloc(&dwb, wbv.size() * 2)); CK(hipMalloc(&dnbr, nbr.size() * 4)); CK(hipMalloc(&dn, 4));
  CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dshift, shift.data(), cout * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dres, res.data(), res.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dwa, wa.data(), wa.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dwb, wbv.data(), wbv.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dnbr, nbr.data(), nbr.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dn, &N, 4, hipMemcpyHostToDevice));
  if (dgr_row_scale(din, cin, cin, 1, dn, N, drs, nullptr) != DGR_OK) { printf("row scale failed\n"); return 1; }

  OdArgs a{};
  a.in = din; a.out = dout; a.wb = (const uint4 *)dwa; a.piece_stride = piece_a; a.shift = dshift; a.res = dres;
  a.row_scale = drs; a.nbr = dnbr; a.n_pad = n_pad; a.n_out = N; a.in_ld = cin; a.in_relu = 1; a.out_ld = cout; a.out_relu = 0;
  a.res_ld = cout; a.res_relu = 1; a.cout = cout; a.w_unscale = w_unscale;
  int64_t blocks = ((N + 63) / 64 + 7) / 8 * 8;
  DgrNbrTable t; t.nbr = dnbr; t.n_pad = n_pad; t.K = 27; t.built = true;
  DgrConvOsLaunch o;
  o.in = din; o.in_ld = cin; o.in_relu = 1; o.out = dout2; o.out_ld = cout; o.out_relu = 0; o.w16 = nullptr; o.shift = dshift;
  o.wb3 = dwb; o.piece_stride = piece_b; o.pieces = 2; o.row_scale = drs; o.w_unscale = w_unscale; o.res = dres; o.res_ld = cout; o.res_relu = 1;
  o.rows_per_block = 64; o.nbr = &t; o.n_out_dev = dn; o.n_out_cap = N; o.cin = cin; o.cin_pad = cin; o.cout = cout;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms_od = 0.f, ms_os = 0.f;
  const char *name = "";
  for (int rep = 0; rep < 21; ++rep) {
    CK(hipEventRecord(e0));
    sparse_conv_od<OD_CIN><<<dim3((unsigned)blocks, (unsigned)(cout / 64)), 256>>>(a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) ms_od += ms;
    CK(hipEventRecord(e0));
    if (dgr_conv_os_launch(o, nullptr, &name) != DGR_OK) { printf("os launch failed\n"); return 1; }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); if (rep) ms_os += ms;
  }
  CK(hipGetLastError());
  printf("dense-tile prototype: %.1f us   %s: %.1f us\n", ms_od * 50.f, name, ms_os * 50.f);
  std::vector<float> y((size_t)N * cout), y2((size_t)N * cout);
  CK(hipMemcpy(y.data(), dout, y.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y2.data(), dout2, y2.size() * 4, hipMemcpyDeviceToHost));
  double e_od = 0, e_os = 0; int bad = 0;
  for (int i = 0; i < N; i += 37) {
    double scale = 0, d1 = 0, d2 = 0;
    for (int jc = 0; jc < cout; ++jc) {
      double s = shift[jc] + fmax(res[(size_t)i * cout + jc], 0.f);
      for (int k = 0; k < K; ++k) {
        const int r = nbr[(size_t)k * n_pad + i];
        if (r < 0) continue;
        for (int c = 0; c < cin; ++c) s += (double)fmaxf(in[(size_t)r * cin + c], 0.f) * W[((size_t)k * cin + c) * cout + jc];
      }
      scale = fmax(scale, fabs(s));
      d1 = fmax(d1, fabs(s - y[(size_t)i * cout + jc])); d2 = fmax(d2, fabs(s - y2[(size_t)i * cout + jc]));
      if (!(y[(size_t)i * cout + jc] == y[(size_t)i * cout + jc])) ++bad;
    }
    e_od = fmax(e_od, d1 / scale); e_os = fmax(e_os, d2 / scale);
  }
  printf("max over sampled rows of max|err|/max|y_row|: prototype %.3e, sparse_conv_os %.3e, NaNs %d\n", e_od, e_os, bad);
  return 0;
}
