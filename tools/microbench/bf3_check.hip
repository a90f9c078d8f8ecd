// Stand-alone check of sparse_conv_bf16x3 (conv_bf3.hip) against an f64 host reference on a synthetic
// rule-major map: K rules, random pair lists, rows spread over 12 orders of magnitude, a row of zeros, a denormal row.
//   hipcc -O3 --offload-arch=gfx950 -I../../include -DCIN=256 -DCOUT=256 -DNUMCUS=256 -DPIECES=2 -o bf3_check bf3_check.hip
// PIECES = 3: bf16 x 3 (six products); PIECES = 2: f16 x 2 under power-of-two row / layer scales (three products).
// Also prints the error of a plain f32 FMA chain on the host, the yardstick both are held to.
#include "../../deepglobalregistration_amd/csrc/conv_bf3.hip"
#include <string.h>
#include <vector>
void dgr_set_error(const char *fmt, ...) { (void)fmt; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  const int cin = CHK_CIN, cout = CHK_COUT, K = 5, N = 1000;
  std::vector<int> counts = {100, 0, 64, 3333, 7};
  std::vector<int32_t> rule_ptr(K + 1, 0), tile_ptr(K + 1, 0), pair_in;
  std::vector<int4> desc;
  for (int k = 0; k < K; ++k) {
    rule_ptr[k + 1] = rule_ptr[k] + counts[k];
    tile_ptr[k + 1] = tile_ptr[k] + (counts[k] + 63) / 64;
    for (int t = 0; t < (counts[k] + 63) / 64; ++t) desc.push_back(make_int4(k, rule_ptr[k] + 64 * t, std::min(64, counts[k] - 64 * t), 0));
    for (int p = 0; p < counts[k]; ++p) pair_in.push_back((p * 37 + k * 11) % N);
  }
  const int P = rule_ptr[K];
  std::vector<float> in((size_t)N * cin), W((size_t)K * cin * cout);
  srand(3);
  for (auto &v : in) v = (float)rand() / RAND_MAX * 4.f - 2.f;
  for (int r = 0; r < N; ++r) {   // row magnitudes 1e-6 .. 1e6; inside a row, every 7th channel 1e-5 of the rest
    const float mag = powf(10.f, (float)(r % 13) - 6.f);
    for (int c = 0; c < cin; ++c) in[(size_t)r * cin + c] *= mag * (c % 7 == 3 ? 1e-5f : 1.f);
  }
  for (int c = 0; c < cin; ++c) { in[(size_t)5 * cin + c] = 0.f; in[(size_t)6 * cin + c] = (c & 1) ? 1e-41f : -3e-42f; }
  for (auto &v : W) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
#ifdef IDENT
  for (int k = 0; k < K; ++k) for (int c = 0; c < cin; ++c) for (int j = 0; j < cout; ++j) W[((size_t)k * cin + c) * cout + j] = (c == j) ? 1.f : 0.f;
#endif
  const int S16 = cin / 16, NB32 = cout / 32;
  const int64_t piece = (int64_t)K * S16 * NB32 * 64;
  std::vector<uint16_t> pieces((size_t)PIECES * piece * 8);
  float wmax = 0.f;
  for (auto v : W) wmax = fmaxf(wmax, fabsf(v));
  int we = 0; (void)frexpf(wmax, &we);
  const float w_scale = ldexpf(1.f, 15 - we), w_unscale = PIECES == 2 ? ldexpf(1.f, we - 15) : 1.f;
  auto f16_bits = [](float x) { _Float16 h = (_Float16)x; uint16_t b; memcpy(&b, &h, 2); return b; };
  auto top16 = [](float x) { uint32_t b; memcpy(&b, &x, 4); return b & 0xffff0000u; };
  auto asf = [](uint32_t b) { float x; memcpy(&x, &b, 4); return x; };
  for (int k = 0; k < K; ++k)
    for (int s = 0; s < S16; ++s)
      for (int nb = 0; nb < NB32; ++nb)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const float x = W[((size_t)k * cin + 16 * s + 8 * (lane >> 5) + e) * cout + 32 * nb + (lane & 31)];
            const uint32_t h = top16(x); const float r1 = x - asf(h); const uint32_t m = top16(r1); const float r2 = r1 - asf(m);
            const size_t o = ((((size_t)k * S16 + s) * NB32 + nb) * 64 + lane) * 8 + e;
            if (PIECES == 2) {
              const float xs = x * w_scale;
              pieces[o] = f16_bits(xs); pieces[piece * 8 + o] = f16_bits(xs - (float)(_Float16)xs);
            } else {
              pieces[o] = h >> 16; pieces[piece * 8 + o] = m >> 16; pieces[2 * piece * 8 + o] = top16(r2) >> 16;
            }
          }
  float *din, *dy; void *dwb; int32_t *dpi, *dtp; int4 *dd;
  CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dy, (size_t)P * cout * 4)); CK(hipMalloc(&dwb, pieces.size() * 2));
  CK(hipMalloc(&dpi, pair_in.size() * 4)); CK(hipMalloc(&dtp, tile_ptr.size() * 4)); CK(hipMalloc(&dd, desc.size() * sizeof(int4)));
  CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dwb, pieces.data(), pieces.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dpi, pair_in.data(), pair_in.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dtp, tile_ptr.data(), tile_ptr.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dd, desc.data(), desc.size() * sizeof(int4), hipMemcpyHostToDevice)); CK(hipMemset(dy, 0xff, (size_t)P * cout * 4));
  float *drs; int32_t *dn;
  CK(hipMalloc(&drs, N * 4)); CK(hipMalloc(&dn, 4)); CK(hipMemcpy(dn, &N, 4, hipMemcpyHostToDevice));
  for (int relu = 0; relu < 2; ++relu) {
    DgrConvLaunch a{};
    a.in = din; a.in_ld = cin; a.in_relu = relu; a.y = dy; a.cin = cin; a.cin_pad = cin; a.cout = cout; a.cout_pad = cout; a.K = K;
    a.pair_in = dpi; a.tile_ptr = dtp; a.tile_desc = dd; a.tile_bound = desc.size();
    const char *name = "";
    if (dgr_row_scale(din, cin, cin, relu, dn, N, drs, nullptr) != DGR_OK) { printf("row scale failed\n"); return 1; }
    if (dgr_conv_bf3_launch(a, dwb, piece, PIECES, w_unscale, drs, NUMCUS, nullptr, &name) != DGR_OK) { printf("launch failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<float> y((size_t)P * cout);
    CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
    // errors are measured per product row against that row's largest |y| (rows differ by 12 orders of magnitude)
    double err = 0, err32 = 0; int bad = 0;
    for (int k = 0; k < K; ++k)
      for (int p = rule_ptr[k]; p < rule_ptr[k + 1]; ++p) {
        double scale = 0, e_row = 0, e32_row = 0;
        for (int j = 0; j < cout; ++j) {
          double s = 0; float s32 = 0.f;
          for (int c = 0; c < cin; ++c) {
            float x = in[(size_t)pair_in[p] * cin + c]; if (relu && x < 0) x = 0;
            const float w = W[((size_t)k * cin + c) * cout + j];
            s += (double)x * w; s32 = fmaf(x, w, s32);
          }
          scale = fmax(scale, fabs(s));
          const double d = fabs(s - y[(size_t)p * cout + j]);
          if (!(d == d)) ++bad;
          e_row = fmax(e_row, d); e32_row = fmax(e32_row, fabs(s - s32));
        }
        if (scale > 1e-30) { err = fmax(err, e_row / scale); err32 = fmax(err32, e32_row / scale); }
        else if (e_row > 1e-36) ++bad;
      }
    printf("%s relu=%d P=%d max over rows of max|err|/max|y_row| = %.3e (host f32 FMA chain: %.3e) bad=%d\n", name, relu, P, err, err32, bad);
#ifdef IDENT
    for (int p : {0, 70}) { printf("p=%d in :", p); for (int c = 0; c < 20; ++c) printf(" %6.3f", in[(size_t)pair_in[p] * cin + c]); printf("\n      got:"); for (int c = 0; c < 20; ++c) printf(" %6.3f", y[(size_t)p * cout + c]); printf("\n"); }
#endif
    for (int p : {0, 1, 63, 64, 100, 164}) {
      double s0 = 0, s1 = 0;
      for (int c = 0; c < cin; ++c) { float x = in[(size_t)pair_in[p] * cin + c]; if (relu && x < 0) x = 0; const int k = p < 100 ? 0 : (p < 164 ? 2 : 3); s0 += (double)x * W[((size_t)k * cin + c) * cout + 0]; s1 += (double)x * W[((size_t)k * cin + c) * cout + 5]; }
      printf("   p=%d ref %.4f %.4f got %.4f %.4f\n", p, s0, s1, y[(size_t)p * cout], y[(size_t)p * cout + 5]);
    }
  }
  return 0;
}
