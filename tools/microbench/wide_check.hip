// Stand-alone harness around the wide-layer kernel (conv_wide.hip): (1) its product rows against an f64 host reference
// on a small rule-major map with adversarial rows (magnitudes over 12 decades, one dominant channel, zero and denormal
// rows, partial tiles, an empty rule), input split by dgr_split_rows; (2) timing on a synthetic map of the size of the
// 6-D block4 layers of the benchmark (75 k rows, 729 offsets, 3.57 M pairs) -- twice: with the input rows of a tile in
// ascending order inside a ~1000-row window (what a spatially sorted coordinate map would give) and with the row
// numbering shuffled (what the first-occurrence order of the real maps gives).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -DCHK_CIN=256 -DCHK_COUT=256 -o wide_check wide_check.hip
//   ./wide_check [pairs_for_timing (0 = skip)] [reps]
#include "../../deepglobalregistration_amd/csrc/conv_wide.hip"
#include <string.h>
#include <algorithm>
#include <random>
#include <vector>
void dgr_set_error(const char *fmt, ...) { (void)fmt; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#ifndef NUMCUS
#define NUMCUS 256
#endif

struct Map {
  std::vector<int32_t> tile_ptr, pair_in;
  std::vector<int4> desc;
  int K;
};
static Map make_map(const std::vector<std::vector<int32_t>> &rules) {
  Map m; m.K = (int)rules.size(); m.tile_ptr.assign(m.K + 1, 0);
  int p = 0;
  for (int k = 0; k < m.K; ++k) {
    const int c = (int)rules[k].size();
    m.tile_ptr[k + 1] = m.tile_ptr[k] + (c + 63) / 64;
    for (int t = 0; t < (c + 63) / 64; ++t) m.desc.push_back(make_int4(k, p + 64 * t, std::min(64, c - 64 * t), 0));
    for (int v : rules[k]) m.pair_in.push_back(v);
    p += c;
  }
  return m;
}

int main(int argc, char **argv) {
  const int cin = CHK_CIN, cout = CHK_COUT;
  const long timing_pairs = argc > 1 ? atol(argv[1]) : 3570000;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  // ------------------------------------------------------------------ (1) accuracy
  {
    const int K = 6, N = 1000;
    std::vector<int> counts = {100, 0, 64, 3333, 7, 129};
    std::vector<std::vector<int32_t>> rules(K);
    for (int k = 0; k < K; ++k) for (int p = 0; p < counts[k]; ++p) rules[k].push_back((p * 37 + k * 11) % N);
    Map m = make_map(rules);
    const int P = (int)m.pair_in.size();
    std::vector<float> in((size_t)N * cin), W((size_t)K * cin * cout);
    srand(3);
    for (auto &v : in) v = (float)rand() / RAND_MAX * 4.f - 2.f;
    for (int r = 0; r < N; ++r) {   // row magnitudes 1e-6 .. 1e6; inside a row, every 7th channel 1e-5 of the rest
      const float mag = powf(10.f, (float)(r % 13) - 6.f);
      for (int c = 0; c < cin; ++c) in[(size_t)r * cin + c] *= mag * (c % 7 == 3 ? 1e-5f : 1.f);
    }
    for (int c = 0; c < cin; ++c) {
      in[(size_t)5 * cin + c] = 0.f;                                  // a row of zeros
      in[(size_t)6 * cin + c] = (c & 1) ? 1e-41f : -3e-42f;           // a denormal row
      if (c != 17) in[(size_t)7 * cin + c] *= ldexpf(1.f, -20);       // one channel 2^20 above the rest
    }
    for (auto &v : W) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
    const int S16 = cin / 16, NB32 = cout / 32;
    const int64_t piece = (int64_t)K * S16 * NB32 * 64;
    std::vector<uint16_t> pieces((size_t)2 * piece * 8);
    float wmax = 0.f;
    for (auto v : W) wmax = fmaxf(wmax, fabsf(v));
    int we = 0; (void)frexpf(wmax, &we);
    const float w_scale = ldexpf(1.f, 15 - we), w_unscale = ldexpf(1.f, we - 15);
    auto f16_bits = [](float x) { _Float16 h = (_Float16)x; uint16_t b; memcpy(&b, &h, 2); return b; };
    for (int k = 0; k < K; ++k)
      for (int s = 0; s < S16; ++s)
        for (int nb = 0; nb < NB32; ++nb)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
              const float xs = W[((size_t)k * cin + 16 * s + 8 * (lane >> 5) + e) * cout + 32 * nb + (lane & 31)] * w_scale;
              const size_t o = ((((size_t)k * S16 + s) * NB32 + nb) * 64 + lane) * 8 + e;
              pieces[o] = f16_bits(xs); pieces[piece * 8 + o] = f16_bits(xs - (float)(_Float16)xs);
            }
    float *din, *dy, *dys, *drs; void *dwb; int32_t *dpi, *dtp, *dn; int4 *dd; unsigned char *dpl;
    CK(hipMalloc(&din, in.size() * 4)); CK(hipMalloc(&dy, (size_t)P * cout * 4)); CK(hipMalloc(&dys, 64 * 256 * 4));
    CK(hipMalloc(&dwb, pieces.size() * 2)); CK(hipMalloc(&dpi, m.pair_in.size() * 4)); CK(hipMalloc(&dtp, m.tile_ptr.size() * 4));
    CK(hipMalloc(&dd, m.desc.size() * sizeof(int4))); CK(hipMalloc(&drs, N * 4)); CK(hipMalloc(&dn, 4)); CK(hipMalloc(&dpl, (size_t)N * 4 * cin));
    CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dwb, pieces.data(), pieces.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dpi, m.pair_in.data(), m.pair_in.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dtp, m.tile_ptr.data(), m.tile_ptr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dd, m.desc.data(), m.desc.size() * sizeof(int4), hipMemcpyHostToDevice)); CK(hipMemcpy(dn, &N, 4, hipMemcpyHostToDevice));
    for (int relu = 0; relu < 2; ++relu)
      for (int cus : {NUMCUS, 8}) {   // 8 workgroups: every block walks many tiles (ring wrap-around, stage reuse)
        CK(hipMemset(dy, 0xff, (size_t)P * cout * 4));
        DgrSplitRows sr; sr.planes = dpl; sr.scale = drs; sr.channels = cin;
        if (dgr_split_rows(din, cin, relu, dn, N, sr, nullptr) != DGR_OK) { printf("split failed\n"); return 1; }
        DgrConvLaunch a{};
        a.in = din; a.in_ld = cin; a.in_relu = relu; a.y = dy; a.cin = cin; a.cin_pad = cin; a.cout = cout; a.cout_pad = cout; a.K = K;
        a.pair_in = dpi; a.tile_ptr = dtp; a.tile_desc = dd; a.tile_bound = m.desc.size();
        const char *name = "";
        if (dgr_conv_wide_launch(a, sr, dwb, piece, w_unscale, cus, nullptr, &name) != DGR_OK) { printf("launch failed\n"); return 1; }
        CK(hipDeviceSynchronize());
        std::vector<float> y((size_t)P * cout);
        CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
        double err = 0, err32 = 0; int bad = 0;
        for (int p = 0; p < P; ++p) {
          int k = 0, acc = 0;
          while (p >= acc + counts[k]) acc += counts[k++];
          double scale = 0, e_row = 0, e32_row = 0;
          for (int j = 0; j < cout; ++j) {
            double s = 0; float s32 = 0.f;
            for (int c = 0; c < cin; ++c) {
              float x = in[(size_t)m.pair_in[p] * cin + c]; if (relu && x < 0) x = 0;
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
        printf("%s relu=%d grid=%d P=%d max over rows of max|err|/max|y_row| = %.3e (host f32 FMA chain: %.3e) bad=%d\n", name, relu, cus, P, err, err32, bad);
      }
    hipFree(din); hipFree(dy); hipFree(dwb); hipFree(dpi); hipFree(dtp); hipFree(dd); hipFree(drs); hipFree(dn); hipFree(dpl); hipFree(dys);
  }
  if (timing_pairs <= 0) return 0;
  // ------------------------------------------------------------------ (2) timing
  for (int shuffled = 0; shuffled < 2; ++shuffled) {
    const int K = 729, N = 75000;
    std::mt19937 rng(1);
    std::vector<std::vector<int32_t>> rules(K);
    const long per = timing_pairs / K;
    for (int k = 0; k < K; ++k) {   // per offset: a sorted random subset of the rows (a same-stride map: in ~ out + shift)
      std::vector<int32_t> &r = rules[k];
      const double keep = (double)per / N;
      std::uniform_real_distribution<double> U(0, 1);
      for (int i = 0; i < N; ++i) if (U(rng) < keep) r.push_back(i);
    }
    if (shuffled) {   // the same map under a random renumbering of the input rows
      std::vector<int32_t> perm(N);
      for (int i = 0; i < N; ++i) perm[i] = i;
      std::shuffle(perm.begin(), perm.end(), rng);
      for (auto &r : rules) for (auto &v : r) v = perm[v];
    }
    Map m = make_map(rules);
    const size_t P = m.pair_in.size();
    const int S16 = cin / 16, NB32 = cout / 32;
    const int64_t piece = (int64_t)K * S16 * NB32 * 64;
    float *dy, *dys, *drs, *din; void *dwb; int32_t *dpi, *dtp, *dn; int4 *dd; unsigned char *dpl;
    CK(hipMalloc(&dy, P * cout * 4)); CK(hipMalloc(&dys, 64 * 256 * 4)); CK(hipMalloc(&dwb, (size_t)2 * piece * 16));
    CK(hipMalloc(&dpi, P * 4)); CK(hipMalloc(&dtp, m.tile_ptr.size() * 4)); CK(hipMalloc(&dd, m.desc.size() * sizeof(int4)));
    CK(hipMalloc(&drs, N * 4)); CK(hipMalloc(&dn, 4)); CK(hipMalloc(&dpl, (size_t)N * 4 * cin)); CK(hipMalloc(&din, (size_t)N * cin * 4));
    {
      std::vector<float> in((size_t)N * cin);
      std::normal_distribution<float> G(0.f, 1.f);
      for (auto &v : in) v = G(rng);
      CK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
      std::vector<uint16_t> w((size_t)2 * piece * 8);
      for (auto &v : w) { _Float16 h = (_Float16)(G(rng) * 1000.f); memcpy(&v, &h, 2); }
      CK(hipMemcpy(dwb, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(dpi, m.pair_in.data(), P * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dtp, m.tile_ptr.data(), m.tile_ptr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dd, m.desc.data(), m.desc.size() * sizeof(int4), hipMemcpyHostToDevice)); CK(hipMemcpy(dn, &N, 4, hipMemcpyHostToDevice));
    DgrSplitRows sr; sr.planes = dpl; sr.scale = drs; sr.channels = cin;
    if (dgr_split_rows(din, cin, 1, dn, N, sr, nullptr) != DGR_OK) { printf("split failed\n"); return 1; }
    DgrConvLaunch a{};
    a.in = din; a.in_ld = cin; a.in_relu = 1; a.y = dy; a.cin = cin; a.cin_pad = cin; a.cout = cout; a.cout_pad = cout; a.K = K;
    a.pair_in = dpi; a.tile_ptr = dtp; a.tile_desc = dd; a.tile_bound = m.desc.size();
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *name = "";
    float best = 1e30f, sum = 0.f;
    for (int i = 0; i < reps + 1; ++i) {
      CK(hipEventRecord(e0, nullptr));
      if (dgr_conv_wide_launch(a, sr, dwb, piece, 1.f, NUMCUS, nullptr, &name) != DGR_OK) { printf("launch failed\n"); return 1; }
      CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      if (i > 0) { best = fminf(best, ms); sum += ms; }
    }
    const double flop = 2.0 * P * cin * cout;
    printf("TIMING (%s input rows) %s P=%zu tiles=%zu: mean %.3f ms, best %.3f ms = %.1f TFLOP/s algorithmic (x3 issued: %.3f of 2500)\n", shuffled ? "shuffled" : "window-sorted", name, P, m.desc.size(),
           sum / reps, best, flop / (sum / reps * 1e-3) / 1e12, 3 * flop / (sum / reps * 1e-3) / 1e12 / 2500.0);
    hipFree(dy); hipFree(dys); hipFree(dwb); hipFree(dpi); hipFree(dtp); hipFree(dd); hipFree(drs); hipFree(dn); hipFree(dpl); hipFree(din);
  }
  return 0;
}
