// Microbenchmark + numerics check: one sparse-conv tile product C[64 x 256] = A[64 x 256] . W[256 x 256]
// (a) with exact-f32 MFMA (v_mfma_f32_32x32x2_f32, what conv.hip does) and
// (b) with the f32 operands split exactly into three bf16 pieces each (x = h + m + l, 8 + 8 + 8 significant bits
//     by truncation) and SIX v_mfma_f32_32x32x16_bf16 products per k-block (hh, hm, mh, hl, lh, mm): the bf16
//     pipe is 16x faster than the f32 pipe, so six products are 2.67x fewer matrix cycles than (a).
// Prints the error of both against an f64 host reference and the sustained TFLOP/s (useful 2 M N K per tile).
//   hipcc -O3 --offload-arch=gfx950 -o bf16x3_tile bf16x3_tile.hip && ./bf16x3_tile
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int C = 256;      // Cin = Cout
constexpr int TM = 64;      // rows per tile

// ---- (a) f32: W fragments Wf[s][nb][lane][c] = W[8 s + 4 (lane >> 5) + c][32 nb + (lane & 31)] (conv.hip layout)
__global__ void __launch_bounds__(256, 2) tile_f32(const float *__restrict__ A, const float *__restrict__ Wf, float *__restrict__ Cout, int tiles) {
  __shared__ __attribute__((aligned(16))) float As[TM][C + 4];
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;   // 1 x 4 waves, each 64 rows x 64 columns
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const float *a = A + (size_t)t * TM * C;
    for (int e = tid; e < TM * C / 4; e += 256) {
      const int r = e / (C / 4), c = (e % (C / 4)) * 4;
      *reinterpret_cast<f32x4 *>(&As[r][c]) = *reinterpret_cast<const f32x4 *>(a + r * C + c);
    }
    __syncthreads();
    f32x16 acc[2][2] = {};
    const f32x4 *wk = reinterpret_cast<const f32x4 *>(Wf) + (wn * 2) * 64 + lane;
    for (int s = 0; s < C / 8; ++s) {
      f32x4 b[2], av[2];
      for (int j = 0; j < 2; ++j) b[j] = wk[(size_t)(s * 8 + j) * 64];
      for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const f32x4 *>(&As[32 * i + (lane & 31)][s * 8 + 4 * (lane >> 5)]);
      for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 2; ++i)
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j][c], av[i][c], acc[i][j], 0, 0, 0);
    }
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        for (int g = 0; g < 4; ++g) {
          const int col = 32 * (wn * 2 + j) + 8 * g + 4 * (lane >> 5);
          f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          *reinterpret_cast<f32x4 *>(Cout + ((size_t)t * TM + 32 * i + (lane & 31)) * C + col) = v;
        }
    __syncthreads();
  }
}

// ---- (b) bf16 x 3.  Weight fragments Wb[piece][s][nb][lane] (16 bytes = 8 bf16): piece p of
//      W[16 s + 8 (lane >> 5) + e][32 nb + (lane & 31)], e = 0..7; activations as three bf16 planes in LDS.
__device__ __forceinline__ void split3(float x, uint32_t &h, uint32_t &m, uint32_t &l) {
  const uint32_t xb = __builtin_bit_cast(uint32_t, x);
  h = xb & 0xffff0000u;
  const float r1 = x - __builtin_bit_cast(float, h);
  const uint32_t rb = __builtin_bit_cast(uint32_t, r1);
  m = rb & 0xffff0000u;
  l = __builtin_bit_cast(uint32_t, r1 - __builtin_bit_cast(float, m));   // <= 8 significant bits: exact in bf16
}

__global__ void __launch_bounds__(256, 2) tile_bf16x3(const float *__restrict__ A, const uint4 *__restrict__ Wb, float *__restrict__ Cout, int tiles) {
  constexpr int LDP = C + 8;   // bf16 elements per plane row (+16 bytes pad)
  __shared__ __attribute__((aligned(16))) unsigned short P[3][TM][LDP];
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  constexpr size_t PIECE = (size_t)(C / 16) * (C / 32) * 64;   // uint4 per weight piece
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const float *a = A + (size_t)t * TM * C;
    for (int e = tid; e < TM * C / 4; e += 256) {
      const int r = e / (C / 4), c = (e % (C / 4)) * 4;
      const f32x4 v = *reinterpret_cast<const f32x4 *>(a + r * C + c);
      uint32_t h[4], m[4], l[4];
      for (int u = 0; u < 4; ++u) split3(v[u], h[u], m[u], l[u]);
      *reinterpret_cast<u32x2 *>(&P[0][r][c]) = u32x2{(h[0] >> 16) | h[1], (h[2] >> 16) | h[3]};
      *reinterpret_cast<u32x2 *>(&P[1][r][c]) = u32x2{(m[0] >> 16) | m[1], (m[2] >> 16) | m[3]};
      *reinterpret_cast<u32x2 *>(&P[2][r][c]) = u32x2{(l[0] >> 16) | (l[1] & 0xffff0000u), (l[2] >> 16) | (l[3] & 0xffff0000u)};
    }
    __syncthreads();
    f32x16 acc[2][2] = {};
    const uint4 *wk = Wb + (wn * 2) * 64 + lane;
    for (int s = 0; s < C / 16; ++s) {
      bf16x8 wh[2], wm[2], wl[2], ah[2], am[2], al[2];
      for (int j = 0; j < 2; ++j) {
        wh[j] = __builtin_bit_cast(bf16x8, wk[0 * PIECE + (size_t)(s * 8 + j) * 64]);
        wm[j] = __builtin_bit_cast(bf16x8, wk[1 * PIECE + (size_t)(s * 8 + j) * 64]);
        wl[j] = __builtin_bit_cast(bf16x8, wk[2 * PIECE + (size_t)(s * 8 + j) * 64]);
      }
      for (int i = 0; i < 2; ++i) {
        const int r = 32 * i + (lane & 31), c = 16 * s + 8 * (lane >> 5);
        ah[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(&P[0][r][c]));
        am[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(&P[1][r][c]));
        al[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(&P[2][r][c]));
      }
      // smallest terms first
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[j], ah[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[j], al[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[j], am[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[j], ah[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[j], am[i], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[j], ah[i], acc[i][j], 0, 0, 0);
        }
    }
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        for (int g = 0; g < 4; ++g) {
          const int col = 32 * (wn * 2 + j) + 8 * g + 4 * (lane >> 5);
          f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          *reinterpret_cast<f32x4 *>(Cout + ((size_t)t * TM + 32 * i + (lane & 31)) * C + col) = v;
        }
    __syncthreads();
  }
}

static uint32_t trunc16(float x) { uint32_t b; memcpy(&b, &x, 4); return b & 0xffff0000u; }
static float asf(uint32_t b) { float x; memcpy(&x, &b, 4); return x; }

int main() {
  const int tiles = 16384;
  std::vector<float> A((size_t)tiles * TM * C), W((size_t)C * C);
  srand(1);
  for (auto &v : A) { v = (float)rand() / RAND_MAX * 2.f - 1.f; v = v > 0.f ? v * 3.f : 0.f; }   // ReLU-like activations
  for (auto &v : W) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
  std::vector<float> Wf((size_t)C * C);
  for (int s = 0; s < C / 8; ++s)
    for (int nb = 0; nb < C / 32; ++nb)
      for (int lane = 0; lane < 64; ++lane)
        for (int c = 0; c < 4; ++c)
          Wf[(((size_t)s * (C / 32) + nb) * 64 + lane) * 4 + c] = W[(size_t)(8 * s + 4 * (lane >> 5) + c) * C + 32 * nb + (lane & 31)];
  const size_t piece = (size_t)(C / 16) * (C / 32) * 64 * 8;   // bf16 elements per piece
  std::vector<unsigned short> Wb(3 * piece);
  for (int s = 0; s < C / 16; ++s)
    for (int nb = 0; nb < C / 32; ++nb)
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const float x = W[(size_t)(16 * s + 8 * (lane >> 5) + e) * C + 32 * nb + (lane & 31)];
          const uint32_t h = trunc16(x);
          const float r1 = x - asf(h);
          const uint32_t m = trunc16(r1);
          const float r2 = r1 - asf(m);
          const size_t o = (((size_t)s * (C / 32) + nb) * 64 + lane) * 8 + e;
          Wb[0 * piece + o] = (unsigned short)(h >> 16);
          Wb[1 * piece + o] = (unsigned short)(m >> 16);
          Wb[2 * piece + o] = (unsigned short)(trunc16(r2) >> 16);
          if (asf(trunc16(r2)) != r2) { printf("split not exact\n"); return 1; }
        }
  float *dA, *dWf, *dC;
  uint4 *dWb;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dWf, Wf.size() * 4)); CK(hipMalloc(&dC, A.size() * 4));
  CK(hipMalloc(&dWb, Wb.size() * 2));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dWf, Wf.data(), Wf.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dWb, Wb.data(), Wb.size() * 2, hipMemcpyHostToDevice));
  std::vector<float> out((size_t)4 * TM * C);
  std::vector<double> ref((size_t)4 * TM * C);
  for (int r = 0; r < 4 * TM; ++r)
    for (int j = 0; j < C; ++j) {
      double s = 0;
      for (int k = 0; k < C; ++k) s += (double)A[(size_t)r * C + k] * (double)W[(size_t)k * C + j];
      ref[(size_t)r * C + j] = s;
    }
  double scale = 0;
  for (double v : ref) scale = fmax(scale, fabs(v));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      if (mode == 0) tile_f32<<<512, 256>>>(dA, dWf, dC, tiles);
      else tile_bf16x3<<<512, 256>>>(dA, dWb, dC, tiles);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) {
        CK(hipMemcpy(out.data(), dC, out.size() * 4, hipMemcpyDeviceToHost));
        double err = 0, bias = 0;
        for (size_t i = 0; i < out.size(); ++i) { err = fmax(err, fabs(out[i] - ref[i])); bias += out[i] - ref[i]; }
        printf("%-10s %.3f ms  %.1f TFLOP/s (useful)  max|err|/max|y| = %.3e  mean signed err/scale = %+.2e\n",
               mode == 0 ? "f32 mfma" : "bf16 x 3", ms, 2.0 * tiles * TM * C * C / (ms * 1e-3) / 1e12, err / scale,
               bias / out.size() / scale);
      }
    }
  }
  return 0;
}
