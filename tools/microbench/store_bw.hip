// Ceiling of the product-row store stream of the wide-layer kernel (conv_wide.hip): persistent workgroups, one per CU,
// W waves each writing whole 1-KB rows (64 lanes x 16 B) of a 3.65-GB buffer in 64-KB tile chunks, by store flavour.
//   hipcc -O3 --offload-arch=gfx950 -o store_bw store_bw.hip && ./store_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int MODE, int UNROLL>
__global__ void __launch_bounds__(512) fill(float *y, long tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  for (long t = blockIdx.x; t < tiles; t += gridDim.x) {
    float *base = y + t * 64 * 256;   // a tile = 64 rows of 1 KB
    for (int r = wave * UNROLL; r < 64; r += nw * UNROLL) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        f32x4 *p = reinterpret_cast<f32x4 *>(base + (r + u) * 256 + lane * 4);
        if (MODE == 0) *p = v;
        else if (MODE == 1) __builtin_nontemporal_store(v, p);
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
      }
    }
  }
}
template <int MODE, int UNROLL>
static int run(float *y, long tiles, int waves, const char *name) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int i = 0; i < 4; ++i) {
    CK(hipEventRecord(e0, nullptr));
    fill<MODE, UNROLL><<<256, 64 * waves>>>(y, tiles);
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (i) best = fminf(best, ms);
  }
  printf("%-10s unroll %d, %d waves/CU: %.3f ms = %.2f TB/s\n", name, UNROLL, waves, best, tiles * 65536.0 / (best * 1e-3) / 1e12);
  return 0;
}
int main() {
  const long tiles = 56088;
  float *y; CK(hipMalloc(&y, tiles * 65536));
  for (int w : {1, 2, 4, 8}) {
    if (run<0, 1>(y, tiles, w, "plain")) return 1;
    if (run<1, 1>(y, tiles, w, "nt")) return 1;
    if (run<2, 1>(y, tiles, w, "sc0 sc1")) return 1;
    if (run<0, 4>(y, tiles, w, "plain")) return 1;
    if (run<1, 4>(y, tiles, w, "nt")) return 1;
  }
  return 0;
}
