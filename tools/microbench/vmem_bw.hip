// Ceiling of the vector-memory LOAD path of one CU on cache-resident data (what bounds the gathers of conv_dense.hip /
// conv_os.hip and the weight stream of conv_wide.hip): 256 workgroups (one per CU) of W waves, every wave issues
// dwordx4 loads (64 lanes x 16 B = 1 KB per instruction, 8 in flight per wave) from a region of `mb` MB in four patterns:
//   linear    lane l reads 16 B at 16 l of a 1-KB row (fully coalesced)
//   rows16    16 random 256-B rows per instruction, lanes (row, chunk) = (l >> 2, l & 3): a quad reads 64 contiguous B
//   operand   16 random 256-B rows per instruction in MFMA operand order (l & 15, l >> 4): a quad touches four rows
//   same      every lane of every instruction reads the same 16 B (pure issue rate)
// Prints bytes / cycle / CU (shader clock from s_memtime is not used: wall clock x 2.4 GHz nominal is misleading under
// DVFS, so the figure is GB/s and B per CU and ns).
//   hipcc -O3 --offload-arch=gfx950 -o vmem_bw vmem_bw.hip && ./vmem_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(512) rd(const float *x, float *sink, unsigned rows_mask, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned seed = (blockIdx.x * 8 + wave) * 2654435761u + 12345u;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      seed = seed * 1664525u + 1013904223u;
      unsigned off;   // in floats
      if (MODE == 0) off = ((seed >> 8) & rows_mask & ~3u) * 64u + lane * 4;              // a 1-KB row = 4 x 256-B rows
      else if (MODE == 1) off = (((seed >> 8) + (lane >> 2) * 7919u) & rows_mask) * 64u + (u & 3) * 16 + (lane & 3) * 4;
      else if (MODE == 2) off = (((seed >> 8) + (lane & 15) * 7919u) & rows_mask) * 64u + (u & 1) * 32 + (lane >> 4) * 8 + ((u >> 1) & 1) * 4;
      else off = ((seed >> 8) & rows_mask) * 64u;
      v[u] = *reinterpret_cast<const f32x4 *>(x + off);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  if (acc.x == 123.456f) sink[threadIdx.x] = acc.y + acc.z + acc.w;
}

template <int MODE>
static int run(const float *x, float *sink, int mb, int waves, const char *name) {
  const unsigned rows = (unsigned)mb * 4096u;   // 256-B rows
  const int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int i = 0; i < 3; ++i) {
    CK(hipEventRecord(e0, nullptr));
    rd<MODE><<<256, 64 * waves>>>(x, sink, rows - 1, iters);
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (i) best = fminf(best, ms);
  }
  const double bytes = 256.0 * waves * iters * 8 * 1024;
  printf("%-8s %4d MB region, %d waves/CU: %.3f ms = %6.2f TB/s = %5.1f B/ns/CU, %5.1f ns per 1-KB instruction per CU\n", name, mb, waves, best,
         bytes / (best * 1e-3) / 1e12, bytes / 256 / (best * 1e6), best * 1e6 / (waves * iters * 8.0));
  return 0;
}
int main() {
  float *x, *sink; CK(hipMalloc(&x, 256u << 20)); CK(hipMalloc(&sink, 4096)); CK(hipMemset(x, 0, 256u << 20));
  for (int mb : {2, 16, 256})
    for (int w : {4, 8}) {
      if (run<0>(x, sink, mb, w, "linear")) return 1;
      if (run<1>(x, sink, mb, w, "rows16")) return 1;
      if (run<2>(x, sink, mb, w, "operand")) return 1;
      if (run<3>(x, sink, mb, w, "same")) return 1;
    }
  return 0;
}
