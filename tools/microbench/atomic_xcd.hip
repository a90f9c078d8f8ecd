// Microbenchmark: throughput of global_atomic_add_f32 when the rows a workgroup updates are
// (a) spread over the whole array, (b) confined to the eighth of the array that belongs to the
// XCD the block runs on (blockIdx % 8 heuristic), (c) same with the real XCC_ID.
// Build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics atomic_xcd.hip -o atomic_xcd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xf;  // HW_REG_XCC_ID
}

// each block: ITER iterations; per iteration a wave adds 1.0 to 2 rows x 32 floats (like the conv epilogue)
template <int MODE>
__global__ void k(float* out, int n_rows, int C, int iters, unsigned* xcc_hist) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned x = MODE == 2 ? xcc_id() : (blockIdx.x & 7);
  if (threadIdx.x == 0 && xcc_hist) atomicAdd(&xcc_hist[(blockIdx.x & 7) * 8 + (xcc_id() & 7)], 1u);
  unsigned s = blockIdx.x * 9781u + wave * 7919u + 12345u;
  const int part = n_rows / 8;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    unsigned r = (s >> 8);
    int row = MODE == 0 ? (int)(r % (unsigned)n_rows) : (int)(x * part + (r % (unsigned)part));
    row = (row & ~1) + (lane >> 5);                      // two adjacent rows per wave instruction
    for (int c = 0; c < C; c += 32) unsafeAtomicAdd(out + (size_t)row * C + c + (lane & 31), 1.0f);
  }
}

int main() {
  const int n_rows = 1 << 15, C = 256, iters = 2000, blocks = 2048, threads = 256;
  float* out; unsigned* hist;
  hipMalloc(&out, (size_t)n_rows * C * 4); hipMalloc(&hist, 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(out, 0, (size_t)n_rows * C * 4); hipMemset(hist, 0, 256);
      hipEventRecord(e0);
      if (mode == 0) k<0><<<blocks, threads>>>(out, n_rows, C, iters, hist);
      if (mode == 1) k<1><<<blocks, threads>>>(out, n_rows, C, iters, hist);
      if (mode == 2) k<2><<<blocks, threads>>>(out, n_rows, C, iters, hist);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<float> h((size_t)n_rows * C); hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
      double sum = 0; for (float v : h) sum += v;
      double expect = (double)blocks * (threads / 64) * iters * 64.0 * (C / 32);
      printf("mode %d rep %d: %.3f ms, %.1f G atomic-lanes/s, sum ok=%d\n", mode, rep, ms,
             expect / (ms * 1e-3) / 1e9, sum == expect);
    }
  }
  unsigned hh[64]; hipMemcpy(hh, hist, 256, hipMemcpyDeviceToHost);
  printf("blockIdx%%8 (rows) vs XCC_ID (cols):\n");
  for (int i = 0; i < 8; ++i) { for (int j = 0; j < 8; ++j) printf("%5u", hh[i * 8 + j]); printf("\n"); }
  return 0;
}
