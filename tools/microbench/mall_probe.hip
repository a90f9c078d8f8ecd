// Does a product-row scratch region that is REUSED stay on the die?  (chunk the 256 -> 256 layers by
// output rows so that a chunk's product rows live in a ~100-MB ring inside the 256-MB Infinity Cache.)  Per region size:
// write it (1-KB rows, 64 lanes x 16 B, plain stores), read it back, write it again, read again -- times and rates; and a
// 2-GB streaming read as the known byte count the rocprofv3 FETCH_SIZE / WRITE_SIZE counters are calibrated on
// (MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request on gfx950; MALL hits "appear to be counted").
//   hipcc -O3 --offload-arch=gfx950 -o mall_probe mall_probe.hip && ./mall_probe
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o f -- ./mall_probe ; rocprofv3 --kernel-trace --pmc WRITE_SIZE ...
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(256) write_rows(float *y, long rows, float tag) {
  const int lane = threadIdx.x & 63;
  const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
  const f32x4 v = {tag, 2.f, 3.f, (float)lane};
  for (long r = w; r < rows; r += nw) *reinterpret_cast<f32x4 *>(y + r * 256 + lane * 4) = v;
}
__global__ void __launch_bounds__(256) read_rows(const float *y, long rows, float *sink) {
  const int lane = threadIdx.x & 63;
  const long w = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long r = w; r + 3 * nw < rows; r += 4 * nw) {   // four independent 1-KB rows in flight per wave
    const f32x4 a = *reinterpret_cast<const f32x4 *>(y + r * 256 + lane * 4);
    const f32x4 b = *reinterpret_cast<const f32x4 *>(y + (r + nw) * 256 + lane * 4);
    const f32x4 c = *reinterpret_cast<const f32x4 *>(y + (r + 2 * nw) * 256 + lane * 4);
    const f32x4 d = *reinterpret_cast<const f32x4 *>(y + (r + 3 * nw) * 256 + lane * 4);
    acc += a + b + c + d;
  }
  if (acc.x == 123456.f) sink[0] = acc.y;   // never true: keeps the loads
}

int main() {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float *sink; CK(hipMalloc(&sink, 64));
  float *big; const long big_bytes = 2l << 30; CK(hipMalloc(&big, big_bytes));
  CK(hipMemset(big, 0, big_bytes));
  auto run = [&](const char *what, auto launch, double bytes) {
    CK(hipEventRecord(e0, nullptr));
    launch();
    CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-28s %8.3f ms  %6.2f TB/s\n", what, ms, bytes / (ms * 1e-3) / 1e12);
    return 0;
  };
  const int grid = 256 * 8;
  printf("calibration: 2-GB streaming read (known bytes: %ld)\n", big_bytes);
  for (int i = 0; i < 2; ++i)
    if (run("read 2 GB", [&] { read_rows<<<grid, 256>>>(big, big_bytes / 1024, sink); }, (double)big_bytes)) return 1;
  for (long mb : {32l, 64l, 100l, 150l, 200l, 400l, 1024l}) {
    const long rows = mb * 1024;   // 1-KB rows
    float *y = big;                 // the region under test: the first `mb` MB of the big buffer
    printf("region %ld MB\n", mb);
    // push whatever the caches hold out of the way first (a read of the far end of the big buffer)
    read_rows<<<grid, 256>>>(big + (1l << 28), (1l << 30) / 1024, sink);
    if (run("write (cold)", [&] { write_rows<<<grid, 256>>>(y, rows, 1.f); }, mb * 1048576.0)) return 1;
    if (run("read back", [&] { read_rows<<<grid, 256>>>(y, rows, sink); }, mb * 1048576.0)) return 1;
    if (run("write again (ring reuse)", [&] { write_rows<<<grid, 256>>>(y, rows, 2.f); }, mb * 1048576.0)) return 1;
    if (run("read back", [&] { read_rows<<<grid, 256>>>(y, rows, sink); }, mb * 1048576.0)) return 1;
    if (run("read again", [&] { read_rows<<<grid, 256>>>(y, rows, sink); }, mb * 1048576.0)) return 1;
  }
  return 0;
}
