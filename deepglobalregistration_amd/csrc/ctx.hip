// Context, error reporting and the grow-only HBM arena of libdgr_hip.so.
#include <stdarg.h>
#include <time.h>
#include <string.h>

#include "dgr_internal.h"

static thread_local char g_err[1024] = "";

void dgr_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *dgr_last_error(void) { return g_err; }
extern "C" const char *dgr_version(void) { return "dgr_hip 0.3 (gfx950)"; }

// ------------------------------------------------------------------------------------------
// arena: sized for 288 GB of HBM -- worst-case kernel-map capacities are reserved instead of
// synchronising with the host to learn exact sizes.
// ------------------------------------------------------------------------------------------
static constexpr size_t kAlign = 256;
static constexpr size_t kMinChunk = (size_t)256 << 20;

void *DgrArena::alloc(size_t bytes) {
  bytes = (bytes + kAlign - 1) / kAlign * kAlign;
  if (bytes == 0) bytes = kAlign;
  while (cur < chunks.size()) {
    if (offset + bytes <= chunks[cur].size) {
      void *p = chunks[cur].base + offset;
      offset += bytes;
      used_total += bytes;
      if (used_total > high_water) high_water = used_total;
      return p;
    }
    ++cur;
    offset = 0;
  }
  size_t sz = bytes > kMinChunk ? bytes : kMinChunk;
  // grow geometrically so that a steady-state call fits one chunk after the next reset
  size_t have = reserved();
  if (sz < have / 2) sz = have / 2;
  char *base = nullptr;
  hipError_t e = hipMalloc((void **)&base, sz);
  if (e != hipSuccess) {
    dgr_set_error("workspace hipMalloc(%zu MiB) failed: %s", sz >> 20, hipGetErrorString(e));
    return nullptr;
  }
  chunks.push_back({base, sz});
  cur = chunks.size() - 1;
  offset = bytes;
  used_total += bytes;
  if (used_total > high_water) high_water = used_total;
  return base;
}

size_t DgrArena::reserved() const {
  size_t s = 0;
  for (auto &c : chunks) s += c.size;
  return s;
}

int DgrArena::reset() {
  if (chunks.size() > 1) {
    // the previous call spilled into several chunks: replace them by one of the high-water size
    DGR_HIP_CHECK(hipDeviceSynchronize());
    for (auto &c : chunks) DGR_HIP_CHECK(hipFree(c.base));
    chunks.clear();
    size_t want = high_water + high_water / 8 + kAlign;
    char *base = nullptr;
    hipError_t e = hipMalloc((void **)&base, want);
    if (e != hipSuccess) {
      dgr_set_error("workspace hipMalloc(%zu MiB) failed: %s", want >> 20, hipGetErrorString(e));
      return DGR_ENOMEM;
    }
    chunks.push_back({base, want});
  }
  cur = 0;
  offset = 0;
  used_total = 0;
  ++generation;
  return DGR_OK;
}

void DgrArena::release() {
  for (auto &c : chunks) (void)hipFree(c.base);
  chunks.clear();
  cur = offset = 0;
}

hipEvent_t DgrEventPool::next() {
  if (used == ev.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) {
      dgr_set_error("hipEventCreate failed");
      return nullptr;
    }
    ev.push_back(e);
  }
  return ev[used++];
}

void DgrEventPool::release() {
  for (auto e : ev) (void)hipEventDestroy(e);
  ev.clear();
  used = 0;
}

// ------------------------------------------------------------------------------------------
extern "C" int dgr_ctx_create(int device, dgr_ctx **out) {
  DGR_REQUIRE(out != nullptr, "dgr_ctx_create: out is NULL");
  int count = 0;
  DGR_HIP_CHECK(hipGetDeviceCount(&count));
  DGR_REQUIRE(device >= 0 && device < count, "dgr_ctx_create: device %d out of range (%d visible)",
              device, count);
  DGR_HIP_CHECK(hipSetDevice(device));
  hipDeviceProp_t prop;
  DGR_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  dgr_ctx *ctx = new dgr_ctx();
  ctx->device = device;
  ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  ctx->all_cus = ctx->num_cus;
  *out = ctx;
  return DGR_OK;
}

extern "C" int dgr_ctx_create_partition_stream(dgr_ctx *ctx, int part, int nparts, dgr_stream *out) {
  DGR_REQUIRE(ctx != nullptr && out != nullptr, "dgr_ctx_create_partition_stream: NULL argument");
  DGR_REQUIRE(nparts == 1 || nparts == 2 || nparts == 4, "partition stream: nparts = %d (1, 2 or 4)", nparts);
  DGR_REQUIRE(part >= 0 && part < nparts, "partition stream: part %d of %d", part, nparts);
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  *out = nullptr;
  if (ctx->part_stream) {
    DGR_HIP_CHECK(hipDeviceSynchronize());
    DGR_HIP_CHECK(hipStreamDestroy(ctx->part_stream));
    ctx->part_stream = nullptr;
    ctx->num_cus = ctx->all_cus;
  }
  if (nparts == 1) return DGR_OK;
  // How a CU mask reaches a multi-XCD part (amdkfd, mqd_symmetrically_map_cu_mask): bit b is slot b / 8 of XCD b % 8.
  // Share p = the slots with slot % nparts == p of every XCD: 16 or 8 CUs of each XCD (all eight L2s stay in use, and the
  // wide conv kernel's "block b runs on XCD b % 8" still holds).  Shares of 8, 16 and 24 slots per XCD ran a persistent
  // one-workgroup-per-CU kernel at the expected rate; shares of 10 and 6 slots ran it 1.6-2x SLOWER than 8 (tools/r06_runs/
  // run37.sh, run38.sh) -- hence only halves and quarters here.
  const int n = ctx->all_cus, xcds = 8;
  DGR_REQUIRE(n % (xcds * 8) == 0, "partition stream: %d compute units are not 8 XCDs x 8 k slots", n);
  const int words = (n + 31) / 32;
  std::vector<uint32_t> mask(words, 0u);
  int mine = 0;
  for (int b = 0; b < n; ++b)
    if ((b / xcds) % nparts == part) {
      mask[b / 32] |= 1u << (b % 32);
      ++mine;
    }
  DGR_HIP_CHECK(hipExtStreamCreateWithCUMask(&ctx->part_stream, (uint32_t)words, mask.data()));
  ctx->num_cus = mine;   // what the context's persistent launches size themselves for
  *out = (dgr_stream)ctx->part_stream;
  return DGR_OK;
}

int dgr_ctx_pinned(dgr_ctx *ctx, size_t bytes, unsigned char **out) {
  if (bytes > ctx->pin_bytes) {
    if (ctx->pin) DGR_HIP_CHECK(hipHostFree(ctx->pin));
    ctx->pin = nullptr;
    ctx->pin_bytes = 0;
    const size_t want = bytes < 4096 ? 4096 : bytes * 2;
    DGR_HIP_CHECK(hipHostMalloc((void **)&ctx->pin, want, hipHostMallocDefault));
    ctx->pin_bytes = want;
  }
  *out = ctx->pin;
  return DGR_OK;
}

int dgr_ctx_wait(dgr_ctx *ctx, hipStream_t stream, long predicted_ns) {
  static const bool spin = getenv("DGR_SPIN_SYNC") != nullptr;
  if (spin) {
    DGR_HIP_CHECK(hipStreamSynchronize(stream));
    return DGR_OK;
  }
  // (hipEventSynchronize on a hipEventBlockingSync event still kept the thread at 100 % here -- measured, round 6: the
  // runtime waits actively unless the DEVICE was created with hipDeviceScheduleBlockingSync, which torch has done before
  // this library is loaded.  So: poll the event, sleeping between polls.)
  if (!ctx->wait_ev) DGR_HIP_CHECK(hipEventCreateWithFlags(&ctx->wait_ev, hipEventDisableTiming));
  DGR_HIP_CHECK(hipEventRecord(ctx->wait_ev, stream));
  // `predicted_ns` > 0 (dgr_register_batch: its previous call's time per input row x this call's rows): ONE sleep for 80 %
  // of it, then polling WITHOUT naps until 110 % of it has passed (a nap costs >= 60 us of timer slack: with naps only, a
  // 6-ms single-pair call took 6.05 ms against 5.86 spinning), then naps.  No prediction: naps of 1/64 of the time waited so
  // far, 10 .. 200 us.  Measured (3 streams x 6 pairs, 49-ms steps): 0.004 s of CPU per step napping, 0.196 s spinning in
  // hipStreamSynchronize, at equal throughput.
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  auto elapsed_ns = [&]() -> long {
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (t1.tv_sec - t0.tv_sec) * 1000000000l + (t1.tv_nsec - t0.tv_nsec);
  };
  if (predicted_ns > 300000) {
    const long ns = predicted_ns / 5 * 4;
    struct timespec first = {ns / 1000000000l, ns % 1000000000l};
    nanosleep(&first, nullptr);
  }
  for (;;) {
    const hipError_t e = hipEventQuery(ctx->wait_ev);
    if (e == hipSuccess) break;
    if (e != hipErrorNotReady) DGR_HIP_CHECK(e);
    const long el = elapsed_ns();
    if (predicted_ns > 300000 && el < predicted_ns + predicted_ns / 10) {   // the last stretch: poll ...
      // ... without naps when the call is short (a nap costs >= 60 us of timer slack: latency of a single pair), with
      // 30-us naps when it is long (a 60-ms batch of a 4-context process kept four threads spinning for 12 ms each)
      if (predicted_ns > 8000000) {
        struct timespec nap = {0, 30000};
        nanosleep(&nap, nullptr);
      }
      continue;
    }
    long ns = el / 64;
    struct timespec nap = {0, ns < 10000 ? 10000 : ns > 200000 ? 200000 : ns};
    nanosleep(&nap, nullptr);
  }
  ctx->last_wait_ns = elapsed_ns();
  return DGR_OK;
}

extern "C" void dgr_ctx_destroy(dgr_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  if (ctx->wait_ev) (void)hipEventDestroy(ctx->wait_ev);
  if (ctx->pin) (void)hipHostFree(ctx->pin);
  if (ctx->part_stream) (void)hipStreamDestroy(ctx->part_stream);
  ctx->arena.release();
  ctx->events.release();
  delete ctx;
}

extern "C" int64_t dgr_ctx_workspace_bytes(dgr_ctx *ctx) {
  return ctx ? (int64_t)ctx->arena.reserved() : 0;
}

extern "C" int dgr_ctx_set_profiling(dgr_ctx *ctx, int enable) {
  DGR_REQUIRE(ctx != nullptr, "ctx is NULL");
  ctx->profiling = enable != 0;
  return DGR_OK;
}

// the version-0.1 entry point under its old name and with its old meaning: exactly the first eight stage times into a bare
// float[8] (a caller built against the 0.1 header keeps working and is never overrun)
extern "C" int dgr_ctx_stage_times(dgr_ctx *ctx, float *times_ms) {
  DGR_REQUIRE(ctx != nullptr && times_ms != nullptr, "bad argument");
  memcpy(times_ms, ctx->stage_ms, (size_t)8 * sizeof(float));
  return DGR_OK;
}

extern "C" int dgr_ctx_stage_times_v2(dgr_ctx *ctx, float *times_ms, int capacity, int *n) {
  DGR_REQUIRE(ctx != nullptr && times_ms != nullptr && capacity >= 0, "bad argument");
  static_assert(sizeof(ctx->stage_ms) == DGR_NUM_STAGE_TIMES * sizeof(float), "stage list and header disagree");
  const int m = capacity < DGR_NUM_STAGE_TIMES ? capacity : DGR_NUM_STAGE_TIMES;
  memcpy(times_ms, ctx->stage_ms, (size_t)m * sizeof(float));
  if (n) *n = m;
  return DGR_OK;
}

extern "C" int64_t dgr_ctx_conv_launches(dgr_ctx *ctx) { return ctx ? ctx->conv_launches : 0; }

extern "C" int dgr_ctx_conv_launch_times(dgr_ctx *ctx, float *times_ms, float *gemm_ms, int64_t capacity, int64_t *n) {
  DGR_REQUIRE(ctx != nullptr && times_ms != nullptr && n != nullptr, "bad argument");
  int64_t m = (int64_t)ctx->conv_span_ms.size();
  if (m > capacity) m = capacity;
  for (int64_t i = 0; i < m; ++i) {
    times_ms[i] = ctx->conv_span_ms[i];
    if (gemm_ms) gemm_ms[i] = i < (int64_t)ctx->gemm_span_ms.size() ? ctx->gemm_span_ms[i] : 0.f;
  }
  *n = m;
  return DGR_OK;
}

extern "C" int dgr_ctx_conv_launch_kernel_us(dgr_ctx *ctx, float *us, int64_t capacity, int64_t *n) {
  DGR_REQUIRE(ctx != nullptr && us != nullptr && n != nullptr, "bad argument");
  int64_t m = (int64_t)ctx->conv_clk_us.size();
  if (m > capacity) m = capacity;
  for (int64_t i = 0; i < m; ++i) us[i] = ctx->conv_clk_us[i];
  *n = m;
  return DGR_OK;
}

extern "C" int dgr_ctx_conv_launch_kinds(dgr_ctx *ctx, char *buf, int64_t capacity, int64_t *n) {
  DGR_REQUIRE(ctx != nullptr && buf != nullptr && n != nullptr && capacity > 0, "bad argument");
  std::string all;
  int64_t m = 0;
  for (const char *k : ctx->conv_kinds) {
    if ((int64_t)(all.size() + strlen(k) + 2) > capacity) break;
    all += k;
    all += '\n';
    ++m;
  }
  memcpy(buf, all.c_str(), all.size() + 1);
  *n = m;
  return DGR_OK;
}
