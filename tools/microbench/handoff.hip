// Stand-alone repro for the cross-workgroup hand-off failure of round 5 (a cluster of workgroups per
// pair exchanged f64 partial sums through memory; next to a SECOND PROCESS on the GPU 0.2-0.9 % of the runs ended a few
// ulps away in every exchange protocol tried).  This program separates the ingredients.  A launch runs G clusters of M
// workgroups (256 threads); per iteration every member publishes NV 64-bit values as tagged 8-byte granules
// {32 data bits, 32-bit tag}, polls the granules of all members, and
//   (1) compares every accepted granule with the value the publisher must have written (integers, recomputed locally):
//       E_mem  = the memory path delivered a value that is not the published one although the tag matched;
//   (2) hands the values of wave 0 to the other waves through LDS behind a workgroup barrier and compares there:
//       E_lds  = a wave read the LDS slot before wave 0 wrote it (a barrier that let a wave through early);
//   (3) adds the exchanged values (as doubles, member order) and compares with the same chain over locally recomputed
//       values: E_arith = arithmetic on correctly delivered values went wrong;
//   (4) watches the wall clock inside the poll loop: a gap of > 100 us between two polls of a resident wave means the
//       wave was context-switched (CWSR) -- `switches` counts them.
// The kernel-only `arith` mode has no exchange at all: per-thread accumulator chains (packed f32 = v_pk_fma_f32, scalar
// f32, f64) over data in memory, workgroup-reduced through LDS, compared bit for bit with the first launch (round 3's
// observation: packed-f32 chains differ next to a second process, f64 and scalar chains do not).
//
//   handoff hammer SECONDS                     competitor: keeps the GPU busy from another process
//   handoff arith LAUNCHES [BLOCKS ITERS]      no exchange (default 512 workgroups x 40 iterations)
//   handoff xchg  LAUNCHES VARIANT             VARIANT bits: 1 agent scope instead of system scope | 2 members of a
//                                              cluster on one XCD (block stride 8) | 4 fine-grained allocation |
//                                              8 release/acquire fences + flag instead of tagged granules |
//                                              16 uncached allocation
// Build: hipcc -O3 --offload-arch=gfx950 handoff.hip -o handoff      (tools/microbench/run_handoff.sh drives it)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

constexpr int NV = 16;           // values per member and iteration
constexpr int MAXM = 8;          // members per cluster at most
constexpr int SLOT = 64;         // granules per member slot (2 * NV used; 512 bytes: no shared cache lines)
constexpr long long SPIN_TICKS = 200000000ll;   // 2 s of the 100-MHz wall clock: give up instead of hanging the box

struct Err { unsigned mem, lds, arith, timeout, switches, maxgap_us; unsigned long long first[8]; };

__device__ __forceinline__ uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}
__device__ __forceinline__ uint64_t value_of(uint32_t nonce, int cluster, int member, int it, int j) {
  return mix(((uint64_t)nonce << 40) ^ ((uint64_t)cluster << 28) ^ ((uint64_t)member << 24) ^ ((uint64_t)it << 5) ^ (uint64_t)j);
}
__device__ __forceinline__ double as_unit_double(uint64_t v) {   // [1, 2): sums round at every add
  return __longlong_as_double((long long)((v >> 12) | 0x3ff0000000000000ull));
}

template <int SCOPE>
__device__ __forceinline__ void st(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE>
__device__ __forceinline__ unsigned long long ld(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }

// VARIANT bit 3 (FENCED): data with plain stores, release fence, flag; reader: flag, acquire fence, plain loads
template <int SCOPE, bool FENCED>
__global__ void __launch_bounds__(256) xchg_kernel(unsigned long long *xbuf, int M, int member_stride, int iters, uint32_t nonce, Err *err) {
  __shared__ unsigned long long lds_vals[2][MAXM * NV];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // cluster c = blocks {c_base + m * member_stride}; member_stride 1: consecutive blocks (round-robin over the XCDs),
  // member_stride 8 with clusters interleaved: all members on XCD (block % 8)
  int cluster, member;
  if (member_stride == 1) { cluster = blockIdx.x / M; member = blockIdx.x % M; }
  else { const int grp = blockIdx.x / (8 * M), r = blockIdx.x % (8 * M); cluster = grp * 8 + (r % 8); member = r / 8; }
  unsigned long long *cb = xbuf + (size_t)cluster * 2 * MAXM * SLOT;
  unsigned mem = 0, ldse = 0, arith = 0, tmo = 0, sw = 0, maxgap = 0;
  unsigned long long firstbad = 0;
  for (int it = 1; it <= iters && !tmo; ++it) {
    unsigned long long *buf = cb + (it & 1) * (MAXM * SLOT);
    const uint32_t tag = (nonce << 20) | (uint32_t)it;
    // publish (wave 0, lanes < NV)
    if (tid < NV) {
      const uint64_t v = value_of(nonce, cluster, member, it, tid);
      if (!FENCED) {
        st<SCOPE>(buf + member * SLOT + 2 * tid, (v & 0xffffffffull) | ((unsigned long long)tag << 32));
        st<SCOPE>(buf + member * SLOT + 2 * tid + 1, (v >> 32) | ((unsigned long long)tag << 32));
      } else {
        buf[member * SLOT + 2 * tid] = v;
      }
    }
    if (FENCED) {
      if (wave == 0) {
        if (SCOPE == __HIP_MEMORY_SCOPE_SYSTEM) __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (tid == 0) st<SCOPE>(buf + member * SLOT + 2 * NV, (unsigned long long)tag);
      }
    }
    // every wave polls for itself: lane l takes values l, l + 64 of the M x NV published ones
    double sum_x = 0.0, sum_l = 0.0;
    for (int base = 0; base < M * NV; base += 64) {
      const int q = base + lane;
      const bool act = q < M * NV;
      const int qm = act ? q / NV : 0, qj = act ? q % NV : 0;
      const unsigned long long *p = buf + qm * SLOT + 2 * qj;
      unsigned long long v0 = 0, v1 = 0;
      long long t_prev = wall_clock64();
      const long long t_start = t_prev;
      for (;;) {
        bool ok;
        if (!FENCED) {
          v0 = ld<SCOPE>(p); v1 = ld<SCOPE>(p + 1);
          ok = !act || ((uint32_t)(v0 >> 32) == tag && (uint32_t)(v1 >> 32) == tag);
        } else {
          const unsigned long long f = ld<SCOPE>(buf + qm * SLOT + 2 * NV);
          ok = !act || (uint32_t)f == tag;
        }
        const long long t = wall_clock64();
        const unsigned gap = (unsigned)((t - t_prev) / 100);
        if (gap > maxgap) maxgap = gap;
        if (gap > 100) ++sw;
        t_prev = t;
        if (__all(ok)) break;
        if (t - t_start > SPIN_TICKS) { tmo = 1; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      if (tmo) break;
      uint64_t got;
      if (FENCED) {
        if (SCOPE == __HIP_MEMORY_SCOPE_SYSTEM) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        got = act ? ((volatile unsigned long long *)p)[0] : 0;
      } else {
        got = (v0 & 0xffffffffull) | (v1 << 32);
      }
      const uint64_t want = act ? value_of(nonce, cluster, qm, it, qj) : 0;
      if (act && got != want) { ++mem; if (!firstbad) firstbad = got ^ want; }
      if (wave == 0 && act) lds_vals[it & 1][q] = got;
      // arithmetic on the delivered values against the same chain on recomputed ones (lane-local, then a wave sum)
      sum_x += as_unit_double(got) * 1.0000001;
      sum_l += as_unit_double(want) * 1.0000001;
    }
    if (tmo) break;
    for (int o = 32; o; o >>= 1) { sum_x += __shfl_xor(sum_x, o); sum_l += __shfl_xor(sum_l, o); }
    if (__double_as_longlong(sum_x) != __double_as_longlong(sum_l) && mem == 0) ++arith;
    // LDS hand-over behind a workgroup barrier (double-buffered by iteration parity: one barrier per iteration)
    __syncthreads();
    if (wave != 0) {
      for (int q = lane; q < M * NV; q += 64)
        if (lds_vals[it & 1][q] != value_of(nonce, cluster, q / NV, it, q % NV)) ++ldse;
    }
  }
  if (mem) atomicAdd(&err->mem, mem);
  if (ldse) atomicAdd(&err->lds, ldse);
  if (arith) atomicAdd(&err->arith, arith);
  if (tmo) atomicAdd(&err->timeout, 1u);
  if (sw) atomicAdd(&err->switches, sw);
  atomicMax(&err->maxgap_us, maxgap);
  if (firstbad) err->first[0] = firstbad;
}

// ---- arithmetic only -------------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) arith_kernel(const float *x, int n, int iters, unsigned long long *out /* [blocks][3] */) {
  __shared__ double slab[2][4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double tot[3] = {0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    f2 acc2 = {0.f, 0.f}; float acc1 = 0.f; double acc8 = 0.0;
    const float *row = x + (size_t)((blockIdx.x * 131 + it * 17) % 64) * n;
    for (int i = tid * 2; i + 1 < n; i += 512) {
      const f2 a = {row[i], row[i + 1]};
      const f2 b = {row[n - 2 - i], row[n - 1 - i]};
      acc2 = __builtin_elementwise_fma(a, b, acc2);   // v_pk_fma_f32 on a loop-carried accumulator pair
      acc1 = __builtin_fmaf(a.x, b.y, acc1);
      acc8 = __builtin_fma((double)a.y, (double)b.x, acc8);
    }
    double v[3] = {(double)acc2.x + (double)acc2.y, (double)acc1, acc8};
    for (int k = 0; k < 3; ++k) for (int o = 32; o; o >>= 1) v[k] += __shfl_xor(v[k], o);
    if (lane == 0) for (int k = 0; k < 3; ++k) slab[it & 1][wave][k] = v[k];
    __syncthreads();
    for (int k = 0; k < 3; ++k) tot[k] += slab[it & 1][0][k] + slab[it & 1][1][k] + slab[it & 1][2][k] + slab[it & 1][3][k];
  }
  if (tid == 0) for (int k = 0; k < 3; ++k) out[blockIdx.x * 3 + k] = (unsigned long long)__double_as_longlong(tot[k]);
}

// ---- competitor ----------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) hammer_kernel(float4 *buf, size_t n4, int rounds) {
  __shared__ float4 stage[512 * 5];    // 40 KB: a few workgroups per CU, registers and LDS both in use
  float4 acc = {0, 0, 0, 0};
  for (int r = 0; r < rounds; ++r)
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 512) {
      float4 v = buf[i];
      stage[threadIdx.x * 5 + (r % 5)] = v;
      __syncthreads();
      const float4 w = stage[((threadIdx.x + 37) % 512) * 5 + (r % 5)];
      acc.x += v.x * w.x; acc.y += v.y * w.y; acc.z += v.z * w.z; acc.w += v.w * w.w;
      __syncthreads();
      buf[i] = acc;
    }
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  if (argc < 3) { printf("usage: handoff hammer S | arith L | xchg L VARIANT\n"); return 1; }
  const char *mode = argv[1];
  if (!strcmp(mode, "hammer")) {
    const double secs = atof(argv[2]);
    const size_t n4 = (size_t)64 << 20;   // 1 GiB
    float4 *buf; CK(hipMalloc(&buf, n4 * 16)); CK(hipMemset(buf, 0, n4 * 16));
    hipStream_t s[3]; for (auto &q : s) CK(hipStreamCreate(&q));
    const double t0 = now_s(); long launches = 0;
    int *h; CK(hipHostMalloc(&h, 4096));
    while (now_s() - t0 < secs) {
      for (int k = 0; k < 3; ++k) {
        hammer_kernel<<<1024 + 512 * k, 512, 0, s[k]>>>(buf + k * (n4 / 4), n4 / 4 / (8 << k), 2);
        CK(hipMemcpyAsync(h + k, buf + k * (n4 / 4), 4, hipMemcpyDeviceToHost, s[k]));
        ++launches;
      }
      CK(hipStreamSynchronize(s[launches % 3]));
    }
    CK(hipDeviceSynchronize());
    printf("hammer: %ld launches in %.1f s\n", launches, now_s() - t0);
    return 0;
  }
  const int L = atoi(argv[2]);
  if (!strcmp(mode, "arith")) {
    // default: 512 workgroups x 40 iterations (the machine to itself); `arith L BLOCKS ITERS`: a FEW long-running workgroups,
    // whose SIMDs a competitor's waves then share (the registration kernel is ONE workgroup, one wave per SIMD, for ~2 ms)
    const int n = 8192, blocks = argc > 3 ? atoi(argv[3]) : 512, iters = argc > 4 ? atoi(argv[4]) : 40;
    std::vector<float> hx((size_t)64 * n);
    uint64_t s = 88172645463325252ull;
    for (auto &v : hx) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (float)((double)(s >> 11) / 9007199254740992.0) - 0.5f; }
    float *x; unsigned long long *out; CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&out, blocks * 3 * 8));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    std::vector<unsigned long long> ref(blocks * 3), cur(blocks * 3);
    long bad[3] = {0, 0, 0}, badl[3] = {0, 0, 0};
    const double t0 = now_s();
    for (int l = 0; l < L; ++l) {
      arith_kernel<<<blocks, 256>>>(x, n, iters, out);
      CK(hipMemcpy(cur.data(), out, cur.size() * 8, hipMemcpyDeviceToHost));
      if (l == 0) { ref = cur; continue; }
      for (int k = 0; k < 3; ++k) {
        long d = 0; for (int b = 0; b < blocks; ++b) d += cur[b * 3 + k] != ref[b * 3 + k];
        bad[k] += d; badl[k] += d != 0;
      }
    }
    printf("arith: %d launches x %d workgroups x %d iterations in %.1f s: workgroup results differing from launch 0 -- "
           "packed f32 %ld (in %ld launches), scalar f32 %ld (%ld), f64 %ld (%ld)\n", L, blocks, iters, now_s() - t0,
           bad[0], badl[0], bad[1], badl[1], bad[2], badl[2]);
    return 0;
  }
  if (!strcmp(mode, "xchg")) {
    const int variant = argc > 3 ? atoi(argv[3]) : 0;
    const int M = 8, G = 16, iters = 200;   // 128 workgroups of 256 threads: co-resident on 256 CUs even next to a competitor
    unsigned long long *xb; Err *err;
    const size_t xbytes = (size_t)G * 2 * MAXM * SLOT * 8;
    if (variant & 4) CK(hipExtMallocWithFlags((void **)&xb, xbytes, hipDeviceMallocFinegrained));
    else if (variant & 16) CK(hipExtMallocWithFlags((void **)&xb, xbytes, hipDeviceMallocUncached));
    else CK(hipMalloc(&xb, xbytes));
    CK(hipMalloc(&err, sizeof(Err)));
    CK(hipMemset(err, 0, sizeof(Err)));
    const int stride = (variant & 2) ? 8 : 1;
    const double t0 = now_s();
    Err h; long bad_launches = 0;
    Err prev; memset(&prev, 0, sizeof prev);
    for (int l = 0; l < L; ++l) {
      CK(hipMemsetAsync(xb, 0, xbytes));
      const uint32_t nonce = (uint32_t)(l + 1) & 0xfff;
      if (variant & 8) {
        if (variant & 1) xchg_kernel<__HIP_MEMORY_SCOPE_AGENT, true><<<G * M, 256>>>(xb, M, stride, iters, nonce, err);
        else xchg_kernel<__HIP_MEMORY_SCOPE_SYSTEM, true><<<G * M, 256>>>(xb, M, stride, iters, nonce, err);
      } else {
        if (variant & 1) xchg_kernel<__HIP_MEMORY_SCOPE_AGENT, false><<<G * M, 256>>>(xb, M, stride, iters, nonce, err);
        else xchg_kernel<__HIP_MEMORY_SCOPE_SYSTEM, false><<<G * M, 256>>>(xb, M, stride, iters, nonce, err);
      }
      if ((l & 63) == 63 || l == L - 1) {
        CK(hipMemcpy(&h, err, sizeof h, hipMemcpyDeviceToHost));
        if (h.timeout) { printf("xchg variant %d: TIMEOUT at launch %d\n", variant, l); break; }
      }
    }
    CK(hipMemcpy(&h, err, sizeof h, hipMemcpyDeviceToHost));
    printf("xchg variant %2d (%s scope, members %s, %s, %s): %d launches x %d clusters x %d members x %d iterations in %.1f s: "
           "E_mem %u  E_lds %u  E_arith %u  timeouts %u  context-switch gaps(>100us) %u  max gap %u us  first xor %016llx\n",
           variant, (variant & 1) ? "agent" : "system", (variant & 2) ? "on one XCD" : "across XCDs",
           (variant & 4) ? "fine-grained" : (variant & 16) ? "uncached" : "coarse-grained", (variant & 8) ? "fence+flag" : "tagged granules",
           L, G, M, iters, now_s() - t0, h.mem, h.lds, h.arith, h.timeout, h.switches, h.maxgap_us, h.first[0]);
    return 0;
  }
  return 1;
}
