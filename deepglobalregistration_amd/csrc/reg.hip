// Confidence gate + weighted Procrustes + robust SE(3) refinement, one workgroup per pair.
// Replaces, for DeepGlobalRegistration.register() step 5 (core/deep_global_registration.py:269-300):
//   * sigmoid / clip / sum gate                     core/deep_global_registration.py:269-281
//   * weighted_procrustes                           core/registration.py:91-113
//   * ortho2rotation / Transformation               core/registration.py:16-64, 116-132
//   * HighDimSmoothL1Loss                           core/loss.py:42-61
//   * GlobalRegistration (Adam 0.1, ExpLR 0.999)    core/registration.py:135-194
//
// The reference runs <=1000 iterations of ~30 tiny autograd kernels with 3 host syncs each.  Here
// the whole optimisation is ONE persistent kernel: correspondences with w > 0 are compacted once
// (clipped weights contribute neither loss nor gradient), every iteration is a single pass over
// them with an analytic gradient (13 block-reduced sums accumulated in f64), and the 9-parameter
// Adam update + the discrete stopping logic run redundantly in every thread.  Per-point
// arithmetic is f32 like the reference; the 3x3 SVD is f64 (one-sided Jacobi) like its LAPACK call.
//
// Round 5: a CLUSTER of 1 .. 8 workgroups per pair (a function of the pair's row count alone, so that a pair's result
// does not depend on its batch).  The iteration is instruction-issue bound on the point pass (one wave per SIMD, ~145
// instructions per row): more waves on the same SIMDs do not help (measured in round 2), more SIMDs do.  Workgroup j of
// a cluster keeps its own share of the rows (256-row chunks j, j + CL, ...), and the cluster exchanges its 13 (pass 1:
// 17) f64 partial sums once per iteration through memory: every workgroup publishes its totals as 8-byte granules
// {32 data bits, 32-bit sequence tag} with system-scope stores (no fence needed: data and tag travel in one word;
// MI355X_MICROARCH.md "handoff-1to1"), polls the granules of all members, and adds them up in member order -- the same
// bits in every workgroup, so all of them take the same Adam step and the same stopping decision.  Deterministic
// (fixed order), independent of scheduling.  A cluster of one skips the exchange and is the round-4 kernel bit for bit.
#include "dgr_internal.h"
#include "svd3.h"

// one wave per SIMD: the 9-parameter update is computed redundantly by every wave, so co-resident
// waves on a SIMD would only serialise it (measured: 512 threads = 10.4 us / iteration)
#ifndef DGR_REG_THREADS
#define DGR_REG_THREADS 256
#endif
constexpr int REG_THREADS = DGR_REG_THREADS;
constexpr int REG_WAVES = REG_THREADS / 64;

constexpr int REG_CLMAX = 8;      // workgroups per pair at most
constexpr int REG_XG = 64;        // granules of a member's slot per exchange buffer: 34 used (17 doubles), padded to 512 bytes
                                  // so that the members' slots -- polled by every member -- do not share cache lines
constexpr int REG_SPIN_LIMIT = 4000000;   // polls of one granule before the kernel gives up (seconds; see reg_poll2)
// workgroups that share a pair: by its row count only (batch-invariant results)
#ifndef DGR_REG_CLUSTER_MAX   // (build-time A/B: -DDGR_REG_CLUSTER_MAX=1 is the one-workgroup kernel of round 4)
#define DGR_REG_CLUSTER_MAX REG_CLMAX
#endif
__host__ __device__ inline int reg_cluster_size(int n) {
  const int cl = n >= 16384 ? 8 : n >= 8192 ? 4 : n >= 4096 ? 2 : 1;
  return cl < DGR_REG_CLUSTER_MAX ? cl : DGR_REG_CLUSTER_MAX;
}

struct RegArgs {
  unsigned long long *xbuf;   // [npairs][2][REG_CLMAX][REG_XG] exchange granules, zeroed before the launch
  int pair_base;              // first pair of this launch (a batch is launched in slices, launch_registration)
  const float *xyz0, *xyz1;
  const int64_t *idx1;  // may be null: xyz1 is already gathered (row aligned with xyz0)
  const float *lw;      // logits (is_logit) or weights
  const int64_t *off0;  // device [npairs+1]
  float *weights_out;   // may be null
  float4 *cA, *cB;      // compacted (x, w) / (y, 0) per pair region
  DgrRegResult *res;
  float clip, q, eps;
  int is_logit, gate, skip_refine, max_iter, max_break;
  double ratio;
  // parity instrumentation (dgr_debug_se3_refine_from; null in the product path): the optimiser state the loop
  // resumes from / ends with -- 30 doubles: prm[9], Adam exp_avg[9], exp_avg_sq[9], iteration, loss_prev, break count
  const double *state_in;
  double *state_out;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
  return v;
}

// Block-wide sum of NV doubles per thread; result visible to every thread in `out`.
template <int NV>
__device__ __forceinline__ void block_sum(const double (&v)[NV], double *lds /* [REG_WAVES*NV + NV] */,
                                          double (&out)[NV]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s = wave_sum(v[i]);
    if (lane == 0) lds[wave * NV + i] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < REG_WAVES; ++w) s += lds[w * NV + threadIdx.x];
    lds[REG_WAVES * NV + threadIdx.x] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) out[i] = lds[REG_WAVES * NV + i];
  __syncthreads();
}

// Per-iteration reduction of <= 16 f64 values per thread in 17 cross-lane exchanges and ONE barrier:
// halving butterfly -- at the xor-32 step each lane hands 8 of its 16 values to its partner and adds
// the 8 it receives, then 4, 2, 1; after four steps every lane owns one value summed over 16 lanes,
// two more steps finish the wave.  Wave results go to a double-buffered LDS slab (`buf` alternates
// per call, which removes the write-after-read barrier); every thread then sums the waves itself.
template <int NV, bool FINISH = true>
__device__ __forceinline__ void block_sum_butterfly(const double (&vin)[NV], double *slab /* [2][REG_WAVES][16] */,
                                                    int buf, double (&out)[NV]) {
  static_assert(NV <= 16, "at most 16 values");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = i < NV ? vin[i] : 0.0;
  const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
  double a[8], b[4], c[2], d;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (b5 ? v[i + 8] : v[i]) + __shfl_xor(b5 ? v[i] : v[i + 8], 32, 64);
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = (b4 ? a[i + 4] : a[i]) + __shfl_xor(b4 ? a[i] : a[i + 4], 16, 64);
#pragma unroll
  for (int i = 0; i < 2; ++i) c[i] = (b3 ? b[i + 2] : b[i]) + __shfl_xor(b3 ? b[i] : b[i + 2], 8, 64);
  d = (b2 ? c[1] : c[0]) + __shfl_xor(b2 ? c[0] : c[1], 4, 64);
  d += __shfl_xor(d, 2, 64);
  d += __shfl_xor(d, 1, 64);
  double *mine = slab + (buf * REG_WAVES + wave) * 16;
  if ((lane & 3) == 0) mine[(b5 ? 8 : 0) + (b4 ? 4 : 0) + (b3 ? 2 : 0) + (b2 ? 1 : 0)] = d;
  __syncthreads();
  if (!FINISH) return;   // cluster path: the caller sums the waves per value (thread i < NV) and exchanges
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < REG_WAVES; ++w) s += slab[(buf * REG_WAVES + w) * 16 + i];
    out[i] = s;
  }
}

// both words of a published double: the two granules are requested together, until both carry the tag
__device__ __forceinline__ double reg_poll2(const unsigned long long *p, uint32_t tag, int &fail) {
  unsigned long long v0, v1;
  int spins = 0;
  for (;;) {
    v0 = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    v1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((uint32_t)(v0 >> 32) == tag && (uint32_t)(v1 >> 32) == tag) break;
    // the members of a cluster are consecutive workgroups of one launch and become resident together; the limit only
    // turns a broken assumption into an error code instead of a hang
    if (++spins > REG_SPIN_LIMIT) { fail = 1; break; }
    __builtin_amdgcn_s_sleep(2);
  }
  return __longlong_as_double((long long)((v1 << 32) | (v0 & 0xffffffffull)));
}

// Cluster-wide sum of the NV workgroup totals tot[0 .. NV) (LDS, written before the last barrier): every thread of
// every member returns with the same out[] = sum over members 0 .. CL-1 in that order.  `seq` counts the exchanges of
// this launch from 1 (tag; buffer = seq & 1: a member is at most one exchange ahead of the slowest one).  xl: LDS
// [REG_CLMAX][17].  One barrier; the caller's next barrier separates these reads of xl from the next exchange's writes.
template <int NV>
__device__ __forceinline__ void cluster_sum(const double *tot, int CL, int j, unsigned long long *xb, uint32_t seq,
                                            double *xl, int *fail_s, double (&out)[NV]) {
  static_assert(2 * NV <= REG_XG && NV <= 17, "granules per member");
  const int tid = threadIdx.x;
  unsigned long long *buf = xb + (seq & 1u) * (REG_CLMAX * REG_XG);
  if (tid < NV) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(tot[tid]);
    const unsigned long long tg = (unsigned long long)seq << 32;
    __hip_atomic_store(buf + j * REG_XG + 2 * tid, (bits & 0xffffffffull) | tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(buf + j * REG_XG + 2 * tid + 1, (bits >> 32) | tg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (tid < CL * NV) {
    const int jj = tid / NV, i = tid - jj * NV;
    int fail = 0;
    // (this member's own totals come back the same way: `tot` is another wave's LDS write, with no barrier in between)
    xl[jj * 17 + i] = reg_poll2(buf + jj * REG_XG + 2 * i, seq, fail);
    if (fail) *fail_s = 1;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double sacc = xl[i];
#pragma unroll
    for (int jj = 1; jj < REG_CLMAX; ++jj)
      if (jj < CL) sacc += xl[jj * 17 + i];
    out[i] = sacc;
  }
}

// ortho2rotation (core/registration.py:16-64) forward; keeps the intermediates for the backward
struct Ortho {
  float x[3], y[3], z[3], b[3];
  float na, nu, coef, inner, norm2;
  bool a_ok, n2_ok, u_ok;
};

__device__ __forceinline__ void ortho_forward(const float p[6], Ortho &o) {
  const float a0 = p[0], a1 = p[1], a2 = p[2];
  o.b[0] = p[3]; o.b[1] = p[4]; o.b[2] = p[5];
  float ra = sqrtf(a0 * a0 + a1 * a1 + a2 * a2);
  o.a_ok = ra >= 1e-8f;
  o.na = fmaxf(ra, 1e-8f);
  o.x[0] = a0 / o.na; o.x[1] = a1 / o.na; o.x[2] = a2 / o.na;
  o.inner = o.x[0] * o.b[0] + o.x[1] * o.b[1] + o.x[2] * o.b[2];
  float n2 = o.x[0] * o.x[0] + o.x[1] * o.x[1] + o.x[2] * o.x[2];
  o.n2_ok = n2 >= 1e-8f;
  o.norm2 = fmaxf(n2, 1e-8f);
  o.coef = o.inner / o.norm2;
  float u0 = o.b[0] - o.coef * o.x[0], u1 = o.b[1] - o.coef * o.x[1], u2 = o.b[2] - o.coef * o.x[2];
  float ru = sqrtf(u0 * u0 + u1 * u1 + u2 * u2);
  o.u_ok = ru >= 1e-8f;
  o.nu = fmaxf(ru, 1e-8f);
  o.y[0] = u0 / o.nu; o.y[1] = u1 / o.nu; o.y[2] = u2 / o.nu;
  o.z[0] = o.x[1] * o.y[2] - o.x[2] * o.y[1];
  o.z[1] = o.x[2] * o.y[0] - o.x[0] * o.y[2];
  o.z[2] = o.x[0] * o.y[1] - o.x[1] * o.y[0];
}

// G[a][b] = dL/dR_ab with R = [x y z] as columns  ->  gradient w.r.t. the 6 parameters
__device__ __forceinline__ void ortho_backward(const Ortho &o, const float G[9], float gp[6]) {
  float gx[3] = {G[0], G[3], G[6]}, gy[3] = {G[1], G[4], G[7]}, gz[3] = {G[2], G[5], G[8]};
  // z = x cross y
  gx[0] += o.y[1] * gz[2] - o.y[2] * gz[1];
  gx[1] += o.y[2] * gz[0] - o.y[0] * gz[2];
  gx[2] += o.y[0] * gz[1] - o.y[1] * gz[0];
  gy[0] += gz[1] * o.x[2] - gz[2] * o.x[1];
  gy[1] += gz[2] * o.x[0] - gz[0] * o.x[2];
  gy[2] += gz[0] * o.x[1] - gz[1] * o.x[0];
  // y = u / max(|u|, 1e-8)
  float gu[3];
  {
    const float d = o.u_ok ? (o.y[0] * gy[0] + o.y[1] * gy[1] + o.y[2] * gy[2]) : 0.f;
    for (int i = 0; i < 3; ++i) gu[i] = (gy[i] - o.y[i] * d) / o.nu;
  }
  // u = b - coef * x
  float gb[3] = {gu[0], gu[1], gu[2]};
  const float gcoef = -(gu[0] * o.x[0] + gu[1] * o.x[1] + gu[2] * o.x[2]);
  for (int i = 0; i < 3; ++i) gx[i] -= o.coef * gu[i];
  // coef = inner / max(x.x, 1e-8)
  const float ginner = gcoef / o.norm2;
  const float gn2 = o.n2_ok ? -gcoef * o.inner / (o.norm2 * o.norm2) : 0.f;
  for (int i = 0; i < 3; ++i) {
    gx[i] += ginner * o.b[i] + 2.f * gn2 * o.x[i];
    gb[i] += ginner * o.x[i];
  }
  // x = a / max(|a|, 1e-8)
  const float d = o.a_ok ? (o.x[0] * gx[0] + o.x[1] * gx[1] + o.x[2] * gx[2]) : 0.f;
  for (int i = 0; i < 3; ++i) gp[i] = (gx[i] - o.x[i] * d) / o.na;
  for (int i = 0; i < 3; ++i) gp[3 + i] = gb[i];
}

// `(X - Y) / quantization_size` (core/loss.py:54): ONE definition for the registration kernel and the debug entry point
// that the golden-vector test of the loss calls.  The reference's CPU path divides (the default here: it is what the
// oracle and the golden vectors do); on its CUDA device ATen multiplies by the f32 reciprocal -- compile with
// -DDGR_REG_RECIP_MUL for that variant (a build-time choice, not a runtime switch).
__device__ __forceinline__ float dgr_residual_scaled(float d, float q) {
#ifdef DGR_REG_RECIP_MUL
  return d * (1.f / q);
#else
  return d / q;
#endif
}

// HighDimSmoothL1Loss per point (core/loss.py:51-61) of s = sum(((X - Y) / q)^2): value and d/ds;
// discontinuous at s == 1 (0.5 vs 0.25) like the reference
__device__ __forceinline__ void dgr_smooth_l1(float s, float eps, float &per, float &dps) {
  if (s < 1.f) {
    per = 0.5f * s;
    dps = 0.5f;
  } else {
    const float rt = sqrtf(s + eps);
    per = 0.5f * (rt - 0.5f);
    dps = 0.25f / rt;
  }
}

__global__ void __launch_bounds__(REG_THREADS) registration_kernel(RegArgs a) {
  __shared__ double red[REG_WAVES * 17 + 17];
  __shared__ double slab[2 * REG_WAVES * 16];
  __shared__ int wave_cnt[REG_WAVES];
  __shared__ float init_Rt[12];
  __shared__ int status_s;
  __shared__ double xl[REG_CLMAX * 17];   // the members' totals of one exchange
  __shared__ double wg_tot[17];           // this workgroup's totals, as published
  __shared__ int fail_s;
  const int p = a.pair_base + blockIdx.x / REG_CLMAX, cj = blockIdx.x % REG_CLMAX;   // pair, member of its cluster
  const int64_t r0 = a.off0[p];
  const int n = (int)(a.off0[p + 1] - r0);
  const int CL = reg_cluster_size(n);
  if (cj >= CL) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // this member's compacted rows: its own region (capacity = its chunks x 256 rows)
  const int chunks = (n + REG_THREADS - 1) / REG_THREADS;
  const int64_t creg = r0 + (int64_t)p * REG_CLMAX * REG_THREADS + (int64_t)cj * ((chunks + CL - 1) / CL) * REG_THREADS;
  float4 *cA = a.cA + creg, *cB = a.cB + creg;
  DgrRegResult *res = a.res + p;
  unsigned long long *xb = a.xbuf + (int64_t)p * (2 * REG_CLMAX * REG_XG);
  uint32_t seq = 0;
  if (tid == 0) fail_s = 0;

  // ---- pass 1: weights (gate), compaction of w > 0 rows, Procrustes sums ---------------------
  double acc[17];
#pragma unroll
  for (int i = 0; i < 17; ++i) acc[i] = 0.0;
  int m = 0;  // compacted count so far (uniform)
  for (int base = cj * REG_THREADS; base < n; base += CL * REG_THREADS) {   // chunks cj, cj + CL, ...
    const int i = base + tid;
    float w = 0.f, x[3] = {0, 0, 0}, y[3] = {0, 0, 0};
    if (i < n) {
      const int64_t r = r0 + i;
      w = a.lw[r];
      if (a.is_logit) {
        w = 1.f / (1.f + expf(-w));
        if (a.clip > 0.f && w < a.clip) w = 0.f;
      }
      if (a.weights_out) a.weights_out[r] = w;
      const int64_t ry = a.idx1 ? a.idx1[r] : r;
#pragma unroll
      for (int d = 0; d < 3; ++d) { x[d] = a.xyz0[r * 3 + d]; y[d] = a.xyz1[ry * 3 + d]; }
      const double wd = w;
      acc[0] += fabs(wd);
      acc[1] += wd;
#pragma unroll
      for (int d = 0; d < 3; ++d) { acc[2 + d] += wd * x[d]; acc[5 + d] += wd * y[d]; }
#pragma unroll
      for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int e = 0; e < 3; ++e) acc[8 + d * 3 + e] += wd * (double)y[d] * (double)x[e];
    }
    // the refinement only ever sees rows with w != 0 (loss and gradient are both scaled by w)
    const bool keep = (i < n) && (w != 0.f);
    const unsigned long long msk = __ballot(keep);
    if (lane == 0) wave_cnt[wave] = __popcll(msk);
    __syncthreads();
    int off = m, tot = 0;
#pragma unroll
    for (int w2 = 0; w2 < REG_WAVES; ++w2) {
      if (w2 < wave) off += wave_cnt[w2];
      tot += wave_cnt[w2];
    }
    if (keep) {
      const int pos = off + __popcll(msk & ((1ull << lane) - 1ull));
      cA[pos] = make_float4(x[0], x[1], x[2], w);
      cB[pos] = make_float4(y[0], y[1], y[2], 0.f);
    }
    m += tot;
    __syncthreads();
  }
  double S[17];
  block_sum<17>(acc, red, S);
  if (CL > 1) {   // the pair's sums: every member's totals, added in member order (the same bits everywhere)
    cluster_sum<17>(red + REG_WAVES * 17, CL, cj, xb, ++seq, xl, &fail_s, S);
    if (fail_s) {
      if (tid == 0 && cj == 0) { res->status = DGR_STATUS_EXCHANGE_TIMEOUT; res->iterations = 0; }
      return;
    }
  }

  // ---- gate + weighted Procrustes (thread 0) ----------------------------------------------------
  if (tid == 0) {
    int status = DGR_STATUS_OK;
    const double wsum = S[1];
    if (a.gate) {
      const double thr = fmax(200.0, 0.05 * (double)n);
      if (!(wsum >= thr)) status = DGR_STATUS_LOW_CONFIDENCE;
    }
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
    if (status == DGR_STATUS_OK) {
      const double inv = 1.0 / (S[0] + (double)a.eps);  // w / (sum|w| + eps)
      const double sn = S[1] * inv;
      double mx[3], my[3], Sxy[9];
      for (int d = 0; d < 3; ++d) { mx[d] = S[2 + d] * inv; my[d] = S[5 + d] * inv; }
      for (int d = 0; d < 3; ++d)
        for (int e = 0; e < 3; ++e) Sxy[d * 3 + e] = S[8 + d * 3 + e] * inv - (2.0 - sn) * my[d] * mx[e];
      bool finite = true;
      for (int i = 0; i < 9; ++i) finite = finite && isfinite(Sxy[i]);
      if (!finite) {
        status = DGR_STATUS_SVD_FAILED;
      } else {
        double U[9], sv[3], V[9];
        svd3(Sxy, U, sv, V);
        const double sg = (det3(U) * det3(V) < 0.0) ? -1.0 : 1.0;
        double Rd[9];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j)
            Rd[i * 3 + j] = U[i * 3 + 0] * V[j * 3 + 0] + U[i * 3 + 1] * V[j * 3 + 1] + sg * U[i * 3 + 2] * V[j * 3 + 2];
        for (int i = 0; i < 9; ++i) R[i] = (float)Rd[i];
        // t = muy - R mux with the f32 R, like the reference
        for (int i = 0; i < 3; ++i)
          t[i] = (float)my[i] - (R[i * 3] * (float)mx[0] + R[i * 3 + 1] * (float)mx[1] + R[i * 3 + 2] * (float)mx[2]);
      }
    }
    for (int i = 0; i < 9; ++i) init_Rt[i] = R[i];
    for (int i = 0; i < 3; ++i) init_Rt[9 + i] = t[i];
    if (cj == 0) {   // (every member computes the same start from the same sums; one writes the record)
      for (int i = 0; i < 9; ++i) res->R[i] = R[i];
      for (int i = 0; i < 3; ++i) res->t[i] = t[i];
      res->wsum = (float)wsum;
      res->status = status;
      res->iterations = 0;
      res->break_count = 0;
      res->loss = 0.f;
    }
    status_s = status;
  }
  __syncthreads();
  if (status_s != DGR_STATUS_OK || a.skip_refine) return;

  // ---- SE(3) refinement: Adam on (rot6d, trans) -------------------------------------------------
  float prm[9];  // rot6d = (R[:,0], R[:,1]) , trans
  prm[0] = init_Rt[0]; prm[1] = init_Rt[3]; prm[2] = init_Rt[6];
  prm[3] = init_Rt[1]; prm[4] = init_Rt[4]; prm[5] = init_Rt[7];
  prm[6] = init_Rt[9]; prm[7] = init_Rt[10]; prm[8] = init_Rt[11];
  float am[9], av[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) { am[i] = 0.f; av[i] = 0.f; }
  const float w1 = (float)S[1];  // loss_fn.w1 = weights.sum(), core/loss.py:48-49
  const float q = a.q;
  const float inv_q = 1.f / q;
#ifdef DGR_REG_TIMING
  long long tsum[3] = {0, 0, 0};
#endif
  double lr = 0.1, b1t = 1.0, b2t = 1.0;
  float loss_prev = 0.f, loss = 0.f;
  int breaks = 0, it = 0, it0 = 0;
  if (a.state_in) {   // resume at iteration it0 from a given optimiser state (the schedules by their own recurrences)
#pragma unroll
    for (int i = 0; i < 9; ++i) { prm[i] = (float)a.state_in[i]; am[i] = (float)a.state_in[9 + i]; av[i] = (float)a.state_in[18 + i]; }
    it0 = (int)a.state_in[27];
    loss_prev = (float)a.state_in[28];
    breaks = (int)a.state_in[29];
    for (int i = 0; i < it0; ++i) { lr *= 0.999; b1t *= 0.9; b2t *= 0.999; }
  }
  for (it = it0; it < a.max_iter; ++it) {
    Ortho o;
    ortho_forward(prm, o);
    const float R00 = o.x[0], R10 = o.x[1], R20 = o.x[2];
    const float R01 = o.y[0], R11 = o.y[1], R21 = o.y[2];
    const float R02 = o.z[0], R12 = o.z[1], R22 = o.z[2];
#ifdef DGR_REG_TIMING
    const long long tc0 = clock64();
#endif
#ifdef DGR_REG_F32_PARTIALS
    // per-thread partial sums in f32 (<= ~100 terms each; the reference sums everything in f32), combined across
    // threads in f64 in a fixed order: 10 % faster per iteration than f64 partials (13 conversions + 13 f64 adds per
    // point less).  NOT the default: with the f32 accumulators the compiler forms packed-f32 chains (v_pk_fma_f32 on
    // loop-carried registers), and that build is reproducible only while the process has the GPU to itself -- next to
    // a second process on the same GPU 4 % of 3000 runs of this kernel on fixed inputs ended a few ulps .. 2e-5 away
    // (the iteration count moving by one), while the f64-partials build and a -fno-slp-vectorize build gave 0 of 3000
    // twice (tools/contention_reg.sh).  The cause below the ISA level was not established.
    float g[13];
#else
    double g[13];
#endif
#pragma unroll
    for (int i = 0; i < 13; ++i) g[i] = 0;
    auto point = [&](const float4 A, const float4 B) {
      const float px = A.x * R00 + A.y * R01 + A.z * R02 + prm[6];
      const float py = A.x * R10 + A.y * R11 + A.z * R12 + prm[7];
      const float pz = A.x * R20 + A.y * R21 + A.z * R22 + prm[8];
      // `(X - Y) / quantization_size` (core/loss.py:54) divides a tensor by a Python scalar: on the reference's CPU
      // path a true division (the default here: it is what the oracle and the golden vectors do), on its CUDA
      // device ATen multiplies by the f32 reciprocal (div_true_kernel_cuda).  -DDGR_REG_RECIP_MUL selects the
      // latter (three ~10-instruction IEEE divisions less per point, -12 % per iteration).  Measured: the two
      // differ by one ulp in s for q = 0.05, which moves points across the s = 1 discontinuity of the loss; after
      // 334 Adam iterations on a 113 k-point pair the result was 1.0e-3 away from the CPU reference with the
      // reciprocal and 3.4e-5 with the division (tests/test_gpu_configs.py, iteration-matched).
      const float rx = dgr_residual_scaled(px - B.x, q), ry = dgr_residual_scaled(py - B.y, q),
                  rz = dgr_residual_scaled(pz - B.z, q);
      const float s = rx * rx + ry * ry + rz * rz;
      float per, dps;
      dgr_smooth_l1(s, a.eps, per, dps);
#ifdef DGR_REG_RECIP_MUL
      const float wk = A.w * dps * 2.f * inv_q;   // DivBackward by the same scalar: again a reciprocal multiply
#else
      const float wk = A.w * dps * 2.f / q;
#endif
      const float gx = wk * rx, gy = wk * ry, gz = wk * rz;
      g[0] += per * A.w;
      g[1] += gx; g[2] += gy; g[3] += gz;
      g[4] += gx * A.x; g[5] += gx * A.y; g[6] += gx * A.z;
      g[7] += gy * A.x; g[8] += gy * A.y; g[9] += gy * A.z;
      g[10] += gz * A.x; g[11] += gz * A.y; g[12] += gz * A.z;
    };
    // four independent row loads in flight per thread (the rows are L2-resident; a dependent
    // one-row-at-a-time loop would pay the L2 latency for every row)
    for (int i = tid; i < m; i += 4 * REG_THREADS) {
      float4 A[4], B[4];
      // unconditional loads (clamped index) so that all eight are in flight together; a guarded load
      // makes the compiler branch and wait per element (measured: ~1000 cycles per row)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = i + j * REG_THREADS;
        const int rc = min(r, m - 1);
        A[j] = cA[rc];
        B[j] = cB[rc];
        if (r >= m) A[j].w = 0.f;   // w = 0 rows contribute exactly 0 to the loss and the gradient
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) point(A[j], B[j]);
    }
#ifdef DGR_REG_TIMING
    const long long tc1 = clock64();
#endif
    double Gs[13];
#ifdef DGR_REG_F32_PARTIALS
    double gd[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) gd[i] = (double)g[i];
    block_sum_butterfly<13>(gd, slab, it & 1, Gs);
#else
    if (CL == 1) {
      block_sum_butterfly<13>(g, slab, it & 1, Gs);
    } else {
      // the waves' sums per value (the same order as above), published, and the members' totals added in member order
      block_sum_butterfly<13, false>(g, slab, it & 1, Gs);
      if (tid < 13) {
        double sw = 0.0;
#pragma unroll
        for (int w = 0; w < REG_WAVES; ++w) sw += slab[((it & 1) * REG_WAVES + w) * 16 + tid];
        wg_tot[tid] = sw;
      }
      // (tid < 13 is one wave: its own LDS writes are visible to its own reads in cluster_sum without a barrier)
      cluster_sum<13>(wg_tot, CL, cj, xb, ++seq, xl, &fail_s, Gs);
      if (fail_s) {
        if (tid == 0 && cj == 0) res->status = DGR_STATUS_EXCHANGE_TIMEOUT;
        return;
      }
    }
#endif
#ifdef DGR_REG_TIMING
    const long long tc2 = clock64();
    tsum[0] += tc1 - tc0; tsum[1] += tc2 - tc1;
#endif
    loss = (float)(Gs[0] / (double)w1);
    if (it == 0) loss_prev = loss;  // loss_prev = loss_fn(T(points), trans_points) before the loop
    if (loss < 1e-7f) break;
    float G[9], grad[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) G[i] = (float)(Gs[4 + i] / (double)w1);
    ortho_backward(o, G, grad);
#pragma unroll
    for (int i = 0; i < 3; ++i) grad[6 + i] = (float)(Gs[1 + i] / (double)w1);
    // torch.optim.Adam (betas 0.9/0.999, eps 1e-8) followed by ExponentialLR(0.999).step()
    b1t *= 0.9;
    b2t *= 0.999;
    const double step_size = lr / (1.0 - b1t);
    const float bc2s = (float)sqrt(1.0 - b2t);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      am[i] = am[i] + (grad[i] - am[i]) * 0.1f;
      av[i] = av[i] * 0.999f + 0.001f * grad[i] * grad[i];
      const float denom = sqrtf(av[i]) / bc2s + 1e-8f;
      prm[i] = prm[i] + (float)(-step_size) * (am[i] / denom);
    }
    lr *= 0.999;
#ifdef DGR_REG_TIMING
    tsum[2] += clock64() - tc2;
#endif
    if (fabs((double)loss_prev - (double)loss) < (double)loss_prev * a.ratio) {
      ++breaks;
      if (breaks >= a.max_break) break;
    }
    loss_prev = loss;
  }
  if (tid == 0 && cj == 0 && a.state_out) {
    for (int i = 0; i < 9; ++i) { a.state_out[i] = prm[i]; a.state_out[9 + i] = am[i]; a.state_out[18 + i] = av[i]; }
    a.state_out[27] = it; a.state_out[28] = loss_prev; a.state_out[29] = breaks;
  }
  if (it >= a.max_iter) it = a.max_iter - 1;  // python's `i` after an exhausted range()
  if (tid == 0 && cj == 0) {
    Ortho o;
    ortho_forward(prm, o);
    res->R[0] = o.x[0]; res->R[1] = o.y[0]; res->R[2] = o.z[0];
    res->R[3] = o.x[1]; res->R[4] = o.y[1]; res->R[5] = o.z[1];
    res->R[6] = o.x[2]; res->R[7] = o.y[2]; res->R[8] = o.z[2];
    res->t[0] = prm[6]; res->t[1] = prm[7]; res->t[2] = prm[8];
    res->iterations = it;
    res->break_count = breaks;
    res->loss = loss;
#ifdef DGR_REG_TIMING
    printf("reg timing: m=%d iters=%d cycles/iter: points %lld reduce %lld update %lld\n", m, it + 1,
           tsum[0] / (it + 1), tsum[1] / (it + 1), tsum[2] / (it + 1));
#endif
  }
}

// needs 2 x float4 x total_rows of scratch; allocated here from the ctx arena.
// off0_host (nullable): the pairs' row offsets on the host, for slicing the batch.  The members of a cluster SPIN while
// they wait for each other, and a member lands on the XCD its workgroup index selects -- exactly one member of an
// 8-cluster per XCD.  Two such launches on different streams could fill each other's slots with spinners (an XCD holds
// 64 workgroups of this kernel) if a launch carried dozens of clusters; a launch therefore carries at most
// REG_MAX_SPINNERS members of multi-member clusters (8 pairs of 27 k rows: at most 8 of an XCD's 64 slots), the rest
// of the batch follows in the next launch on the same stream.  Without the host offsets every pair counts as 8 members.
constexpr int REG_MAX_SPINNERS = 64;
static int launch_registration(dgr_ctx *ctx, const float *xyz0, const float *xyz1, const int64_t *idx1,
                               const float *lw, int is_logit, float clip, const int64_t *off0_dev, const int64_t *off0_host,
                               int npairs, int64_t total_rows, float q, int max_iter, int max_break,
                               double ratio, int skip_refine, int gate, float eps, float *weights_out,
                               DgrRegResult *results_dev, hipStream_t stream, const double *state_in = nullptr,
                               double *state_out = nullptr) {
  RegArgs a;
  a.xyz0 = xyz0; a.xyz1 = xyz1; a.idx1 = idx1; a.lw = lw; a.off0 = off0_dev;
  a.weights_out = weights_out; a.res = results_dev;
  // per-member compaction regions: a pair's rows + one chunk of slack per member
  const int64_t crows = total_rows + (int64_t)npairs * REG_CLMAX * REG_THREADS;
  DGR_ALLOC(a.cA, ctx->arena, float4, crows);
  DGR_ALLOC(a.cB, ctx->arena, float4, crows);
  DGR_ALLOC(a.xbuf, ctx->arena, unsigned long long, (int64_t)npairs * 2 * REG_CLMAX * REG_XG);
  DGR_HIP_CHECK(hipMemsetAsync(a.xbuf, 0, (size_t)npairs * 2 * REG_CLMAX * REG_XG * sizeof(unsigned long long), stream));
  a.clip = clip; a.q = q; a.eps = eps;
  a.is_logit = is_logit; a.gate = gate; a.skip_refine = skip_refine;
  a.max_iter = max_iter; a.max_break = max_break; a.ratio = ratio;
  a.state_in = state_in; a.state_out = state_out;
  for (int p0 = 0; p0 < npairs;) {
    int p1 = p0, spinners = 0;
    while (p1 < npairs && p1 - p0 < 4096) {
      const int cl = off0_host ? reg_cluster_size((int)(off0_host[p1 + 1] - off0_host[p1])) : REG_CLMAX;
      const int add = cl > 1 ? cl : 0;
      if (p1 > p0 && spinners + add > REG_MAX_SPINNERS) break;
      spinners += add;
      ++p1;
    }
    a.pair_base = p0;
    // REG_CLMAX consecutive workgroups per pair; those beyond the pair's cluster size exit at once
    registration_kernel<<<(p1 - p0) * REG_CLMAX, REG_THREADS, 0, stream>>>(a);
    DGR_LAUNCH_CHECK();
    p0 = p1;
  }
  return DGR_OK;
}

int dgr_registration_launch_ctx(dgr_ctx *ctx, const float *xyz0, const float *xyz1, const int64_t *idx1,
                                const float *lw, int is_logit, float clip, const int64_t *off0_dev,
                                const int64_t *off0_host, int npairs, int64_t total_rows, float q, int max_iter, int max_break,
                                double ratio, int skip_refine, int gate, float eps, float *weights_out,
                                DgrRegResult *results_dev, hipStream_t stream) {
  return launch_registration(ctx, xyz0, xyz1, idx1, lw, is_logit, clip, off0_dev, off0_host, npairs, total_rows, q,
                             max_iter, max_break, ratio, skip_refine, gate, eps, weights_out, results_dev,
                             stream);
}

static int run_single(dgr_ctx *ctx, const float *X, const float *Y, const float *w, int64_t N, float q,
                      int max_iter, int max_break, double ratio, int skip_refine, float eps,
                      DgrRegResult *host_res, hipStream_t stream, const double *state_in_host = nullptr,
                      double *state_out_host = nullptr) {
  DGR_REQUIRE(ctx && X && Y && w, "registration: NULL argument");
  DGR_REQUIRE(N > 0 && N < (1ll << 31), "registration: N=%lld out of range", (long long)N);
  DGR_REQUIRE(skip_refine || max_iter >= 1, "GlobalRegistration: max_iter must be >= 1");
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  DGR_CHECK(ctx->arena.reset());
  int64_t *off;
  DgrRegResult *res;
  DGR_ALLOC(off, ctx->arena, int64_t, 2);
  DGR_ALLOC(res, ctx->arena, DgrRegResult, 1);
  const int64_t h_off[2] = {0, N};
  DGR_HIP_CHECK(hipMemcpyAsync(off, h_off, sizeof(h_off), hipMemcpyHostToDevice, stream));
  double *st_in = nullptr, *st_out = nullptr;
  if (state_in_host) {
    DGR_ALLOC(st_in, ctx->arena, double, 30);
    DGR_HIP_CHECK(hipMemcpyAsync(st_in, state_in_host, 30 * sizeof(double), hipMemcpyHostToDevice, stream));
  }
  if (state_out_host) {
    DGR_ALLOC(st_out, ctx->arena, double, 30);
    DGR_HIP_CHECK(hipMemsetAsync(st_out, 0, 30 * sizeof(double), stream));
  }
  DGR_CHECK(launch_registration(ctx, X, Y, nullptr, w, 0, 0.f, off, h_off, 1, N, q, max_iter, max_break, ratio,
                                skip_refine, 0, eps, nullptr, res, stream, st_in, st_out));
  DGR_HIP_CHECK(hipMemcpyAsync(host_res, res, sizeof(DgrRegResult), hipMemcpyDeviceToHost, stream));
  if (state_out_host) DGR_HIP_CHECK(hipMemcpyAsync(state_out_host, st_out, 30 * sizeof(double), hipMemcpyDeviceToHost, stream));
  DGR_HIP_CHECK(hipStreamSynchronize(stream));
  if (host_res->status == DGR_STATUS_EXCHANGE_TIMEOUT) {
    dgr_set_error("registration: the workgroups sharing a pair lost each other (exchange timed out)");
    return DGR_EINTERNAL;
  }
  if (host_res->status == DGR_STATUS_SVD_FAILED) {
    dgr_set_error("weighted Procrustes: non-finite covariance, SVD failed");
    return DGR_ESVD;
  }
  return DGR_OK;
}

extern "C" int dgr_weighted_procrustes(dgr_ctx *ctx, const float *X, const float *Y, const float *w,
                                       int64_t N, float eps, float *R9, float *t3, dgr_stream stream) {
  DGR_REQUIRE(R9 && t3, "dgr_weighted_procrustes: NULL output");
  DgrRegResult r;
  DGR_CHECK(run_single(ctx, X, Y, w, N, 1.f, 1, 1, 0.0, 1, eps, &r, (hipStream_t)stream));
  for (int i = 0; i < 9; ++i) R9[i] = r.R[i];
  for (int i = 0; i < 3; ++i) t3[i] = r.t[i];
  return DGR_OK;
}

extern "C" int dgr_se3_refine(dgr_ctx *ctx, const float *X, const float *Y, const float *w, int64_t N,
                              float quantization_size, int max_iter, int max_break_count,
                              double break_threshold_ratio, float *R9, float *t3, int32_t *iterations,
                              float *loss, int32_t *break_count, dgr_stream stream) {
  DGR_REQUIRE(R9 && t3, "dgr_se3_refine: NULL output");
  DgrRegResult r;
  const float eps = 1.1920928955078125e-07f;  // np.finfo(np.float32).eps, core/loss.py:44
  DGR_CHECK(run_single(ctx, X, Y, w, N, quantization_size, max_iter, max_break_count, break_threshold_ratio, 0,
                       eps, &r, (hipStream_t)stream));
  for (int i = 0; i < 9; ++i) R9[i] = r.R[i];
  for (int i = 0; i < 3; ++i) t3[i] = r.t[i];
  if (iterations) *iterations = r.iterations;
  if (loss) *loss = r.loss;
  if (break_count) *break_count = r.break_count;
  return DGR_OK;
}

// Parity instrumentation: the refinement loop of dgr_se3_refine resumed at iteration state_in[27] from a given
// optimiser state (30 host doubles: prm[9] = rot6d + trans, Adam exp_avg[9], exp_avg_sq[9], iteration, loss_prev, break
// count) and run up to iteration max_iter; the state it ends with in state_out.  tests/helpers.py holds W such steps
// against W steps of the reference algorithm in f32 and in f64 FROM THE SAME STATE (core/registration.py:168-190).
extern "C" int dgr_debug_se3_refine_from(dgr_ctx *ctx, const float *X, const float *Y, const float *w, int64_t N,
                                         float quantization_size, int max_iter, int max_break_count,
                                         double break_threshold_ratio, const double *state_in, double *state_out,
                                         dgr_stream stream) {
  DGR_REQUIRE(state_in && state_out, "dgr_debug_se3_refine_from: NULL state");
  DGR_REQUIRE(state_in[27] >= 0 && state_in[27] < max_iter, "dgr_debug_se3_refine_from: iteration %g outside [0, max_iter)", state_in[27]);
  DgrRegResult r;
  const float eps = 1.1920928955078125e-07f;
  return run_single(ctx, X, Y, w, N, quantization_size, max_iter, max_break_count, break_threshold_ratio, 0, eps, &r,
                    (hipStream_t)stream, state_in, state_out);
}

// ---- debug entry points: the device functions of the registration kernel on their own, so that the parity tests
//      can hold them against the reference-generated vectors directly (tests/golden/ortho6d*.npz, loss.npz) ----
__global__ void debug_ortho_kernel(const float *__restrict__ P, int64_t n, const float *__restrict__ G,
                                   float *__restrict__ R, float *__restrict__ dP) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p[6];
  for (int d = 0; d < 6; ++d) p[d] = P[i * 6 + d];
  Ortho o;
  ortho_forward(p, o);
  for (int a = 0; a < 3; ++a) {   // R = [x y z] as columns
    R[i * 9 + 3 * a + 0] = o.x[a];
    R[i * 9 + 3 * a + 1] = o.y[a];
    R[i * 9 + 3 * a + 2] = o.z[a];
  }
  if (G && dP) {
    float g[9], gp[6];
    for (int d = 0; d < 9; ++d) g[d] = G[i * 9 + d];
    ortho_backward(o, g, gp);
    for (int d = 0; d < 6; ++d) dP[i * 6 + d] = gp[d];
  }
}

__global__ void debug_loss_kernel(const float *__restrict__ X, const float *__restrict__ Y, int64_t n, float q,
                                  float eps, float *__restrict__ per_point) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float rx = dgr_residual_scaled(X[i * 3] - Y[i * 3], q), ry = dgr_residual_scaled(X[i * 3 + 1] - Y[i * 3 + 1], q),
              rz = dgr_residual_scaled(X[i * 3 + 2] - Y[i * 3 + 2], q);
  float per, dps;
  dgr_smooth_l1(rx * rx + ry * ry + rz * rz, eps, per, dps);
  per_point[i] = per;
}

extern "C" int dgr_debug_ortho2rotation(dgr_ctx *ctx, const float *p6, int64_t n, const float *grad_R9, float *R9_out,
                                        float *grad_p6_out, dgr_stream stream) {
  DGR_REQUIRE(ctx && p6 && R9_out && n > 0, "dgr_debug_ortho2rotation: bad argument");
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  debug_ortho_kernel<<<(int)dgr_ceil_div(n, 64), 64, 0, (hipStream_t)stream>>>(p6, n, grad_R9, R9_out, grad_p6_out);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}

extern "C" int dgr_debug_smooth_l1(dgr_ctx *ctx, const float *X, const float *Y, int64_t n, float quantization_size,
                                   float *per_point_out, dgr_stream stream) {
  DGR_REQUIRE(ctx && X && Y && per_point_out && n > 0 && quantization_size > 0, "dgr_debug_smooth_l1: bad argument");
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  debug_loss_kernel<<<(int)dgr_ceil_div(n, 64), 64, 0, (hipStream_t)stream>>>(X, Y, n, quantization_size,
                                                                              1.1920928955078125e-07f, per_point_out);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}
