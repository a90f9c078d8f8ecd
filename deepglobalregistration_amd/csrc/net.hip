// ResUNetBN2C (3-D FCGF net and 6-D inlier net) on top of the sparse-conv kernel.
// Replaces model.load_model('ResUNetBN2C')(...) + load_state_dict + eval + forward:
// constructor model/resunet.py:428-596, channel tables :662-665, forward :598-649, residual block
// model/residual_block.py:83-134, eval batch norm model/common.py:11-21.
//
// Load time: eval-mode batch norm is folded into the kernels (scale) and a per-channel shift; the
// kernels are re-tiled into the MFMA B-operand order documented in conv.hip.
// Forward: every conv is one of the kernel families of conv.hip / conv_os.hip / conv_wide.hip (chosen per layer in
// Fwd::conv); rule-major layers write per-pair product rows that a deterministic reduction sums on top of the folded
// shift (+ residual).  ReLU is never a separate pass: a tensor carries a "ReLU pending" flag and the consumer applies
// max(x,0) while gathering (or the producer bakes it into the split rows it writes for a wide-layer consumer).
// ME.cat is free: producers write into column ranges of a pre-concatenated buffer (row stride = total channels).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <memory>
#include <string>

#include "dgr_internal.h"

static const int CH[5] = {0, 32, 64, 128, 256};     // model/resunet.py:664
static const int TR[5] = {0, 64, 64, 64, 128};      // model/resunet.py:665
static constexpr float BN_EPS = 1e-5f;

struct DgrLayer {
  std::string name;
  int K, cin, cout, cin_pad, cout_pad;
  float *w = nullptr;      // device, tiled
  float *w16 = nullptr;    // device, 16x16x4 fragment order (3-D K = 27 layers: output-stationary conv, conv_os.hip)
  void *w16b = nullptr;    // device, the same as two f16 pieces in 16x16x32 fragment order (conv_os.hip)
  void *w16d = nullptr;    // device, the pieces once more in the channel order of the dense-tile kernel's gather (conv_dense.hip)
  int64_t w16b_piece = 0;
  float *wc = nullptr;     // device, conv1 weights in the operand order of conv1_grid_mfma (3-D conv1 with one input channel)
  float *wq = nullptr;     // device, conv1 weights quad-major for conv_cin6_quad_kernel (6-D conv1 with six input channels)
  void *wb = nullptr;      // device, two f16 pieces in 32x32x16 fragment order (wide layers, conv_wide.hip)
  int64_t wb_piece = 0;    // 16-byte units per piece
  int pieces = 2;          // wb / w16b: two f16 pieces of 2^e W; w_unscale = 2^-e
  float w_unscale = 1.f;
  float *shift = nullptr;  // device [cout] or nullptr
};

struct LayerRun {  // bookkeeping of the last forward, for dgr_net_layer_stats / dgr_net_rerun_layer
  const int32_t *rule_ptr = nullptr;
  const int32_t *n_in = nullptr, *n_out = nullptr;
  int K = 1;
  DgrConvLaunch launch;   // the exact launch of phase 1
  DgrSplitRows split_in, split_out;   // wide-layer kernel: the input as split rows; what the reduction also writes
  int out_relu = 0;
  bool split_only = false;
  bool small_cin = false;  // conv1 ran through the output-stationary kernel instead
  const int32_t *fused_pairs = nullptr;  // conv1 fused with its neighbour search: device pair counter
  bool os = false;                       // ran through the output-stationary kernel
  DgrConvOsLaunch os_launch;
  DgrNbrTable nbr;
  DgrKernelMap km;  // copy (the map set itself lives on the forward's stack)
  bool has_reduce = false;  // phase 2 parameters
  const int32_t *red_ptr = nullptr, *red_pos = nullptr;
  int64_t n_out_cap = 0;
  const float *res = nullptr;
  int res_ld = 0, res_relu = 0;
};

struct DgrTensorRef {
  const float *ptr = nullptr;
  int ld = 0, cols = 0;
  const int32_t *n_dev = nullptr;
  const int32_t *canon = nullptr;   // rows of a coarse 6-D map: library numbering -> first-occurrence numbering
};

// The immutable half of a net: the folded, re-tiled weights in HBM.  Shared (reference-counted) by every dgr_net made
// from it with dgr_net_share -- one context per HIP stream, ONE weight set per device.
struct DgrWeights {
  std::vector<DgrLayer> layers;  // 23 convs in forward order
  int64_t param_bytes = 0;
  int device = 0;
  // Runs on whichever host thread drops the last reference (e.g. Python's GC inside a worker thread): the thread's current
  // device is restored afterwards; the synchronisation stalls every stream of THIS device once, when the last sharer goes.
  ~DgrWeights() {
    int cur = -1;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    for (auto &l : layers) {
      if (l.w) (void)hipFree(l.w);
      if (l.w16) (void)hipFree(l.w16);
      if (l.wb) (void)hipFree(l.wb);
      if (l.wc) (void)hipFree(l.wc);
      if (l.wq) (void)hipFree(l.wq);
      if (l.w16b) (void)hipFree(l.w16b);
      if (l.w16d) (void)hipFree(l.w16d);
      if (l.shift) (void)hipFree(l.shift);
    }
    if (cur >= 0 && cur != device) (void)hipSetDevice(cur);
  }
};

// ... and the per-context half: what the last forward of THIS net object left behind
struct dgr_net {
  dgr_ctx *ctx;
  int D, cin, cout, conv1_ks, normalize;
  std::shared_ptr<DgrWeights> W;
  LayerRun runs[23];
  std::map<std::string, DgrTensorRef> inter;
  uint64_t run_generation = 0;   // arena generation `runs` / `inter` point into
};

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static const dgr_weight_desc *find_desc(const dgr_weight_desc *w, int n, const std::string &name) {
  for (int i = 0; i < n; ++i)
    if (name == w[i].name) return &w[i];
  return nullptr;
}

// ---- the same re-tilings ON THE DEVICE (dgr_net_create_device: the checkpoint's tensors are already in HBM, e.g. out of
// the RCCL broadcast buffer of a multi-GPU start -- no copy back to the host, no host-side loops over 236 M parameters).
// Every kernel below computes element o of a destination layout exactly as the host loops of make_layer do (same index
// arithmetic, the same left-to-right f32 products, the same f16 roundings): the two paths give bit-identical weight sets
// (tests/test_gpu_device_weights.py).
template <class F>
__global__ void __launch_bounds__(256) dgr_fill_kernel(int64_t n, F f) {
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256) f(o);
}
template <class F>
static int dgr_fill(int64_t n, F f) {
  if (n <= 0) return DGR_OK;
  const int64_t blocks = std::min<int64_t>((n + 255) / 256, 1 << 16);
  dgr_fill_kernel<<<(int)blocks, 256>>>(n, f);
  DGR_LAUNCH_CHECK();
  return DGR_OK;
}
__device__ __forceinline__ uint16_t dgr_f16_bits_dev(float x) {
  const _Float16 h = (_Float16)x;
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float dgr_f16_val_dev(float x) { return (float)(_Float16)x; }
__global__ void __launch_bounds__(256) dgr_absmax_scaled_kernel(const float *__restrict__ w, const float *__restrict__ scale,
                                                                int64_t n, int cout, uint32_t *out_bits) {
  float mx = 0.f;
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256)
    mx = fmaxf(mx, fabsf(__fmul_rn(w[o], scale[o % cout])));
  for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(mx));   // non-negative floats order like their bits
}

// make_layer's second half for a kernel tensor that is ALREADY in HBM (src = [K, cin, cout] f32, device): every layout the
// conv kernels read, produced by the fill kernels above; scale / shift = the folded batch norm (host, cout floats).
static int make_layer_device(dgr_net *net, DgrLayer L, const float *src, const std::vector<float> &scale,
                             const std::vector<float> &shift, bool has_shift, const std::string &name) {
  const int K = L.K, cin = L.cin, cout = L.cout;
  static const bool exact_f32 = getenv("DGR_EXACT_F32") != nullptr;
  L.pieces = 2;
  float *sc = nullptr;       // scale[] on the device
  uint32_t *mxb = nullptr;
  DGR_HIP_CHECK(hipMalloc((void **)&sc, (size_t)cout * sizeof(float)));
  DGR_HIP_CHECK(hipMemcpy(sc, scale.data(), (size_t)cout * sizeof(float), hipMemcpyHostToDevice));
  DGR_HIP_CHECK(hipMalloc((void **)&mxb, sizeof(uint32_t)));
  DGR_HIP_CHECK(hipMemset(mxb, 0, sizeof(uint32_t)));
  struct Tmp {   // freed on every exit (after the fill kernels have run: hipFree synchronises)
    float *a; uint32_t *b;
    ~Tmp() { (void)hipFree(a); (void)hipFree(b); }
  } tmp{sc, mxb};
  float w_scale = 1.f;
  {
    const int64_t n = (int64_t)K * cin * cout;
    dgr_absmax_scaled_kernel<<<(int)std::min<int64_t>((n + 255) / 256, 4096), 256>>>(src, sc, n, cout, mxb);
    DGR_LAUNCH_CHECK();
    uint32_t bits = 0;
    DGR_HIP_CHECK(hipMemcpy(&bits, mxb, sizeof(bits), hipMemcpyDeviceToHost));
    float mx;
    memcpy(&mx, &bits, 4);
    int e = 0;
    if (mx > 0.f && std::isfinite(mx)) (void)frexpf(mx, &e);
    e = std::min(std::max(e, -100), 100);
    w_scale = ldexpf(1.f, 15 - e);
    L.w_unscale = ldexpf(1.f, e - 15);
  }
  const bool use_wide = K > 1 && !exact_f32 && dgr_conv_wide_supported(L.cin_pad, cin, cout) && !(net->D == 3 && K == 27);
  if (use_wide) {
    const int S16 = cin / 16, NB32 = cout / 32;
    L.wb_piece = (int64_t)K * S16 * NB32 * 64;
    const int64_t ne = L.wb_piece * 8;
    uint16_t *dst;
    DGR_HIP_CHECK(hipMalloc((void **)&dst, (size_t)2 * ne * sizeof(uint16_t)));
    L.wb = dst;
    DGR_CHECK(dgr_fill(ne, [=] __device__(int64_t o) {
      const int e = (int)(o & 7), lane = (int)((o >> 3) & 63);
      const int64_t r = o >> 9;
      const int nb = (int)(r % NB32), s2 = (int)((r / NB32) % S16), k = (int)(r / ((int64_t)NB32 * S16));
      const int col = 32 * nb + (lane & 31), row = 16 * s2 + 8 * (lane >> 5) + e;
      const float xs = __fmul_rn(__fmul_rn(src[((size_t)k * cin + row) * cout + col], sc[col]), w_scale);
      dst[o] = dgr_f16_bits_dev(xs);
      dst[ne + o] = dgr_f16_bits_dev(xs - dgr_f16_val_dev(xs));
    }));
    net->W->param_bytes += (size_t)2 * ne * sizeof(uint16_t);
  } else {
    const int S = L.cin_pad / 8, NBLK = L.cout_pad / 32;
    const int64_t ne = (int64_t)K * S * NBLK * 256;
    float *dst;
    DGR_HIP_CHECK(hipMalloc((void **)&dst, (size_t)ne * sizeof(float)));
    L.w = dst;
    DGR_CHECK(dgr_fill(ne, [=] __device__(int64_t o) {
      const int c = (int)(o & 3), lane = (int)((o >> 2) & 63);
      const int64_t r = o >> 8;
      const int nb = (int)(r % NBLK), s2 = (int)((r / NBLK) % S), k = (int)(r / ((int64_t)NBLK * S));
      const int row = 8 * s2 + 4 * (lane >> 5) + c, col = 32 * nb + (lane & 31);
      dst[o] = (row < cin && col < cout) ? __fmul_rn(src[((size_t)k * cin + row) * cout + col], sc[col]) : 0.f;
    }));
    net->W->param_bytes += (size_t)ne * sizeof(float);
  }
  if (net->D == 3 && K == 27 && L.cin_pad % 16 == 0 && cout % 32 == 0) {
    const int GT = L.cin_pad / 16, NB = cout / 16;
    {
      const int64_t ne = (int64_t)K * GT * NB * 256;
      float *dst;
      DGR_HIP_CHECK(hipMalloc((void **)&dst, (size_t)ne * sizeof(float)));
      L.w16 = dst;
      DGR_CHECK(dgr_fill(ne, [=] __device__(int64_t o) {
        const int c = (int)(o & 3), lane = (int)((o >> 2) & 63);
        const int64_t r = o >> 8;
        const int jb = (int)(r % NB), g = (int)((r / NB) % GT), k = (int)(r / ((int64_t)NB * GT));
        const int col = 16 * jb + (lane & 15), row = 16 * g + 4 * (lane >> 4) + c;
        dst[o] = row < cin ? __fmul_rn(src[((size_t)k * cin + row) * cout + col], sc[col]) : 0.f;
      }));
      net->W->param_bytes += (size_t)ne * sizeof(float);
    }
    if (cin % 32 == 0) {
      const int S32 = cin / 32;
      L.w16b_piece = (int64_t)K * S32 * NB * 64;
      const int64_t ne = L.w16b_piece * 8;
      uint16_t *pcs;
      DGR_HIP_CHECK(hipMalloc((void **)&pcs, (size_t)2 * ne * sizeof(uint16_t)));
      L.w16b = pcs;
      DGR_CHECK(dgr_fill(ne, [=] __device__(int64_t o) {
        const int e = (int)(o & 7), lane = (int)((o >> 3) & 63);
        const int64_t r = o >> 9;
        const int jb = (int)(r % NB), sI = (int)((r / NB) % S32), k = (int)(r / ((int64_t)NB * S32));
        const int col = 16 * jb + (lane & 15), row = 32 * sI + 8 * (lane >> 4) + e;
        const float xs = __fmul_rn(__fmul_rn(src[((size_t)k * cin + row) * cout + col], sc[col]), w_scale);
        pcs[o] = dgr_f16_bits_dev(xs);
        pcs[ne + o] = dgr_f16_bits_dev(xs - dgr_f16_val_dev(xs));
      }));
      net->W->param_bytes += (size_t)2 * ne * sizeof(uint16_t);
      if (net->D == 3 && K == 27 && (dgr_conv_dense_supported(cin, L.cin_pad, cout) || dgr_conv_up_supported(cin, L.cin_pad, cout))) {
        uint16_t *pd;
        DGR_HIP_CHECK(hipMalloc((void **)&pd, (size_t)2 * ne * sizeof(uint16_t)));
        L.w16d = pd;
        DGR_CHECK(dgr_fill(2 * ne, [=] __device__(int64_t o) {   // (fragments of both pieces alike)
          const int e = (int)(o & 3), h = (int)((o >> 2) & 1);
          const int64_t f = o >> 3, base = f & ~(int64_t)63;
          const int lane = (int)(f & 63), col = lane & 15, lq = lane >> 4;
          const int64_t srcf = base + col + 16 * ((lq >> 1) + 2 * h);
          pd[o] = pcs[srcf * 8 + 4 * (lq & 1) + e];
        }));
        net->W->param_bytes += (size_t)2 * ne * sizeof(uint16_t);
      }
    }
  }
  if (net->D == 3 && name == "conv1" && cin == 1 && cout == 32 && K <= 343) {
    int ks = 1;
    while (ks * ks * ks < K) ++ks;
    const int ks2 = ks * ks, steps = (ks2 + 3) / 4;
    const int64_t ne = (int64_t)ks * steps * 128;
    float *dst;
    DGR_HIP_CHECK(hipMalloc((void **)&dst, (size_t)ne * sizeof(float)));
    L.wc = dst;
    DGR_CHECK(dgr_fill(ne, [=] __device__(int64_t o) {
      const int j = (int)(o & 1), lane = (int)((o >> 1) & 63);
      const int64_t r = o >> 7;
      const int s2 = (int)(r % steps), kz = (int)(r / steps);
      const int kk = 4 * s2 + (lane >> 4), col = (lane & 15) + 16 * j;
      dst[o] = kk < ks2 ? __fmul_rn(src[(size_t)(kz * ks2 + kk) * cout + col], sc[col]) : 0.f;
    }));
  }
  if (name == "conv1" && cin == 6 && cout == 32 && K > 1) {
    const int64_t ne = (int64_t)K * 192;
    float *dst;
    DGR_HIP_CHECK(hipMalloc((void **)&dst, (size_t)ne * sizeof(float)));
    L.wq = dst;
    DGR_CHECK(dgr_fill(ne, [=] __device__(int64_t o) {
      const int e = (int)(o & 3), q = (int)((o >> 2) & 3);
      const int i = (int)((o >> 4) % 12), k = (int)(o / 192);
      const int ci = i >> 1, co = 8 * q + 4 * (i & 1) + e;
      dst[o] = __fmul_rn(src[((size_t)k * cin + ci) * cout + co], sc[co]);
    }));
    net->W->param_bytes += (size_t)ne * sizeof(float);
  }
  if (has_shift) {
    DGR_HIP_CHECK(hipMalloc((void **)&L.shift, cout * sizeof(float)));
    DGR_HIP_CHECK(hipMemcpy(L.shift, shift.data(), cout * sizeof(float), hipMemcpyHostToDevice));
  }
  DGR_HIP_CHECK(hipDeviceSynchronize());   // the fills read sc / the caller's tensors: done before either can go away
  net->W->layers.push_back(L);
  return DGR_OK;
}

static int make_layer(dgr_net *net, const dgr_weight_desc *descs, int nd, const std::string &name, int K,
                      int cin, int cout, const char *bn, bool bias, bool dev = false) {
  DgrLayer L;
  L.name = name;
  L.K = K; L.cin = cin; L.cout = cout;
  L.cin_pad = round_up(cin, 8);
  L.cout_pad = cout <= 32 ? 32 : cout <= 64 ? 64 : cout <= 128 ? 128 : 256;
  // conv1 is the one layer whose Cin the caller chooses: zero-pad to the next width the rule-major kernel is built for
  if (L.cout_pad == 32 && L.cin_pad > 8) {
    DGR_REQUIRE(cin <= 64, "%s: %d input channels (at most 64 into a 32-channel layer)", name.c_str(), cin);
    L.cin_pad = cin <= 32 ? 32 : 64;
  }
  DGR_REQUIRE(cout <= 256 && cin <= 256, "%s: channel count above 256 not supported", name.c_str());
  const dgr_weight_desc *kd = find_desc(descs, nd, name + ".kernel");
  DGR_REQUIRE(kd != nullptr, "state_dict is missing '%s.kernel'", name.c_str());
  // kernel-volume-1 convs are stored [Cin,Cout] by ME 0.5.x and [1,Cin,Cout] by 0.4-era checkpoints
  DGR_REQUIRE(kd->numel == (int64_t)K * cin * cout, "'%s.kernel' has %lld elements, expected %d x %d x %d",
              name.c_str(), (long long)kd->numel, K, cin, cout);
  std::vector<float> scale(cout, 1.f), shift(cout, 0.f);
  bool has_shift = false;
  // the per-channel tensors (batch norm, bias: cout floats each) are folded on the host either way; `dev`: fetched first
  std::vector<std::vector<float>> small;
  auto host_of = [&](const dgr_weight_desc *d) -> const float * {
    if (!dev) return d->data;
    small.emplace_back((size_t)d->numel);
    if (hipMemcpy(small.back().data(), d->data, (size_t)d->numel * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
    return small.back().data();
  };
  if (bn) {
    std::string p = std::string(bn) + ".bn.";
    const dgr_weight_desc *g = find_desc(descs, nd, p + "weight"), *b = find_desc(descs, nd, p + "bias"),
                          *m = find_desc(descs, nd, p + "running_mean"),
                          *v = find_desc(descs, nd, p + "running_var");
    DGR_REQUIRE(g && b && m && v, "state_dict is missing batch-norm tensors '%s*'", p.c_str());
    DGR_REQUIRE(g->numel == cout && b->numel == cout && m->numel == cout && v->numel == cout,
                "batch-norm '%s' has the wrong width", p.c_str());
    const float *gd = host_of(g), *bd2 = host_of(b), *md = host_of(m), *vd = host_of(v);
    DGR_REQUIRE(gd && bd2 && md && vd, "batch-norm '%s': device tensors could not be read", p.c_str());
    for (int c = 0; c < cout; ++c) {
      float s = gd[c] / sqrtf(vd[c] + BN_EPS);
      scale[c] = s;
      shift[c] = bd2[c] - md[c] * s;
    }
    has_shift = true;
  }
  if (bias) {
    const dgr_weight_desc *bd = find_desc(descs, nd, name + ".bias");
    DGR_REQUIRE(bd && bd->numel == cout, "state_dict is missing '%s.bias' [1,%d]", name.c_str(), cout);
    const float *bh = host_of(bd);
    DGR_REQUIRE(bh, "'%s.bias': device tensor could not be read", name.c_str());
    for (int c = 0; c < cout; ++c) shift[c] += bh[c];
    has_shift = true;
  }
  if (dev) return make_layer_device(net, L, kd->data, scale, shift, has_shift, name);
  // DGR_EXACT_F32=1: every conv on v_mfma_f32_*_f32 with the f32 operands themselves (the reference's arithmetic,
  // conv.hip / conv_os.hip) -- the mode the split-operand kernels are measured against.  Default: two f16 pieces per
  // operand under power-of-two scales, three products per MAC (conv_wide.hip, conv_os.hip).
  static const bool exact_f32 = getenv("DGR_EXACT_F32") != nullptr;
  L.pieces = 2;
  // the layer's weight scale: the largest |w| (batch norm folded in) lands in [2^14, 2^15)
  float w_scale = 1.f;
  {
    float mx = 0.f;
    for (int k = 0; k < K; ++k)
      for (int r = 0; r < cin; ++r)
        for (int c = 0; c < cout; ++c) mx = std::max(mx, fabsf(kd->data[((size_t)k * cin + r) * cout + c] * scale[c]));
    int e = 0;
    if (mx > 0.f && std::isfinite(mx)) (void)frexpf(mx, &e);   // mx in [2^(e-1), 2^e)
    e = std::min(std::max(e, -100), 100);
    w_scale = ldexpf(1.f, 15 - e);
    L.w_unscale = ldexpf(1.f, e - 15);
  }
  auto f16_bits = [](float x) { _Float16 h = (_Float16)x; uint16_t b; memcpy(&b, &h, 2); return b; };
  auto f16_val = [](float x) { return (float)(_Float16)x; };
  const bool use_wide = K > 1 && !exact_f32 && dgr_conv_wide_supported(L.cin_pad, cin, cout) && !(net->D == 3 && K == 27);
  if (use_wide) {
    const int S16 = cin / 16, NB32 = cout / 32;
    L.wb_piece = (int64_t)K * S16 * NB32 * 64;
    std::vector<uint16_t> pieces((size_t)2 * L.wb_piece * 8);
    for (int k = 0; k < K; ++k) {
      const float *src = kd->data + (size_t)k * cin * cout;
      for (int s = 0; s < S16; ++s)
        for (int nb = 0; nb < NB32; ++nb)
          for (int lane = 0; lane < 64; ++lane) {
            const int col = 32 * nb + (lane & 31);
            const size_t o = ((((size_t)k * S16 + s) * NB32 + nb) * 64 + lane) * 8;
            for (int e = 0; e < 8; ++e) {
              const float xs = src[(size_t)(16 * s + 8 * (lane >> 5) + e) * cout + col] * scale[col] * w_scale;
              pieces[o + e] = f16_bits(xs);
              pieces[(size_t)L.wb_piece * 8 + o + e] = f16_bits(xs - f16_val(xs));
            }
          }
    }
    DGR_HIP_CHECK(hipMalloc(&L.wb, pieces.size() * sizeof(uint16_t)));
    DGR_HIP_CHECK(hipMemcpy(L.wb, pieces.data(), pieces.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    net->W->param_bytes += pieces.size() * sizeof(uint16_t);
  } else {
  const int S = L.cin_pad / 8, NBLK = L.cout_pad / 32;
  const size_t per_k = (size_t)S * NBLK * 256;
  std::vector<float> tiled((size_t)K * per_k);
  for (int k = 0; k < K; ++k) {
    const float *src = kd->data + (size_t)k * cin * cout;
    float *dst = tiled.data() + (size_t)k * per_k;
    for (int s = 0; s < S; ++s)
      for (int nb = 0; nb < NBLK; ++nb)
        for (int lane = 0; lane < 64; ++lane) {
          const int col = 32 * nb + (lane & 31);
          float *d = dst + ((size_t)(s * NBLK + nb) * 64 + lane) * 4;
          for (int c = 0; c < 4; ++c) {
            const int row = 8 * s + 4 * (lane >> 5) + c;
            d[c] = (row < cin && col < cout) ? src[(size_t)row * cout + col] * scale[col] : 0.f;
          }
        }
  }
  DGR_HIP_CHECK(hipMalloc((void **)&L.w, tiled.size() * sizeof(float)));
  DGR_HIP_CHECK(hipMemcpy(L.w, tiled.data(), tiled.size() * sizeof(float), hipMemcpyHostToDevice));
  net->W->param_bytes += tiled.size() * sizeof(float);
  }
  if (net->D == 3 && K == 27 && L.cin_pad % 16 == 0 && cout % 32 == 0) {
    // second copy in v_mfma_f32_16x16x4_f32 operand order (conv_os.hip): W16[k][g][jb][lane][c]
    const int GT = L.cin_pad / 16, NB = cout / 16;
    std::vector<float> t16((size_t)K * GT * NB * 256);
    for (int k = 0; k < K; ++k) {
      const float *src = kd->data + (size_t)k * cin * cout;
      for (int g = 0; g < GT; ++g)
        for (int jb = 0; jb < NB; ++jb)
          for (int lane = 0; lane < 64; ++lane) {
            float *d = t16.data() + ((((size_t)k * GT + g) * NB + jb) * 64 + lane) * 4;
            const int col = 16 * jb + (lane & 15);
            for (int c = 0; c < 4; ++c) {
              const int row = 16 * g + 4 * (lane >> 4) + c;
              d[c] = row < cin ? src[(size_t)row * cout + col] * scale[col] : 0.f;
            }
          }
    }
    DGR_HIP_CHECK(hipMalloc((void **)&L.w16, t16.size() * sizeof(float)));
    DGR_HIP_CHECK(hipMemcpy(L.w16, t16.data(), t16.size() * sizeof(float), hipMemcpyHostToDevice));
    net->W->param_bytes += t16.size() * sizeof(float);
    if (cin % 32 == 0) {
      // WB[piece][k][s][jb][lane] = 8 bf16 = piece of W[k][32 s + 8 (lane >> 4) + e][16 jb + (lane & 15)]
      const int S32 = cin / 32;
      L.w16b_piece = (int64_t)K * S32 * NB * 64;
      std::vector<uint16_t> pcs((size_t)2 * L.w16b_piece * 8);
      for (int k = 0; k < K; ++k) {
        const float *src = kd->data + (size_t)k * cin * cout;
        for (int sI = 0; sI < S32; ++sI)
          for (int jb = 0; jb < NB; ++jb)
            for (int lane = 0; lane < 64; ++lane) {
              const int col = 16 * jb + (lane & 15);
              const size_t o = ((((size_t)k * S32 + sI) * NB + jb) * 64 + lane) * 8;
              for (int e = 0; e < 8; ++e) {
                const float xs = src[(size_t)(32 * sI + 8 * (lane >> 4) + e) * cout + col] * scale[col] * w_scale;
                pcs[o + e] = f16_bits(xs);
                pcs[(size_t)L.w16b_piece * 8 + o + e] = f16_bits(xs - f16_val(xs));
              }
            }
      }
      DGR_HIP_CHECK(hipMalloc(&L.w16b, pcs.size() * sizeof(uint16_t)));
      DGR_HIP_CHECK(hipMemcpy(L.w16b, pcs.data(), pcs.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      net->W->param_bytes += pcs.size() * sizeof(uint16_t);
      if (net->D == 3 && K == 27 && (dgr_conv_dense_supported(cin, L.cin_pad, cout) || dgr_conv_up_supported(cin, L.cin_pad, cout))) {
        // (conv_up.hip, the transposed convs' kernel, gathers the same way)
        // conv_dense.hip reads its weight operands straight from memory: the quad-coalesced gather hands lane (col, lq)
        // the channels 4 lq .. + 3 and 16 + 4 lq .. + 3 of a 32-channel step, so its 16-byte operand is half (lq & 1) of the
        // natural fragments of lanes (col, lq >> 1) and (col, (lq >> 1) + 2)
        std::vector<uint16_t> pd(pcs.size());
        for (size_t f = 0; f < pcs.size() / 8; ++f) {
          const size_t base = f & ~(size_t)63;
          const int lane = (int)(f & 63), col = lane & 15, lq = lane >> 4;
          for (int h = 0; h < 2; ++h) {
            const size_t srcf = base + col + 16 * ((lq >> 1) + 2 * h);
            for (int e = 0; e < 4; ++e) pd[f * 8 + 4 * h + e] = pcs[srcf * 8 + 4 * (lq & 1) + e];
          }
        }
        DGR_HIP_CHECK(hipMalloc(&L.w16d, pd.size() * sizeof(uint16_t)));
        DGR_HIP_CHECK(hipMemcpy(L.w16d, pd.data(), pd.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        net->W->param_bytes += pd.size() * sizeof(uint16_t);
      }
    }
  }
  if (net->D == 3 && name == "conv1" && cin == 1 && cout == 32 && K <= 343) {
    // conv1_grid_mfma (conv.hip): per z-slab kz and 4-offset step s the A operands of the two v_mfma_f32_16x16x4_f32,
    // wt[kz][s][lane] = {W[k][lane & 15], W[k][16 + (lane & 15)]}, k = kz ks^2 + 4 s + (lane >> 4); zero past the slab
    int ks = 1;
    while (ks * ks * ks < K) ++ks;
    const int ks2 = ks * ks, steps = (ks2 + 3) / 4;
    std::vector<float> wc((size_t)ks * steps * 64 * 2, 0.f);
    for (int kz = 0; kz < ks; ++kz)
      for (int s = 0; s < steps; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const int kk = 4 * s + (lane >> 4);
          if (kk >= ks2) continue;
          const int k = kz * ks2 + kk;
          float *d = wc.data() + (((size_t)kz * steps + s) * 64 + lane) * 2;
          d[0] = kd->data[(size_t)k * cout + (lane & 15)] * scale[lane & 15];
          d[1] = kd->data[(size_t)k * cout + 16 + (lane & 15)] * scale[16 + (lane & 15)];
        }
    DGR_HIP_CHECK(hipMalloc((void **)&L.wc, wc.size() * sizeof(float)));
    DGR_HIP_CHECK(hipMemcpy(L.wc, wc.data(), wc.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if (name == "conv1" && cin == 6 && cout == 32 && K > 1) {
    // conv_cin6_quad_kernel (conv.hip): wq[k][i][q][e] = W[k][i / 2][8 q + 4 (i % 2) + e], batch norm folded in
    std::vector<float> wq((size_t)K * 192);
    for (int k = 0; k < K; ++k)
      for (int i = 0; i < 12; ++i)
        for (int q = 0; q < 4; ++q)
          for (int e = 0; e < 4; ++e) {
            const int ci = i >> 1, co = 8 * q + 4 * (i & 1) + e;
            wq[(size_t)k * 192 + (i * 4 + q) * 4 + e] = kd->data[((size_t)k * cin + ci) * cout + co] * scale[co];
          }
    DGR_HIP_CHECK(hipMalloc((void **)&L.wq, wq.size() * sizeof(float)));
    DGR_HIP_CHECK(hipMemcpy(L.wq, wq.data(), wq.size() * sizeof(float), hipMemcpyHostToDevice));
    net->W->param_bytes += wq.size() * sizeof(float);
  }
  if (has_shift) {
    DGR_HIP_CHECK(hipMalloc((void **)&L.shift, cout * sizeof(float)));
    DGR_HIP_CHECK(hipMemcpy(L.shift, shift.data(), cout * sizeof(float), hipMemcpyHostToDevice));
  }
  net->W->layers.push_back(L);
  return DGR_OK;
}

static int net_create(dgr_ctx *ctx, int D, int in_channels, int out_channels, int conv1_kernel_size, int normalize_feature,
                      const dgr_weight_desc *weights, int n_weights, dgr_net **out, bool dev) {
  DGR_REQUIRE(ctx && weights && out, "dgr_net_create: NULL argument");
  if (dev) {
    // every tensor must really live on this context's device: the fill kernels dereference the pointers
    for (int i = 0; i < n_weights; ++i) {
      hipPointerAttribute_t at;
      const hipError_t e = hipPointerGetAttributes(&at, weights[i].data);
      if (e != hipSuccess) (void)hipGetLastError();
      DGR_REQUIRE(e == hipSuccess && at.type == hipMemoryTypeDevice && at.device == ctx->device,
                  "dgr_net_create_device: '%s' is not a device pointer on device %d (use dgr_net_create for host tensors)",
                  weights[i].name ? weights[i].name : "?", ctx->device);
    }
  }
  DGR_REQUIRE(D == 3 || D == 6, "dgr_net_create: D=%d (ResUNetBN2C is used with D=3 and D=6)", D);
  DGR_REQUIRE(in_channels >= 1 && in_channels <= 256 && out_channels >= 1 && out_channels <= 64,
              "dgr_net_create: unsupported channel counts in=%d out=%d", in_channels, out_channels);
  DGR_REQUIRE(conv1_kernel_size % 2 == 1, "conv1 kernel size must be odd");
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  dgr_net *net = new dgr_net();
  net->ctx = ctx;
  net->W = std::make_shared<DgrWeights>();
  net->W->device = ctx->device;
  net->D = D; net->cin = in_channels; net->cout = out_channels;
  net->conv1_ks = conv1_kernel_size; net->normalize = normalize_feature;
  int k3 = 1, k1 = 1;
  for (int d = 0; d < D; ++d) { k3 *= 3; k1 *= conv1_kernel_size; }
  int rc = DGR_OK;
  auto L = [&](const std::string &name, int K, int ci, int co, const char *bn, bool bias = false) {
    if (rc == DGR_OK) rc = make_layer(net, weights, n_weights, name, K, ci, co, bn, bias, dev);
  };
  auto block = [&](const std::string &b, int c) {
    L(b + ".conv1", k3, c, c, (b + ".norm1").c_str());
    L(b + ".conv2", k3, c, c, (b + ".norm2").c_str());
  };
  L("conv1", k1, in_channels, CH[1], "norm1");          block("block1", CH[1]);
  L("conv2", k3, CH[1], CH[2], "norm2");                block("block2", CH[2]);
  L("conv3", k3, CH[2], CH[3], "norm3");                block("block3", CH[3]);
  L("conv4", k3, CH[3], CH[4], "norm4");                block("block4", CH[4]);
  L("conv4_tr", k3, CH[4], TR[4], "norm4_tr");          block("block4_tr", TR[4]);
  L("conv3_tr", k3, CH[3] + TR[4], TR[3], "norm3_tr");  block("block3_tr", TR[3]);
  L("conv2_tr", k3, CH[2] + TR[3], TR[2], "norm2_tr");  block("block2_tr", TR[2]);
  L("conv1_tr", 1, CH[1] + TR[2], TR[1], nullptr);      // no bias, no BN: residual_block.py:38-44
  L("final", 1, TR[1], out_channels, nullptr, true);    // the only bias: resunet.py:589-596
  if (rc != DGR_OK) {
    dgr_net_destroy(net);
    return rc;
  }
  *out = net;
  return DGR_OK;
}

extern "C" int dgr_net_create(dgr_ctx *ctx, int D, int in_channels, int out_channels,
                              int conv1_kernel_size, int normalize_feature,
                              const dgr_weight_desc *weights, int n_weights, dgr_net **out) {
  return net_create(ctx, D, in_channels, out_channels, conv1_kernel_size, normalize_feature, weights, n_weights, out, false);
}

extern "C" int dgr_net_create_device(dgr_ctx *ctx, int D, int in_channels, int out_channels,
                                     int conv1_kernel_size, int normalize_feature,
                                     const dgr_weight_desc *weights, int n_weights, dgr_net **out) {
  return net_create(ctx, D, in_channels, out_channels, conv1_kernel_size, normalize_feature, weights, n_weights, out, true);
}

extern "C" void dgr_net_destroy(dgr_net *net) {
  if (!net) return;
  delete net;   // the weights go with their last sharer (DgrWeights::~DgrWeights synchronises the device first)
}

extern "C" int dgr_net_share(dgr_ctx *ctx, const dgr_net *src, dgr_net **out) {
  DGR_REQUIRE(ctx && src && out, "dgr_net_share: NULL argument");
  DGR_REQUIRE(ctx->device == src->W->device, "dgr_net_share: the weights live on device %d, the context on device %d",
              src->W->device, ctx->device);
  dgr_net *net = new dgr_net();
  net->ctx = ctx;
  net->D = src->D; net->cin = src->cin; net->cout = src->cout;
  net->conv1_ks = src->conv1_ks; net->normalize = src->normalize;
  net->W = src->W;
  *out = net;
  return DGR_OK;
}

extern "C" int dgr_net_sharers(const dgr_net *net) { return net ? (int)net->W.use_count() : 0; }
extern "C" int64_t dgr_net_param_bytes(const dgr_net *net) { return net ? net->W->param_bytes : 0; }
extern "C" int dgr_net_num_layers(const dgr_net *net) { return net ? (int)net->W->layers.size() : 0; }

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
struct Tensor {
  float *ptr;
  int ld;
  int relu;  // consumers must apply ReLU when reading
  DgrSplitRows split;   // set for tensors a wide layer gathers: written by the tensor's producer (ReLU applied)
  bool split_only = false;   // ... and nobody reads the f32 rows (a block's middle tensor): they are not written
  // 3-D net: per row the bits of its largest |x| (after the pending ReLU), left behind by the tensor's producer for the
  // split-operand consumers (conv_os.hip / conv_dense.hip); amax2: the same array of the concatenation this tensor is a
  // column range of.  Zero-initialised, filled by atomicMax.
  uint32_t *amax = nullptr, *amax2 = nullptr;
  // 3-D net: the rows as ready-made operand pieces of the dense-tile kernel (conv_dense.hip, "dense split rows": 4 x
  // channels bytes per row), written by the tensor's producer; dsplit_only: nobody reads the f32 rows (a block's middle
  // tensor) -- they are not written and `dsplit` is the tensor's own buffer
  unsigned char *dsplit = nullptr;
  bool dsplit_only = false;
};

struct Fwd {
  dgr_ctx *ctx;
  dgr_net *net;
  hipStream_t stream;
  DgrMapSet ms;
  bool prof;
  bool l2_next = false;   // the next identity-map conv writes unit-norm rows (the `final` conv of a normalising net)

  float *ybuf = nullptr;  // per-pair product rows, sized for the largest layer of this forward

  // one conv: Y[pair] = in[pair_in] W[k] (MFMA), then out[o] = shift (+res) + sum of the row's Y rows
  int conv(int li, const Tensor &in, const DgrKernelMap *km, bool swapped, int lvl_in, int lvl_out,
           const Tensor &out, const Tensor *res) {
    const DgrLayer &L = net->W->layers[li];
    const DgrCoordMap &cin_map = ms.cm[lvl_in], &cout_map = ms.cm[lvl_out];
    DgrConvLaunch a;
    a.in = in.ptr; a.in_ld = in.ld; a.in_relu = in.relu;
    a.out = out.ptr; a.out_ld = out.ld;
    a.y = ybuf; a.shift = L.shift;
    a.w = L.w;
    a.cin = L.cin; a.cin_pad = L.cin_pad; a.cout = L.cout; a.cout_pad = L.cout_pad; a.K = L.K;
    if (km && ms.use_nbr) {
      a.pair_in = a.pair_out = a.tile_ptr = a.rule_ptr = nullptr;   // output-stationary path below
      a.tile_desc = nullptr; a.n_rows_dev = nullptr; a.tile_bound = 0;
    } else if (km) {
      a.pair_in = swapped ? km->pair_out : km->pair_in;
      a.pair_out = swapped ? km->pair_in : km->pair_out;
      a.tile_ptr = km->tile_ptr; a.rule_ptr = km->rule_ptr; a.tile_desc = km->tile_desc;
      a.n_rows_dev = nullptr;
      a.tile_bound = km->pair_cap / DGR_TILE_M + km->K;
      DGR_REQUIRE(km->K == L.K, "layer %s: kernel volume mismatch", L.name.c_str());
      DGR_REQUIRE(!swapped || km->in_ptr, "layer %s: map has no in-major CSR", L.name.c_str());
    } else {
      DGR_REQUIRE(res == nullptr, "identity conv with residual not supported");
      a.pair_in = a.pair_out = a.tile_ptr = a.rule_ptr = nullptr;
      a.tile_desc = nullptr;
      a.n_rows_dev = cout_map.n_dev;
      a.tile_bound = dgr_ceil_div(cout_map.n_cap, DGR_TILE_M);
      a.l2_normalize = l2_next ? 1 : 0;
      l2_next = false;
    }
    hipEvent_t e0 = nullptr, em = nullptr, e1 = nullptr;
    if (prof) {
      e0 = ctx->events.next(); em = ctx->events.next(); e1 = ctx->events.next();
      if (!e0 || !em || !e1) return DGR_EHIP;
      DGR_HIP_CHECK(hipEventRecord(e0, stream));
    }
    if (km && ms.use_nbr) {
      // D = 3: output-stationary fused conv over the dense neighbour table of this (in map, out map) pair
      const DgrNbrTable *t = nullptr;
      for (int l = 0; l < 4; ++l)
        if (km == &ms.same[l]) t = &ms.nsame[l];
      for (int l = 0; l < 3; ++l)
        if (km == &ms.down[l]) t = swapped ? &ms.nup[l] : &ms.ndown[l];
      DGR_REQUIRE(t && t->built && L.w16, "layer %s: no neighbour table / 16x16 weights", L.name.c_str());
      DgrConvOsLaunch o;
      o.in = in.ptr; o.in_ld = in.ld; o.in_relu = in.relu;
      o.out = out.ptr; o.out_ld = out.ld; o.out_relu = out.relu;
      // rows per workgroup and input channels per pipeline phase of the list-based kernel, measured per layer shape on
      // the benchmark's clouds (tools/r06_runs/run15-17.sh): the coarse levels have few rows and need small blocks to fill
      // 256 CUs; with Cin >= 128 a phase of 128 channels halves the number of barrier-separated phases per tile, and at
      // level 2 that pays for 64-row blocks (half the weight fragments through the vector-memory path); the one wide layer
      // that writes level 0 (conv2_tr) is faster with 64
      o.rows_per_block = lvl_out <= 1 ? 64 : lvl_out == 2 ? (L.cin_pad >= 128 ? 64 : 32) : 16;
      o.phase_channels = (L.cin_pad >= 256 || (L.cin_pad == 128 && lvl_out >= 2)) ? 128 : 64;
      if (const char *e = getenv(lvl_out == 2 ? "DGR_OS_MB2" : lvl_out == 3 ? "DGR_OS_MB3" : "DGR_OS_MB01")) o.rows_per_block = atoi(e);
      if (const char *e = getenv("DGR_OS_CK")) o.phase_channels = atoi(e);
      o.w16 = L.w16; o.shift = L.shift;
      o.wb3 = L.w16b; o.piece_stride = L.w16b_piece;
      o.wbd = L.w16d;
      o.w_unscale = L.w_unscale; o.n_in_cap = cin_map.n_cap;
      static const bool os_f32 = getenv("DGR_EXACT_F32") != nullptr;   // (conv_os.hip reads the same switch)
      if (L.w16b && !os_f32) {
        o.row_amax = in.amax;
        if (!o.row_amax) {   // a tensor that did not come out of one of the conv kernels (the single-layer debug entry)
          uint32_t *mx;
          DGR_ALLOC(mx, ctx->arena, uint32_t, cin_map.n_cap);
          DGR_CHECK(dgr_row_amax(in.ptr, in.ld, L.cin, in.relu, cin_map.n_dev, cin_map.n_cap, mx, stream));
          o.row_amax = mx;
        }
      }
      o.out_amax = out.amax; o.out_amax2 = out.amax2;
      o.res = res ? res->ptr : nullptr; o.res_ld = res ? res->ld : 0; o.res_relu = res ? res->relu : 0;
      o.nbr = t; o.n_out_dev = cout_map.n_dev; o.n_out_cap = cout_map.n_cap;
      o.cin = L.cin; o.cin_pad = L.cin_pad; o.cout = L.cout;
      // same-stride layers with C <= 64 (the two finest levels of ResUNetBN2C): dense tiles, no pair lists
      static const bool os_lists = getenv("DGR_OS_LISTS") != nullptr;   // A/B + the bit-identity test of the two kernels
      bool same_stride = false;
      for (int l = 0; l < 4; ++l) same_stride = same_stride || t == &ms.nsame[l];
      // (its buffer loads address the input with 32-bit byte offsets: tensors of 2 GB and more stay on the list kernel)
      o.dense = same_stride && o.row_amax && o.wbd && !os_lists && dgr_conv_dense_supported(L.cin, L.cin_pad, L.cout) &&
                cin_map.n_cap * (int64_t)in.ld * 4 < (1ll << 31);
      DGR_REQUIRE(o.dense || (!in.dsplit_only && !out.dsplit_only), "layer %s: dense split rows outside the dense-tile kernel", L.name.c_str());
      // transposed convs (the strided map used swapped) with Cin, Cout multiples of 64: by parity class of the output rows
      static const bool no_up = getenv("DGR_NO_UP") != nullptr;   // A/B
      o.up = swapped && !o.dense && !no_up && !os_lists && o.row_amax && o.wbd && t->perm && res == nullptr &&
             dgr_conv_up_supported(L.cin, L.cin_pad, L.cout) && cin_map.n_cap * (int64_t)in.ld * 4 < (1ll << 31);
      if (o.dense) {
        o.in_dsplit = in.dsplit;
        o.out_dsplit = out.dsplit;
        if (out.dsplit_only) o.out = nullptr;
      }
      const char *kname = "sparse_conv_os";
      DGR_CHECK(dgr_conv_os_launch(o, stream, &kname));
      if (prof) {
        DGR_HIP_CHECK(hipEventRecord(e1, stream));
        ctx->conv_spans.push_back({e0, e1});
        ctx->gemm_spans.push_back({e0, e1});
        ctx->conv_kinds.push_back(kname);
      }
      LayerRun &r = net->runs[li];
      r = LayerRun();
      r.os = true;
      r.os_launch = o;
      r.nbr = *t;
      r.os_launch.nbr = &r.nbr;
      r.n_in = cin_map.n_dev; r.n_out = cout_map.n_dev; r.K = L.K;
      r.n_out_cap = cout_map.n_cap;
      return DGR_OK;
    }
    const bool small_cin = km && !swapped && !res && L.cin <= 8 && L.cout == 32 && L.cin_pad == 8;
    // (Cin = 6 / 1 -- the inlier net's two input widths -- run the thread-per-voxel variant, conv.hip)
    const char *kname = (L.cin == 6 && L.wq) ? "conv_cin6_quad_kernel" : L.cin == 1 ? "conv_small_cin_row_kernel" : "conv_small_cin_kernel";
    unsigned long long *clk = nullptr;
    const bool wide = L.wb && km;
    if (small_cin)
      DGR_CHECK(dgr_conv_small_cin(in.ptr, in.ld, in.relu, L.cin, L.w, L.wq, L.shift, *km, cout_map.n_dev, cout_map.n_cap,
                                   out.ptr, out.ld, stream));
    else if (wide) {
      DGR_REQUIRE(in.split.planes, "layer %s: the wide-layer kernel needs its input as split rows", L.name.c_str());
      if (prof) {   // the kernel stamps its own start / end (dgr_ctx_conv_launch_kernel_us)
        DGR_ALLOC(a.clk, ctx->arena, unsigned long long, 2);
        DGR_HIP_CHECK(hipMemsetAsync(a.clk, 0xff, sizeof(unsigned long long), stream));       // start: atomicMin
        DGR_HIP_CHECK(hipMemsetAsync(a.clk + 1, 0, sizeof(unsigned long long), stream));      // end: atomicMax
        clk = a.clk;
      }
      DGR_CHECK(dgr_conv_wide_launch(a, in.split, L.wb, L.wb_piece, L.w_unscale, ctx->num_cus, stream, &kname));
    } else
      DGR_CHECK(dgr_conv_launch(a, ctx->num_cus, stream, &kname));
    if (prof) DGR_HIP_CHECK(hipEventRecord(em, stream));   // end of the MFMA phase
    DGR_REQUIRE(!out.split.planes || (km && !small_cin), "layer %s: only a reduction can write split rows", L.name.c_str());
    if (km && !small_cin)
      DGR_CHECK(dgr_reduce_rows(ybuf, L.cout, swapped ? km->in_ptr : km->out_ptr, swapped ? km->in_pos : km->out_pos,
                                cout_map.n_dev, cout_map.n_cap, out.split_only ? nullptr : out.ptr, out.ld, L.shift, res ? res->ptr : nullptr,
                                res ? res->ld : 0, res ? res->relu : 0, stream, out.split.planes ? &out.split : nullptr,
                                out.relu));
    if (prof) {
      DGR_HIP_CHECK(hipEventRecord(e1, stream));
      ctx->conv_spans.push_back({e0, e1});
      ctx->gemm_spans.push_back({e0, em});
      ctx->conv_kinds.push_back(kname);
      ctx->conv_clks.resize(ctx->conv_spans.size(), nullptr);
      ctx->conv_clks.back() = clk;
    }
    LayerRun &r = net->runs[li];
    a.clk = nullptr;   // (the recorded launch is re-run by dgr_net_rerun_layer after the arena moved on)
    r.launch = a;
    r.split_in = wide ? in.split : DgrSplitRows();
    r.split_out = out.split;
    r.out_relu = out.relu;
    r.split_only = out.split_only;
    r.has_reduce = km != nullptr;
    r.small_cin = small_cin;
    if (km) r.km = *km;
    if (km) {
      r.red_ptr = swapped ? km->in_ptr : km->out_ptr;
      r.red_pos = swapped ? km->in_pos : km->out_pos;
    }
    r.n_out_cap = cout_map.n_cap;
    r.res = res ? res->ptr : nullptr; r.res_ld = res ? res->ld : 0; r.res_relu = res ? res->relu : 0;
    r.rule_ptr = km ? km->rule_ptr : nullptr;
    r.n_in = cin_map.n_dev; r.n_out = cout_map.n_dev; r.K = L.K;
    return DGR_OK;
  }
};

int dgr_resunet_forward_impl(dgr_ctx *ctx, dgr_net *net, const int32_t *coords, const float *feats,
                             int64_t N, float *out, hipStream_t stream) {
  DgrArena &A = ctx->arena;
  Fwd f;
  f.ctx = ctx; f.net = net; f.stream = stream; f.prof = ctx->profiling;
  hipEvent_t m0 = nullptr, m1 = nullptr;
  if (f.prof) {
    m0 = ctx->events.next(); m1 = ctx->events.next();
    DGR_HIP_CHECK(hipEventRecord(m0, stream));
  }
  f.ms.overflow = ctx->flag_dev;
  // FCGF conv1 (ks^3 offsets, <= 8 input channels) is fused with its neighbour search: no map for it (any odd kernel
  // size <= 7, also 3); such a 3-D net needs no rule-major map at all: the K = 27 layers run output-stationary over
  // dense neighbour tables (conv_os.hip).  Any other 3-D shape takes the rule-major two-phase path of conv.hip.
  const bool use_nbr = net->D == 3 && net->cin <= 8 && net->conv1_ks <= 7;
  const bool conv1_fused = use_nbr;
  DGR_CHECK(dgr_build_maps(A, coords, N, net->D, net->conv1_ks, &f.ms, stream, conv1_fused, /*lean=*/true, use_nbr));
  if (f.prof) {
    DGR_HIP_CHECK(hipEventRecord(m1, stream));
    (net->D == 3 ? ctx->map3_spans : ctx->map6_spans).push_back({m0, m1});
  }
  {
    // Y capacity: the largest (pair capacity x Cout) over the layers of this net
    const DgrMapSet &m = f.ms;
    int64_t need = m.use_nbr ? 64 : m.conv1.pair_cap * 32;
    const int64_t same_c[4] = {64, 64, 128, 256}, down_c[3] = {64, 128, 256};  // widest Cout per map
    for (int l = 0; l < 4; ++l) need = std::max(need, m.same[l].pair_cap * same_c[l]);
    for (int l = 0; l < 3; ++l) need = std::max(need, m.down[l].pair_cap * down_c[l]);
    DGR_ALLOC(f.ybuf, A, float, need);
  }
  const DgrMapSet &ms = f.ms;
  const int64_t n1 = ms.cm[0].n_cap, n2 = ms.cm[1].n_cap, n4 = ms.cm[2].n_cap, n8 = ms.cm[3].n_cap;
  auto buf = [&](int64_t rows, int cols) -> float * { return A.get<float>((size_t)rows * cols); };
  // activations (names follow model/resunet.py:598-649)
  float *t1 = buf(n1, 32), *y1 = buf(n1, 32), *cat1 = buf(n1, 96);
  float *t2 = buf(n2, 64), *y2 = buf(n2, 64), *cat2 = buf(n2, 128);
  float *t4 = buf(n4, 128), *y4 = buf(n4, 128), *cat4 = buf(n4, 256);
  float *t8 = buf(n8, 256), *y8 = buf(n8, 256), *s8 = buf(n8, 256);
  float *u4 = buf(n4, 128), *v4 = buf(n4, 128);
  float *u2 = buf(n2, 64), *v2 = buf(n2, 64);
  float *u1 = buf(n1, 64), *v1 = buf(n1, 64);
  float *h = buf(n1, 64);
  // unit-norm output features (model/resunet.py:643-647): fused into the `final` conv's epilogue when a row's channels
  // fit one 32-channel block (every FCGF checkpoint of the reference: 16 or 32), else a pass of its own
  const bool fuse_l2 = net->normalize && net->cout <= 32;
  float *fin = (net->normalize && !fuse_l2) ? buf(n1, net->cout) : out;
  if (!t1 || !y1 || !cat1 || !t2 || !y2 || !cat2 || !t4 || !y4 || !cat4 || !t8 || !y8 || !s8 || !u4 ||
      !v4 || !u2 || !v2 || !u1 || !v1 || !h || !fin)
    return DGR_ENOMEM;

  const Tensor X{const_cast<float *>(feats), net->cin, 0};
  Tensor T1{t1, 32, 0}, Y1{y1, 32, 1}, S1{cat1 + 64, 96, 1};
  Tensor T2{t2, 64, 0}, Y2{y2, 64, 1};
  Tensor S2{cat2 + 64, 128, 1};
  Tensor T4{t4, 128, 0}, Y4{y4, 128, 1}, S4{cat4 + 128, 256, 1};
  Tensor T8{t8, 256, 0}, Y8{y8, 256, 1}, S8{s8, 256, 1};
  Tensor U4{u4, 128, 0}, V4{v4, 128, 1};
  Tensor U2{u2, 64, 0}, V2{v2, 64, 1};
  Tensor U1{u1, 64, 0}, V1{v1, 64, 1};
  Tensor S4T{cat4, 256, 1};
  // tensors that a wide layer (conv_wide.hip) gathers are also written as split rows by their producer
  {
    // (middle: the tensor between the two convs of a residual block -- its only reader is the block's second conv)
    struct { Tensor *t; int consumer, channels; int64_t rows; bool middle; } sp[] = {
        {&T2, 4, 64, n2, false}, {&Y2, 5, 64, n2, true},
        {&S2, 6, 64, n2, false}, {&T4, 7, 128, n4, false}, {&Y4, 8, 128, n4, true}, {&S4, 9, 128, n4, false},
        {&T8, 10, 256, n8, false}, {&Y8, 11, 256, n8, true}, {&S8, 12, 256, n8, false}, {&U4, 13, 128, n4, false},
        {&V4, 14, 128, n4, true}, {&U2, 16, 64, n2, false}, {&V2, 17, 64, n2, true}, {&U1, 19, 64, n1, false},
        {&V1, 20, 64, n1, true}};
    for (auto &e : sp) {
      if (!net->W->layers[e.consumer].wb) continue;
      e.t->split.channels = e.channels;
      e.t->split_only = e.middle;
      DGR_ALLOC(e.t->split.planes, A, unsigned char, (size_t)e.rows * 4 * e.channels);
      DGR_ALLOC(e.t->split.scale, A, float, e.rows);
    }
  }
  Tensor S2T{cat2, 128, 1};
  const Tensor S1T{cat1, 96, 1};
  const Tensor H{h, 64, 1}, FIN{fin, net->cout, 0};
  Tensor CAT4{cat4, 256, 1}, CAT2{cat2, 128, 1};
  if (use_nbr) {
    // row maxima of every tensor a split-operand 3-D conv reads, from ONE cleared allocation: the producers' epilogues
    // fill them (round 3 ran a row-scale pass per consuming layer: 20 launches per forward)
    const int64_t words = 5 * n1 + 6 * n2 + 6 * n4 + 3 * n8;
    uint32_t *pool;
    DGR_ALLOC(pool, A, uint32_t, words);
    DGR_HIP_CHECK(hipMemsetAsync(pool, 0, (size_t)words * sizeof(uint32_t), stream));
    auto take = [&](int64_t n) { uint32_t *p = pool; pool += n; return p; };
    for (Tensor *t : {&T1, &Y1, &S1, &U1, &V1}) t->amax = take(n1);
    for (Tensor *t : {&T2, &Y2, &S2, &CAT2, &U2, &V2}) t->amax = take(n2);
    for (Tensor *t : {&T4, &Y4, &S4, &CAT4, &U4, &V4}) t->amax = take(n4);
    for (Tensor *t : {&T8, &Y8, &S8}) t->amax = take(n8);
    S2.amax2 = CAT2.amax; S2T.amax2 = CAT2.amax;   // both halves of a concatenation feed its row maxima
    S4.amax2 = CAT4.amax; S4T.amax2 = CAT4.amax;
  }

  if (use_nbr && !getenv("DGR_NO_DSPLIT")) {
    // the middle tensor of a residual block at the two finest levels goes from dense-tile kernel to dense-tile kernel:
    // written as that kernel's operand pieces, and as nothing else (the conditions are those of `o.dense` in Fwd::conv)
    static const bool os_f32 = getenv("DGR_EXACT_F32") != nullptr, os_lists = getenv("DGR_OS_LISTS") != nullptr;
    auto dense_layer = [&](int l, int64_t rows) {
      const DgrLayer &L = net->W->layers[l];
      return L.w16b && L.w16d && !os_f32 && !os_lists && dgr_conv_dense_supported(L.cin, L.cin_pad, L.cout) &&
             rows * (int64_t)L.cin * 4 < (1ll << 31);
    };
    struct { Tensor *t; int producer; int64_t rows; } mid[] = {{&Y1, 1, n1}, {&Y2, 4, n2}, {&V2, 16, n2}, {&V1, 19, n1}};
    for (auto &e : mid)
      if (dense_layer(e.producer, e.rows) && dense_layer(e.producer + 1, e.rows) && e.t->ld == net->W->layers[e.producer].cout) {
        e.t->dsplit = reinterpret_cast<unsigned char *>(e.t->ptr);
        e.t->dsplit_only = true;
      }
  }

  int li = 0;
  // encoder
  if (conv1_fused) {                                                   // conv1 + norm1
    const DgrLayer &L0 = net->W->layers[0];
    int32_t *pc;
    DGR_ALLOC(pc, A, int32_t, 1);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (f.prof) {
      e0 = ctx->events.next(); e1 = ctx->events.next();
      DGR_HIP_CHECK(hipEventRecord(e0, stream));
    }
    const char *k1name = "conv1_grid_kernel";
    DGR_CHECK(dgr_conv1_probe(A, ms.cm[0], net->conv1_ks, feats, net->cin, net->cin, L0.w, L0.shift, t1, 32, pc, stream,
                              L0.wc, &k1name, T1.amax));
    if (f.prof) {
      DGR_HIP_CHECK(hipEventRecord(e1, stream));
      ctx->conv_spans.push_back({e0, e1});
      ctx->gemm_spans.push_back({e0, e1});
      ctx->conv_kinds.push_back(k1name);
    }
    LayerRun &r0 = net->runs[0];
    r0 = LayerRun();
    r0.n_in = r0.n_out = ms.cm[0].n_dev;
    r0.K = L0.K;
    r0.fused_pairs = pc;
    li++;
  } else {
    DGR_CHECK(f.conv(li++, X, &ms.conv1, false, 0, 0, T1, nullptr));
  }
  DGR_CHECK(f.conv(li++, T1, &ms.same[0], false, 0, 0, Y1, nullptr));  // block1
  DGR_CHECK(f.conv(li++, Y1, &ms.same[0], false, 0, 0, S1, &T1));
  DGR_CHECK(f.conv(li++, S1, &ms.down[0], false, 0, 1, T2, nullptr));  // conv2 (stride 2) + norm2
  DGR_CHECK(f.conv(li++, T2, &ms.same[1], false, 1, 1, Y2, nullptr));  // block2
  DGR_CHECK(f.conv(li++, Y2, &ms.same[1], false, 1, 1, S2, &T2));
  DGR_CHECK(f.conv(li++, S2, &ms.down[1], false, 1, 2, T4, nullptr));  // conv3
  DGR_CHECK(f.conv(li++, T4, &ms.same[2], false, 2, 2, Y4, nullptr));  // block3
  DGR_CHECK(f.conv(li++, Y4, &ms.same[2], false, 2, 2, S4, &T4));
  DGR_CHECK(f.conv(li++, S4, &ms.down[2], false, 2, 3, T8, nullptr));  // conv4
  DGR_CHECK(f.conv(li++, T8, &ms.same[3], false, 3, 3, Y8, nullptr));  // block4
  DGR_CHECK(f.conv(li++, Y8, &ms.same[3], false, 3, 3, S8, &T8));
  // decoder: transposed convs reuse the strided maps with in/out swapped
  DGR_CHECK(f.conv(li++, S8, &ms.down[2], true, 3, 2, U4, nullptr));   // conv4_tr + norm4_tr
  DGR_CHECK(f.conv(li++, U4, &ms.same[2], false, 2, 2, V4, nullptr));  // block4_tr
  DGR_CHECK(f.conv(li++, V4, &ms.same[2], false, 2, 2, S4T, &U4));     // -> cat4[:, :128]
  DGR_CHECK(f.conv(li++, CAT4, &ms.down[1], true, 2, 1, U2, nullptr));  // conv3_tr
  DGR_CHECK(f.conv(li++, U2, &ms.same[1], false, 1, 1, V2, nullptr));   // block3_tr
  DGR_CHECK(f.conv(li++, V2, &ms.same[1], false, 1, 1, S2T, &U2));
  DGR_CHECK(f.conv(li++, CAT2, &ms.down[0], true, 1, 0, U1, nullptr));  // conv2_tr
  DGR_CHECK(f.conv(li++, U1, &ms.same[0], false, 0, 0, V1, nullptr));   // block2_tr
  DGR_CHECK(f.conv(li++, V1, &ms.same[0], false, 0, 0, S1T, &U1));
  const Tensor CAT1{cat1, 96, 1};
  DGR_CHECK(f.conv(li++, CAT1, nullptr, false, 0, 0, H, nullptr));      // conv1_tr (k=1), ReLU pending
  f.l2_next = fuse_l2;
  DGR_CHECK(f.conv(li++, H, nullptr, false, 0, 0, FIN, nullptr));       // final (k=1) + bias (+ normalisation)
  if (net->normalize && !fuse_l2)
    DGR_CHECK(dgr_l2_normalize_rows(fin, net->cout, out, net->cout, net->cout, 0, ms.cm[0].n_dev, n1, stream));

  auto &I = net->inter;
  net->run_generation = ctx->arena.generation;
  I.clear();
  I["s1"] = {S1.ptr, 96, 32, ms.cm[0].n_dev};
  I["s2"] = {S2.ptr, 128, 64, ms.cm[1].n_dev, ms.cm[1].canon};
  I["s4"] = {S4.ptr, 256, 128, ms.cm[2].n_dev, ms.cm[2].canon};
  I["s8"] = {S8.ptr, 256, 256, ms.cm[3].n_dev, ms.cm[3].canon};
  I["s4_tr"] = {cat4, 256, 128, ms.cm[2].n_dev, ms.cm[2].canon};
  I["s2_tr"] = {cat2, 128, 64, ms.cm[1].n_dev, ms.cm[1].canon};
  I["s1_tr"] = {cat1, 96, 64, ms.cm[0].n_dev};

  return DGR_OK;
}

void dgr_ctx_begin_profile(dgr_ctx *ctx) {
  ctx->events.used = 0;
  ctx->conv_spans.clear();
  ctx->gemm_spans.clear();
  ctx->conv_kinds.clear();
  ctx->conv_clks.clear();
  ctx->map3_spans.clear();
  ctx->map6_spans.clear();
  ctx->conv_launches = 0;
}

int dgr_ctx_collect_profile(dgr_ctx *ctx) {
  auto total = [](const std::vector<std::pair<hipEvent_t, hipEvent_t>> &v, float *out) -> int {
    float s = 0.f, t = 0.f;
    for (auto &sp : v) {
      DGR_HIP_CHECK(hipEventElapsedTime(&t, sp.first, sp.second));
      s += t;
    }
    *out = s;
    return DGR_OK;
  };
  DGR_CHECK(total(ctx->map3_spans, &ctx->stage_ms[5]));
  DGR_CHECK(total(ctx->map6_spans, &ctx->stage_ms[6]));
  DGR_CHECK(total(ctx->conv_spans, &ctx->stage_ms[7]));
  ctx->conv_launches = (int64_t)ctx->conv_spans.size();
  ctx->conv_span_ms.clear();
  ctx->gemm_span_ms.clear();
  for (auto &sp : ctx->conv_spans) {
    float t = 0.f;
    DGR_HIP_CHECK(hipEventElapsedTime(&t, sp.first, sp.second));
    ctx->conv_span_ms.push_back(t);
  }
  for (auto &sp : ctx->gemm_spans) {
    float t = 0.f;
    DGR_HIP_CHECK(hipEventElapsedTime(&t, sp.first, sp.second));
    ctx->gemm_span_ms.push_back(t);
  }
  ctx->conv_clk_us.assign(ctx->conv_spans.size(), 0.f);
  for (size_t i = 0; i < ctx->conv_clks.size() && i < ctx->conv_clk_us.size(); ++i)
    if (ctx->conv_clks[i]) {
      unsigned long long c[2] = {0, 0};
      DGR_HIP_CHECK(hipMemcpy(c, ctx->conv_clks[i], sizeof(c), hipMemcpyDeviceToHost));
      if (c[1] > c[0]) ctx->conv_clk_us[i] = (float)((double)(c[1] - c[0]) * 0.01);   // 100-MHz ticks
    }
  return DGR_OK;
}

int dgr_flag_error(int32_t flag) {
  if (flag == 1) {
    dgr_set_error("duplicate coordinates in the sparse tensor input");
    return DGR_EINVAL;
  }
  if (flag == 2) {
    dgr_set_error("kernel-map capacity exceeded (more than the reserved pairs per output row)");
    return DGR_ENOMEM;
  }
  return DGR_OK;
}

static int check_flag(dgr_ctx *ctx, const int32_t *flag_dev, hipStream_t stream) {
  // the flag word lands in pinned host memory: the copy is asynchronous and goes in FRONT of the call's one wait (a
  // pageable destination made the copy block inside the runtime until the stream had drained, and a second
  // synchronisation followed it)
  unsigned char *pin;
  DGR_CHECK(dgr_ctx_pinned(ctx, 64, &pin));
  DGR_HIP_CHECK(hipMemcpyAsync(pin, flag_dev, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  DGR_CHECK(dgr_ctx_wait(ctx, stream));
  return dgr_flag_error(*reinterpret_cast<volatile int32_t *>(pin));
}

int dgr_ctx_new_flag(dgr_ctx *ctx, hipStream_t stream) {
  DGR_ALLOC(ctx->flag_dev, ctx->arena, int32_t, 1);
  DGR_HIP_CHECK(hipMemsetAsync(ctx->flag_dev, 0, sizeof(int32_t), stream));
  return DGR_OK;
}

int dgr_ctx_check_flag(dgr_ctx *ctx, hipStream_t stream) {
  if (!ctx->flag_dev) return DGR_OK;
  return check_flag(ctx, ctx->flag_dev, stream);
}

// the fused pipeline rewinds the arena region of a forward and reuses it: its maps / activations are gone
void dgr_net_invalidate_runs(dgr_net *net) { net->run_generation = ~0ull; }
int dgr_net_out_channels(const dgr_net *net) { return net->cout; }
int dgr_net_in_channels(const dgr_net *net) { return net->cin; }
int dgr_net_dim(const dgr_net *net) { return net->D; }
const dgr_ctx *dgr_net_ctx(const dgr_net *net) { return net->ctx; }

extern "C" int dgr_resunet_forward(dgr_ctx *ctx, dgr_net *net, const int32_t *coords, const float *feats,
                                   int64_t N, float *out, dgr_stream stream_) {
  DGR_REQUIRE(ctx && net && coords && feats && out, "dgr_resunet_forward: NULL argument");
  DGR_REQUIRE(net->ctx == ctx, "dgr_resunet_forward: the net object belongs to another context (one dgr_net per context: "
                               "dgr_net_share gives a second context its own over the same weights)");
  DGR_REQUIRE(N > 0, "dgr_resunet_forward: empty sparse tensor (N=%lld)", (long long)N);
  hipStream_t stream = (hipStream_t)stream_;
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  DGR_CHECK(ctx->arena.reset());
  DGR_CHECK(dgr_ctx_new_flag(ctx, stream));
  dgr_ctx_begin_profile(ctx);
  DGR_CHECK(dgr_resunet_forward_impl(ctx, net, coords, feats, N, out, stream));
  DGR_CHECK(dgr_ctx_check_flag(ctx, stream));  // synchronises the stream
  if (ctx->profiling) {
    memset(ctx->stage_ms, 0, sizeof(ctx->stage_ms));
    DGR_CHECK(dgr_ctx_collect_profile(ctx));
  }
  return DGR_OK;
}

// One conv layer of a network on a caller-supplied feature matrix, through exactly the kernels the forward uses for it
// (split rows + the wide-layer kernel + the reduction in the 6-D net, the output-stationary kernel in the 3-D net) over
// the same-stride 3^D kernel map of `coords`.  For layers with a 3^D kernel and Cin >= 32.  Test instrument: it lets the
// parity tests feed the split-operand kernels rows of any magnitude pattern (tests/test_gpu_split_f64.py).
extern "C" int dgr_debug_conv_layer(dgr_ctx *ctx, dgr_net *net, int layer, const int32_t *coords, const float *in,
                                    int in_relu, int64_t N, float *out, dgr_stream stream_) {
  DGR_REQUIRE(ctx && net && coords && in && out && N > 0, "dgr_debug_conv_layer: bad argument");
  DGR_REQUIRE(layer >= 0 && layer < (int)net->W->layers.size(), "layer %d out of range", layer);
  const DgrLayer &L = net->W->layers[layer];
  int k3 = 1;
  for (int d = 0; d < net->D; ++d) k3 *= 3;
  DGR_REQUIRE(L.K == k3 && L.cin >= 32, "dgr_debug_conv_layer: layer %s is not a 3^D conv with >= 32 input channels", L.name.c_str());
  hipStream_t stream = (hipStream_t)stream_;
  DGR_HIP_CHECK(hipSetDevice(ctx->device));
  DGR_CHECK(ctx->arena.reset());
  DGR_CHECK(dgr_ctx_new_flag(ctx, stream));
  dgr_ctx_begin_profile(ctx);
  DgrArena &A = ctx->arena;
  Fwd f;
  f.ctx = ctx; f.net = net; f.stream = stream; f.prof = false;
  f.ms.overflow = ctx->flag_dev;
  DGR_CHECK(dgr_build_maps(A, coords, N, net->D, 3, &f.ms, stream, false, /*lean=*/true, /*nbr_tables=*/net->D == 3));
  const int64_t n_cap = f.ms.cm[0].n_cap;
  if (!f.ms.use_nbr) DGR_ALLOC(f.ybuf, A, float, f.ms.same[0].pair_cap * L.cout);
  Tensor tin{const_cast<float *>(in), L.cin, in_relu}, tout{out, L.cout, 0};
  if (L.wb) {
    tin.split.channels = L.cin;
    DGR_ALLOC(tin.split.planes, A, unsigned char, (size_t)n_cap * 4 * L.cin);
    DGR_ALLOC(tin.split.scale, A, float, n_cap);
    DGR_CHECK(dgr_split_rows(in, L.cin, in_relu, f.ms.cm[0].n_dev, n_cap, tin.split, stream));
  }
  DGR_CHECK(f.conv(layer, tin, &f.ms.same[0], false, 0, 0, tout, nullptr));
  DGR_CHECK(dgr_ctx_check_flag(ctx, stream));   // synchronises the stream
  dgr_net_invalidate_runs(net);
  return DGR_OK;
}

extern "C" int dgr_net_get_intermediate(dgr_ctx *ctx, dgr_net *net, const char *name, float *host_out,
                                        int64_t capacity, int64_t *rows, int64_t *cols) {
  DGR_REQUIRE(ctx && net && name && rows && cols, "NULL argument");
  auto it = net->inter.find(name);
  DGR_REQUIRE(it != net->inter.end() && it->second.n_dev, "no intermediate named '%s' (run a forward first)", name);
  DGR_REQUIRE(net->run_generation == ctx->arena.generation,
              "the activations of the last forward are gone: a later call on this context reused its workspace");
  const DgrTensorRef &t = it->second;
  int32_t n = 0;
  DGR_HIP_CHECK(hipDeviceSynchronize());
  DGR_HIP_CHECK(hipMemcpy(&n, t.n_dev, sizeof(int32_t), hipMemcpyDeviceToHost));
  *rows = n;
  *cols = t.cols;
  if (host_out) {
    DGR_REQUIRE(capacity >= (int64_t)n * t.cols, "host buffer too small");
    if (!t.canon) {
      DGR_HIP_CHECK(hipMemcpy2D(host_out, (size_t)t.cols * sizeof(float), t.ptr, (size_t)t.ld * sizeof(float),
                                (size_t)t.cols * sizeof(float), (size_t)n, hipMemcpyDeviceToHost));
    } else {   // rows back in first-occurrence order (coarse 6-D maps are numbered in bucket order)
      std::vector<float> tmp((size_t)n * t.cols);
      std::vector<int32_t> canon(n);
      DGR_HIP_CHECK(hipMemcpy2D(tmp.data(), (size_t)t.cols * sizeof(float), t.ptr, (size_t)t.ld * sizeof(float),
                                (size_t)t.cols * sizeof(float), (size_t)n, hipMemcpyDeviceToHost));
      DGR_HIP_CHECK(hipMemcpy(canon.data(), t.canon, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
      for (int32_t p = 0; p < n; ++p)
        memcpy(host_out + (size_t)canon[p] * t.cols, tmp.data() + (size_t)p * t.cols, (size_t)t.cols * sizeof(float));
    }
    for (int64_t i = 0; i < (int64_t)n * t.cols; ++i) host_out[i] = host_out[i] > 0.f ? host_out[i] : 0.f;
  }
  return DGR_OK;
}

extern "C" int dgr_net_layer_stats(dgr_ctx *ctx, dgr_net *net, int layer, int64_t stats[8]) {
  DGR_REQUIRE(ctx && net && stats, "NULL argument");
  DGR_REQUIRE(layer >= 0 && layer < (int)net->W->layers.size(), "layer %d out of range", layer);
  const LayerRun &r = net->runs[layer];
  DGR_REQUIRE(r.n_in && r.n_out, "run a forward first");
  DGR_REQUIRE(net->run_generation == ctx->arena.generation,
              "the kernel maps of the last forward are gone: a later call on this context reused its workspace");
  const DgrLayer &L = net->W->layers[layer];
  DGR_HIP_CHECK(hipDeviceSynchronize());
  int32_t n_in = 0, n_out = 0;
  DGR_HIP_CHECK(hipMemcpy(&n_in, r.n_in, sizeof(int32_t), hipMemcpyDeviceToHost));
  DGR_HIP_CHECK(hipMemcpy(&n_out, r.n_out, sizeof(int32_t), hipMemcpyDeviceToHost));
  int64_t P = n_out, kne = 1;
  if (r.os) {
    int64_t counts[27];
    DGR_CHECK(dgr_nbr_counts(r.nbr, r.n_out, counts));
    P = 0;
    kne = 0;
    for (int k = 0; k < 27; ++k) { P += counts[k]; kne += counts[k] > 0; }
  } else if (r.fused_pairs) {
    int32_t pc = 0;
    DGR_HIP_CHECK(hipMemcpy(&pc, r.fused_pairs, sizeof(int32_t), hipMemcpyDeviceToHost));
    P = pc;
    kne = r.K;
  } else if (r.rule_ptr) {
    std::vector<int32_t> rp(r.K + 1);
    DGR_HIP_CHECK(hipMemcpy(rp.data(), r.rule_ptr, (size_t)(r.K + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
    P = rp[r.K];
    kne = 0;
    for (int k = 0; k < r.K; ++k) kne += rp[k + 1] > rp[k];
  }
  stats[0] = P; stats[1] = kne; stats[2] = n_in; stats[3] = n_out;
  stats[4] = L.cin; stats[5] = L.cout; stats[6] = L.K; stats[7] = 0;
  return DGR_OK;
}

// Re-run one conv layer of the last stage-wise forward `reps` times (its kernel maps and activations
// are still in the arena) and report the mean duration of phase 1 (MFMA) and phase 2 (reduce) in ms.
// Kernel-tuning instrument; not part of the inference path.
extern "C" int dgr_net_rerun_layer(dgr_ctx *ctx, dgr_net *net, int layer, int reps, float *gemm_ms, float *reduce_ms) {
  DGR_REQUIRE(ctx && net && gemm_ms && reduce_ms && reps > 0, "bad argument");
  DGR_REQUIRE(layer >= 0 && layer < (int)net->W->layers.size(), "layer %d out of range", layer);
  const LayerRun &r = net->runs[layer];
  DGR_REQUIRE(r.n_in && r.n_out, "run a forward first");
  DGR_REQUIRE(net->run_generation == ctx->arena.generation,
              "the kernel maps of the last forward are gone: a later call on this context reused its workspace");
  DGR_REQUIRE(!r.fused_pairs, "layer %d ran fused with its neighbour search; not re-runnable", layer);
  const DgrLayer &L = net->W->layers[layer];
  hipEvent_t e0, e1, e2;
  DGR_HIP_CHECK(hipEventCreate(&e0)); DGR_HIP_CHECK(hipEventCreate(&e1)); DGR_HIP_CHECK(hipEventCreate(&e2));
  float tg = 0.f, tr = 0.f;
  for (int i = 0; i < reps + 1; ++i) {  // first iteration = warm-up
    DGR_HIP_CHECK(hipEventRecord(e0, nullptr));
    if (r.os)
      DGR_CHECK(dgr_conv_os_launch(r.os_launch, nullptr));
    else if (r.small_cin)
      DGR_CHECK(dgr_conv_small_cin(r.launch.in, r.launch.in_ld, r.launch.in_relu, L.cin, L.w, L.wq, L.shift, r.km, r.n_out,
                                   r.n_out_cap, r.launch.out, r.launch.out_ld, nullptr));
    else if (L.wb && r.has_reduce)
      DGR_CHECK(dgr_conv_wide_launch(r.launch, r.split_in, L.wb, L.wb_piece, L.w_unscale, ctx->num_cus, nullptr));
    else
      DGR_CHECK(dgr_conv_launch(r.launch, ctx->num_cus, nullptr));
    DGR_HIP_CHECK(hipEventRecord(e1, nullptr));
    if (r.has_reduce && !r.small_cin && !r.os)
      DGR_CHECK(dgr_reduce_rows(r.launch.y, L.cout, r.red_ptr, r.red_pos, r.n_out, r.n_out_cap,
                                r.split_only ? nullptr : r.launch.out, r.launch.out_ld, L.shift, r.res, r.res_ld, r.res_relu, nullptr,
                                r.split_out.planes ? &r.split_out : nullptr, r.out_relu));
    DGR_HIP_CHECK(hipEventRecord(e2, nullptr));
    DGR_HIP_CHECK(hipEventSynchronize(e2));
    float a = 0.f, b = 0.f;
    DGR_HIP_CHECK(hipEventElapsedTime(&a, e0, e1));
    DGR_HIP_CHECK(hipEventElapsedTime(&b, e1, e2));
    if (i > 0) { tg += a; tr += b; }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
  *gemm_ms = tg / reps;
  *reduce_ms = tr / reps;
  return DGR_OK;
}
