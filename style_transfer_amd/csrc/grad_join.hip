// Gradient "joins" of the loss VGG's backward in the bf16 compute mode (trainer.py:472 through the feature
// extractors of trainer.py:165-199): the data-gradient launches hand each other the gradient in the bf16 channel-group
// layout G8 (st_conv3x3_dgrad_bf16_g8).  Where a layer's output was CAPTURED (a loss tap, fp32 NCHW) or feeds a max
// pool, the gradient that reaches it has more than one source and crosses a ReLU; this kernel forms it in ONE pass
// and writes the G8 operand of the layer's data-gradient launch:
//
//     out = gate(x) * (chain + add_post) + add_pre,      gate(x) = [x > 0]
//
//   chain      gradient from the next layer: fp32 NCHW of the tap's shape, OR (`chain_pooled`) the gradient w.r.t. the
//              output of the floor-mode 2x2 max pool that follows the tap, routed to the first maximum of each window
//              in scan order exactly like st_maxpool2x2_bwd (x = the pool's input);
//   add_post   the loss gradient w.r.t. the post-ReLU tap;  add_pre  the loss gradient w.r.t. the pre-ReLU tap (the
//              last index of an IntermediateLayerGetter is captured before the in-place ReLU, loss_network.py);
//   x          the tap itself (post-ReLU output, or the pre-ReLU tensor: same sign pattern and, where any entry of a
//              window is positive, the same arg-max); NULL = no gate.
// Every term is optional.  fp32 sums in this order, one round-to-nearest-even to bf16: the same values the separate
// max-pool-backward / accumulate / ReLU-backward kernels followed by the data-gradient kernel's staging produce.
// HBM-bound: 4 B per source element read once, 2 B written.
#include <algorithm>

#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct JoinParams {
  const float* chain;
  const float* chain_pooled;
  const float* x;
  const float* add_post;
  const float* add_pre;
  // the loss gradient as per-plane affine coefficients [n * c][4] = (k0, k1, mean, -): k0 + k1 * (t - mean) of the
  // tap t = aff_x (st_meanstd_loss_bwd_coeffs) in place of a materialised add_post / add_pre tensor
  const float* aff_x;
  const float* aff_post;
  const float* aff_pre;
  u32x4* out;
  int c, h, w;
  long long total;   // threads' worth of work items
};

__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
  bf16x8 b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[j];
  return __builtin_bit_cast(u32x4, b);
}

// item = (image, channel group, PX consecutive pixels of the flattened plane)
template <int PX>
__global__ __launch_bounds__(256) void grad_join_kernel(JoinParams p) {
  const long long hw = (long long)p.h * p.w;
  const long long per_plane = hw / PX;
  const int cgs = p.c >> 3;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < p.total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long pix = (e % per_plane) * PX;
    const long long t = e / per_plane;
    const int cg = int(t % cgs);
    const long long b = t / cgs;
    const long long base = (b * p.c + cg * 8) * hw + pix;
    float v[PX][8];
    const bool has_post = p.add_post || p.aff_post, has_pre = p.add_pre || p.aff_pre;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long off = base + j * hw;
      float a[PX], g[PX], q[PX], r[PX], t[PX];
      if constexpr (PX == 4) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const f32x4 a4 = p.chain ? *reinterpret_cast<const f32x4*>(p.chain + off) : z;
        const f32x4 q4 = p.add_post ? *reinterpret_cast<const f32x4*>(p.add_post + off) : z;
        const f32x4 g4 = p.x ? *reinterpret_cast<const f32x4*>(p.x + off) : z;
        const f32x4 r4 = p.add_pre ? *reinterpret_cast<const f32x4*>(p.add_pre + off) : z;
        const f32x4 t4 = p.aff_x ? (p.aff_x == p.x ? g4 : *reinterpret_cast<const f32x4*>(p.aff_x + off)) : z;
#pragma unroll
        for (int k = 0; k < 4; ++k) { a[k] = a4[k]; q[k] = q4[k]; g[k] = g4[k]; r[k] = r4[k]; t[k] = t4[k]; }
      } else {
        a[0] = p.chain ? p.chain[off] : 0.f;
        q[0] = p.add_post ? p.add_post[off] : 0.f;
        g[0] = p.x ? p.x[off] : 0.f;
        r[0] = p.add_pre ? p.add_pre[off] : 0.f;
        t[0] = p.aff_x ? p.aff_x[off] : 0.f;
      }
      if (p.aff_post) {      // meanstd_loss_bwd_kernel's expression on its three per-plane scalars
        const f32x4 cf = reinterpret_cast<const f32x4*>(p.aff_post)[b * p.c + cg * 8 + j];
#pragma unroll
        for (int k = 0; k < PX; ++k) q[k] = cf[0] + cf[1] * (t[k] - cf[2]);
      }
      if (p.aff_pre) {
        const f32x4 cf = reinterpret_cast<const f32x4*>(p.aff_pre)[b * p.c + cg * 8 + j];
#pragma unroll
        for (int k = 0; k < PX; ++k) r[k] = cf[0] + cf[1] * (t[k] - cf[2]);
      }
#pragma unroll
      for (int k = 0; k < PX; ++k) {
        float s = a[k];
        if (has_post) s = p.chain ? s + q[k] : q[k];
        if (p.x) s = g[k] > 0.f ? s : 0.f;
        if (has_pre) s = (p.chain || has_post) ? s + r[k] : r[k];
        v[k][j] = s;
      }
    }
    u32x4* o = p.out + (b * cgs + cg) * hw + pix;
#pragma unroll
    for (int k = 0; k < PX; ++k) o[k] = pack8(v[k]);
  }
}

// item = (image, channel group, one 2x2 window of the pool's input); h and w even
__global__ __launch_bounds__(256) void grad_join_pool_kernel(JoinParams p) {
  const int ho = p.h >> 1, wo = p.w >> 1;
  const long long hw = (long long)p.h * p.w, hwo = (long long)ho * wo;
  const int cgs = p.c >> 3;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < p.total;
       e += (long long)gridDim.x * blockDim.x) {
    const int ox = int(e % wo);
    long long t = e / wo;
    const int oy = int(t % ho);
    t /= ho;
    const int cg = int(t % cgs);
    const long long b = t / cgs;
    const long long win = (long long)(2 * oy) * p.w + 2 * ox;     // first pixel of the window inside a plane
    float v[4][8];                                                 // window positions 0..3 in scan order
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long plane = b * p.c + cg * 8 + j;
      const long long off = plane * hw + win;
      const f32x2 x0 = *reinterpret_cast<const f32x2*>(p.x + off);
      const f32x2 x1 = *reinterpret_cast<const f32x2*>(p.x + off + p.w);
      const float g = p.chain_pooled[plane * hwo + (long long)oy * wo + ox];
      const float xs[4] = {x0[0], x0[1], x1[0], x1[1]};
      float m = xs[0];
      int arg = 0;
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (xs[k] > m || xs[k] != xs[k]) { m = xs[k]; arg = k; }   // st_maxpool2x2_bwd's (= torch's) tie rule
      float q[4] = {0.f, 0.f, 0.f, 0.f}, r[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.add_post) {
        const f32x2 q0 = *reinterpret_cast<const f32x2*>(p.add_post + off);
        const f32x2 q1 = *reinterpret_cast<const f32x2*>(p.add_post + off + p.w);
        q[0] = q0[0]; q[1] = q0[1]; q[2] = q1[0]; q[3] = q1[1];
      }
      if (p.add_pre) {
        const f32x2 r0 = *reinterpret_cast<const f32x2*>(p.add_pre + off);
        const f32x2 r1 = *reinterpret_cast<const f32x2*>(p.add_pre + off + p.w);
        r[0] = r0[0]; r[1] = r0[1]; r[2] = r1[0]; r[3] = r1[1];
      }
      // affine loss gradients of the tap (here always the pool's input x itself: the entry point checks aff_x == x)
      if (p.aff_post) {
        const f32x4 cf = reinterpret_cast<const f32x4*>(p.aff_post)[plane];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = cf[0] + cf[1] * (xs[k] - cf[2]);
      }
      if (p.aff_pre) {
        const f32x4 cf = reinterpret_cast<const f32x4*>(p.aff_pre)[plane];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = cf[0] + cf[1] * (xs[k] - cf[2]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float s = arg == k ? g : 0.f;
        if (p.add_post || p.aff_post) s += q[k];
        s = xs[k] > 0.f ? s : 0.f;
        if (p.add_pre || p.aff_pre) s += r[k];
        v[k][j] = s;
      }
    }
    u32x4* o = p.out + (b * cgs + cg) * hw + win;
    o[0] = pack8(v[0]);
    o[1] = pack8(v[1]);
    o[p.w] = pack8(v[2]);
    o[p.w + 1] = pack8(v[3]);
  }
}

// Floor-mode 2x2 max pool of an fp32 NCHW tensor written straight in G8 (torchvision vgg19 'M' between a captured
// layer and the next conv of a G8 chain): rounding is monotonic, so round(max) == max(round) -- the same bits the
// next conv's staging would make of the fp32 pool output.  item = (image, channel group, output pixel); h, w even.
__global__ __launch_bounds__(256) void maxpool2x2_g8_kernel(const float* __restrict__ x, u32x4* __restrict__ y, int c,
                                                           int h, int w, long long total) {
  const int ho = h >> 1, wo = w >> 1;
  const long long hw = (long long)h * w;
  const int cgs = c >> 3;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int ox = int(e % wo);
    long long t = e / wo;
    const int oy = int(t % ho);
    t /= ho;
    const int cg = int(t % cgs);
    const long long b = t / cgs;
    const float* src = x + (b * c + cg * 8) * hw + (long long)(2 * oy) * w + 2 * ox;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x2 r0 = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(src + j * hw));
      const f32x2 r1 = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(src + j * hw + w));
      v[j] = fmaxf(fmaxf(r0[0], r0[1]), fmaxf(r1[0], r1[1]));
    }
    y[e] = pack8(v);
  }
}

inline int grid_for(long long total) {
  return (int)std::max<long long>(1, std::min<long long>(st::cdivll(total, 256), 256 * 16));
}

}  // namespace

static int grad_join_impl(const float* chain, const float* chain_pooled, const float* x_gate, const float* add_post,
                          const float* add_pre, const float* aff_x, const float* aff_post, const float* aff_pre,
                          void* out_g8, int n, int c, int h, int w, st_stream_t stream) {
  ST_REQUIRE(out_g8, "st_grad_join_g8: null output");
  ST_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && c % 8 == 0, "st_grad_join_g8: bad dims n=%d c=%d h=%d w=%d (c %% 8 == 0)",
             n, c, h, w);
  ST_REQUIRE(!(chain && chain_pooled), "st_grad_join_g8: chain and chain_pooled exclude each other");
  ST_REQUIRE(chain || chain_pooled || add_post || add_pre || aff_post || aff_pre, "st_grad_join_g8: no gradient source");
  ST_REQUIRE(!(add_post && aff_post) && !(add_pre && aff_pre), "st_grad_join_g8: a loss gradient is a tensor OR coefficients");
  ST_REQUIRE(!(aff_post || aff_pre) || aff_x, "st_grad_join_g8_affine: coefficients need the tap tensor aff_x");
  ST_REQUIRE(reinterpret_cast<uintptr_t>(out_g8) % 16 == 0 && reinterpret_cast<uintptr_t>(aff_post) % 16 == 0 &&
             reinterpret_cast<uintptr_t>(aff_pre) % 16 == 0, "st_grad_join_g8: out_g8 / coefficients must be 16-byte aligned");
  JoinParams p;
  p.chain = chain; p.chain_pooled = chain_pooled; p.x = x_gate; p.add_post = add_post; p.add_pre = add_pre;
  p.aff_x = aff_x; p.aff_post = aff_post; p.aff_pre = aff_pre;
  p.out = (u32x4*)out_g8; p.c = c; p.h = h; p.w = w;
  hipStream_t s = (hipStream_t)stream;
  const long long hw = (long long)h * w;
  if (chain_pooled) {
    ST_REQUIRE(x_gate, "st_grad_join_g8: chain_pooled needs the pool's input in x_gate");
    ST_REQUIRE(!aff_x || aff_x == x_gate, "st_grad_join_g8_affine: with chain_pooled the tap is the pool's input");
    ST_REQUIRE(h % 2 == 0 && w % 2 == 0, "st_grad_join_g8: chain_pooled needs even h, w (got %dx%d)", h, w);
    for (const float* q : {x_gate, add_post, add_pre})
      ST_REQUIRE(reinterpret_cast<uintptr_t>(q) % 8 == 0, "st_grad_join_g8: 8-byte aligned fp32 tensors expected");
    p.total = (long long)n * (c / 8) * (hw / 4);
    hipLaunchKernelGGL(grad_join_pool_kernel, dim3(grid_for(p.total)), dim3(256), 0, s, p);
    return st::check_launch("st_grad_join_g8");
  }
  bool vec = hw % 4 == 0;
  for (const float* q : {chain, x_gate, add_post, add_pre, aff_x}) vec = vec && reinterpret_cast<uintptr_t>(q) % 16 == 0;
  if (vec) {
    p.total = (long long)n * (c / 8) * (hw / 4);
    hipLaunchKernelGGL(grad_join_kernel<4>, dim3(grid_for(p.total)), dim3(256), 0, s, p);
  } else {
    p.total = (long long)n * (c / 8) * hw;
    hipLaunchKernelGGL(grad_join_kernel<1>, dim3(grid_for(p.total)), dim3(256), 0, s, p);
  }
  return st::check_launch("st_grad_join_g8");
}

extern "C" int st_grad_join_g8(const float* chain, const float* chain_pooled, const float* x_gate,
                               const float* add_post, const float* add_pre, void* out_g8, int n, int c, int h,
                               int w, st_stream_t stream) {
  return grad_join_impl(chain, chain_pooled, x_gate, add_post, add_pre, nullptr, nullptr, nullptr, out_g8, n, c, h, w,
                        stream);
}

extern "C" int st_grad_join_g8_affine(const float* chain, const float* chain_pooled, const float* x_gate,
                                      const float* add_post, const float* add_pre, const float* aff_x,
                                      const float* aff_post, const float* aff_pre, void* out_g8, int n, int c, int h,
                                      int w, st_stream_t stream) {
  return grad_join_impl(chain, chain_pooled, x_gate, add_post, add_pre, aff_x, aff_post, aff_pre, out_g8, n, c, h, w,
                        stream);
}

// Max-pool backward from the winner codes of st_conv3x3_pool_fwd_bf16_g8_codes: the G8 gradient w.r.t. the pooled
// activation goes to the window position named by bits 0-1 of the channel's code when bit 2 (maximum > 0) is set; the
// other three positions are zero.  item = (image, channel group, pooled pixel): 16 + 8 bytes in, 64 bytes out.
__global__ __launch_bounds__(256) void grad_pool_codes_kernel(const u32x4* __restrict__ g, const u32x2* __restrict__ codes,
                                                              u32x4* __restrict__ out, int ho, int wo, long long total) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int ox = int(e % wo);
    long long t = e / wo;
    const int oy = int(t % ho);
    const long long plane = t / ho;                    // image * groups + group
    const u32x4 gv = g[e];
    const u32x2 cv = codes[e];
    u32x4 o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned code = (cv[j >> 2] >> (8 * (j & 3))) & 0xffu;
      const unsigned lane = gv[j >> 1] & ((j & 1) ? 0xffff0000u : 0x0000ffffu);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (code == (4u | unsigned(k))) o[k][j >> 1] |= lane;
    }
    u32x4* dst = out + (plane * (2 * ho) + 2 * oy) * (long long)(2 * wo) + 2 * ox;
    dst[0] = o[0];
    dst[1] = o[1];
    dst[2 * wo] = o[2];
    dst[2 * wo + 1] = o[3];
  }
}

extern "C" int st_grad_pool_codes_g8(const void* g_pooled_g8, const void* codes, void* out_g8, int n, int c, int h,
                                     int w, st_stream_t stream) {
  ST_REQUIRE(g_pooled_g8 && codes && out_g8, "st_grad_pool_codes_g8: null pointer");
  ST_REQUIRE(n > 0 && c > 0 && c % 8 == 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0,
             "st_grad_pool_codes_g8: bad dims n=%d c=%d h=%d w=%d (c %% 8 == 0, even h and w)", n, c, h, w);
  ST_REQUIRE(reinterpret_cast<uintptr_t>(g_pooled_g8) % 16 == 0 && reinterpret_cast<uintptr_t>(out_g8) % 16 == 0 &&
                 reinterpret_cast<uintptr_t>(codes) % 8 == 0,
             "st_grad_pool_codes_g8: 16-byte aligned G8 tensors and 8-byte aligned codes expected");
  const long long total = (long long)n * (c / 8) * (h / 2) * (w / 2);
  hipLaunchKernelGGL(grad_pool_codes_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const u32x4*)g_pooled_g8, (const u32x2*)codes, (u32x4*)out_g8, h / 2, w / 2, total);
  return st::check_launch("st_grad_pool_codes_g8");
}

extern "C" int st_maxpool2x2_fwd_g8(const float* x, void* y_g8, int n, int c, int h, int w, st_stream_t stream) {
  ST_REQUIRE(x && y_g8, "st_maxpool2x2_fwd_g8: null pointer");
  ST_REQUIRE(n > 0 && c > 0 && c % 8 == 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0,
             "st_maxpool2x2_fwd_g8: bad dims n=%d c=%d h=%d w=%d (c %% 8 == 0, even h and w)", n, c, h, w);
  ST_REQUIRE(reinterpret_cast<uintptr_t>(x) % 8 == 0 && reinterpret_cast<uintptr_t>(y_g8) % 16 == 0,
             "st_maxpool2x2_fwd_g8: x must be 8-byte and y_g8 16-byte aligned");
  const long long total = (long long)n * (c / 8) * (h / 2) * (w / 2);
  hipLaunchKernelGGL(maxpool2x2_g8_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x,
                     (u32x4*)y_g8, c, h, w, total);
  return st::check_launch("st_maxpool2x2_fwd_g8");
}
