// HBM-bound elementwise kernels around the conv stacks:
//   * 1x1 conv 3->3 on the raw image          (generator.py:46)
//   * 2x2 stride-2 max pool, ceil / floor     (generator.py:53,60,73; torchvision vgg19 'M')
#include <algorithm>

#include "common.h"

namespace {

__global__ __launch_bounds__(256) void conv1x1_3to3_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ b,
                                                          float* __restrict__ y, long long hw,
                                                          long long total) {
  float wr[9], br[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) wr[i] = w[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) br[i] = b ? b[i] : 0.f;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long img = e / hw, pix = e - img * hw;
    const float* xp = x + img * 3 * hw + pix;
    float* yp = y + img * 3 * hw + pix;
    const float x0 = xp[0], x1 = xp[hw], x2 = xp[2 * hw];
#pragma unroll
    for (int o = 0; o < 3; ++o)
      yp[o * hw] = fmaf(wr[o * 3 + 2], x2, fmaf(wr[o * 3 + 1], x1, fmaf(wr[o * 3], x0, 0.f))) + br[o];
  }
}

__global__ __launch_bounds__(256) void maxpool2x2_kernel(const float* __restrict__ x,
                                                        float* __restrict__ y, int h, int w,
                                                        int ho, int wo, long long total) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int ox = int(e % wo);
    long long t = e / wo;
    const int oy = int(t % ho);
    const long long plane = t / ho;
    const float* xp = x + plane * h * w;
    const int iy = oy * 2, ix = ox * 2;
    const bool y1 = iy + 1 < h, x1 = ix + 1 < w;   // partial windows only in ceil mode
    float m = xp[(size_t)iy * w + ix];
    if (x1) m = fmaxf(m, xp[(size_t)iy * w + ix + 1]);
    if (y1) {
      m = fmaxf(m, xp[(size_t)(iy + 1) * w + ix]);
      if (x1) m = fmaxf(m, xp[(size_t)(iy + 1) * w + ix + 1]);
    }
    y[e] = m;
  }
}

// Backward of the 2x2/2 max pool: each thread owns one (ceil-mode) window of x, routes dy to
// the first maximum in scan order (torch's tie rule: strictly-greater replaces) and zeroes the
// rest, so dx is written exactly once everywhere (floor-mode remainders get 0).
__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ dy,
                                                            float* __restrict__ dx, int h, int w,
                                                            int ho, int wo, int hc, int wc,
                                                            long long total) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int ox = int(e % wc);
    long long t = e / wc;
    const int oy = int(t % hc);
    const long long plane = t / hc;
    const float* xp = x + plane * h * w;
    float* dp = dx + plane * h * w;
    const int iy = oy * 2, ix = ox * 2;
    const bool y1 = iy + 1 < h, x1 = ix + 1 < w;
    const float g = (oy < ho && ox < wo) ? dy[(plane * ho + oy) * wo + ox] : 0.f;
    float m = xp[(size_t)iy * w + ix];
    int arg = 0;
    if (x1) { const float v = xp[(size_t)iy * w + ix + 1]; if (v > m || v != v) { m = v; arg = 1; } }
    if (y1) {
      const float v = xp[(size_t)(iy + 1) * w + ix];
      if (v > m || v != v) { m = v; arg = 2; }
      if (x1) { const float v2 = xp[(size_t)(iy + 1) * w + ix + 1]; if (v2 > m || v2 != v2) { m = v2; arg = 3; } }
    }
    dp[(size_t)iy * w + ix] = arg == 0 ? g : 0.f;
    if (x1) dp[(size_t)iy * w + ix + 1] = arg == 1 ? g : 0.f;
    if (y1) {
      dp[(size_t)(iy + 1) * w + ix] = arg == 2 ? g : 0.f;
      if (x1) dp[(size_t)(iy + 1) * w + ix + 1] = arg == 3 ? g : 0.f;
    }
  }
}

inline int grid_for(long long total) {
  return (int)std::max<long long>(1, std::min<long long>(st::cdivll(total, 256), 256 * 8));
}

}  // namespace

extern "C" int st_conv1x1_3to3_fwd(const float* x, const float* w, const float* b, float* y,
                                   int n, int hw, st_stream_t stream) {
  ST_REQUIRE(x && w && y, "st_conv1x1_3to3_fwd: null pointer");
  ST_REQUIRE(n > 0 && hw > 0, "st_conv1x1_3to3_fwd: bad dims n=%d hw=%d", n, hw);
  const long long total = (long long)n * hw;
  hipLaunchKernelGGL(conv1x1_3to3_kernel, dim3(grid_for(total)), dim3(256), 0,
                     (hipStream_t)stream, x, w, b, y, (long long)hw, total);
  return st::check_launch("st_conv1x1_3to3_fwd");
}

extern "C" int st_maxpool2x2_fwd(const float* x, float* y, int nc, int h, int w, int ceil_mode,
                                 st_stream_t stream) {
  ST_REQUIRE(x && y, "st_maxpool2x2_fwd: null pointer");
  ST_REQUIRE(nc > 0 && h > 0 && w > 0, "st_maxpool2x2_fwd: bad dims nc=%d h=%d w=%d", nc, h, w);
  const int ho = ceil_mode ? (h + 1) / 2 : h / 2, wo = ceil_mode ? (w + 1) / 2 : w / 2;
  ST_REQUIRE(ho > 0 && wo > 0, "st_maxpool2x2_fwd: empty output for %dx%d floor mode", h, w);
  const long long total = (long long)nc * ho * wo;
  hipLaunchKernelGGL(maxpool2x2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     x, y, h, w, ho, wo, total);
  return st::check_launch("st_maxpool2x2_fwd");
}

extern "C" int st_maxpool2x2_bwd(const float* x, const float* dy, float* dx, int nc, int h, int w,
                                 int ceil_mode, st_stream_t stream) {
  ST_REQUIRE(x && dy && dx, "st_maxpool2x2_bwd: null pointer");
  ST_REQUIRE(nc > 0 && h > 0 && w > 0, "st_maxpool2x2_bwd: bad dims nc=%d h=%d w=%d", nc, h, w);
  const int hc = (h + 1) / 2, wc = (w + 1) / 2;
  const int ho = ceil_mode ? hc : h / 2, wo = ceil_mode ? wc : w / 2;
  ST_REQUIRE(ho > 0 && wo > 0, "st_maxpool2x2_bwd: empty output for %dx%d floor mode", h, w);
  const long long total = (long long)nc * hc * wc;
  hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3(grid_for(total)), dim3(256), 0,
                     (hipStream_t)stream, x, dy, dx, h, w, ho, wo, hc, wc, total);
  return st::check_launch("st_maxpool2x2_bwd");
}

// ---- SymmetricPadding2D, any (left, right, top, bottom) (generator.py:14-39) -------------
namespace {

// numpy "symmetric" reflection about -0.5 and n-0.5: period 2n triangular wave.
__device__ __forceinline__ int sym_index(int i, int n) {
  int m = i % (2 * n);
  if (m < 0) m += 2 * n;
  return m >= n ? 2 * n - 1 - m : m;
}

__global__ __launch_bounds__(256) void sympad_kernel(const float* __restrict__ x,
                                                    float* __restrict__ y, int h, int w, int left,
                                                    int top, int ho, int wo, long long total) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int ox = int(e % wo);
    long long t = e / wo;
    const int oy = int(t % ho);
    const long long plane = t / ho;
    y[e] = x[plane * h * w + (size_t)sym_index(oy - top, h) * w + sym_index(ox - left, w)];
  }
}

}  // namespace

extern "C" int st_sympad_fwd(const float* x, float* y, int nc, int h, int w, int left, int right,
                             int top, int bottom, st_stream_t stream) {
  ST_REQUIRE(x && y, "st_sympad_fwd: null pointer");
  ST_REQUIRE(nc > 0 && h > 0 && w > 0, "st_sympad_fwd: bad dims nc=%d h=%d w=%d", nc, h, w);
  ST_REQUIRE(left >= 0 && right >= 0 && top >= 0 && bottom >= 0, "st_sympad_fwd: negative padding");
  const int ho = h + top + bottom, wo = w + left + right;
  const long long total = (long long)nc * ho * wo;
  hipLaunchKernelGGL(sympad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y,
                     h, w, left, top, ho, wo, total);
  return st::check_launch("st_sympad_fwd");
}

// ---- transforms.Normalize(mean, std) on [n][c][hw] (trainer.py:299, 440-442, 707-708) ----------
namespace {

struct NormArgs {
  float mean[8];
  float inv_std[8];
};

__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       int c, long long hw, long long total, NormArgs a) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int ch = int((e / hw) % c);
    y[e] = (x[e] - a.mean[ch]) * a.inv_std[ch];
  }
}

}  // namespace

extern "C" int st_normalize_fwd(const float* x, float* y, int n, int c, int hw, const float* mean_host,
                                const float* std_host, st_stream_t stream) {
  ST_REQUIRE(x && y && std_host, "st_normalize_fwd: null pointer");
  ST_REQUIRE(n > 0 && c > 0 && c <= 8 && hw > 0, "st_normalize_fwd: bad dims n=%d c=%d hw=%d", n, c, hw);
  NormArgs a;
  for (int i = 0; i < c; ++i) {
    a.mean[i] = mean_host ? mean_host[i] : 0.f;
    ST_REQUIRE(std_host[i] != 0.f, "st_normalize_fwd: std[%d] == 0", i);
    a.inv_std[i] = 1.f / std_host[i];
  }
  const long long total = (long long)n * c * hw;
  hipLaunchKernelGGL(normalize_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, c,
                     (long long)hw, total, a);
  return st::check_launch("st_normalize_fwd");
}

// ---- stand-alone ReLU (only where a pre-ReLU feature must be kept: shared loss-VGG pass) --------
namespace {
__global__ __launch_bounds__(256) void relu_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                  long long n) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
    y[e] = fmaxf(x[e], 0.f);
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                      float* __restrict__ dx, long long n) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
    dx[e] = y[e] > 0.f ? dy[e] : 0.f;
}
}  // namespace

extern "C" int st_relu_fwd(const float* x, float* y, long long n, st_stream_t stream) {
  ST_REQUIRE(x && y && n > 0, "st_relu_fwd: bad arguments");
  hipLaunchKernelGGL(relu_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  return st::check_launch("st_relu_fwd");
}

extern "C" int st_relu_bwd(const float* y, const float* dy, float* dx, long long n, st_stream_t stream) {
  ST_REQUIRE(y && dy && dx && n > 0, "st_relu_bwd: bad arguments");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, n);
  return st::check_launch("st_relu_bwd");
}
