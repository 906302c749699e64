// HBM-bound loss reductions and their gradients for gfx950:
//   * MSE               AdaINContentLoss (losses.py:93-101), nn.MSELoss terms of losses.py:35-36
//   * mean/std loss     AdaINStyleLoss   (losses.py:14-39)
//   * TV loss           TVLoss           (losses.py:104-116)
// Every scalar is produced by a fixed-order two-stage reduction (per-block partials in a
// caller workspace, then one block sums them): deterministic, no float atomics.
// Backward kernels take the upstream gradient as a DEVICE scalar (`gout`, nullable = 1) times a
// host-side `scale` (the loss weight), and either overwrite or accumulate into dx.
#include <algorithm>
#include <cstdint>

#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int MAX_PARTS = 1024;

inline int parts_for(long long n) {
  return (int)std::max<long long>(1, std::min<long long>(st::cdivll(n, NT * 8), MAX_PARTS));
}

// stage 2: out[k] = scale[k] * sum_i partials[k][i]   (nk <= 4 scalars, one block)
__global__ __launch_bounds__(NT) void finalize_kernel(const float* __restrict__ partials, int nparts,
                                                     int stride, int nk, float s0, float s1,
                                                     float* __restrict__ out, int combine) {
  __shared__ float red[NT / 64];
  float tot[2] = {0.f, 0.f};
  for (int k = 0; k < nk; ++k) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += NT) s += partials[(size_t)k * stride + i];
    tot[k] = st::block_sum<NT>(s, red);
  }
  if (threadIdx.x == 0) {
    if (combine) out[0] = tot[0] * s0 + tot[1] * s1;
    else { out[0] = tot[0] * s0; if (nk > 1) out[1] = tot[1] * s1; }
  }
}

// ---- MSE -----------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void sqdiff_partial_kernel(const float* __restrict__ a,
                                                           const float* __restrict__ b, long long n,
                                                           float* __restrict__ partials) {
  __shared__ float red[NT / 64];
  float s = 0.f;
  for (long long e = blockIdx.x * (long long)NT + threadIdx.x; e < n; e += (long long)gridDim.x * NT) {
    const float d = a[e] - b[e];
    s += d * d;
  }
  s = st::block_sum<NT>(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(NT) void mse_bwd_kernel(const float* __restrict__ a,
                                                    const float* __restrict__ b,
                                                    const float* __restrict__ gout, float coef,
                                                    float* __restrict__ da, long long n, int accumulate) {
  const float c = coef * (gout ? gout[0] : 1.f);
  for (long long e = blockIdx.x * (long long)NT + threadIdx.x; e < n; e += (long long)gridDim.x * NT) {
    const float g = c * (a[e] - b[e]);
    da[e] = accumulate ? da[e] + g : g;
  }
}

// ---- TV ------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void tv_partial_kernel(const float* __restrict__ x, int h, int w,
                                                       long long total, float* __restrict__ partials,
                                                       int stride) {
  __shared__ float red[NT / 64];
  float sh = 0.f, sw = 0.f;
  for (long long e = blockIdx.x * (long long)NT + threadIdx.x; e < total; e += (long long)gridDim.x * NT) {
    const int xx = int(e % w);
    const int yy = int((e / w) % h);
    const float v = x[e];
    if (yy + 1 < h) { const float d = x[e + w] - v; sh += d * d; }
    if (xx + 1 < w) { const float d = x[e + 1] - v; sw += d * d; }
  }
  sh = st::block_sum<NT>(sh, red);
  sw = st::block_sum<NT>(sw, red);
  if (threadIdx.x == 0) { partials[blockIdx.x] = sh; partials[stride + blockIdx.x] = sw; }
}

__global__ __launch_bounds__(NT) void tv_bwd_kernel(const float* __restrict__ x,
                                                   const float* __restrict__ gout, float ch, float cw,
                                                   float* __restrict__ dx, int h, int w, long long total,
                                                   int accumulate) {
  const float g0 = gout ? gout[0] : 1.f;
  for (long long e = blockIdx.x * (long long)NT + threadIdx.x; e < total; e += (long long)gridDim.x * NT) {
    const int xx = int(e % w);
    const int yy = int((e / w) % h);
    const float v = x[e];
    float gh = 0.f, gw = 0.f;
    if (yy > 0) gh += v - x[e - w];
    if (yy + 1 < h) gh -= x[e + w] - v;
    if (xx > 0) gw += v - x[e - 1];
    if (xx + 1 < w) gw -= x[e + 1] - v;
    const float g = g0 * (ch * gh + cw * gw);
    dx[e] = accumulate ? dx[e] + g : g;
  }
}

// ---- mean/std (style) loss -----------------------------------------------------------------
__global__ __launch_bounds__(NT) void meanstd_loss_partial_kernel(const float* __restrict__ ma,
                                                                 const float* __restrict__ sa,
                                                                 const float* __restrict__ mb,
                                                                 const float* __restrict__ sb, int planes,
                                                                 float* __restrict__ partials) {
  __shared__ float red[NT / 64];
  float s = 0.f;
  for (int e = blockIdx.x * NT + threadIdx.x; e < planes; e += gridDim.x * NT) {
    const float dm = ma[e] - mb[e], ds = sa[e] - sb[e];
    s += dm * dm + ds * ds;
  }
  s = st::block_sum<NT>(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// dx_i (+)= c * [ 2(ma-mb)/P / hw  +  2(sa-sb)/P * (x_i - ma) / ((hw-1) sa) ]
// Grid (planes, slices of BWD_SLICE elements): a 1 MB plane (64 channels at 512 x 512) is worked on by 16 blocks with
// 16-B accesses instead of one block with dword accesses (that form ran at 2.5 TB/s on the training shapes).
constexpr int BWD_SLICE = 16384;

template <bool V4>
__global__ __launch_bounds__(NT) void meanstd_loss_bwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ ma,
                                                             const float* __restrict__ sa,
                                                             const float* __restrict__ mb,
                                                             const float* __restrict__ sb,
                                                             const float* __restrict__ gout, float scale,
                                                             float* __restrict__ dx, int planes, int hw,
                                                             int accumulate) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const size_t plane = blockIdx.x;
  const float c = scale * (gout ? gout[0] : 1.f) * 2.f / float(planes);
  const float mean = ma[plane], sd = sa[plane];
  const float k0 = c * (mean - mb[plane]) / float(hw);
  const float k1 = c * (sd - sb[plane]) / (float(hw - 1) * sd);
  const float* xp = x + plane * hw;
  float* dp = dx + plane * hw;
  const int i0 = blockIdx.y * BWD_SLICE, i1 = min(i0 + BWD_SLICE, hw);
  if (V4) {
    for (int i = i0 / 4 + threadIdx.x; i < i1 / 4; i += NT) {
      const f4 v = reinterpret_cast<const f4*>(xp)[i];
      f4 g;
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = k0 + k1 * (v[k] - mean);
      if (accumulate) {
        const f4 d = reinterpret_cast<const f4*>(dp)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = d[k] + g[k];
      }
      reinterpret_cast<f4*>(dp)[i] = g;
    }
  } else {
    for (int i = i0 + threadIdx.x; i < i1; i += NT) {
      const float g = k0 + k1 * (xp[i] - mean);
      dp[i] = accumulate ? dp[i] + g : g;
    }
  }
}

}  // namespace

extern "C" size_t st_reduce_workspace_bytes(void) { return size_t(2) * MAX_PARTS * sizeof(float); }

extern "C" int st_mse_fwd(const float* a, const float* b, long long n, float* workspace, float* out,
                          st_stream_t stream) {
  ST_REQUIRE(a && b && workspace && out, "st_mse_fwd: null pointer");
  ST_REQUIRE(n > 0, "st_mse_fwd: n=%lld", n);
  const int parts = parts_for(n);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sqdiff_partial_kernel, dim3(parts), dim3(NT), 0, s, a, b, n, workspace);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(NT), 0, s, workspace, parts, MAX_PARTS, 1,
                     1.f / float(n), 0.f, out, 0);
  return st::check_launch("st_mse_fwd");
}

extern "C" int st_mse_bwd(const float* a, const float* b, const float* gout, float scale, float* da,
                          long long n, int accumulate, st_stream_t stream) {
  ST_REQUIRE(a && b && da, "st_mse_bwd: null pointer");
  ST_REQUIRE(n > 0, "st_mse_bwd: n=%lld", n);
  const int blocks = (int)std::min<long long>(st::cdivll(n, NT), 2048);
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, a, b, gout,
                     scale * 2.f / float(n), da, n, accumulate);
  return st::check_launch("st_mse_bwd");
}

extern "C" int st_tv_fwd(const float* x, int n, int c, int h, int w, float* workspace, float* out,
                         st_stream_t stream) {
  ST_REQUIRE(x && workspace && out, "st_tv_fwd: null pointer");
  ST_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, "st_tv_fwd: bad dims %d %d %d %d", n, c, h, w);
  const long long total = (long long)n * c * h * w;
  const int parts = parts_for(total);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(tv_partial_kernel, dim3(parts), dim3(NT), 0, s, x, h, w, total, workspace, MAX_PARTS);
  // losses.py:107-112: (2 * (sum_h / count_h + sum_w / count_w) / batch) * 100
  const float count_h = float(c) * float(h - 1) * float(w), count_w = float(c) * float(h) * float(w - 1);
  // an empty slice (h == 1 or w == 1) contributes 0 in the reference (sum of an empty tensor)
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(NT), 0, s, workspace, parts, MAX_PARTS, 2,
                     count_h > 0.f ? 200.f / (count_h * float(n)) : 0.f,
                     count_w > 0.f ? 200.f / (count_w * float(n)) : 0.f, out, 1);
  return st::check_launch("st_tv_fwd");
}

extern "C" int st_tv_bwd(const float* x, const float* gout, float scale, float* dx, int n, int c, int h,
                         int w, int accumulate, st_stream_t stream) {
  ST_REQUIRE(x && dx, "st_tv_bwd: null pointer");
  ST_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, "st_tv_bwd: bad dims %d %d %d %d", n, c, h, w);
  const long long total = (long long)n * c * h * w;
  const float count_h = float(c) * float(h - 1) * float(w), count_w = float(c) * float(h) * float(w - 1);
  const int blocks = (int)std::min<long long>(st::cdivll(total, NT), 2048);
  hipLaunchKernelGGL(tv_bwd_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, x, gout,
                     count_h > 0.f ? scale * 400.f / (count_h * float(n)) : 0.f,
                     count_w > 0.f ? scale * 400.f / (count_w * float(n)) : 0.f, dx, h, w, total, accumulate);
  return st::check_launch("st_tv_bwd");
}

extern "C" int st_meanstd_loss_fwd(const float* mean_a, const float* std_a, const float* mean_b,
                                   const float* std_b, int planes, float* workspace, float* out,
                                   st_stream_t stream) {
  ST_REQUIRE(mean_a && std_a && mean_b && std_b && workspace && out, "st_meanstd_loss_fwd: null pointer");
  ST_REQUIRE(planes > 0, "st_meanstd_loss_fwd: planes=%d", planes);
  const int parts = parts_for(planes);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(meanstd_loss_partial_kernel, dim3(parts), dim3(NT), 0, s, mean_a, std_a, mean_b,
                     std_b, planes, workspace);
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(NT), 0, s, workspace, parts, MAX_PARTS, 1,
                     1.f / float(planes), 0.f, out, 0);
  return st::check_launch("st_meanstd_loss_fwd");
}

extern "C" int st_meanstd_loss_bwd(const float* x, const float* mean_a, const float* std_a,
                                   const float* mean_b, const float* std_b, const float* gout, float scale,
                                   float* dx, int planes, int hw, int accumulate, st_stream_t stream) {
  ST_REQUIRE(x && mean_a && std_a && mean_b && std_b && dx, "st_meanstd_loss_bwd: null pointer");
  ST_REQUIRE(planes > 0 && hw > 0, "st_meanstd_loss_bwd: bad dims planes=%d hw=%d", planes, hw);
  const dim3 grid((unsigned)planes, (unsigned)((hw + BWD_SLICE - 1) / BWD_SLICE));
  ST_REQUIRE(grid.y <= 65535u, "st_meanstd_loss_bwd: plane too large (hw=%d)", hw);
  const bool v4 = hw % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(dx) % 16 == 0;
  if (v4)
    hipLaunchKernelGGL(meanstd_loss_bwd_kernel<true>, grid, dim3(NT), 0, (hipStream_t)stream, x, mean_a, std_a, mean_b,
                       std_b, gout, scale, dx, planes, hw, accumulate);
  else
    hipLaunchKernelGGL(meanstd_loss_bwd_kernel<false>, grid, dim3(NT), 0, (hipStream_t)stream, x, mean_a, std_a,
                       mean_b, std_b, gout, scale, dx, planes, hw, accumulate);
  return st::check_launch("st_meanstd_loss_bwd");
}
