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

// Row-strip form of both TV kernels (w % 4 == 0, 16-B aligned planes): a group of LPR threads owns a strip of
// TV_ROWS rows of one plane and walks down it; a thread keeps the float4 of its 4 columns (plus the column to the
// right) of the previous row(s) in registers, so every element is loaded ONCE from HBM (+ 1 / TV_ROWS for the strip's
// extra row) with 16-B accesses -- the element-wise kernels above issue 3 (forward) / 5 (backward) dword loads and a
// 64-bit modulo per element (measured 1.1 TB/s at [16,3,2048,2048]).
constexpr int TV_ROWS = 32;
constexpr int TV_BATCH = 8;     // rows whose loads are in flight together per thread
typedef float tvf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sq(float v) { return v * v; }

// value of column 4 (the first column of the thread to the right): a lane shuffle inside the wave; only the last
// thread of a row group (its neighbour is the next column pass or another group) and lane 63 load it
template <int LPR>
__device__ __forceinline__ float right_of(const float* p, float mine0, bool right, int t) {
  const float nb = __shfl_down(mine0, 1, 64);
  const bool own = t == LPR - 1 || (threadIdx.x & 63) == 63;
  return !right ? 0.f : (own ? p[4] : nb);
}
template <int LPR>
__device__ __forceinline__ float left_of(const float* p, float mine3, bool left, int t) {
  const float nb = __shfl_up(mine3, 1, 64);
  const bool own = t == 0 || (threadIdx.x & 63) == 0;
  return !left ? 0.f : (own ? p[-1] : nb);
}

template <int LPR>   // threads per row (power of two <= NT); NT / LPR strips per block at a time
__global__ __launch_bounds__(NT) void tv_rows_partial_kernel(const float* __restrict__ x, int h, int w,
                                                            int strips_per_plane, long long nstrips,
                                                            float* __restrict__ partials, int stride) {
  __shared__ float red[NT / 64];
  constexpr int GROUPS = NT / LPR;
  const int g = threadIdx.x / LPR, t = threadIdx.x % LPR;
  const int w4 = w >> 2;
  const int passes = (w4 + LPR - 1) / LPR;          // uniform trip count: every lane takes part in the shuffles
  float sh = 0.f, sw = 0.f;
  for (long long s = (long long)blockIdx.x * GROUPS + g; s < nstrips; s += (long long)gridDim.x * GROUPS) {
    const long long plane = s / strips_per_plane;
    const int y0 = int(s - plane * strips_per_plane) * TV_ROWS, y1 = min(y0 + TV_ROWS, h);
    const int yend = min(y1, h - 1);                // rows y0 .. yend-1 have a row below them
    const float* xp = x + plane * (long long)h * w;
    for (int ps0 = 0; ps0 < passes; ++ps0) {
      // Rotate the order of the column passes from strip to strip: with a power-of-two row pitch every row of a
      // 4 KB column band maps to the same half of the HBM channels, and blocks that all start at band 0 leave the
      // other half idle (measured 3.7 -> TB/s at 2048-wide rows).
      const int ps = int((ps0 + s) % passes);
      const int c4 = ps * LPR + t;
      const bool live = c4 < w4;
      const int col = min(c4, w4 - 1) * 4;          // dead lanes shadow the last column group, contribute nothing
      const bool right = col + 4 < w;
      const float* p = xp + (size_t)y0 * w + col;
      tvf4 prev = __builtin_nontemporal_load(reinterpret_cast<const tvf4*>(p));
      float prev_r = right_of<LPR>(p, prev[0], right, t);
      float ah = 0.f, aw = 0.f;
      // batches of TV_BATCH rows: all their 16-B loads are issued before the first one is consumed (read-once data:
      // non-temporal loads); row indices past the strip are clamped and their contribution masked (uniform)
      for (int y = y0; y < yend; y += TV_BATCH) {
        tvf4 rows[TV_BATCH];
        float rr[TV_BATCH];
#pragma unroll
        for (int k = 0; k < TV_BATCH; ++k) {
          const int yy = min(y + 1 + k, h - 1);
          rows[k] = __builtin_nontemporal_load(reinterpret_cast<const tvf4*>(xp + (size_t)yy * w + col));
        }
#pragma unroll
        for (int k = 0; k < TV_BATCH; ++k) {
          const int yy = min(y + 1 + k, h - 1);
          rr[k] = right_of<LPR>(xp + (size_t)yy * w + col, rows[k][0], right, t);
        }
#pragma unroll
        for (int k = 0; k < TV_BATCH; ++k) {
          if (y + k < yend) {
            const tvf4 cur = rows[k];
            aw += sq(prev[1] - prev[0]) + sq(prev[2] - prev[1]) + sq(prev[3] - prev[2]) +
                  (right ? sq(prev_r - prev[3]) : 0.f);
            ah += sq(cur[0] - prev[0]) + sq(cur[1] - prev[1]) + sq(cur[2] - prev[2]) + sq(cur[3] - prev[3]);
            prev = cur;
            prev_r = rr[k];
          }
        }
      }
      if (y1 == h)                                   // the image's last row: horizontal term only
        aw += sq(prev[1] - prev[0]) + sq(prev[2] - prev[1]) + sq(prev[3] - prev[2]) + (right ? sq(prev_r - prev[3]) : 0.f);
      if (live) { sh += ah; sw += aw; }
    }
  }
  sh = st::block_sum<NT>(sh, red);
  sw = st::block_sum<NT>(sw, red);
  if (threadIdx.x == 0) { partials[blockIdx.x] = sh; partials[stride + blockIdx.x] = sw; }
}

template <int LPR>
__global__ __launch_bounds__(NT) void tv_rows_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout,
                                                        float ch, float cw, float* __restrict__ dx, int h, int w,
                                                        int strips_per_plane, long long nstrips, int accumulate) {
  constexpr int GROUPS = NT / LPR;
  const int g = threadIdx.x / LPR, t = threadIdx.x % LPR;
  const int w4 = w >> 2;
  const int passes = (w4 + LPR - 1) / LPR;
  const float g0 = gout ? gout[0] : 1.f;
  const float kh = g0 * ch, kw = g0 * cw;
  for (long long s = (long long)blockIdx.x * GROUPS + g; s < nstrips; s += (long long)gridDim.x * GROUPS) {
    const long long plane = s / strips_per_plane;
    const int y0 = int(s - plane * strips_per_plane) * TV_ROWS, y1 = min(y0 + TV_ROWS, h);
    const float* xp = x + plane * (long long)h * w;
    float* dp = dx + plane * (long long)h * w;
    for (int ps0 = 0; ps0 < passes; ++ps0) {
      // Rotate the order of the column passes from strip to strip: with a power-of-two row pitch every row of a
      // 4 KB column band maps to the same half of the HBM channels, and blocks that all start at band 0 leave the
      // other half idle (measured 3.7 -> TB/s at 2048-wide rows).
      const int ps = int((ps0 + s) % passes);
      const int c4 = ps * LPR + t;
      const bool live = c4 < w4;
      const int col = min(c4, w4 - 1) * 4;
      const bool left = col > 0, right = col + 4 < w;
      const float* p = xp + (size_t)y0 * w + col;
      tvf4 cur = *reinterpret_cast<const tvf4*>(p);
      tvf4 up = y0 > 0 ? *reinterpret_cast<const tvf4*>(p - w) : cur;     // no row above: difference 0
      float cur_l = left_of<LPR>(p, cur[3], left, t), cur_r = right_of<LPR>(p, cur[0], right, t);
#pragma unroll 4
      for (int y = y0; y < y1; ++y) {
        const bool below = y + 1 < h;                                     // uniform over the strip's threads
        const float* pn = below ? p + w : p;
        const tvf4 down = *reinterpret_cast<const tvf4*>(pn);             // no row below: down == cur, difference 0
        const float down_l = left_of<LPR>(pn, down[3], left, t), down_r = right_of<LPR>(pn, down[0], right, t);
        const float l = left ? cur_l : cur[0], r = right ? cur_r : cur[3];   // absent neighbours alias the pixel
        tvf4 out;
        out[0] = kh * ((cur[0] - up[0]) - (down[0] - cur[0])) + kw * ((cur[0] - l) - (cur[1] - cur[0]));
        out[1] = kh * ((cur[1] - up[1]) - (down[1] - cur[1])) + kw * ((cur[1] - cur[0]) - (cur[2] - cur[1]));
        out[2] = kh * ((cur[2] - up[2]) - (down[2] - cur[2])) + kw * ((cur[2] - cur[1]) - (cur[3] - cur[2]));
        out[3] = kh * ((cur[3] - up[3]) - (down[3] - cur[3])) + kw * ((cur[3] - cur[2]) - (r - cur[3]));
        float* q = dp + (size_t)y * w + col;
        if (live) {
          if (accumulate) out += *reinterpret_cast<const tvf4*>(q);
          *reinterpret_cast<tvf4*>(q) = out;
        }
        up = cur; cur = down; cur_l = down_l; cur_r = down_r;
        p = pn;
      }
    }
  }
}

template <int LPR>
void launch_tv_rows(bool bwd, const float* x, const float* gout, float ch, float cw, float* dx, float* partials,
                    int h, int w, long long planes, int accumulate, int* parts_out, hipStream_t s) {
  const int strips_per_plane = st::cdiv(h, TV_ROWS);
  const long long nstrips = planes * strips_per_plane;
  constexpr int GROUPS = NT / LPR;
  const int blocks = (int)std::max<long long>(1, std::min<long long>(st::cdivll(nstrips, GROUPS), bwd ? 4096 : MAX_PARTS));
  if (bwd)
    hipLaunchKernelGGL(tv_rows_bwd_kernel<LPR>, dim3(blocks), dim3(NT), 0, s, x, gout, ch, cw, dx, h, w,
                       strips_per_plane, nstrips, accumulate);
  else
    hipLaunchKernelGGL(tv_rows_partial_kernel<LPR>, dim3(blocks), dim3(NT), 0, s, x, h, w, strips_per_plane, nstrips,
                       partials, MAX_PARTS);
  if (parts_out) *parts_out = blocks;
}

// picks the threads-per-row instantiation for the row width; false when the row form does not apply
bool tv_rows(bool bwd, const float* x, const float* gout, float ch, float cw, float* dx, float* partials, int h, int w,
             long long planes, int accumulate, int* parts_out, hipStream_t s) {
  const bool ok = w % 4 == 0 && w >= 8 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                  (!bwd || reinterpret_cast<uintptr_t>(dx) % 16 == 0);
  if (!ok) return false;
  const int w4 = w / 4;
  if (w4 > 128) launch_tv_rows<256>(bwd, x, gout, ch, cw, dx, partials, h, w, planes, accumulate, parts_out, s);
  else if (w4 > 64) launch_tv_rows<128>(bwd, x, gout, ch, cw, dx, partials, h, w, planes, accumulate, parts_out, s);
  else if (w4 > 32) launch_tv_rows<64>(bwd, x, gout, ch, cw, dx, partials, h, w, planes, accumulate, parts_out, s);
  else if (w4 > 16) launch_tv_rows<32>(bwd, x, gout, ch, cw, dx, partials, h, w, planes, accumulate, parts_out, s);
  else launch_tv_rows<16>(bwd, x, gout, ch, cw, dx, partials, h, w, planes, accumulate, parts_out, s);
  return true;
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

// The per-plane scalars of meanstd_loss_bwd_kernel -- (k0, k1, mean, 0), the SAME expressions -- for a consumer that
// evaluates the gradient k0 + k1 * (x - mean) itself while it reads x anyway (st_grad_join_g8_affine).
__global__ __launch_bounds__(256) void meanstd_loss_coeffs_kernel(const float* __restrict__ ma,
                                                                 const float* __restrict__ sa,
                                                                 const float* __restrict__ mb,
                                                                 const float* __restrict__ sb,
                                                                 const float* __restrict__ gout, float scale,
                                                                 float* __restrict__ coeffs, int planes, int hw) {
  const int plane = blockIdx.x * blockDim.x + threadIdx.x;
  if (plane >= planes) return;
  const float c = scale * (gout ? gout[0] : 1.f) * 2.f / float(planes);
  const float mean = ma[plane], sd = sa[plane];
  const float k0 = c * (mean - mb[plane]) / float(hw);
  const float k1 = c * (sd - sb[plane]) / (float(hw - 1) * sd);
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 o;
  o[0] = k0; o[1] = k1; o[2] = mean; o[3] = 0.f;
  reinterpret_cast<f4*>(coeffs)[plane] = o;
}

}  // namespace

extern "C" int st_meanstd_loss_bwd_coeffs(const float* mean_a, const float* std_a, const float* mean_b,
                                          const float* std_b, const float* gout, float scale, float* coeffs,
                                          int planes, int hw, st_stream_t stream) {
  ST_REQUIRE(mean_a && std_a && mean_b && std_b && coeffs, "st_meanstd_loss_bwd_coeffs: null pointer");
  ST_REQUIRE(planes > 0 && hw > 1, "st_meanstd_loss_bwd_coeffs: planes=%d hw=%d", planes, hw);
  ST_REQUIRE(reinterpret_cast<uintptr_t>(coeffs) % 16 == 0, "st_meanstd_loss_bwd_coeffs: coeffs must be 16-byte aligned");
  hipLaunchKernelGGL(meanstd_loss_coeffs_kernel, dim3(st::cdiv(planes, 256)), dim3(256), 0, (hipStream_t)stream, mean_a,
                     std_a, mean_b, std_b, gout, scale, coeffs, planes, hw);
  return st::check_launch("st_meanstd_loss_bwd_coeffs");
}

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
  int parts = parts_for(total);
  hipStream_t s = (hipStream_t)stream;
  if (!tv_rows(false, x, nullptr, 0.f, 0.f, nullptr, workspace, h, w, (long long)n * c, 0, &parts, s))
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
  const float kh = count_h > 0.f ? scale * 400.f / (count_h * float(n)) : 0.f;
  const float kw = count_w > 0.f ? scale * 400.f / (count_w * float(n)) : 0.f;
  if (!tv_rows(true, x, gout, kh, kw, dx, nullptr, h, w, (long long)n * c, accumulate, nullptr, (hipStream_t)stream))
    hipLaunchKernelGGL(tv_bwd_kernel, dim3(blocks), dim3(NT), 0, (hipStream_t)stream, x, gout, kh, kw, dx, h, w,
                       total, accumulate);
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
