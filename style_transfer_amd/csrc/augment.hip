// GPU-side training augmentations: the per-image tensor transforms the reference composes after ToTensor
// (src/train.py:116-135) -- RGBToGrayscaleStacked and AddGaussianNoise of src/utils/custom_transform.py:9-46,
// torchvision's RandomRotation (nearest, same canvas, zero fill), RandomAdjustSharpness, ColorJitter and
// RandomHorizontalFlip -- applied to a whole batch [n][3][h][w] that already sits in HBM, with per-image parameters
// (which image gets which transform, angles, factors, the jitter order) drawn on the host and passed as small device
// arrays.  In the reference these run per image on the CPU inside DataLoader workers; once a training step takes
// 26 ms for 8 images that pipeline, not the step, bounds the epoch.
//
// All kernels are single-pass and HBM-bound (a 3 x 256 x 256 image is 786 KB); grid = (blocks per image, n).
// Per-image parameter p[i] < 0 (or flag 0) = "this image is left alone": the kernel copies it.
#include <algorithm>
#include <cfloat>

#include "common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
// torchvision rgb_to_grayscale (ColorJitter / sharpness paths): 0.2989 R + 0.587 G + 0.114 B
__device__ __forceinline__ float gray_tv(float r, float g, float b) { return 0.2989f * r + 0.587f * g + 0.114f * b; }

// ---- RGBToGrayscaleStacked (custom_transform.py:32-46; weights 0.299 / 0.587 / 0.144 as written there) --------
__global__ __launch_bounds__(NT) void aug_gray_stack_kernel(const float* __restrict__ x, float* __restrict__ y, int hw) {
  const size_t base = (size_t)blockIdx.y * 3 * hw;
  for (int p = blockIdx.x * NT + threadIdx.x; p < hw; p += gridDim.x * NT) {
    const float r = x[base + p] * 0.299f, g = x[base + hw + p] * 0.587f, b = x[base + 2 * (size_t)hw + p] * 0.144f;
    const float v = (r + g) + b;                       // torch.sum over the channel dim: sequential
    y[base + p] = v;
    y[base + hw + p] = v;
    y[base + 2 * (size_t)hw + p] = v;
  }
}

// ---- RandomRotation: F.affine_grid + F.grid_sample(nearest, zeros, align_corners=False) ------------------------
// theta = [[cos, sin h/w, 0], [-sin w/h, cos, 0]]; output pixel (x, y) samples the input at the rounded
// (half-to-even, like nearbyint) un-normalised source coordinate; outside the image: 0.
__global__ __launch_bounds__(NT) void aug_rotate_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ angle_rad, int c, int h, int w) {
  const int img = blockIdx.y;
  const float a = angle_rad[img];
  const float cs = cosf(a), sn = sinf(a);
  const float t01 = sn * float(h) / float(w), t10 = -sn * float(w) / float(h);
  const size_t base = (size_t)img * c * h * w;
  const int hw = h * w;
  for (int p = blockIdx.x * NT + threadIdx.x; p < hw; p += gridDim.x * NT) {
    const int py = p / w, px = p - py * w;
    const float xn = (2.f * float(px) + 1.f) / float(w) - 1.f, yn = (2.f * float(py) + 1.f) / float(h) - 1.f;
    const float xs = cs * xn + t01 * yn, ys = t10 * xn + cs * yn;
    const float fx = ((xs + 1.f) * float(w) - 1.f) * 0.5f, fy = ((ys + 1.f) * float(h) - 1.f) * 0.5f;
    const int ix = int(rintf(fx)), iy = int(rintf(fy));
    const bool in = ix >= 0 && ix < w && iy >= 0 && iy < h;
    for (int ch = 0; ch < c; ++ch)
      y[base + (size_t)ch * hw + p] = in ? x[base + (size_t)ch * hw + (size_t)iy * w + ix] : 0.f;
  }
}

// ---- per-image min / max (AddGaussianNoise's two min-max renormalisations): one block per image ---------------
__global__ __launch_bounds__(1024) void aug_minmax_kernel(const float* __restrict__ x, float* __restrict__ mm, int elems) {
  __shared__ float smin[16], smax[16];
  const float* p = x + (size_t)blockIdx.x * elems;
  float lo = FLT_MAX, hi = -FLT_MAX;
  for (int e = threadIdx.x; e < elems; e += 1024) {
    const float v = p[e];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off, 64));
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
  }
  if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; ++k) { lo = fminf(lo, smin[k]); hi = fmaxf(hi, smax[k]); }
    lo = fminf(lo, smin[0]); hi = fmaxf(hi, smax[0]);
    mm[2 * blockIdx.x] = lo;
    mm[2 * blockIdx.x + 1] = hi;
  }
}

// np.interp(v, (lo, hi), (0, 1)): linear map with the ends pinned; lo == hi maps everything to the upper value
__device__ __forceinline__ float unit_interp(float v, float lo, float hi) {
  if (!(hi > lo)) return 1.f;
  return fminf(fmaxf((v - lo) / (hi - lo), 0.f), 1.f);
}

// AddGaussianNoise, first half (custom_transform.py:18-23): t = interp(x, (min, max), (0, 1)) + noise, for the
// images whose sigma[i] >= 0; `noise` holds N(mean, sigma_i) samples.  Others are copied.
__global__ __launch_bounds__(NT) void aug_noise_add_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                                          const float* __restrict__ mm, const float* __restrict__ sigma,
                                                          float* __restrict__ t, int elems) {
  const int img = blockIdx.y;
  const bool on = sigma[img] >= 0.f;
  const float lo = mm[2 * img], hi = mm[2 * img + 1];
  const size_t base = (size_t)img * elems;
  for (int e = blockIdx.x * NT + threadIdx.x; e < elems; e += gridDim.x * NT) {
    const float v = x[base + e];
    t[base + e] = on ? unit_interp(v, lo, hi) + noise[base + e] : v;
  }
}

// second half (custom_transform.py:24-26): interp to (0, 255), truncate to uint8, ToTensor (/ 255)
__global__ __launch_bounds__(NT) void aug_noise_quant_kernel(float* __restrict__ t, const float* __restrict__ mm,
                                                            const float* __restrict__ sigma, int elems) {
  const int img = blockIdx.y;
  if (!(sigma[img] >= 0.f)) return;
  const float lo = mm[2 * img], hi = mm[2 * img + 1];
  const size_t base = (size_t)img * elems;
  for (int e = blockIdx.x * NT + threadIdx.x; e < elems; e += gridDim.x * NT) {
    const float u = unit_interp(t[base + e], lo, hi) * 255.f;
    t[base + e] = floorf(u) / 255.f;
  }
}

// ---- RandomAdjustSharpness: blend(img, blur3x3(img) on the interior, factor), clamp -----------------------------
__global__ __launch_bounds__(NT) void aug_sharpness_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          const float* __restrict__ factor, int c, int h, int w) {
  const int img = blockIdx.y;
  const float f = factor[img];                          // < 0: not selected
  const size_t base = (size_t)img * c * h * w;
  const int total = c * h * w;
  const bool tiny = h <= 2 || w <= 2;
  for (int e = blockIdx.x * NT + threadIdx.x; e < total; e += gridDim.x * NT) {
    const int px = e % w, py = (e / w) % h;
    const float v = x[base + e];
    float out = v;
    if (f >= 0.f && !tiny) {
      float blur = v;
      if (px > 0 && px < w - 1 && py > 0 && py < h - 1) {
        const float* q = x + base + e;
        const float s = q[-w - 1] + q[-w] + q[-w + 1] + q[-1] + q[1] + q[w - 1] + q[w] + q[w + 1];
        blur = (s + 5.f * v) / 13.f;
      }
      out = clamp01(f * v + (1.f - f) * blur);
    }
    y[base + e] = out;
  }
}

// ---- ColorJitter: one of the four adjustments per launch ("slot"), chosen per image -------------------------
// code[i]: 0 brightness, 1 contrast, 2 saturation, 3 hue, anything else: copy.  param[i]: the factor (hue: the shift).
// gmean[i]: mean of the grayscale image (contrast), computed by aug_gray_mean_kernel on the slot's INPUT.
__global__ __launch_bounds__(1024) void aug_gray_mean_kernel(const float* __restrict__ x, const int* __restrict__ code,
                                                            float* __restrict__ gmean, int hw) {
  __shared__ float red[16];
  const int img = blockIdx.x;
  if (code[img] != 1) return;                           // only contrast needs it (uniform per block)
  const float* p = x + (size_t)img * 3 * hw;
  float s = 0.f;
  for (int e = threadIdx.x; e < hw; e += 1024) s += gray_tv(p[e], p[hw + e], p[2 * (size_t)hw + e]);
  s = st::wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += red[k];
    gmean[img] = t / float(hw);
  }
}

__device__ __forceinline__ void hue_shift(float& r, float& g, float& b, float shift) {
  // torchvision _rgb2hsv / _hsv2rgb
  const float maxc = fmaxf(r, fmaxf(g, b)), minc = fminf(r, fminf(g, b));
  const bool eqc = maxc == minc;
  const float cr = maxc - minc;
  const float s = cr / (eqc ? 1.f : maxc);
  const float crd = eqc ? 1.f : cr;
  const float rc = (maxc - r) / crd, gc = (maxc - g) / crd, bc = (maxc - b) / crd;
  const float hr = (maxc == r) ? (bc - gc) : 0.f;
  const float hg = (maxc == g && maxc != r) ? (2.f + rc - bc) : 0.f;
  const float hb = (maxc != g && maxc != r) ? (4.f + gc - rc) : 0.f;
  float hh = fmodf((hr + hg + hb) / 6.f + 1.f, 1.f);
  hh = hh + shift;
  hh = hh - floorf(hh);                                 // python's % 1.0
  const float v = maxc;
  const float i6 = floorf(hh * 6.f);
  const float f = hh * 6.f - i6;
  int i = int(i6) % 6;
  if (i < 0) i += 6;
  const float p = clamp01(v * (1.f - s)), q = clamp01(v * (1.f - s * f)), t = clamp01(v * (1.f - s * (1.f - f)));
  switch (i) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

__global__ __launch_bounds__(NT) void aug_jitter_slot_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            const int* __restrict__ code, const float* __restrict__ param,
                                                            const float* __restrict__ gmean, int hw) {
  const int img = blockIdx.y;
  const int cd = code[img];
  const float f = param[img];
  const float gm = cd == 1 ? gmean[img] : 0.f;
  const size_t base = (size_t)img * 3 * hw;
  for (int p = blockIdx.x * NT + threadIdx.x; p < hw; p += gridDim.x * NT) {
    float r = x[base + p], g = x[base + hw + p], b = x[base + 2 * (size_t)hw + p];
    if (cd == 0) {                                      // brightness: blend with black
      r = clamp01(f * r); g = clamp01(f * g); b = clamp01(f * b);
    } else if (cd == 1) {                               // contrast: blend with the mean grey level
      const float k = (1.f - f) * gm;
      r = clamp01(f * r + k); g = clamp01(f * g + k); b = clamp01(f * b + k);
    } else if (cd == 2) {                               // saturation: blend with the pixel's grey level
      const float k = (1.f - f) * gray_tv(r, g, b);
      r = clamp01(f * r + k); g = clamp01(f * g + k); b = clamp01(f * b + k);
    } else if (cd == 3) {
      hue_shift(r, g, b, f);
    }
    y[base + p] = r;
    y[base + hw + p] = g;
    y[base + 2 * (size_t)hw + p] = b;
  }
}

// ---- RandomHorizontalFlip --------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void aug_hflip_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      const int* __restrict__ flip, int c, int h, int w) {
  const int img = blockIdx.y;
  const bool fl = flip[img] != 0;
  const size_t base = (size_t)img * c * h * w;
  const int total = c * h * w;
  for (int e = blockIdx.x * NT + threadIdx.x; e < total; e += gridDim.x * NT) {
    const int px = e % w;
    y[base + e] = x[base + (fl ? e - px + (w - 1 - px) : e)];
  }
}

inline dim3 grid_for(long long per_image, int n) {
  return dim3((unsigned)std::max<long long>(1, std::min<long long>(st::cdivll(per_image, NT), 256)), (unsigned)n);
}

}  // namespace

#define ST_AUG_CHECK(name, cond) ST_REQUIRE(cond, name ": bad arguments (null pointer or non-positive size)")

extern "C" int st_aug_gray_stack(const float* x, float* y, int n, int hw, st_stream_t stream) {
  ST_AUG_CHECK("st_aug_gray_stack", x && y && n > 0 && hw > 0);
  hipLaunchKernelGGL(aug_gray_stack_kernel, grid_for(hw, n), dim3(NT), 0, (hipStream_t)stream, x, y, hw);
  return st::check_launch("st_aug_gray_stack");
}

extern "C" int st_aug_rotate(const float* x, float* y, const float* angle_rad_dev, int n, int c, int h, int w,
                             st_stream_t stream) {
  ST_AUG_CHECK("st_aug_rotate", x && y && angle_rad_dev && n > 0 && c > 0 && h > 0 && w > 0);
  ST_REQUIRE(x != y, "st_aug_rotate: in-place rotation is not possible");
  hipLaunchKernelGGL(aug_rotate_kernel, grid_for((long long)h * w, n), dim3(NT), 0, (hipStream_t)stream, x, y,
                     angle_rad_dev, c, h, w);
  return st::check_launch("st_aug_rotate");
}

extern "C" int st_aug_gaussian_noise(const float* x, const float* noise, const float* sigma_dev, float* y,
                                     float* minmax_ws, int n, int elems_per_image, st_stream_t stream) {
  ST_AUG_CHECK("st_aug_gaussian_noise", x && noise && sigma_dev && y && minmax_ws && n > 0 && elems_per_image > 0);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(aug_minmax_kernel, dim3(n), dim3(1024), 0, s, x, minmax_ws, elems_per_image);
  hipLaunchKernelGGL(aug_noise_add_kernel, grid_for(elems_per_image, n), dim3(NT), 0, s, x, noise, minmax_ws, sigma_dev,
                     y, elems_per_image);
  hipLaunchKernelGGL(aug_minmax_kernel, dim3(n), dim3(1024), 0, s, y, minmax_ws, elems_per_image);
  hipLaunchKernelGGL(aug_noise_quant_kernel, grid_for(elems_per_image, n), dim3(NT), 0, s, y, minmax_ws, sigma_dev,
                     elems_per_image);
  return st::check_launch("st_aug_gaussian_noise");
}

extern "C" int st_aug_sharpness(const float* x, float* y, const float* factor_dev, int n, int c, int h, int w,
                                st_stream_t stream) {
  ST_AUG_CHECK("st_aug_sharpness", x && y && factor_dev && n > 0 && c > 0 && h > 0 && w > 0);
  ST_REQUIRE(x != y, "st_aug_sharpness: in-place filtering is not possible");
  hipLaunchKernelGGL(aug_sharpness_kernel, grid_for((long long)c * h * w, n), dim3(NT), 0, (hipStream_t)stream, x, y,
                     factor_dev, c, h, w);
  return st::check_launch("st_aug_sharpness");
}

extern "C" int st_aug_color_jitter_slot(const float* x, float* y, const int* code_dev, const float* param_dev,
                                        float* gray_mean_ws, int n, int hw, st_stream_t stream) {
  ST_AUG_CHECK("st_aug_color_jitter_slot", x && y && code_dev && param_dev && gray_mean_ws && n > 0 && hw > 0);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(aug_gray_mean_kernel, dim3(n), dim3(1024), 0, s, x, code_dev, gray_mean_ws, hw);
  hipLaunchKernelGGL(aug_jitter_slot_kernel, grid_for(hw, n), dim3(NT), 0, s, x, y, code_dev, param_dev, gray_mean_ws, hw);
  return st::check_launch("st_aug_color_jitter_slot");
}

extern "C" int st_aug_hflip(const float* x, float* y, const int* flip_dev, int n, int c, int h, int w,
                            st_stream_t stream) {
  ST_AUG_CHECK("st_aug_hflip", x && y && flip_dev && n > 0 && c > 0 && h > 0 && w > 0);
  ST_REQUIRE(x != y, "st_aug_hflip: in-place flip is not possible");
  hipLaunchKernelGGL(aug_hflip_kernel, grid_for((long long)c * h * w, n), dim3(NT), 0, (hipStream_t)stream, x, y,
                     flip_dev, c, h, w);
  return st::check_launch("st_aug_hflip");
}
