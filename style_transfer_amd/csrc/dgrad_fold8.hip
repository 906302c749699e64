// Adjoint of the edge-replicate pad (+ optional nearest x2 upsample) on the padded data gradient of a decoder layer
// (generator.py:85-130 backward), eight channels per work item, for the bf16 compute mode's G8 chain:
//
//   dxp [n*c][H+2][Wp] fp32 (the data-gradient kernel's padded-domain result)
//     -> s = sum of dxp over every padded position the forward read from source pixel (i, j)     (as dgrad_fold4)
//     -> s = 0 where the layer's input activation (the ReLU output of the layer below) is <= 0     (optional gate)
//   and s, the gradient w.r.t. the output of the layer below, leaves in every form its consumers take:
//     dx   fp32 NCHW                       (the bottom of the chain: the gradient handed back to AdaIN)
//     g8   bf16 [n][c/8][h][w][8]          (operand of the layer below's data-gradient launch, staged by LDS-DMA)
//     g16  bf16 [n][c][h][w]               (operand "G" of the layer below's weight-gradient GEMM)
//     bias [slab][c] fp32 partial sums     (the layer below's bias gradient; summed by the fixed-order slab reduce)
// -- one pass instead of fold + (weight-gradient prep: read fp32, write bf16, row sums) + the fp32 staging of the
// next data-gradient launch.  Same fp32 arithmetic per element as dgrad_fold4_kernel (csrc/conv3x3.hip).
// HBM-bound: reads 4 B (x4 under the upsample) per element, writes 2 + 2 B.
#include <algorithm>

#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct Fold8Params {
  const float* dxp;
  const float* gate_f32;     // fp32 NCHW [n][c][hs][ws] or null
  const u32x4* gate_g8;      // G8 [n][c/8][hs][ws][8] or null
  float* dx;                 // optional outputs
  u32x4* g8;
  unsigned short* g16;
  float* bias;               // [n * gridDim.x][c]
  int c, H, W, hs, ws, iters;
};

template <bool UP>
__global__ __launch_bounds__(256) void dgrad_fold8_kernel(Fold8Params p) {
  __shared__ float red[4][8];
  const int Wp = st::padded_row(p.W);
  const int q = p.ws >> 2;                       // items per source row
  const int items = p.hs * q;
  const int cgs = p.c >> 3;
  const int cg = blockIdx.y, b = blockIdx.z;
  const size_t plane_p = size_t(p.H + 2) * Wp, plane_s = size_t(p.hs) * p.ws;
  const float* src0 = p.dxp + (size_t(b) * p.c + cg * 8) * plane_p;
  float bsum[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bsum[j] = 0.f;
  for (int it = 0; it < p.iters; ++it) {
    const int item = (blockIdx.x * p.iters + it) * 256 + threadIdx.x;
    if (item >= items) break;
    const int i = item / q;
    const int j0 = (item - i * q) * 4;
    int u0 = UP ? 2 * i : i, u1 = UP ? 2 * i + 1 : i;
    if (u0 == 0) u0 = -1;
    if (u1 == p.H - 1) u1 = p.H;
    const bool left = j0 == 0, right = j0 + 4 == p.ws;
    float s[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j][0] = s[j][1] = s[j][2] = s[j][3] = 0.f;
    // rows outermost: the loads of all eight channels of a padded row are independent and issue back to back (with the
    // channel loop outside, each channel's loads waited behind the row loop's dynamic trip count: 8 serial latencies)
    for (int u = u0; u <= u1; ++u) {
      const float* r0 = src0 + size_t(u + 1) * Wp + (UP ? 2 * j0 : j0);
      f4 a[8], bb[8];
      float e0[8], e1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* r = r0 + j * plane_p;
        a[j] = *reinterpret_cast<const f4*>(r);
        if (UP) bb[j] = *reinterpret_cast<const f4*>(r + 4);
        e0[j] = r[UP ? 8 : 4];
        e1[j] = right ? r[UP ? 9 : 5] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (UP) {
          s[j][0] += a[j][1] + a[j][2] + (left ? a[j][0] : 0.f);
          s[j][1] += a[j][3] + bb[j][0];
          s[j][2] += bb[j][1] + bb[j][2];
          s[j][3] += bb[j][3] + e0[j] + e1[j];
        } else {
          s[j][0] += a[j][1] + (left ? a[j][0] : 0.f);
          s[j][1] += a[j][2];
          s[j][2] += a[j][3];
          s[j][3] += e0[j] + e1[j];
        }
      }
    }
    const size_t pix = size_t(i) * p.ws + j0;                       // first of the 4 pixels inside a source plane
    if (p.gate_g8) {
      const u32x4* gp = p.gate_g8 + (size_t(b) * cgs + cg) * plane_s + pix;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32x4 g = gp[k];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned hb = (j & 1) ? (g[j >> 1] >> 16) : (g[j >> 1] & 0xffffu);
          if (!((hb - 1u) < 0x7f80u)) s[j][k] = 0.f;                // keep where the bf16 activation is > 0
        }
      }
    } else if (p.gate_f32) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f4 g = __builtin_nontemporal_load(
            reinterpret_cast<const f4*>(p.gate_f32 + (size_t(b) * p.c + cg * 8 + j) * plane_s + pix));
#pragma unroll
        for (int k = 0; k < 4; ++k) s[j][k] = g[k] > 0.f ? s[j][k] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bsum[j] += (s[j][0] + s[j][1]) + (s[j][2] + s[j][3]);
      const size_t o = (size_t(b) * p.c + cg * 8 + j) * plane_s + pix;
      if (p.dx) {
        f4 v; v[0] = s[j][0]; v[1] = s[j][1]; v[2] = s[j][2]; v[3] = s[j][3];
        *reinterpret_cast<f4*>(p.dx + o) = v;
      }
      if (p.g16) {
        bf16x4 h;
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] = (__bf16)s[j][k];
        *reinterpret_cast<u32x2*>(p.g16 + o) = __builtin_bit_cast(u32x2, h);
      }
    }
    if (p.g8) {
      u32x4* op = p.g8 + (size_t(b) * cgs + cg) * plane_s + pix;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bf16x8 h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (__bf16)s[j][k];
        op[k] = __builtin_bit_cast(u32x4, h);
      }
    }
  }
  if (p.bias) {      // fixed order: lanes (shuffles), then the 4 waves
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = st::wave_sum(bsum[j]);
      if ((threadIdx.x & 63) == 0) red[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
      const int j = threadIdx.x;
      p.bias[(size_t(b) * gridDim.x + blockIdx.x) * p.c + cg * 8 + j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    }
  }
}

}  // namespace

namespace st {

// blocks along a plane (= bias slabs per image) for an [hs][ws] source plane
int fold8_blocks_per_plane(int hs, int ws, int* iters_out) {
  const int items = hs * (ws / 4);
  const int iters = std::max(1, cdiv(items, 256 * 32));
  if (iters_out) *iters_out = iters;
  return cdiv(items, 256 * iters);
}

// h, w: the conv-input dims (after the upsample); gate (either form) / outputs may be null.  bias: [n * bpp][c].
int launch_dgrad_fold8(const float* dxp, const float* gate_f32, const void* gate_g8, float* dx, void* g8, void* g16,
                       float* bias, int n, int c, int h, int w, int up, hipStream_t s) {
  const int hs = h >> up, ws = w >> up;
  ST_REQUIRE(c % 8 == 0 && ws % 4 == 0, "dgrad fold (8-channel form): c=%d, source width %d", c, ws);
  ST_REQUIRE(n <= 65535 && c / 8 <= 65535, "dgrad fold (8-channel form): n=%d, c=%d", n, c);
  Fold8Params p;
  p.dxp = dxp; p.gate_f32 = gate_f32; p.gate_g8 = (const u32x4*)gate_g8; p.dx = dx; p.g8 = (u32x4*)g8;
  p.g16 = (unsigned short*)g16; p.bias = bias; p.c = c; p.H = h; p.W = w; p.hs = hs; p.ws = ws;
  const int bpp = fold8_blocks_per_plane(hs, ws, &p.iters);
  dim3 grid(bpp, c / 8, n);
  if (up)
    hipLaunchKernelGGL(dgrad_fold8_kernel<true>, grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(dgrad_fold8_kernel<false>, grid, dim3(256), 0, s, p);
  return check_launch("st_conv3x3_dgrad(fold8)");
}

}  // namespace st
