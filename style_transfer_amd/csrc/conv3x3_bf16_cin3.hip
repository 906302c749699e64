// bf16-mode 3x3 convolution of a 3-channel image (the encoder's and the loss VGG's conv1_1: generator.py:47-49,
// trainer.py:165-199 -- torchvision vgg19.features[0]) for gfx950 (MI355X).
//
// The general bf16 kernel (conv3x3_bf16.hip, TAIL form) pads the 3 input channels to a 16-channel chunk: nine MFMAs
// of K = 16 per accumulator tile with 13 of 16 k empty, and a block per tile whose load / convert / MFMA / store
// phases simply add up (0.135 ms for 8 x 512 x 512 against 0.045 ms of output traffic).  Here the layer is a GEMM with
// K = 27 -> 32, k = channel * 9 + tap:
//   * two v_mfma_f32_32x32x16_bf16 per 32-channel x 32-column accumulator tile;
//   * the weight fragments (2 channel tiles x 2 k steps x 8 bf16) stay in registers for the whole block;
//   * the pixel operand of a lane -- 8 consecutive k = 8 (channel, tap) positions of one pixel -- is gathered from a bf16
//     halo image in LDS (ds_read_u16 x 8, packed in pairs) once per 32-column tile and feeds both channel tiles;
//   * persistent blocks of 4 waves, tile = 4 rows x 128 columns (wave = row); the next tile's halo is requested before
//     the current tile's MFMAs; one accumulator tile (16 registers) at a time, so many waves share a SIMD and one wave's
//     stores drain under the others' MFMAs.
// Rounding: x and w to bf16 (round to nearest even, as the TAIL form and oracle/ref_ops.conv_compute("bf16")), fp32
// accumulation, bias and ReLU in fp32; a G8 result is rounded to bf16 (the next layer's staging would do the same).
// The k order differs from the TAIL form's (tap-major there), so the two forms agree to fp32 accumulation order, not
// bit for bit; both public entry points that take a 3-channel fp32 image (st_conv3x3_fwd_bf16 and
// st_conv3x3_fwd_bf16_g8) route here, so the per-layer and the chained paths of one network still see the same bits.
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int TW = 128, TH = 4;                  // tile: 4 rows x 128 columns (wave = row, accumulator tile = 32 columns)
constexpr int HCS = TW + 4;                      // halo row stride in elements (TW + 2 used)
constexpr int PLANE = (TH + 2) * HCS;
constexpr int NH = 3 * (TH + 2) * (TW + 2), NHP = (NH + 255) / 256;

__device__ __forceinline__ unsigned short bf16_bits(float v) {
  return __builtin_bit_cast(unsigned short, (__bf16)v);
}

// halo offset of k = channel * 9 + tap (k >= 27: any valid position, its weight is zero)
__host__ __device__ constexpr int koff(int k) {
  return k < 27 ? (k / 9) * PLANE + ((k % 9) / 3) * HCS + (k % 9) % 3 : 0;
}

template <bool YG>
__global__ __launch_bounds__(256, 4) void conv3x3_bf16_cin3_kernel(
    const float* __restrict__ x, const unsigned short* __restrict__ wp, const float* __restrict__ bias,
    void* __restrict__ yout, int H, int W, int cout, int cout_pad, int pad_zero, int relu, int tiles_x, int tiles_y,
    int ntiles, int vec) {
  __shared__ unsigned short xs[3 * PLANE];
  __shared__ __attribute__((aligned(16))) float patch[YG ? 4 : 4 * 32 * 32];
  __shared__ float bsh[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const size_t plane = size_t(H) * W;
  if (tid < 64) bsh[tid] = (bias && tid < cout) ? bias[tid] : 0.f;

  int hpk[NHP];       // (channel << 16) | (row << 8) | column of this thread's halo elements
#pragma unroll
  for (int j = 0; j < NHP; ++j) {
    const int i = min(tid + j * 256, NH - 1);
    const int c = i / ((TH + 2) * (TW + 2)), rem = i - c * ((TH + 2) * (TW + 2));
    const int r = rem / (TW + 2);
    hpk[j] = (c << 16) | (r << 8) | (rem - r * (TW + 2));
  }
  float hv[NHP];
  auto load_halo = [&](int t) {
    const int tx = t % tiles_x;
    const int rest = t / tiles_x;
    const int ty = rest % tiles_y, b = rest / tiles_y;
    const float* xb = x + size_t(b) * 3 * plane;
#pragma unroll
    for (int j = 0; j < NHP; ++j) {
      const int gy = ty * TH + ((hpk[j] >> 8) & 255) - 1, gx = tx * TW + (hpk[j] & 255) - 1;
      const bool keep = (gy >= 0 && gy < H && gx >= 0 && gx < W) || !pad_zero;
      const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
      const float v = xb[(hpk[j] >> 16) * plane + size_t(cy) * W + cx];
      hv[j] = keep ? v : 0.f;
    }
  };
  int tile = blockIdx.x;
  load_halo(tile);

  // weight fragments: step j of lane (li, lh), channel tile mt = W[mt * 32 + li][k = j * 16 + lh * 8 + e], e = 0..7;
  // packed layout of st_conv3x3_pack_weights_bf16 for cin <= 16: element ((tap * 2 + 0) * cout_pad + o) * 8 + c
  bf16x8 wa[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      unsigned short e8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = j * 16 + lh * 8 + e;
        const int c = min(k / 9, 2), tap = k - (k / 9) * 9;
        const int o = mt * 32 + li;
        e8[e] = (k < 27 && o < cout) ? wp[(size_t((k < 27 ? tap : 0) * 2) * cout_pad + o) * 8 + c] : (unsigned short)0;
      }
      u32x4 pk;
#pragma unroll
      for (int q = 0; q < 4; ++q) pk[q] = unsigned(e8[2 * q]) | (unsigned(e8[2 * q + 1]) << 16);
      wa[mt][j] = __builtin_bit_cast(bf16x8, pk);
    }
  // halo offsets of this lane's 8 k per step (selected by lh)
  int bo[2][8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) bo[j][e] = lh ? koff(j * 16 + 8 + e) : koff(j * 16 + e);

  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const unsigned short* bp = xs + wave * HCS + li;
  for (; tile < ntiles; tile += gridDim.x) {
    const int tx = tile % tiles_x;
    const int rest = tile / tiles_x;
    const int ty = rest % tiles_y, b = rest / tiles_y;
    const int x0 = tx * TW, y0 = ty * TH;
    __syncthreads();                       // the previous tile's operand reads are done
#pragma unroll
    for (int j = 0; j < NHP; ++j)
      if (NH % 256 == 0 || tid + j * 256 < NH)
        xs[(hpk[j] >> 16) * PLANE + ((hpk[j] >> 8) & 255) * HCS + (hpk[j] & 255)] = bf16_bits(hv[j]);
    __syncthreads();
    if (tile + int(gridDim.x) < ntiles) load_halo(tile + int(gridDim.x));

    const int yw = y0 + wv;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      bf16x8 bv[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        u32x4 pk;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          pk[q] = unsigned(bp[bo[j][2 * q] + nt * 32]) | (unsigned(bp[bo[j][2 * q + 1] + nt * 32]) << 16);
        bv[j] = __builtin_bit_cast(bf16x8, pk);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[mt][0], bv[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[mt][1], bv[1], acc, 0, 0, 0);
        // accumulator register r of lane (li, lh) = channel mt * 32 + (r & 3) + 8 (r >> 2) + 4 lh, column li
        if constexpr (YG) {
          // G8 [n][cout/8][H][W][8]: registers 4g .. 4g + 3 are half (lh) of the 16-byte unit of pixel li, channel
          // group mt * 4 + g: 8-byte stores, a wave instruction writes 512 contiguous bytes
          const int xo = x0 + nt * 32 + li;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bf16x4 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float v = acc[4 * g + e] + bsh[mt * 32 + 8 * g + 4 * lh + e];
              if (relu) v = fmaxf(v, 0.f);
              h[e] = (__bf16)v;
            }
            const int cg = mt * 4 + g;
            if (cg * 8 < cout && yw < H && xo < W) {
              unsigned char* dst = reinterpret_cast<unsigned char*>(yout) +
                                   (((size_t(b) * (cout >> 3) + cg) * H + yw) * W + xo) * 16 + lh * 8;
              *reinterpret_cast<u32x2*>(dst) = __builtin_bit_cast(u32x2, h);
            }
          }
        } else {
          // fp32 NCHW through a per-wave 4 KB LDS patch: 16 bytes per lane, an instruction writes 8 channel rows
          float* pt = patch + wave * (32 * 32);
          float* y = reinterpret_cast<float*>(yout);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[r] + bsh[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
            if (relu) v = fmaxf(v, 0.f);
            pt[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = v;
          }
          // [scalar base of (image, channel group of 8, row, first column)] + [one per-lane byte offset]: no per-store
          // vector address arithmetic for the compiler to hoist into registers
          const unsigned st_off = unsigned((size_t(lane >> 3) * plane + (lane & 7) * 4) * sizeof(float));
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(pt + (q * 8 + (lane >> 3)) * 32 + (lane & 7) * 4);
            const int cg = mt * 32 + q * 8;                    // + lane / 8: the same side of cout (a multiple of 8)
            unsigned char* base = reinterpret_cast<unsigned char*>(y + (size_t(b) * cout + cg) * plane +
                                                                   size_t(yw) * W + x0 + nt * 32);
            if (cg < cout && yw < H) {
              const int xo = x0 + nt * 32 + (lane & 7) * 4;
              if (vec) {
                if (xo < W) *reinterpret_cast<f32x4*>(base + st_off) = v;
              } else {                                         // w % 4 != 0 or an unaligned y: element by element
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  if (xo + e < W) reinterpret_cast<float*>(base + st_off)[e] = v[e];
              }
            }
          }
        }
      }
    }
  }
}

}  // namespace

namespace st {

// true when this kernel takes the layer (then *rc is its launch status): x fp32 NCHW with 3 channels, cout <= 64 and a
// multiple of 8, no upsample / pool; y fp32 NCHW or G8 (both forms for EVERY such shape: the two stay bit-identical).  ST_BF16_CIN3_TAIL=1: the
// general kernel's TAIL form (A/B measurements).
bool conv3x3_bf16_cin3_fwd(const float* x, const void* w_packed, const float* bias, void* y, bool y_g8, int n, int cin,
                           int h, int w, int cout, int cout_pad, int pad_zero, int relu, hipStream_t s, int* rc) {
  if (cin != 3 || cout > 64 || cout % 8 != 0 || getenv("ST_BF16_CIN3_TAIL") != nullptr) return false;
  if (y_g8 && reinterpret_cast<uintptr_t>(y) % 16 != 0) return false;
  const int vec = (w % 4 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0) ? 1 : 0;      // fp32 form: 16-byte stores
  const int tiles_x = cdiv(w, TW), tiles_y = cdiv(h, TH);
  const long long ntiles = (long long)n * tiles_x * tiles_y;
  if (ntiles > 0x7fffffffLL) return false;
  const dim3 grid((unsigned)std::min<long long>(ntiles, 1024));
  if (y_g8)
    hipLaunchKernelGGL(conv3x3_bf16_cin3_kernel<true>, grid, dim3(256), 0, s, x, (const unsigned short*)w_packed, bias, y,
                       h, w, cout, cout_pad, pad_zero, relu, tiles_x, tiles_y, (int)ntiles, vec);
  else
    hipLaunchKernelGGL(conv3x3_bf16_cin3_kernel<false>, grid, dim3(256), 0, s, x, (const unsigned short*)w_packed, bias, y,
                       h, w, cout, cout_pad, pad_zero, relu, tiles_x, tiles_y, (int)ntiles, vec);
  *rc = check_launch("st_conv3x3_fwd_bf16(cin=3)");
  return true;
}

}  // namespace st
