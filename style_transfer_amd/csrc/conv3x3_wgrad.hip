// Weight (and bias) gradient of the fused [nearest x2 upsample] -> pad 1 -> Conv2d(3x3) -> [ReLU]
// layer as an fp32 MFMA GEMM that reduces over pixels (decoder training: generator.py:85-130
// trained by trainer.py:286-291, 471-473).
//
//   dW[o][c][tap] = sum_{n,y,x} G[n][o][y][x] * Xpad[n][c][y+ky-1][x+kx-1],   G = dy * (y > 0)
//
// GEMM view: D[o][j] = sum_pix G[o][pix] * Xp[pix][j],  j = c*9 + tap  (== the OIHW order, so an
// accumulator register row is a contiguous run of dW).
//   * v_mfma_f32_32x32x2_f32, M = output channels, N = (c, tap) columns, K = pixels.
//   * A block (12 waves) owns 128 output channels x 32 input channels (288 = 9 x 32 columns) and
//     walks a contiguous range of 64-pixel tiles (2 rows x 32); wave w owns channel tile w%4 and
//     column tiles 3*(w/4)..+2: 3 accumulator tiles (48 registers), 1 A + 3 B reads per 3 MFMAs.
//     The small register footprint lets two blocks (6 waves/SIMD) share a CU, so one block's
//     staging hides under the other's MFMA phase.
//   * LDS: gated dy tile [128][64 + 1] (odd stride: the A read walks channels), input halo
//     [32][4 x 35 (+pad)] with row stride 35 and plane stride == 9 (mod 32), which puts the 32
//     columns j of a B read on 32 distinct banks.
//   * Split-K over pixel tiles: every block writes its partial slab; a second kernel sums the
//     slabs in a fixed order (deterministic, no float atomics).  The bias gradient is the row sum
//     of G, accumulated while staging.
#include <algorithm>
#include <cstdint>

#include <cstdlib>

#include "common.h"

namespace {

constexpr int NTHREADS = 768;      // 12 waves
constexpr int BO = 128;           // output channels per block
constexpr int CI = 32;            // input channels per block -> 288 columns = 9 MFMA tiles
constexpr int NJT = 9;
constexpr int WJT = 3;            // column tiles per wave
constexpr int TR = 2;             // pixel-tile rows
constexpr int TW = 32;            // pixel-tile width
constexpr int TP = TR * TW;       // 64 pixels = 32 MFMA k-steps
constexpr int GS = TP + 1;        // odd row stride of the dy tile
constexpr int HC = 35;            // halo row stride (== 3 mod 32)
constexpr int HR = TR + 2;
constexpr int PLANE = 169;        // >= HR*HC = 140 and == 9 (mod 32)
constexpr int G_ELEMS = BO * GS;
constexpr int X_ELEMS = CI * PLANE;
constexpr int NXP = (CI * HR * (TW + 2) + NTHREADS - 1) / NTHREADS;   // halo loads per thread
constexpr int NG4 = BO * TR * (TW / 4);                                // float4 of one dy tile
constexpr int NGP = (NG4 + NTHREADS - 1) / NTHREADS;
constexpr size_t LDS_BYTES = size_t(G_ELEMS + X_ELEMS) * sizeof(float);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WgradParams {
  const float* dy;      // [n][cout][H][W]
  const float* ygate;   // nullable, same shape: dy counts only where ygate > 0
  const float* x;       // [n][cin][hs][ws]
  float* slabs;         // [nsplit][cout][cin*9]
  float* bias_slabs;    // [nsplit][cout] (nullable)
  int n, cout, cin, hs, ws, H, W;
  int up, pad_zero;
  int tiles_x, tiles_y, ntiles;   // pixel tiles per image / total
  int tiles_per_split, nsplit;
  // staged bf16 path (conv3x3_wgrad_bf16s_kernel): operands prepared once per call by the wgrad_prep_* kernels
  const unsigned short* g16;   // [n][cout][H][W] bf16, gate applied
  const unsigned short* x16;   // [n][cin][H + 2][wp] bf16: pad (and x2 upsample) materialised, pixel gx at column gx + 8
  int wp;                      // row pitch of x16 in elements (multiple of 8)
  // XCD-aware block order (1-D grid of gx * gy * roundup(nsplit, 8) blocks): workgroups are dealt round-robin to the
  // 8 XCDs, each with its own L2.  With the plain (cin chunk, cout tile, split) grid the gx blocks that read the SAME
  // dy tile sit on gx different XCDs and every L2 fetches it again; here all gx * gy blocks of one pixel range share
  // an XCD, so dy and x leave HBM once.
  int gx = 0, gy = 0;          // 0: plain 3-D grid
};

// BO_ output channels x CI_ input channels per block; BO_ / 32 channel tiles x (CI_ * 9 / 32) column tiles are
// dealt to the 12 waves as (1 channel tile) x (3 column tiles) each: <128, 32> (cout > 64), <64, 64> (cout <= 64:
// no output-channel rows wasted on padding; narrower M would have to widen N beyond the layer's cin).
template <int BO_, int CI_>
struct WShape {
  static constexpr int OT = BO_ / 32;                  // channel tiles
  static constexpr int NJT_ = CI_ * 9 / 32;            // column tiles
  static constexpr int GROUPS = 12 / OT;
  static_assert(BO_ % 32 == 0 && OT * GROUPS == 12 && NJT_ == GROUPS * WJT, "12 waves x 3 column tiles");
  static constexpr int G_ELEMS_ = BO_ * GS;
  static constexpr int X_ELEMS_ = CI_ * PLANE;
  static constexpr int NXE = CI_ * HR * (TW + 2);      // halo elements
  static constexpr int NXP_ = (NXE + NTHREADS - 1) / NTHREADS;
  static constexpr int NG4_ = BO_ * TR * (TW / 4);     // float4 of one dy tile
  static constexpr int NGP_ = (NG4_ + NTHREADS - 1) / NTHREADS;
  static constexpr size_t LDS = size_t(G_ELEMS_ + X_ELEMS_) * sizeof(float);
};

template <bool GATE, int BO_, int CI_>
__global__ __launch_bounds__(NTHREADS) void conv3x3_wgrad_kernel(WgradParams p) {
  using S = WShape<BO_, CI_>;
  constexpr int NXP_ = S::NXP_, NGP_ = S::NGP_;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Gs = smem;
  float* Xs = smem + S::G_ELEMS_;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int c0 = blockIdx.x * CI_;
  const int o0 = blockIdx.y * BO_;
  const int split = blockIdx.z;
  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.ntiles);

  // ---- staging maps ----------------------------------------------------------------------
  // dy tile = 256 row segments (128 channels x 2 tile rows) of 32 pixels = 8 float4.  Eight
  // consecutive lanes cover one 128-B segment, so a wave instruction reads 8 whole lines.
  // float4 index g = tid + i*768 -> segment r = g/8 (channel r>>1, tile row r&1), float4 q = g%8.
  const int gq = tid & 7, gr0 = tid >> 3;
  const bool vec4 = (p.W & 3) == 0;   // rows 16-B aligned (the entry point checks the bases)
  // halo: element e = tid + i*768 of a [CI_][HR][TW+2] box; (channel, row, col) are recomputed from e with
  // constant divisors where they are needed (a per-thread table costs NXP_ more registers than the 168 allow)
#define ST_WG_XE(i_)                                                         \
  const int e = min(tid + (i_) * NTHREADS, S::NXE - 1);                      \
  const int c = e / (HR * (TW + 2));                                         \
  const int rem_ = e - c * (HR * (TW + 2));                                  \
  const int row = rem_ / (TW + 2), col = rem_ - row * (TW + 2);
  const size_t plane_src = size_t(p.hs) * p.ws;
  const size_t plane_out = size_t(p.H) * p.W;

  f32x16 acc[WJT];
#pragma unroll
  for (int t = 0; t < WJT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bias_acc[NGP_];
#pragma unroll
  for (int i = 0; i < NGP_; ++i) bias_acc[i] = 0.f;

  // per-lane operand bases.  A: Gs[(ot*32 + li)][pix]; B: column j = jt*32 + li -> (c, tap)
  const int ot = wave % S::OT, jt0 = (wave / S::OT) * WJT;
  const float* aPtr = Gs + (ot * 32 + li) * GS + lh;
  int bOff[WJT];
#pragma unroll
  for (int jt = 0; jt < WJT; ++jt) {
    const int j = (jt0 + jt) * 32 + li;
    const int c = j / 9, tap = j - c * 9;
    bOff[jt] = c * PLANE + (tap / 3) * HC + (tap % 3) + lh;
  }

  for (int t = t_begin; t < t_end; ++t) {
    int rem = t;
    const int tx = rem % p.tiles_x;
    rem /= p.tiles_x;
    const int ty = rem % p.tiles_y;
    const int b = rem / p.tiles_y;
    const int x0 = tx * TW, y0 = ty * TR;

    // -- global loads (unconditional, clamped addresses) --
    f32x4 gv[NGP_];
#pragma unroll
    for (int i = 0; i < NGP_; ++i) {
      const int r = min(gr0 + (NTHREADS / 8) * i, BO_ * TR - 1);   // tail threads re-load the last segment
      const int o = min(o0 + (r >> 1), p.cout - 1);
      const int cy = min(y0 + (r & 1), p.H - 1);
      const size_t base = (size_t(b) * p.cout + o) * plane_out + size_t(cy) * p.W;
      if (vec4) {
        const int cx = min(x0 + gq * 4, p.W - 4);
        gv[i] = *reinterpret_cast<const f32x4*>(p.dy + base + cx);
        if (GATE) {
          const f32x4 yv = *reinterpret_cast<const f32x4*>(p.ygate + base + cx);
#pragma unroll
          for (int k = 0; k < 4; ++k) gv[i][k] = yv[k] > 0.f ? gv[i][k] : 0.f;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int cx = min(x0 + gq * 4 + k, p.W - 1);
          gv[i][k] = p.dy[base + cx];
          if (GATE) gv[i][k] = p.ygate[base + cx] > 0.f ? gv[i][k] : 0.f;
        }
      }
    }
    float xv[NXP_];
#pragma unroll
    for (int i = 0; i < NXP_; ++i) {
      ST_WG_XE(i)
      int cy = min(max(y0 + row - 1, 0), p.H - 1), cx = min(max(x0 + col - 1, 0), p.W - 1);
      if (p.up) { cy >>= 1; cx >>= 1; }
      xv[i] = p.x[(size_t(b) * p.cin + min(c0 + c, p.cin - 1)) * plane_src + size_t(cy) * p.ws + cx];
    }

    __syncthreads();   // previous tile's MFMA phase is done with LDS
#pragma unroll
    for (int i = 0; i < NGP_; ++i) {
      const int r = gr0 + (NTHREADS / 8) * i;
      if (S::NG4_ % NTHREADS != 0 && r >= BO_ * TR) break;
      const bool row_ok = (o0 + (r >> 1)) < p.cout && (y0 + (r & 1)) < p.H;
      float* dst = Gs + (r >> 1) * GS + (r & 1) * TW + gq * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float g = (row_ok && (x0 + gq * 4 + k) < p.W) ? gv[i][k] : 0.f;
        dst[k] = g;
        bias_acc[i] += g;
      }
    }
#pragma unroll
    for (int i = 0; i < NXP_; ++i) {
      ST_WG_XE(i)
      const int gy = y0 + row - 1, gx = x0 + col - 1;
      const bool inb = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
      const unsigned m = ((inb || !p.pad_zero) && (c0 + c) < p.cin) ? 0xffffffffu : 0u;
      if (tid + i * NTHREADS < S::NXE) Xs[c * PLANE + row * HC + col] = __uint_as_float(__float_as_uint(xv[i]) & m);
    }
    __syncthreads();

    // -- 32 k-steps of 2 pixels: pixel 2s+lh = (row, col) of the tile --
#pragma unroll
    for (int s = 0; s < TP / 2; ++s) {
      const int row = (2 * s) / TW, col = (2 * s) % TW;
      const float a = aPtr[2 * s];
      const int poff = row * HC + col;
#pragma unroll
      for (int jt = 0; jt < WJT; ++jt) {
        const float bv = Xs[bOff[jt] + poff];
        acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[jt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: partial slab, register row r = one output channel, 32 contiguous columns --
  float* slab = p.slabs + size_t(split) * p.cout * p.cin * 9;
#pragma unroll
  for (int jt = 0; jt < WJT; ++jt) {
    const int jg = c0 * 9 + (jt0 + jt) * 32 + li;     // column in the [cin*9] row
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (o < p.cout && jg < p.cin * 9) slab[size_t(o) * p.cin * 9 + jg] = acc[jt][r];
    }
  }
  if (p.bias_slabs && blockIdx.x == 0) {
    // a channel's 2 segments x 8 float4 sit in 16 consecutive lanes (tid/8 = 2*ch + row)
#pragma unroll
    for (int i = 0; i < NGP_; ++i) {
      float v = bias_acc[i];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      const int r = gr0 + (NTHREADS / 8) * i;
      const int o = o0 + (r >> 1);
      if ((tid & 15) == 0 && r < BO_ * TR && o < p.cout) p.bias_slabs[size_t(split) * p.cout + o] = v;
    }
  }
}

#undef ST_WG_XE

// out[e] = sum_s slabs[s][e], fixed order.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs,
                                                         float* __restrict__ out, long long n,
                                                         int nsplit) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n;
       e += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += slabs[(size_t)k * n + e];
    out[e] = s;
  }
}

// Many short slabs (the bias partial sums of the G8 decoder chain: hundreds of [cout] rows): one wave per output
// element, lanes stride over the slabs, fixed-order shuffle tree -- slab_reduce_kernel would walk them serially.
__global__ __launch_bounds__(64) void slab_reduce_wave_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                             int n, int nslabs) {
  const int e = blockIdx.x;
  float s = 0.f;
  for (int k = threadIdx.x; k < nslabs; k += 64) s += slabs[(size_t)k * n + e];
  s = st::wave_sum(s);
  if (threadIdx.x == 0) out[e] = s;
}

// Thousands of rows (the fused chain epilogue writes one per wave and pixel tile): first stage of a two-stage sum.
// A block sums `rows_per_seg` consecutive rows for 16 consecutive elements -- thread = (element, one of 16 row lanes),
// so a row contributes one full 64-B sector -- and writes one row of partial[seg][n]; the wave kernel sums those.
__global__ __launch_bounds__(256) void slab_reduce_rows_kernel(const float* __restrict__ slabs,
                                                              float* __restrict__ partial, int n, int nslabs,
                                                              int rows_per_seg) {
  __shared__ float red[16][17];
  const int e = blockIdx.x * 16 + (threadIdx.x & 15), rl = threadIdx.x >> 4, seg = blockIdx.y;
  const int k0 = seg * rows_per_seg, k1 = min(k0 + rows_per_seg, nslabs);
  float s = 0.f;
  if (e < n)
    for (int k = k0 + rl; k < k1; k += 16) s += slabs[(size_t)k * n + e];
  red[rl][threadIdx.x & 15] = s;
  __syncthreads();
  if (threadIdx.x < 16 && e < n) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][threadIdx.x];
    partial[(size_t)seg * n + e] = t;
  }
}

// ================================================================================================
// bf16 compute mode (BASELINE configs[3]): the same GEMM with bf16 operands and fp32 accumulation,
// v_mfma_f32_32x32x16_bf16, K step = 16 consecutive pixels of one tile row.
//   * A lane supplies 8 consecutive K values, so both LDS images are pixel-innermost bf16:
//       G  [128 o][4 rows x 32 px]                  row stride 272 B
//       Xk [3 kx][32 c][6 halo rows][32 px]         one copy per horizontal tap shift, so that the 8
//                                                   pixels of every (c, ky, kx) column start 16-B aligned
//     strides (rows 80 B, channels 496 B, copies 15984 B) found by exhaustive search: the 16 lanes of
//     every ds_read_b128 group hit 16 distinct 4-bank sets for all nine column tiles.
//   * 12 waves as in the fp32 kernel (wave = 1 channel tile x 3 column tiles), 128-pixel tiles
//     (8 K steps), next tile's global loads in flight during the MFMA phase.
//   * fp32 slabs / fixed-order reduction / bias gradient exactly as above.
namespace bf {

constexpr int TRB = 4;                      // pixel-tile rows
constexpr int TPB = TRB * TW;               // 128 pixels = 8 K steps
constexpr int HRB = TRB + 2;                // halo rows
constexpr int G_ROW_B = TPB * 2 + 16;       // 272 bytes per channel row of G
constexpr int X_ROW_B = 80, X_CH_B = 496, X_COPY_B = 15984;
constexpr int G_BYTES = BO * G_ROW_B;       // 34816
constexpr size_t LDS_BYTES_B = size_t(G_BYTES) + 3 * X_COPY_B;   // 82768
constexpr int NG4B = BO * TRB * (TW / 4);   // float4 of one dy tile = 4096
constexpr int NGPB = (NG4B + NTHREADS - 1) / NTHREADS;           // 6
static_assert(CI * HRB * 4 == NTHREADS, "one halo staging item per thread");

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <bool GATE>
__global__ __launch_bounds__(NTHREADS) void conv3x3_wgrad_bf16_kernel(WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smemb[];
  unsigned char* Gs = smemb;
  unsigned char* Xs = smemb + G_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int c0 = blockIdx.x * CI;
  const int o0 = blockIdx.y * BO;
  const int split = blockIdx.z;
  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.ntiles);

  // halo staging item of this thread: channel xc, halo row xr, 8-pixel group xq
  const int xq = tid & 3, xr = (tid >> 2) % HRB, xc = tid / (4 * HRB);
  const bool vec4 = (p.W & 3) == 0;
  const size_t plane_src = size_t(p.hs) * p.ws;
  const size_t plane_out = size_t(p.H) * p.W;

  f32x16 acc[WJT];
#pragma unroll
  for (int t = 0; t < WJT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bias_acc[NGPB];
#pragma unroll
  for (int i = 0; i < NGPB; ++i) bias_acc[i] = 0.f;

  const int ot = wave & 3, jt0 = (wave >> 2) * WJT;
  const unsigned char* aPtr = Gs + (ot * 32 + li) * G_ROW_B + lh * 16;
  int bOff[WJT];
#pragma unroll
  for (int jt = 0; jt < WJT; ++jt) {
    const int j = (jt0 + jt) * 32 + li;
    const int c = j / 9, tap = j - c * 9;
    bOff[jt] = (tap % 3) * X_COPY_B + c * X_CH_B + (tap / 3) * X_ROW_B + lh * 16;
  }

  f32x4 gv[NGPB];
  float xv[10];

#define ST_WG_LOAD(t_)                                                                          \
  {                                                                                             \
    int rem = (t_);                                                                             \
    const int tx = rem % p.tiles_x;                                                             \
    rem /= p.tiles_x;                                                                           \
    const int ty = rem % p.tiles_y;                                                             \
    const int b = rem / p.tiles_y;                                                              \
    const int x0 = tx * TW, y0 = ty * TRB;                                                      \
    _Pragma("unroll") for (int i = 0; i < NGPB; ++i) {                                          \
      const int g = min(tid + i * NTHREADS, NG4B - 1);                                          \
      const int seg = g >> 3, q4 = g & 7;                                                       \
      const int o = min(o0 + (seg >> 2), p.cout - 1);                                           \
      const int cy = min(y0 + (seg & 3), p.H - 1);                                              \
      const size_t base = (size_t(b) * p.cout + o) * plane_out + size_t(cy) * p.W;              \
      if (vec4) {                                                                               \
        const int cx = min(x0 + q4 * 4, p.W - 4);                                               \
        gv[i] = *reinterpret_cast<const f32x4*>(p.dy + base + cx);                              \
        if (GATE) {                                                                             \
          const f32x4 yv = *reinterpret_cast<const f32x4*>(p.ygate + base + cx);                \
          _Pragma("unroll") for (int k = 0; k < 4; ++k) gv[i][k] = yv[k] > 0.f ? gv[i][k] : 0.f; \
        }                                                                                       \
      } else {                                                                                  \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                         \
          const int cx = min(x0 + q4 * 4 + k, p.W - 1);                                         \
          gv[i][k] = p.dy[base + cx];                                                           \
          if (GATE) gv[i][k] = p.ygate[base + cx] > 0.f ? gv[i][k] : 0.f;                       \
        }                                                                                       \
      }                                                                                         \
    }                                                                                           \
    {                                                                                           \
      const int gy = y0 + xr - 1;                                                               \
      int cy = min(max(gy, 0), p.H - 1);                                                        \
      if (p.up) cy >>= 1;                                                                       \
      const float* xrow = p.x + (size_t(b) * p.cin + min(c0 + xc, p.cin - 1)) * plane_src +     \
                          size_t(cy) * p.ws;                                                    \
      _Pragma("unroll") for (int k = 0; k < 10; ++k) {                                          \
        int cx = min(max(x0 + xq * 8 + k - 1, 0), p.W - 1);                                     \
        if (p.up) cx >>= 1;                                                                     \
        xv[k] = xrow[cx];                                                                       \
      }                                                                                         \
    }                                                                                           \
  }

#define ST_WG_STORE(t_)                                                                         \
  {                                                                                             \
    int rem = (t_);                                                                             \
    const int tx = rem % p.tiles_x;                                                             \
    rem /= p.tiles_x;                                                                           \
    const int ty = rem % p.tiles_y;                                                             \
    const int x0 = tx * TW, y0 = ty * TRB;                                                      \
    _Pragma("unroll") for (int i = 0; i < NGPB; ++i) {                                          \
      const int g = tid + i * NTHREADS;                                                         \
      if (NG4B % NTHREADS != 0 && g >= NG4B) break;                                             \
      const int seg = g >> 3, q4 = g & 7;                                                       \
      const bool row_ok = (o0 + (seg >> 2)) < p.cout && (y0 + (seg & 3)) < p.H;                 \
      bf16x4 pk;                                                                                \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                           \
        const float gg = (row_ok && (x0 + q4 * 4 + k) < p.W) ? gv[i][k] : 0.f;                  \
        pk[k] = (__bf16)gg;                                                                     \
        bias_acc[i] += gg;                                                                      \
      }                                                                                         \
      *reinterpret_cast<u32x2*>(Gs + (seg >> 2) * G_ROW_B + ((seg & 3) * TW + q4 * 4) * 2) =    \
          __builtin_bit_cast(u32x2, pk);                                                        \
    }                                                                                           \
    {                                                                                           \
      const int gy = y0 + xr - 1;                                                               \
      const bool rowin = gy >= 0 && gy < p.H;                                                   \
      const bool chan_ok = (c0 + xc) < p.cin;                                                   \
      float xm[10];                                                                             \
      _Pragma("unroll") for (int k = 0; k < 10; ++k) {                                          \
        const int gx = x0 + xq * 8 + k - 1;                                                     \
        const bool inb = rowin && gx >= 0 && gx < p.W;                                          \
        const unsigned m = ((inb || !p.pad_zero) && chan_ok) ? 0xffffffffu : 0u;                \
        xm[k] = __uint_as_float(__float_as_uint(xv[k]) & m);                                    \
      }                                                                                         \
      _Pragma("unroll") for (int kx = 0; kx < 3; ++kx) {                                        \
        bf16x8 pk;                                                                              \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) pk[k] = (__bf16)xm[k + kx];               \
        *reinterpret_cast<u32x4*>(Xs + kx * X_COPY_B + xc * X_CH_B + xr * X_ROW_B + xq * 16) =  \
            __builtin_bit_cast(u32x4, pk);                                                      \
      }                                                                                         \
    }                                                                                           \
  }

  if (t_begin < t_end) ST_WG_LOAD(t_begin);
  for (int t = t_begin; t < t_end; ++t) {
    __syncthreads();   // previous tile's MFMA phase is done with LDS
    ST_WG_STORE(t);
    __syncthreads();
    if (t + 1 < t_end) ST_WG_LOAD(t + 1);   // in flight during the MFMA phase
#pragma unroll
    for (int s = 0; s < TPB / 16; ++s) {
      // K step s = pixels s*16 .. s*16+15 of the tile: row s/2, columns (s%2)*16 ..
      const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(aPtr + s * 32));
      const int poff = (s >> 1) * X_ROW_B + (s & 1) * 32;
#pragma unroll
      for (int jt = 0; jt < WJT; ++jt) {
        const bf16x8 bv = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Xs + bOff[jt] + poff));
        acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv, acc[jt], 0, 0, 0);
      }
    }
  }
#undef ST_WG_LOAD
#undef ST_WG_STORE

  // ---- epilogue: partial slab, register row r = one output channel, 32 contiguous columns --
  float* slab = p.slabs + size_t(split) * p.cout * p.cin * 9;
#pragma unroll
  for (int jt = 0; jt < WJT; ++jt) {
    const int jg = c0 * 9 + (jt0 + jt) * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (o < p.cout && jg < p.cin * 9)
        slab[size_t(o) * p.cin * 9 + jg] = acc[jt][r];
    }
  }
  if (p.bias_slabs && blockIdx.x == 0) {
    // a channel's 4 rows x 8 float4 sit in 32 consecutive lanes (g / 32 = channel)
#pragma unroll
    for (int i = 0; i < NGPB; ++i) {
      float v = bias_acc[i];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      const int g = tid + i * NTHREADS;
      const int o = o0 + (g >> 5);
      if ((tid & 31) == 0 && g < NG4B && o < p.cout) p.bias_slabs[size_t(split) * p.cout + o] = v;
    }
  }
}

// ---- staged form: both operands prepared ONCE per call ------------------------------------------------------
// Measured on the decoder layers at B = 8 (tools/bench_conv_layers.py --mode wgrad --compute bf16): the kernel above
// takes 0.52 ms for 256 -> 256 at 128 x 128 with or WITHOUT its MFMA loop, 0.18 ms without its global loads, 0.13 ms
// without any staging: it is bound by operand traffic.  Every block re-reads the fp32 dy tile and its fp32 gate for
// each of the cin / 32 channel chunks (and x for each of the cout / 128): 2.5 GB per call for 154 GFLOP.  Here two
// elementwise passes write what the GEMM actually consumes --
//   G  = bf16(dy * (ygate > 0))                [n][cout][H][W]           (+ the bias gradient: row sums, in fp32)
//   Xp = bf16(pad1(upsample?(x)))              [n][cin][H + 2][W + 16]   pixel gx at column gx + 8
// -- and the GEMM stages plain 16-B vectors: no gate, no clamps, no conversion, a quarter of the bytes for dy and
// half for x (0.67 GB per call for that layer).  The three horizontally shifted copies of an x row segment come
// from one aligned 16-B load + its two neighbour dwords by v_alignbit.  Needs W % 32 == 0 and H % 4 == 0 (whole
// tiles; every decoder layer), otherwise the kernel above runs.
constexpr int XPAD = 8;             // left / right padding columns of an Xp row (keeps 16-B alignment)

template <bool GATE>
__global__ __launch_bounds__(256) void wgrad_prep_g_kernel(const float* __restrict__ dy,
                                                          const float* __restrict__ ygate,
                                                          unsigned short* __restrict__ g16,
                                                          float* __restrict__ bias_part, int hw) {
  __shared__ float red[4];
  const size_t base = size_t(blockIdx.x) * hw;          // one (image, channel) plane per block; hw % 8 == 0
  float sum = 0.f;
  for (int i = threadIdx.x * 8; i < hw; i += 256 * 8) {
    f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dy + base + i));
    f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(dy + base + i + 4));
    if (GATE) {
      const f32x4 ya = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ygate + base + i));
      const f32x4 yb = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ygate + base + i + 4));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a[k] = ya[k] > 0.f ? a[k] : 0.f;
        b[k] = yb[k] > 0.f ? b[k] : 0.f;
      }
    }
    bf16x8 pk;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pk[k] = (__bf16)a[k];
      pk[4 + k] = (__bf16)b[k];
      sum += a[k] + b[k];
    }
    *reinterpret_cast<u32x4*>(g16 + base + i) = __builtin_bit_cast(u32x4, pk);
  }
  sum = st::block_sum<256>(sum, red);
  if (bias_part && threadIdx.x == 0) bias_part[blockIdx.x] = sum;     // [image][channel]: a slab per image
}

// One item = 8 consecutive columns (16 B) of one padded row; interior items read two aligned float4 (one, doubled,
// under the x2 upsample), the two edge items of a row hold one halo pixel each.
__global__ __launch_bounds__(256) void wgrad_prep_x_kernel(const float* __restrict__ x,
                                                          unsigned short* __restrict__ x16, long long planes,
                                                          int hs, int ws, int H, int W, int wp, int up,
                                                          int pad_zero) {
  const int groups = wp / 8;
  const long long per_plane = (long long)(H + 2) * groups, total = planes * per_plane;
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += gridDim.x * 256LL) {
    const long long plane = idx / per_plane;
    const int rem = int(idx - plane * per_plane);
    const int py = rem / groups, g = rem - py * groups;
    const float* xp = x + plane * size_t(hs) * ws;
    const int gy = py - 1;
    const bool rowin = gy >= 0 && gy < H;
    int cy = min(max(gy, 0), H - 1);
    if (up) cy >>= 1;
    const float* row = xp + size_t(cy) * ws;
    const bool keep = rowin || !pad_zero;
    bf16x8 pk;
    if (g >= 1 && g < groups - 1) {
      const int gx0 = (g - 1) * 8;                                  // 0 .. W - 8, a multiple of 8
      float v[8];
      if (up) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(row + (gx0 >> 1));
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = a[k >> 1];
      } else {
        const f32x4 a = *reinterpret_cast<const f32x4*>(row + gx0);
        const f32x4 b = *reinterpret_cast<const f32x4*>(row + gx0 + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = a[k]; v[4 + k] = b[k]; }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) pk[k] = (__bf16)(keep ? v[k] : 0.f);
    } else {
      // g == 0: column 7 is pixel -1; g == groups - 1: column 0 is pixel W (replicated edge, or zero)
      const float e = (keep && !pad_zero) ? row[g == 0 ? 0 : ((W - 1) >> up)] : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) pk[k] = (__bf16)0.f;
      if (g == 0) pk[7] = (__bf16)e; else pk[0] = (__bf16)e;
    }
    *reinterpret_cast<u32x4*>(x16 + plane * size_t(H + 2) * wp + size_t(py) * wp + g * 8) = __builtin_bit_cast(u32x4, pk);
  }
}

// The same image from a G8 activation (bf16 [n][c/8][hs][ws][8], what the layers of a bf16-mode chain hand each
// other): one item = 8 channels x 8 consecutive columns of one padded row -- 8 (4 under the upsample) 16-B pixel
// reads, an 8 x 8 transpose in registers, one 16-B store per channel.  No rounding: the values already are bf16 (the
// round-to-nearest-even wgrad_prep_x_kernel would apply to the fp32 form of the same activation).
__global__ __launch_bounds__(256) void wgrad_prep_x_g8_kernel(const u32x4* __restrict__ xg,
                                                             unsigned short* __restrict__ x16, long long groups_total,
                                                             int cgs, int hs, int ws, int H, int W, int wp, int up,
                                                             int pad_zero) {
  const int groups = wp / 8;
  const long long per_cg = (long long)(H + 2) * groups, total = groups_total * per_cg;
  for (long long idx = blockIdx.x * 256LL + threadIdx.x; idx < total; idx += gridDim.x * 256LL) {
    const long long bcg = idx / per_cg;                  // image * cgs + channel group
    const int rem = int(idx - bcg * per_cg);
    const int py = rem / groups, g = rem - py * groups;
    const int gy = py - 1;
    const bool rowin = gy >= 0 && gy < H;
    int cy = min(max(gy, 0), H - 1);
    if (up) cy >>= 1;
    const u32x4* row = xg + (bcg * hs + cy) * (long long)ws;
    const bool keep = rowin || !pad_zero;
    u32x4 px[8];                                         // px[k] = the 8 channels of column k of this item
    const u32x4 zero = {0u, 0u, 0u, 0u};
    if (g >= 1 && g < groups - 1) {
      const int gx0 = (g - 1) * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) px[k] = keep ? row[up ? ((gx0 + k) >> 1) : (gx0 + k)] : zero;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) px[k] = zero;
      const u32x4 e = (keep && !pad_zero) ? row[g == 0 ? 0 : ((W - 1) >> up)] : zero;
      if (g == 0) px[7] = e; else px[0] = e;
    }
    const long long cg = bcg % cgs, b = bcg / cgs;
#pragma unroll
    for (int j = 0; j < 8; ++j) {                        // channel j of the group: one halfword of every pixel
      u32x4 o;
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) {
        const unsigned lo = (j & 1) ? (px[2 * k2][j >> 1] >> 16) : (px[2 * k2][j >> 1] & 0xffffu);
        const unsigned hi = (j & 1) ? (px[2 * k2 + 1][j >> 1] >> 16) : (px[2 * k2 + 1][j >> 1] & 0xffffu);
        o[k2] = lo | (hi << 16);
      }
      const long long plane = (b * cgs + cg) * 8 + j;
      *reinterpret_cast<u32x4*>(x16 + plane * size_t(H + 2) * wp + size_t(py) * wp + g * 8) = o;
    }
  }
}

// Block shape <BO_, CI_>: 12 waves = (BO_ / 32 channel tiles) x (CI_ * 9 / 32 / 3 groups of three column tiles):
// <128, 32> for cout > 64, <64, 64> for cout <= 64 (no MFMA rows spent on padding; the decoder's 128 -> 64 and
// 64 -> 64 layers).  The copy stride of the wider X image keeps the residue (mod 256 B) that makes every
// ds_read_b128 group conflict-free (the channel stride repeats its residues every 16 channels).
template <int BO_, int CI_>
struct SShape {
  static constexpr int OT = BO_ / 32, GROUPS = 12 / OT;
  static_assert(OT * GROUPS == 12 && CI_ * 9 / 32 == GROUPS * WJT && (CI_ * 9) % 32 == 0, "12 waves x 3 column tiles");
  static constexpr int G_BYTES_ = BO_ * G_ROW_B;
  static constexpr int XCOPY = CI_ == 32 ? X_COPY_B : 31856;
  static_assert(XCOPY >= CI_ * X_CH_B && XCOPY % 256 == X_COPY_B % 256, "X copy stride");
  static constexpr int NG = BO_ * TRB * 4, NGS_ = (NG + NTHREADS - 1) / NTHREADS;        // 16-B items of a G tile
  static constexpr int NX = CI_ * HRB * 4, NXS_ = (NX + NTHREADS - 1) / NTHREADS;        // ... of the X halo
  static constexpr size_t LDS = size_t(G_BYTES_) + 3 * size_t(XCOPY);
};

template <int BO_, int CI_>
__global__ __launch_bounds__(NTHREADS) void conv3x3_wgrad_bf16s_kernel(WgradParams p) {
  typedef SShape<BO_, CI_> S;
  extern __shared__ __attribute__((aligned(16))) unsigned char smemb[];
  unsigned char* Gs = smemb;
  unsigned char* Xs = smemb + S::G_BYTES_;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  int bx = blockIdx.x, by = blockIdx.y, split = blockIdx.z;
  if (p.gx > 0) {
    const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3, gs = p.gx * p.gy;
    const int g = k % gs;
    split = (k / gs) * 8 + xcd;
    bx = g % p.gx;
    by = g / p.gx;
    if (split >= p.nsplit) return;     // the whole block (padding of the split count to a multiple of 8)
  }
  const int c0 = bx * CI_;
  const int o0 = by * BO_;
  const int t_begin = split * p.tiles_per_split;
  const int t_end = min(t_begin + p.tiles_per_split, p.ntiles);
  const size_t xplane = size_t(p.H + 2) * p.wp;
  const size_t gplane = size_t(p.H) * p.W;

  f32x16 acc[WJT];
#pragma unroll
  for (int t = 0; t < WJT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int ot = wave % S::OT, jt0 = (wave / S::OT) * WJT;
  const unsigned char* aPtr = Gs + (ot * 32 + li) * G_ROW_B + lh * 16;
  int bOff[WJT];
#pragma unroll
  for (int jt = 0; jt < WJT; ++jt) {
    const int j = (jt0 + jt) * 32 + li;
    const int c = j / 9, tap = j - c * 9;
    bOff[jt] = (tap % 3) * S::XCOPY + c * X_CH_B + (tap / 3) * X_ROW_B + lh * 16;
  }

  // Two register sets and barriers that do NOT drain vmcnt: the global loads of tile t + 2 are issued while tile t is
  // multiplied and stay in flight across the two barriers of tile t + 1.  (__syncthreads() waits for vmcnt(0): with it
  // a load had one MFMA phase, ~1.3 us, to arrive whatever the prefetch distance, and the kernel ran at load latency.)
  struct Stage {
    u32x4 gv[S::NGS_], xv[S::NXS_];
    unsigned xl[S::NXS_];
  };
  Stage st0, st1;
#define ST_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  // Addresses: a load is [uniform tile base, kept in scalar registers and advanced tile by tile] + [a per-thread byte
  // offset that does not depend on the tile, computed once] -- one instruction per load (scalar base + 32-bit vector
  // offset) and no vector ALU work.  With the addresses formed per tile (tile decode by division, clamps, 64-bit
  // multiply-adds per item: ~160-250 vector instructions per wave and tile at 4 cycles each) the three waves of a
  // SIMD serialised on that arithmetic: in-kernel stamps had the LAST wave of a SIMD issue its loads 2 000 (<128, 32>)
  // to 4 400 (<64, 64>) cycles after the barrier, 40-60 % of a tile's 5 100-6 800 cycles, with the matrix pipe waiting.
  unsigned goff[S::NGS_], xoff[S::NXS_], xloff[S::NXS_];
#pragma unroll
  for (int i = 0; i < S::NGS_; ++i) {
    const int g = min(tid + i * NTHREADS, S::NG - 1);
    const int seg = g >> 2, q = g & 3;                           // seg = channel * 4 + tile row
    const int o = min(o0 + (seg >> 2), p.cout - 1);
    goff[i] = unsigned((size_t(o) * gplane + size_t(seg & 3) * p.W + q * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < S::NXS_; ++i) {
    const int it = min(tid + i * NTHREADS, S::NX - 1);           // channel xc, halo row xr, 8-pixel group xq
    const int xq = it & 3, xr = (it >> 2) % HRB, xc = it / (4 * HRB);
    xoff[i] = unsigned((size_t(min(c0 + xc, p.cin - 1)) * xplane + size_t(xr) * p.wp + XPAD + xq * 8) * 2);
    // One halo dword per item instead of two neighbour dwords: the four lanes xq = 0..3 of a row hold consecutive
    // 8-pixel groups, so the left / right neighbour pixels of the inner groups come from the adjacent LANE at store
    // time (DPP row_shr / row_shl) and only xq == 0 / xq == 3 fetch the pixels outside the 32-pixel tile row.  (Two
    // plain neighbour loads got merged by hipcc with the 16-byte load into a misaligned pair whose registers were
    // re-sorted right behind the loads: s_waitcnt vmcnt(4); v_mov ... -- a wait for the whole memory latency in
    // front of every MFMA phase.)
    xloff[i] = xq == 0 ? xoff[i] - 4u : (xq == 3 ? xoff[i] + 16u : xoff[i] + 4u * unsigned(xq));
  }
  // the tile the next load() call fetches: calls come in tile order t_begin, t_begin + 1, ...
  int lx = t_begin % p.tiles_x, ly = (t_begin / p.tiles_x) % p.tiles_y, lb = t_begin / (p.tiles_x * p.tiles_y);
  auto load = [&](Stage& r) {
    const size_t gbase = (size_t(lb) * p.cout * gplane + size_t(ly * TRB) * p.W + size_t(lx * TW)) * 2;
    const size_t xbase = (size_t(lb) * p.cin * xplane + size_t(ly * TRB) * p.wp + size_t(lx * TW)) * 2;
    const unsigned char* gb = reinterpret_cast<const unsigned char*>(p.g16) + gbase;
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x16) + xbase;
#pragma unroll
    for (int i = 0; i < S::NGS_; ++i) r.gv[i] = *reinterpret_cast<const u32x4*>(gb + goff[i]);
#pragma unroll
    for (int i = 0; i < S::NXS_; ++i) {
      r.xv[i] = *reinterpret_cast<const u32x4*>(xb + xoff[i]);
      r.xl[i] = *reinterpret_cast<const unsigned*>(xb + xloff[i]);
    }
    if (++lx == p.tiles_x) {
      lx = 0;
      if (++ly == p.tiles_y) { ly = 0; ++lb; }
    }
  };
  static_assert(S::NX % NTHREADS == 0, "the DPP neighbour exchange in store() needs whole waves of x items");
  auto store = [&](const Stage& r) {
#pragma unroll
    for (int i = 0; i < S::NGS_; ++i) {
      const int g = tid + i * NTHREADS;
      if (g >= S::NG) break;
      const int seg = g >> 2, q = g & 3;
      u32x4 v = r.gv[i];
      if (o0 + (seg >> 2) >= p.cout) v = u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(Gs + (seg >> 2) * G_ROW_B + ((seg & 3) * TW + q * 8) * 2) = v;
    }
#pragma unroll
    for (int i = 0; i < S::NXS_; ++i) {
      const int it = tid + i * NTHREADS;
      if (it >= S::NX) break;
      const int xq = it & 3, xr = (it >> 2) % HRB, xc = it / (4 * HRB);
      u32x4 mid = r.xv[i];
      // lane - 1 / lane + 1 of the same 16-lane row: the neighbouring 8-pixel group (it = tid + i * 768, so xq is the
      // lane's low two bits); executed by all lanes before any of them leaves the loop
      const unsigned from_left = __builtin_amdgcn_update_dpp(0u, mid[3], 0x111, 0xF, 0xF, false);
      const unsigned from_right = __builtin_amdgcn_update_dpp(0u, mid[0], 0x101, 0xF, 0xF, false);
      unsigned l = xq == 0 ? r.xl[i] : from_left, rt = xq == 3 ? r.xl[i] : from_right;
      if (c0 + xc >= p.cin) { mid = u32x4{0u, 0u, 0u, 0u}; l = 0u; rt = 0u; }
      // dword k of `mid` holds pixels 2k (low half) and 2k + 1; one pixel to the left / right by v_alignbit
      const unsigned s01 = __builtin_amdgcn_alignbit(mid[1], mid[0], 16);
      const unsigned s12 = __builtin_amdgcn_alignbit(mid[2], mid[1], 16);
      const unsigned s23 = __builtin_amdgcn_alignbit(mid[3], mid[2], 16);
      const u32x4 left = {__builtin_amdgcn_alignbit(mid[0], l, 16), s01, s12, s23};
      const u32x4 right = {s01, s12, s23, __builtin_amdgcn_alignbit(rt, mid[3], 16)};
      unsigned char* dst = Xs + xc * X_CH_B + xr * X_ROW_B + xq * 16;
      *reinterpret_cast<u32x4*>(dst) = left;                       // kx = 0: pixels x - 1 ..
      *reinterpret_cast<u32x4*>(dst + S::XCOPY) = mid;             // kx = 1
      *reinterpret_cast<u32x4*>(dst + 2 * S::XCOPY) = right;       // kx = 2: pixels x + 1 ..
    }
  };

  auto mfma_tile = [&]() {
#pragma unroll
    for (int s = 0; s < TPB / 16; ++s) {
      const bf16x8 a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(aPtr + s * 32));
      const int poff = (s >> 1) * X_ROW_B + (s & 1) * 32;
#pragma unroll
      for (int jt = 0; jt < WJT; ++jt) {
        const bf16x8 bv = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Xs + bOff[jt] + poff));
        acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv, acc[jt], 0, 0, 0);
      }
    }
  };
  if (t_begin < t_end) load(st0);
  if (t_begin + 1 < t_end) load(st1);
  for (int t = t_begin; t < t_end; t += 2) {
    ST_LDS_BARRIER();   // previous tile's MFMA phase is done with LDS
    store(st0);
    ST_LDS_BARRIER();
    if (t + 2 < t_end) load(st0);          // tile t + 2: in flight during two MFMA phases
    mfma_tile();
    if (t + 1 < t_end) {
      ST_LDS_BARRIER();
      store(st1);
      ST_LDS_BARRIER();
      if (t + 3 < t_end) load(st1);        // tile t + 3
      mfma_tile();
    }
  }
#undef ST_LDS_BARRIER

  float* slab = p.slabs + size_t(split) * p.cout * p.cin * 9;
#pragma unroll
  for (int jt = 0; jt < WJT; ++jt) {
    const int jg = c0 * 9 + (jt0 + jt) * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = o0 + ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (o < p.cout && jg < p.cin * 9)
        slab[size_t(o) * p.cin * 9 + jg] = acc[jt][r];
    }
  }
}

}  // namespace bf

struct Plan {
  int tiles_x, tiles_y, ntiles, nsplit, tiles_per_split;
};

// block shape of the fp32 kernel for a layer (see WShape)
inline void wgrad_shape(int cout, int* bo, int* ci) {
  if (cout <= 64) { *bo = 64; *ci = 64; }
  else { *bo = BO; *ci = CI; }
}

Plan make_plan(int n, int cout, int cin, int H, int W, int tile_rows = TR, int bo = BO, int ci = CI) {
  Plan pl;
  pl.tiles_x = st::cdiv(W, TW);
  pl.tiles_y = st::cdiv(H, tile_rows);
  const long long nt = (long long)n * pl.tiles_x * pl.tiles_y;
  pl.ntiles = (int)std::min<long long>(nt, 0x7fffffff);
  const int ctiles = st::cdiv(cin, ci) * st::cdiv(cout, bo);
  // one block per CU, exactly one round of blocks: measured on MI355X (B = 8, 256^2 and 512^2 decoder
  // shapes) 256 blocks 6.1 ms, 512 blocks 6.8, 1024 blocks 7.9, 384 blocks 8.2 (ragged second round);
  // fewer splits also means fewer partial slabs to write and re-read
  int want = std::max(1, 256 / ctiles);
  int nsplit = std::max(1, std::min(want, pl.ntiles / 4));
  nsplit = std::min(nsplit, 512);
  pl.tiles_per_split = st::cdiv(pl.ntiles, nsplit);
  pl.nsplit = st::cdiv(pl.ntiles, pl.tiles_per_split);
  return pl;
}

}  // namespace

namespace st {
// csrc/conv3x3_small.hip: vector-ALU weight gradient of the 3-output-channel layer
int wgrad_small_nsplit(int n, int cout, int cin, int H, int W);
int wgrad_small_launch(const float* x, const float* dy, const float* ygate, float* slabs, float* bias_slabs, int n,
                       int cin, int hs, int ws, int H, int W, int up, int pad_zero, hipStream_t s, bool x_g8);
}  // namespace st

// cout == 3: the vector-ALU kernel writes its partial slabs, the fixed-order reduction below sums them
static int wgrad_small(const float* x, const float* dy, const float* y_gate, float* dw, float* dbias,
                       float* workspace, int n, int cin, int hin, int win, int cout, int up, int pad_mode,
                       hipStream_t s, bool* handled, bool x_g8 = false) {
  const int H = hin << up, W = win << up;
  const int nsplit = st::wgrad_small_nsplit(n, cout, cin, H, W);
  *handled = nsplit > 0;
  if (!*handled) return ST_OK;
  float* bias_slabs = dbias ? workspace + size_t(nsplit) * cout * cin * 9 : nullptr;
  int rc = st::wgrad_small_launch(x, dy, y_gate, workspace, bias_slabs, n, cin, hin, win, H, W, up,
                                  pad_mode == ST_PAD_ZERO, s, x_g8);
  if (rc != ST_OK) return rc;
  const long long nw = (long long)cout * cin * 9;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((int)std::min<long long>(st::cdivll(nw, 256), 2048)),
                     dim3(256), 0, s, workspace, dw, nw, nsplit);
  if (dbias)
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(1), dim3(256), 0, s, bias_slabs, dbias, (long long)cout, nsplit);
  return st::check_launch("st_conv3x3_wgrad(cout=3, reduce)");
}

extern "C" size_t st_conv3x3_wgrad_workspace_bytes(int n, int cout, int cin, int h, int w) {
  if (n <= 0 || cout <= 0 || cin <= 0 || h <= 0 || w <= 0) return 0;
  // covers both the fp32 (2-row tiles) and the bf16 (4-row tiles) split plans
  int bo, ci;
  wgrad_shape(cout, &bo, &ci);
  const int nsplit = std::max({make_plan(n, cout, cin, h, w, TR, bo, ci).nsplit,
                               make_plan(n, cout, cin, h, w, bf::TRB).nsplit,
                               make_plan(n, cout, cin, h, w, bf::TRB, bo, ci).nsplit,
                               st::wgrad_small_nsplit(n, cout, cin, h, w)});
  return size_t(nsplit) * (size_t(cout) * cin * 9 + cout) * sizeof(float);
}

extern "C" int st_conv3x3_wgrad_cout3_xg8(const void* x_g8, const float* dy, float* dw, float* dbias,
                                          float* workspace, int n, int cin, int hin, int win, int upsample2x,
                                          int pad_mode, st_stream_t stream) {
  ST_REQUIRE(x_g8 && dy && dw && workspace, "st_conv3x3_wgrad_cout3_xg8: null pointer");
  ST_REQUIRE(n > 0 && cin > 0 && cin % 8 == 0 && hin > 0 && win > 0,
             "st_conv3x3_wgrad_cout3_xg8: bad dims n=%d cin=%d (multiple of 8) h=%d w=%d", n, cin, hin, win);
  ST_REQUIRE(pad_mode == ST_PAD_REPLICATE || pad_mode == ST_PAD_ZERO, "st_conv3x3_wgrad_cout3_xg8: bad pad_mode %d",
             pad_mode);
  ST_REQUIRE(reinterpret_cast<uintptr_t>(x_g8) % 16 == 0 && reinterpret_cast<uintptr_t>(dy) % 16 == 0,
             "st_conv3x3_wgrad_cout3_xg8: x_g8 / dy must be 16-byte aligned");
  bool handled = false;
  const int rc = wgrad_small(reinterpret_cast<const float*>(x_g8), dy, nullptr, dw, dbias, workspace, n, cin, hin, win,
                             3, upsample2x ? 1 : 0, pad_mode, (hipStream_t)stream, &handled, true);
  ST_REQUIRE(handled, "st_conv3x3_wgrad_cout3_xg8: too many pixel tiles");
  return rc;
}

extern "C" int st_conv3x3_wgrad(const float* x, const float* dy, const float* y_gate, float* dw,
                                float* dbias, float* workspace, int n, int cin, int hin, int win,
                                int cout, int upsample2x, int pad_mode, st_stream_t stream) {
  ST_REQUIRE(x && dy && dw && workspace, "st_conv3x3_wgrad: null pointer");
  ST_REQUIRE(n > 0 && cin > 0 && cout > 0 && hin > 0 && win > 0,
             "st_conv3x3_wgrad: bad dims n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, hin, win);
  ST_REQUIRE(pad_mode == ST_PAD_REPLICATE || pad_mode == ST_PAD_ZERO,
             "st_conv3x3_wgrad: bad pad_mode %d", pad_mode);
  ST_REQUIRE(reinterpret_cast<uintptr_t>(dy) % 16 == 0 &&
             (!y_gate || reinterpret_cast<uintptr_t>(y_gate) % 16 == 0),
             "st_conv3x3_wgrad: dy / y_gate must be 16-byte aligned");
  const int up = upsample2x ? 1 : 0;
  const int H = hin << up, W = win << up;
  {
    bool handled = false;
    const int rc_small = wgrad_small(x, dy, y_gate, dw, dbias, workspace, n, cin, hin, win, cout, up, pad_mode,
                                     (hipStream_t)stream, &handled);
    if (handled) return rc_small;
  }
  ST_REQUIRE((long long)n * st::cdiv(W, TW) * st::cdiv(H, TR) <= 0x7fffffffLL,
             "st_conv3x3_wgrad: too many pixel tiles");
  int bo, ci;
  wgrad_shape(cout, &bo, &ci);
  const Plan pl = make_plan(n, cout, cin, H, W, TR, bo, ci);
  WgradParams p;
  p.dy = dy; p.ygate = y_gate; p.x = x;
  p.slabs = workspace;
  p.bias_slabs = dbias ? workspace + size_t(pl.nsplit) * cout * cin * 9 : nullptr;
  p.n = n; p.cout = cout; p.cin = cin; p.hs = hin; p.ws = win; p.H = H; p.W = W;
  p.up = up; p.pad_zero = pad_mode == ST_PAD_ZERO;
  p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles;
  p.tiles_per_split = pl.tiles_per_split; p.nsplit = pl.nsplit;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(st::cdiv(cin, ci), st::cdiv(cout, bo), pl.nsplit);
#define ST_WG_LAUNCH(BO_, CI_)                                                                              \
  {                                                                                                         \
    if (y_gate)                                                                                             \
      hipLaunchKernelGGL((conv3x3_wgrad_kernel<true, BO_, CI_>), grid, dim3(NTHREADS), (WShape<BO_, CI_>::LDS), s, p); \
    else                                                                                                    \
      hipLaunchKernelGGL((conv3x3_wgrad_kernel<false, BO_, CI_>), grid, dim3(NTHREADS), (WShape<BO_, CI_>::LDS), s, p); \
  }
  if (bo == 64) ST_WG_LAUNCH(64, 64)
  else ST_WG_LAUNCH(128, 32)
#undef ST_WG_LAUNCH
  int rc = st::check_launch("st_conv3x3_wgrad");
  if (rc != ST_OK) return rc;
  const long long nw = (long long)cout * cin * 9;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((int)std::min<long long>(st::cdivll(nw, 256), 2048)),
                     dim3(256), 0, s, p.slabs, dw, nw, pl.nsplit);
  if (dbias)
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(st::cdiv(cout, 256)), dim3(256), 0, s,
                       p.bias_slabs, dbias, (long long)cout, pl.nsplit);
  return st::check_launch("st_conv3x3_wgrad(reduce)");
}

// staged bf16 path: whole 4 x 32 pixel tiles, 16-B rows
// one image's operand planes must stay below 4 GiB: the kernel adds 32-bit per-thread byte offsets to a scalar tile base
static bool wgrad_bf16_staged_ok(int cout, int cin, int H, int W) {
  return cout != 3 && W % 32 == 0 && H % bf::TRB == 0 && (long long)cout * H * W * 2 < (1LL << 32) &&
         (long long)cin * (H + 2) * (W + 2 * bf::XPAD) * 2 < (1LL << 32);
}
static size_t align16(size_t bytes) { return (bytes + 15) & ~size_t(15); }

extern "C" size_t st_conv3x3_wgrad_bf16_workspace_bytes(int n, int cout, int cin, int h, int w) {
  size_t bytes = align16(st_conv3x3_wgrad_workspace_bytes(n, cout, cin, h, w));
  if (bytes == 0 || !wgrad_bf16_staged_ok(cout, cin, h, w)) return bytes;
  bytes += align16(size_t(n) * cout * sizeof(float));                                   // bias rows per image
  bytes += align16(size_t(n) * cout * h * w * sizeof(unsigned short));                   // G
  bytes += align16(size_t(n) * cin * (h + 2) * (w + 2 * bf::XPAD) * sizeof(unsigned short));   // Xp
  return bytes;
}

extern "C" int st_conv3x3_wgrad_bf16(const float* x, const float* dy, const float* y_gate, float* dw,
                                     float* dbias, float* workspace, int n, int cin, int hin, int win,
                                     int cout, int upsample2x, int pad_mode, st_stream_t stream) {
  ST_REQUIRE(x && dy && dw && workspace, "st_conv3x3_wgrad_bf16: null pointer");
  ST_REQUIRE(n > 0 && cin > 0 && cout > 0 && hin > 0 && win > 0,
             "st_conv3x3_wgrad_bf16: bad dims n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, hin, win);
  ST_REQUIRE(pad_mode == ST_PAD_REPLICATE || pad_mode == ST_PAD_ZERO,
             "st_conv3x3_wgrad_bf16: bad pad_mode %d", pad_mode);
  ST_REQUIRE(reinterpret_cast<uintptr_t>(dy) % 16 == 0 &&
             (!y_gate || reinterpret_cast<uintptr_t>(y_gate) % 16 == 0),
             "st_conv3x3_wgrad_bf16: dy / y_gate must be 16-byte aligned");
  const int up = upsample2x ? 1 : 0;
  const int H = hin << up, W = win << up;
  {
    bool handled = false;
    const int rc_small = wgrad_small(x, dy, y_gate, dw, dbias, workspace, n, cin, hin, win, cout, up, pad_mode,
                                     (hipStream_t)stream, &handled);
    if (handled) return rc_small;
  }
  ST_REQUIRE((long long)n * st::cdiv(W, TW) * st::cdiv(H, bf::TRB) <= 0x7fffffffLL,
             "st_conv3x3_wgrad_bf16: too many pixel tiles");
  const bool staged = wgrad_bf16_staged_ok(cout, cin, H, W) && (long long)n * cout <= 0x7fffffffLL &&
                      (long long)n * cin <= 0x7fffffffLL && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                      reinterpret_cast<uintptr_t>(workspace) % 16 == 0;
  int sbo = BO, sci = CI;
  if (staged) wgrad_shape(cout, &sbo, &sci);          // <64, 64> for cout <= 64, as in the fp32 kernel
  const Plan pl = make_plan(n, cout, cin, H, W, bf::TRB, sbo, sci);
  WgradParams p;
  p.dy = dy; p.ygate = y_gate; p.x = x;
  p.slabs = workspace;
  p.bias_slabs = dbias ? workspace + size_t(pl.nsplit) * cout * cin * 9 : nullptr;
  p.n = n; p.cout = cout; p.cin = cin; p.hs = hin; p.ws = win; p.H = H; p.W = W;
  p.up = up; p.pad_zero = pad_mode == ST_PAD_ZERO;
  p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles;
  p.tiles_per_split = pl.tiles_per_split; p.nsplit = pl.nsplit;
  p.g16 = nullptr; p.x16 = nullptr; p.wp = 0;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(st::cdiv(cin, sci), st::cdiv(cout, sbo), pl.nsplit);
  if (staged) {
    // operands prepared once (see the staged kernels): workspace = [slabs | bias rows per image | G | Xp]
    unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace);
    size_t off = align16(st_conv3x3_wgrad_workspace_bytes(n, cout, cin, H, W));
    float* bias_part = reinterpret_cast<float*>(wsb + off);
    off += align16(size_t(n) * cout * sizeof(float));
    unsigned short* g16 = reinterpret_cast<unsigned short*>(wsb + off);
    off += align16(size_t(n) * cout * H * W * sizeof(unsigned short));
    unsigned short* x16 = reinterpret_cast<unsigned short*>(wsb + off);
    const int wp = W + 2 * bf::XPAD;
    if (y_gate)
      hipLaunchKernelGGL(bf::wgrad_prep_g_kernel<true>, dim3((unsigned)(n * cout)), dim3(256), 0, s, dy, y_gate, g16,
                         dbias ? bias_part : nullptr, H * W);
    else
      hipLaunchKernelGGL(bf::wgrad_prep_g_kernel<false>, dim3((unsigned)(n * cout)), dim3(256), 0, s, dy, y_gate, g16,
                         dbias ? bias_part : nullptr, H * W);
    {
      const long long items = (long long)n * cin * (H + 2) * (wp / 8);
      hipLaunchKernelGGL(bf::wgrad_prep_x_kernel, dim3((unsigned)std::min<long long>(st::cdivll(items, 256), 16384)),
                         dim3(256), 0, s, x, x16, (long long)n * cin, hin, win, H, W, wp, up, p.pad_zero);
    }
    p.g16 = g16; p.x16 = x16; p.wp = wp;
    p.bias_slabs = nullptr;
    p.gx = (int)grid.x; p.gy = (int)grid.y;
    const dim3 sgrid(grid.x * grid.y * unsigned((pl.nsplit + 7) / 8 * 8));
    if (sbo == 64)
      ST_LAUNCH_DYN_LDS("st_conv3x3_wgrad_bf16", (bf::conv3x3_wgrad_bf16s_kernel<64, 64>), sgrid, dim3(NTHREADS),
                        (bf::SShape<64, 64>::LDS), s, p);
    else
      ST_LAUNCH_DYN_LDS("st_conv3x3_wgrad_bf16", (bf::conv3x3_wgrad_bf16s_kernel<128, 32>), sgrid, dim3(NTHREADS),
                        (bf::SShape<128, 32>::LDS), s, p);
    p.gx = p.gy = 0;
    int rc = st::check_launch("st_conv3x3_wgrad_bf16(staged)");
    if (rc != ST_OK) return rc;
    const long long nw = (long long)cout * cin * 9;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((int)std::min<long long>(st::cdivll(nw, 256), 2048)),
                       dim3(256), 0, s, p.slabs, dw, nw, pl.nsplit);
    if (dbias)      // the per-image row sums are "slabs" of [cout]: same fixed-order reduction
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(st::cdiv(cout, 256)), dim3(256), 0, s, bias_part, dbias,
                         (long long)cout, n);
    return st::check_launch("st_conv3x3_wgrad_bf16(staged, reduce)");
  }
  if (y_gate)   // 82.8 KB of dynamic LDS (> the 64 KB default limit)
    ST_LAUNCH_DYN_LDS("st_conv3x3_wgrad_bf16", bf::conv3x3_wgrad_bf16_kernel<true>, grid, dim3(NTHREADS),
                      bf::LDS_BYTES_B, s, p);
  else
    ST_LAUNCH_DYN_LDS("st_conv3x3_wgrad_bf16", bf::conv3x3_wgrad_bf16_kernel<false>, grid, dim3(NTHREADS),
                      bf::LDS_BYTES_B, s, p);
  int rc = st::check_launch("st_conv3x3_wgrad_bf16");
  if (rc != ST_OK) return rc;
  const long long nw = (long long)cout * cin * 9;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((int)std::min<long long>(st::cdivll(nw, 256), 2048)),
                     dim3(256), 0, s, p.slabs, dw, nw, pl.nsplit);
  if (dbias)
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(st::cdiv(cout, 256)), dim3(256), 0, s,
                       p.bias_slabs, dbias, (long long)cout, pl.nsplit);
  return st::check_launch("st_conv3x3_wgrad_bf16(reduce)");
}

// Weight gradient inside the bf16 mode's G8 decoder chain: operand G arrives READY (bf16 [n][cout][H][W], written by
// st_conv3x3_dgrad_bf16_g8_chain of the layer above together with the bias partial sums), x is fp32 NCHW or G8.
// Staged shapes only (W % 32 == 0, H % 4 == 0, cout != 3).
extern "C" size_t st_conv3x3_wgrad_bf16_pre_workspace_bytes(int n, int cout, int cin, int h, int w) {
  size_t bytes = align16(st_conv3x3_wgrad_workspace_bytes(n, cout, cin, h, w));
  if (bytes == 0 || !wgrad_bf16_staged_ok(cout, cin, h, w)) return 0;
  bytes += align16(size_t(n) * cin * (h + 2) * (w + 2 * bf::XPAD) * sizeof(unsigned short));   // Xp
  return bytes;
}

extern "C" int st_conv3x3_wgrad_bf16_pre(const void* x, int x_g8, const void* g16, const float* bias_slabs,
                                         int n_bias_slabs, float* dw, float* dbias, float* workspace, int n, int cin,
                                         int hin, int win, int cout, int upsample2x, int pad_mode,
                                         st_stream_t stream) {
  ST_REQUIRE(x && g16 && dw && workspace, "st_conv3x3_wgrad_bf16_pre: null pointer");
  ST_REQUIRE(n > 0 && cin > 0 && cout > 0 && hin > 0 && win > 0,
             "st_conv3x3_wgrad_bf16_pre: bad dims n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, hin, win);
  ST_REQUIRE(pad_mode == ST_PAD_REPLICATE || pad_mode == ST_PAD_ZERO, "st_conv3x3_wgrad_bf16_pre: bad pad_mode %d",
             pad_mode);
  ST_REQUIRE(!dbias || (bias_slabs && n_bias_slabs > 0), "st_conv3x3_wgrad_bf16_pre: dbias needs the bias partial sums");
  const int up = upsample2x ? 1 : 0;
  const int H = hin << up, W = win << up;
  ST_REQUIRE(wgrad_bf16_staged_ok(cout, cin, H, W), "st_conv3x3_wgrad_bf16_pre: needs W %% 32 == 0, H %% 4 == 0, cout != 3 "
             "and operand planes of one image below 4 GiB (got %dx%d, cout=%d, cin=%d)", H, W, cout, cin);
  ST_REQUIRE(!x_g8 || cin % 8 == 0, "st_conv3x3_wgrad_bf16_pre: a G8 x needs cin %% 8 == 0");
  ST_REQUIRE(reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(g16) % 16 == 0 &&
             reinterpret_cast<uintptr_t>(workspace) % 16 == 0, "st_conv3x3_wgrad_bf16_pre: 16-byte aligned buffers expected");
  ST_REQUIRE((long long)n * st::cdiv(W, TW) * st::cdiv(H, bf::TRB) <= 0x7fffffffLL && (long long)n * cin <= 0x7fffffffLL,
             "st_conv3x3_wgrad_bf16_pre: too many pixel tiles");
  int sbo, sci;
  wgrad_shape(cout, &sbo, &sci);
  const Plan pl = make_plan(n, cout, cin, H, W, bf::TRB, sbo, sci);
  WgradParams p;
  p.dy = nullptr; p.ygate = nullptr; p.x = nullptr;
  p.slabs = workspace;
  p.bias_slabs = nullptr;
  p.n = n; p.cout = cout; p.cin = cin; p.hs = hin; p.ws = win; p.H = H; p.W = W;
  p.up = up; p.pad_zero = pad_mode == ST_PAD_ZERO;
  p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.ntiles = pl.ntiles;
  p.tiles_per_split = pl.tiles_per_split; p.nsplit = pl.nsplit;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(st::cdiv(cin, sci), st::cdiv(cout, sbo), pl.nsplit);
  unsigned char* wsb = reinterpret_cast<unsigned char*>(workspace);
  unsigned short* x16 = reinterpret_cast<unsigned short*>(wsb + align16(st_conv3x3_wgrad_workspace_bytes(n, cout, cin, H, W)));
  const int wp = W + 2 * bf::XPAD;
  if (x_g8) {
    const long long bcgs = (long long)n * (cin / 8);
    const long long items = bcgs * (H + 2) * (wp / 8);
    hipLaunchKernelGGL(bf::wgrad_prep_x_g8_kernel, dim3((unsigned)std::min<long long>(st::cdivll(items, 256), 16384)),
                       dim3(256), 0, s, (const bf::u32x4*)x, x16, bcgs, cin / 8, hin, win, H, W, wp, up, p.pad_zero);
  } else {
    const long long items = (long long)n * cin * (H + 2) * (wp / 8);
    hipLaunchKernelGGL(bf::wgrad_prep_x_kernel, dim3((unsigned)std::min<long long>(st::cdivll(items, 256), 16384)),
                       dim3(256), 0, s, (const float*)x, x16, (long long)n * cin, hin, win, H, W, wp, up, p.pad_zero);
  }
  p.g16 = (unsigned short*)g16; p.x16 = x16; p.wp = wp;
  p.gx = (int)grid.x; p.gy = (int)grid.y;
  const dim3 sgrid(grid.x * grid.y * unsigned((pl.nsplit + 7) / 8 * 8));
  if (getenv("ST_WGRAD_NO_XCD")) { p.gx = p.gy = 0; }
  if (sbo == 64)
    ST_LAUNCH_DYN_LDS("st_conv3x3_wgrad_bf16_pre", (bf::conv3x3_wgrad_bf16s_kernel<64, 64>), p.gx ? sgrid : grid, dim3(NTHREADS),
                      (bf::SShape<64, 64>::LDS), s, p);
  else
    ST_LAUNCH_DYN_LDS("st_conv3x3_wgrad_bf16_pre", (bf::conv3x3_wgrad_bf16s_kernel<128, 32>), p.gx ? sgrid : grid, dim3(NTHREADS),
                      (bf::SShape<128, 32>::LDS), s, p);
  int rc = st::check_launch("st_conv3x3_wgrad_bf16_pre");
  if (rc != ST_OK) return rc;
  const long long nw = (long long)cout * cin * 9;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3((int)std::min<long long>(st::cdivll(nw, 256), 2048)), dim3(256), 0, s,
                     p.slabs, dw, nw, pl.nsplit);
  if (dbias) {
    constexpr int SEGS = 64;
    if (n_bias_slabs > 8 * SEGS) {
      // many rows: two stages (the partial rows live in the slab area of the workspace, free again by now --
      // the weight slabs were consumed by the launch above; SEGS * cout floats always fit one weight slab)
      const int rows_per_seg = st::cdiv(n_bias_slabs, SEGS);
      hipLaunchKernelGGL(slab_reduce_rows_kernel, dim3(st::cdiv(cout, 16), SEGS), dim3(256), 0, s, bias_slabs, p.slabs,
                         cout, n_bias_slabs, rows_per_seg);
      hipLaunchKernelGGL(slab_reduce_wave_kernel, dim3(cout), dim3(64), 0, s, p.slabs, dbias, cout,
                         st::cdiv(n_bias_slabs, rows_per_seg));
    } else {
      hipLaunchKernelGGL(slab_reduce_wave_kernel, dim3(cout), dim3(64), 0, s, bias_slabs, dbias, cout, n_bias_slabs);
    }
  }
  return st::check_launch("st_conv3x3_wgrad_bf16_pre(reduce)");
}
