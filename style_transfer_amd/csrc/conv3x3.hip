// 3x3 stride-1 convolution as an fp32 MFMA implicit GEMM for gfx950 (MI355X).
//
// Replaces SymmetricPadding2D((1,1,1,1)) -> Conv2d(3x3) -> ReLU of the reference encoder /
// decoder (generator.py:47-76, 89-117), with the nearest x2 upsample that precedes three
// decoder convs (generator.py:92,105,112) folded into the load address math, and the
// zero-padded convs of the torchvision VGG19 loss network (trainer.py:165-199).
//
// GEMM view (per image):  D[cout][pixel] = sum_k  Wp[k][cout] * X[k][pixel],  k = (cin, tap).
//   * M (MFMA rows, register index of D) = output channels, N (MFMA cols, lane) = pixels, so
//     each accumulator register is one NCHW row segment: 32 consecutive x of one channel
//     -> every store instruction writes two full 128-B lines.
//   * v_mfma_f32_32x32x2_f32: exact fp32 (an fmaf chain), 64 cycles / SIMD.
//   * A block owns BM channels x (TH rows x 32 columns) pixels.  Per K chunk of KC = 8 input
//     channels it stages the (TH+2) x 34 input halo of each channel ONCE in LDS (edge
//     replicate / zero pad / upsample resolved here) and re-reads it for all nine taps,
//     plus the 72 x BM weight slab.  Operand reads are conflict-free ds_read_b32 with
//     compile-time offsets (32 lanes = 32 consecutive floats).
//   * The next chunk's global loads are issued before the MFMA phase and written to LDS
//     after it (register prefetch), several blocks per CU overlap staging with MFMA.
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace {

constexpr int KC = 8;        // input channels per K chunk
constexpr int TW = 32;       // pixel-tile width == one MFMA N tile
constexpr int HC = TW + 2;   // halo row length
constexpr int CPAD = 128;    // packed weights: cout padded to a multiple of this
constexpr int NTHREADS = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvParams {
  const float* x;
  const float* gate;       // GATE kernels: same shape as x; x is read as 0 where gate <= 0
  const float* wp;
  const float* bias;
  float* y;
  int n, cin, hs, ws;      // source tensor dims
  int cout, H, W;          // conv-input spatial dims, after the optional upsample
  int Ho, Wo;              // output dims: (H, W) forward; (H+2, W+2) for the padded data gradient
  int pool;                // 1: the epilogue writes maxpool2x2(ceil_mode) of the activation instead of it
  int ldo;                 // output row stride in floats (Wo, or Wo rounded up to 4 in the padded workspace)
  int shift;               // output pixel (oy, ox) is centred on input pixel (oy - shift, ox - shift)
  int cout_pad, nchunks;
  int up, pad_zero, relu;
  int tiles_x, tiles_y, ctiles, xcd_swizzle;
};

// SUB = the sub-pixel form of a conv that reads its input through a nearest x2 upsample (generator.py:92,105,112):
// output pixel (2i + a, 2j + b) only sees the 2 x 2 source pixels (i - 1 + a .. i + a) x (j - 1 + b .. j + b), with
// the 3 x 3 filter's rows / columns that fall on the same source pixel summed beforehand (st_conv3x3_pack_weights_up2)
// -- four 2 x 2 convolutions ("classes" (a, b)) over the LOW-resolution input instead of one 3 x 3 convolution over the
// upsampled one: 16 instead of 36 multiply-adds per source pixel and channel pair.  A block owns BM channels x (TH x 32)
// SOURCE pixels; its four waves are the four classes (same channels, same source tile, same staged halo).
template <int MT, int NT, int WM, int WP, bool SUB = false>
struct Cfg {
  static constexpr int BM = SUB ? 32 * MT : 32 * MT * WM;
  static constexpr int TH = SUB ? NT : NT * WP;
  static constexpr int HR = TH + 2;
  static constexpr int PLANE = HR * HC;
  static constexpr int NTAP = SUB ? 16 : 9;       // weight rows per input channel: (class, 2 x 2 tap) or 3 x 3 tap
  static constexpr int W_ELEMS = NTAP * KC * BM;
  static constexpr int X_ELEMS = KC * PLANE;
  static constexpr int NP = (PLANE + NTHREADS - 1) / NTHREADS;
  static constexpr int NW4 = (W_ELEMS / 4 + NTHREADS - 1) / NTHREADS;
  static constexpr size_t LDS_BYTES = size_t(W_ELEMS + X_ELEMS) * sizeof(float);
};

// OCC = waves per SIMD the register allocation aims at (= blocks per CU: a block has one wave per SIMD).
// The block is single-buffered in LDS (MFMA phase, barrier, LDS-store phase, barrier), so it is the OTHER blocks
// on the CU that keep the matrix pipe busy during a block's staging phase: at 3 blocks per CU (<= 168 registers,
// 3 x 43 KB of LDS) the large layers run at 135-137 TFLOP/s instead of 121-126 at 2 (220 registers).
template <int MT, int NT, int WM, int WP, bool GATE, int OCC, bool SUB = false>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void conv3x3_f32_kernel(ConvParams p) {
  using C = Cfg<MT, NT, WM, WP, SUB>;
  static_assert(WM * WP == 4, "four waves per block");
  static_assert(!SUB || !GATE, "sub-pixel form: forward only");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ws = smem;
  float* Xs = smem + C::W_ELEMS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = SUB ? 0 : wave / WP, wp = SUB ? 0 : wave % WP;
  const int cls_a = wave >> 1, cls_b = wave & 1;     // SUB: this wave's output parity class (row, column)

  // Block -> (pixel tile, channel tile).  Blocks are dealt round-robin over the 8 XCDs (each with
  // its own L2), so ids that differ by 8 share an XCD: the channel tiles of one pixel tile get
  // consecutive slots on ONE XCD and the 2..4 reads of the same activation tile hit its L2.
  int bt, ct;
  {
    const int id = blockIdx.x;
    if (p.xcd_swizzle) {
      const int k = id >> 3;
      ct = k % p.ctiles;
      bt = (k / p.ctiles) * 8 + (id & 7);
    } else {
      ct = id % p.ctiles;
      bt = id / p.ctiles;
    }
  }
  const int tx = bt % p.tiles_x;
  bt /= p.tiles_x;
  const int ty = bt % p.tiles_y;
  const int b = bt / p.tiles_y;
  const int m0 = ct * C::BM;
  const int x0 = tx * TW, y0 = ty * C::TH;

  // ---- per-thread source offsets of the halo positions this thread stages -------------
  // Every load below is unconditional (clamped, always in-bounds address) and the zero
  // cases (zero padding, channel tail) are selects, so the prefetch registers stay in
  // VGPRs and the loads issue back to back.
  int soff[C::NP];
  unsigned smask[C::NP];   // all-ones: keep the loaded value; 0: zero padding
#pragma unroll
  for (int i = 0; i < C::NP; ++i) {
    const int pidx = tid + i * NTHREADS;
    const int r = pidx / HC, cc = pidx - r * HC;
    const int gy = y0 + r - 1 - p.shift, gx = x0 + cc - 1 - p.shift;
    const bool inb = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
    if (p.up) { cy >>= 1; cx >>= 1; }
    soff[i] = cy * p.ws + cx;
    smask[i] = (inb || !p.pad_zero) ? 0xffffffffu : 0u;
  }
  const size_t plane_src = size_t(p.hs) * p.ws;
  const float* xb = p.x + size_t(b) * p.cin * plane_src;
  const float* gb = GATE ? p.gate + size_t(b) * p.cin * plane_src : nullptr;

  float xr[KC][C::NP];
  float gr[GATE ? KC : 1][C::NP];   // ReLU gate prefetch (data-gradient kernels)
  f32x4 wr[C::NW4];
  // weight slab addressing: float4 index e -> (row, c4); clamp e so the tail threads of the
  // last pass re-load a valid element (their LDS write is skipped)
  int woff[C::NW4];
#pragma unroll
  for (int i = 0; i < C::NW4; ++i) {
    const int e = min(tid + i * NTHREADS, C::W_ELEMS / 4 - 1);
    const int row = e / (C::BM / 4), c4 = e - row * (C::BM / 4);
    woff[i] = row * p.cout_pad + c4 * 4;
  }

#define ST_LOAD_CHUNK(chunk_)                                                          \
  {                                                                                    \
    _Pragma("unroll") for (int c = 0; c < KC; ++c) {                                   \
      const int cg = (chunk_) * KC + c;                                                \
      const float* src = xb + size_t(min(cg, p.cin - 1)) * plane_src;                  \
      _Pragma("unroll") for (int i = 0; i < C::NP; ++i) xr[c][i] = src[soff[i]];       \
      if (GATE) {                                                                      \
        const float* gsrc = gb + size_t(min(cg, p.cin - 1)) * plane_src;               \
        _Pragma("unroll") for (int i = 0; i < C::NP; ++i) gr[c][i] = gsrc[soff[i]];    \
      }                                                                                \
    }                                                                                  \
    const float* wsrc = p.wp + size_t(chunk_) * (C::NTAP * KC) * p.cout_pad + m0;      \
    _Pragma("unroll") for (int i = 0; i < C::NW4; ++i)                                 \
        wr[i] = *reinterpret_cast<const f32x4*>(wsrc + woff[i]);                      \
  }

  // Zero padding / channel tail are applied at the LDS store, as bit masks: a select on a
  // uniform condition at the load makes hipcc branch around it with a vmcnt(0) per element,
  // and a mask at the load is waited for before the MFMA phase instead of after it.
#define ST_STORE_CHUNK(chunk_)                                                         \
  {                                                                                    \
    _Pragma("unroll") for (int c = 0; c < KC; ++c) {                                   \
      const unsigned cmask = ((chunk_) * KC + c) < p.cin ? 0xffffffffu : 0u;           \
      _Pragma("unroll") for (int i = 0; i < C::NP; ++i) {                              \
        const int pidx = tid + i * NTHREADS;                                           \
        unsigned m = smask[i] & cmask;                                                 \
        if (GATE) m &= gr[c][i] > 0.f ? 0xffffffffu : 0u;                              \
        if (C::PLANE % NTHREADS == 0 || pidx < C::PLANE)                               \
          Xs[c * C::PLANE + pidx] = __uint_as_float(__float_as_uint(xr[c][i]) & m);    \
      }                                                                                \
    }                                                                                  \
    _Pragma("unroll") for (int i = 0; i < C::NW4; ++i) {                               \
      const int e = tid + i * NTHREADS;                                                \
      if ((C::W_ELEMS / 4) % NTHREADS == 0 || e < C::W_ELEMS / 4)                      \
        reinterpret_cast<f32x4*>(Ws)[e] = wr[i];                                      \
    }                                                                                  \
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const float* aPtr = Ws + lh * C::BM + wm * (MT * 32) + li + (SUB ? wave * (4 * KC * C::BM) : 0);
  const float* bPtr = Xs + lh * C::PLANE + (wp * NT) * HC + li + (SUB ? cls_a * HC + cls_b : 0);

  ST_LOAD_CHUNK(0);
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    if (chunk) __syncthreads();  // every wave is done reading the previous chunk
    ST_STORE_CHUNK(chunk);
    __syncthreads();
    if (chunk + 1 < p.nchunks) ST_LOAD_CHUNK(chunk + 1);  // in flight during the MFMA phase
#pragma unroll
    for (int tap = 0; tap < (SUB ? 4 : 9); ++tap) {
      // SUB: 2 x 2 taps of this wave's class; source row y0 + nt - 1 + a + dy is halo row nt + a + dy (a, b in bPtr)
      const int dy = SUB ? tap / 2 : tap / 3, dx = SUB ? tap % 2 : tap % 3;
#pragma unroll
      for (int cp = 0; cp < KC / 2; ++cp) {
        float a[MT], bb[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[mt] = aPtr[(tap * KC + 2 * cp) * C::BM + mt * 32];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bb[nt] = bPtr[(2 * cp) * C::PLANE + (nt + dy) * HC + dx];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], bb[nt], acc[mt][nt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: bias + ReLU, NCHW store (register r of a tile = one channel row) -----
  const int xx = x0 + li;
  if constexpr (SUB) {
    // source pixel (yy, xx) of class (a, b) is output pixel (2 yy + a, 2 xx + b): dword stores at stride 2 whose
    // other halves come from the partner wave (b ^ 1) of this block -- the lines are completed in the same L2
    if (xx < p.W) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (co < p.cout) {
            const float bv = p.bias ? p.bias[co] : 0.f;
            float* yrow = p.y + (size_t(b) * p.cout + co) * p.Ho * p.ldo + 2 * xx + cls_b;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int yy = y0 + nt;
              if (yy < p.H) {
                float v = acc[mt][nt][r] + bv;
                if (p.relu) v = fmaxf(v, 0.f);
                yrow[size_t(2 * yy + cls_a) * p.ldo] = v;
              }
            }
          }
        }
      }
    }
  } else if (NT % 2 == 0 && p.pool) {     // (the 2-row-tile instantiation, NT = 1, is never launched with pool)
    // fused MaxPool2d((2,2),(2,2),ceil_mode=True) (generator.py:53,60,73): rows 2k, 2k+1 are two
    // accumulator tiles of this wave, columns 2k, 2k+1 are neighbouring lanes; bias and ReLU commute
    // with the max.  Even lanes write the pooled row (the full-resolution tensor never exists).
    const int Hp = (p.Ho + 1) >> 1, Wp = (p.Wo + 1) >> 1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + wm * (MT * 32) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const float bv = (p.bias && co < p.cout) ? p.bias[co] : 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; nt += 2) {
          const int yy = y0 + wp * NT + nt;
          float v = acc[mt][nt][r];
          if (yy + 1 < p.Ho) v = fmaxf(v, acc[mt][(nt + 1) % NT][r]);
          const float vn = __shfl_xor(v, 1, 64);
          if (xx + 1 < p.Wo) v = fmaxf(v, vn);
          v += bv;
          if (p.relu) v = fmaxf(v, 0.f);
          if (!(li & 1) && co < p.cout && yy < p.Ho && xx < p.Wo)
            p.y[((size_t(b) * p.cout + co) * Hp + (yy >> 1)) * Wp + (xx >> 1)] = v;
        }
      }
    }
  } else if (xx < p.Wo) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + wm * (MT * 32) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < p.cout) {
          const float bv = p.bias ? p.bias[co] : 0.f;
          float* yrow = p.y + (size_t(b) * p.cout + co) * p.Ho * p.ldo + xx;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int yy = y0 + wp * NT + nt;
            if (yy < p.Ho) {
              float v = acc[mt][nt][r] + bv;
              if (p.relu) v = fmaxf(v, 0.f);
              yrow[size_t(yy) * p.ldo] = v;
            }
          }
        }
      }
    }
  }
}

#undef ST_LOAD_CHUNK
#undef ST_STORE_CHUNK

// OIHW -> packed [cin_pad/KC][9][KC][cout_pad] (K-major, output channel contiguous).
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                    int O, int I, int flip, int cout, int cin, int cout_pad,
                                    long long total) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int o = int(e % cout_pad);
    long long k = e / cout_pad;
    const int c = int(k % KC);
    k /= KC;
    const int tap = int(k % 9);
    const int cg = int(k / 9) * KC + c;
    float v = 0.f;
    if (o < cout && cg < cin) {
      v = flip ? w[(size_t(cg) * I + o) * 9 + (8 - tap)] : w[(size_t(o) * I + cg) * 9 + tap];
    }
    wp[e] = v;
  }
}

// Adjoint of the edge-replicate pad (+ optional nearest x2 upsample) on the padded data
// gradient dxp [planes][H+2][W+2]:  dx[i][j] = sum of dxp over every padded position that the
// forward pass read from source pixel (i, j).
__global__ __launch_bounds__(256) void dgrad_fold_kernel(const float* __restrict__ dxp,
                                                        float* __restrict__ dx,
                                                        const float* __restrict__ gate, int H, int W,
                                                        int up, int hs, int ws, long long total) {
  const int Wp = st::padded_row(W);   // row stride of the padded workspace
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int j = int(e % ws);
    long long t = e / ws;
    const int i = int(t % hs);
    const long long plane = t / hs;
    const float* src = dxp + plane * (long long)(H + 2) * Wp;
    // conv-input rows covered by source row i, then widened over the replicated border
    int u0 = up ? 2 * i : i, u1 = up ? 2 * i + 1 : i;
    int v0 = up ? 2 * j : j, v1 = up ? 2 * j + 1 : j;
    if (u0 == 0) u0 = -1;
    if (u1 == H - 1) u1 = H;
    if (v0 == 0) v0 = -1;
    if (v1 == W - 1) v1 = W;
    float s = 0.f;
    for (int u = u0; u <= u1; ++u)
      for (int v = v0; v <= v1; ++v) s += src[(size_t)(u + 1) * Wp + (v + 1)];
    // optional fused ReLU backward of the PRODUCER of this layer's input (gate = that activation)
    dx[e] = (gate && !(gate[e] > 0.f)) ? 0.f : s;
  }
}

// Vector form of dgrad_fold_kernel for ws % 4 == 0: a thread produces 4 consecutive source pixels of one row
// (16-B gate load, 16-B store) from aligned 16-B reads of the padded rows; 32-bit index arithmetic.  The scalar
// kernel spends its time on 64-bit div/mod and dword accesses (measured at half the HBM rate on the 512 x 512
// decoder layers, ~10 % of a bf16 training step).
template <bool UP>
__global__ __launch_bounds__(256) void dgrad_fold4_kernel(const float* __restrict__ dxp, float* __restrict__ dx,
                                                         const float* __restrict__ gate, int H, int W, int hs,
                                                         int ws, unsigned total4) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int Wp = st::padded_row(W);
  const unsigned q = unsigned(ws) >> 2;
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += gridDim.x * blockDim.x) {
    const unsigned row = e / q;
    const int j0 = int(e - row * q) * 4;
    const unsigned plane = row / unsigned(hs);
    const int i = int(row - plane * unsigned(hs));
    const float* src = dxp + size_t(plane) * size_t(H + 2) * Wp;
    int u0 = UP ? 2 * i : i, u1 = UP ? 2 * i + 1 : i;
    if (u0 == 0) u0 = -1;
    if (u1 == H - 1) u1 = H;
    const bool left = j0 == 0, right = j0 + 4 == ws;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int u = u0; u <= u1; ++u) {
      const float* r = src + size_t(u + 1) * Wp + (UP ? 2 * j0 : j0);
      const f4 a = *reinterpret_cast<const f4*>(r);
      if (UP) {
        const f4 b = *reinterpret_cast<const f4*>(r + 4);
        s0 += a[1] + a[2] + (left ? a[0] : 0.f);
        s1 += a[3] + b[0];
        s2 += b[1] + b[2];
        s3 += b[3] + r[8] + (right ? r[9] : 0.f);
      } else {
        s0 += a[1] + (left ? a[0] : 0.f);
        s1 += a[2];
        s2 += a[3];
        s3 += r[4] + (right ? r[5] : 0.f);
      }
    }
    const size_t o = size_t(e) * 4;
    if (gate) {
      const f4 g = __builtin_nontemporal_load(reinterpret_cast<const f4*>(gate + o));
      s0 = g[0] > 0.f ? s0 : 0.f; s1 = g[1] > 0.f ? s1 : 0.f; s2 = g[2] > 0.f ? s2 : 0.f; s3 = g[3] > 0.f ? s3 : 0.f;
    }
    f4 out; out[0] = s0; out[1] = s1; out[2] = s2; out[3] = s3;
    *reinterpret_cast<f4*>(dx + o) = out;
  }
}

// OIHW -> the sub-pixel layout [cin_pad / KC][(class a, b)(tap dy', dx')][KC][cout_pad]: filter rows / columns that
// read the same source pixel under the x2 nearest upsample are summed (a = 0: {0}, {1, 2}; a = 1: {0, 1}, {2}).
__global__ void pack_weights_up2_kernel(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin,
                                        int cout_pad, long long total) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int o = int(e % cout_pad);
    long long k = e / cout_pad;
    const int c = int(k % KC);
    k /= KC;
    const int t = int(k % 16);
    const int cg = int(k / 16) * KC + c;
    const int a = t >> 3, bq = (t >> 2) & 1, dy = (t >> 1) & 1, dx = t & 1;
    float v = 0.f;
    if (o < cout && cg < cin) {
      const int r0 = a == 0 ? (dy == 0 ? 0 : 1) : (dy == 0 ? 0 : 2), r1 = a == 0 ? (dy == 0 ? 0 : 2) : (dy == 0 ? 1 : 2);
      const int c0 = bq == 0 ? (dx == 0 ? 0 : 1) : (dx == 0 ? 0 : 2), c1 = bq == 0 ? (dx == 0 ? 0 : 2) : (dx == 0 ? 1 : 2);
      const float* wk = w + (size_t(o) * cin + cg) * 9;
      for (int r = r0; r <= r1; ++r)
        for (int q = c0; q <= c1; ++q) v += wk[r * 3 + q];
    }
    wp[e] = v;
  }
}

template <int MT, int NT, int OCC>
int launch_conv_sub(ConvParams& p, hipStream_t stream) {
  using C = Cfg<MT, NT, 2, 2, true>;
  p.tiles_x = st::cdiv(p.W, TW);
  p.tiles_y = st::cdiv(p.H, C::TH);
  const long long ptiles = (long long)p.n * p.tiles_x * p.tiles_y;
  p.ctiles = st::cdiv(p.cout, C::BM);
  ST_REQUIRE(ptiles * p.ctiles <= 0x7fffffffLL, "st_conv3x3_up2_fwd: too many tiles (%lld)", ptiles * p.ctiles);
  p.xcd_swizzle = (ptiles % 8 == 0) ? 1 : 0;
  dim3 grid((unsigned)(ptiles * p.ctiles));
  hipLaunchKernelGGL((conv3x3_f32_kernel<MT, NT, 2, 2, false, OCC, true>), grid, dim3(NTHREADS), C::LDS_BYTES, stream, p);
  return st::check_launch("st_conv3x3_up2_fwd");
}

template <int MT, int NT, int WM, int WP, bool GATE>
int launch_conv(ConvParams& p, hipStream_t stream) {
  using C = Cfg<MT, NT, WM, WP>;
  p.tiles_x = st::cdiv(p.Wo, TW);
  p.tiles_y = st::cdiv(p.Ho, C::TH);
  const long long ptiles = (long long)p.n * p.tiles_x * p.tiles_y;
  p.ctiles = st::cdiv(p.cout, C::BM);
  ST_REQUIRE(ptiles * p.ctiles <= 0x7fffffffLL, "st_conv3x3: too many tiles (%lld)", ptiles * p.ctiles);
  p.xcd_swizzle = (ptiles % 8 == 0) ? 1 : 0;
  dim3 grid((unsigned)(ptiles * p.ctiles));
  // 3 blocks per CU unless the grid is between one and two rounds of 768 slots (then the second, mostly empty
  // round costs more than the third block gains: 512->256 at 64x64, batch 16: 110 vs 124 TFLOP/s); the gated
  // (data-gradient) form and the 32-channel tile need > 168 registers and stay at 2.
  const long long blocks = ptiles * p.ctiles;
  // (the gated form needs 20 B of scratch at 168 registers: worth it on large grids only, +6 % at batch 16, 512 x 512)
  const bool occ3 = MT == 2 && !(blocks > 768 && blocks < 1536) && (!GATE || blocks >= 1536);
  if constexpr (MT == 2) {
    if (occ3) {
      hipLaunchKernelGGL((conv3x3_f32_kernel<MT, NT, WM, WP, GATE, 3>), grid, dim3(NTHREADS), C::LDS_BYTES, stream, p);
      return st::check_launch("st_conv3x3");
    }
  }
  hipLaunchKernelGGL((conv3x3_f32_kernel<MT, NT, WM, WP, GATE, 2>), grid, dim3(NTHREADS),
                     C::LDS_BYTES, stream, p);
  return st::check_launch("st_conv3x3");
}

template <bool GATE>
int dispatch_conv(ConvParams& p, hipStream_t s) {
  if (p.cout > 64) {
    {
      // small grids (training shapes: 512 -> 512 at 32 x 32, batch 8 = 256 tiles of 4 rows): 2-row tiles double
      // the number of blocks so that every CU holds 2-3 of them
      // (measured at batch 8, 256 x 256: 512 -> 256 at 32 x 32, 128 blocks: 0.337 -> 0.197 ms; from 256 blocks up
      // the 4-row tile is as fast or faster)
      const long long blocks4 = (long long)p.n * st::cdiv(p.Wo, TW) * st::cdiv(p.Ho, 4) * st::cdiv(p.cout, 128);
      if (blocks4 <= 192 && !p.pool) return launch_conv<2, 1, 2, 2, GATE>(p, s);
    }
    return launch_conv<2, 2, 2, 2, GATE>(p, s);   // 128 ch x (4 rows x 32)
  }
  if (p.cout > 32) return launch_conv<2, 2, 1, 4, GATE>(p, s);   //  64 ch x (8 rows x 32)
  return launch_conv<1, 4, 1, 4, GATE>(p, s);                    //  32 ch x (16 rows x 32)
}

}  // namespace

namespace st {
// csrc/conv3x3_small.hip: vector-ALU kernels for the 3-channel layers
bool conv3x3_small_fwd(const float* x, const float* w_packed, const float* bias, float* y, int n, int cin,
                       int h, int w, int cout, int cout_pad, int pad_zero, int relu, hipStream_t s, int* rc);
// Adjoint of the replicate pad (+ upsample) on the padded data gradient; shared with the bf16 path.
int launch_dgrad_fold(const float* dxp, float* dx, const float* gate, int n, int cin, int h, int w, int up,
                      hipStream_t s) {
  const int hs = h >> up, ws = w >> up;
  const long long total = (long long)n * cin * hs * ws;
  if (ws % 4 == 0 && total / 4 < (1LL << 31) && reinterpret_cast<uintptr_t>(dxp) % 16 == 0 &&
      reinterpret_cast<uintptr_t>(dx) % 16 == 0 && (!gate || reinterpret_cast<uintptr_t>(gate) % 16 == 0)) {
    const unsigned total4 = unsigned(total / 4);
    const int blocks4 = (int)std::min<long long>(st::cdivll(total4, 256), 256 * 16);
    if (up)
      hipLaunchKernelGGL(dgrad_fold4_kernel<true>, dim3(blocks4), dim3(256), 0, s, dxp, dx, gate, h, w, hs, ws, total4);
    else
      hipLaunchKernelGGL(dgrad_fold4_kernel<false>, dim3(blocks4), dim3(256), 0, s, dxp, dx, gate, h, w, hs, ws, total4);
    return st::check_launch("st_conv3x3_dgrad(fold)");
  }
  const int blocks = (int)std::min<long long>(st::cdivll(total, 256), 256 * 8);
  hipLaunchKernelGGL(dgrad_fold_kernel, dim3(blocks), dim3(256), 0, s, dxp, dx, gate, h, w, up, hs, ws, total);
  return st::check_launch("st_conv3x3_dgrad(fold)");
}
}  // namespace st

extern "C" size_t st_conv3x3_packed_elems(int cout, int cin) {
  if (cout <= 0 || cin <= 0) return 0;
  const size_t cin_pad = size_t(st::cdiv(cin, KC)) * KC;
  const size_t cout_pad = size_t(st::cdiv(cout, CPAD)) * CPAD;
  return cin_pad * 9 * cout_pad;
}

extern "C" int st_conv3x3_pack_weights(const float* w_oihw, float* w_packed, int cout, int cin,
                                       int transpose_flip, st_stream_t stream) {
  ST_REQUIRE(w_oihw && w_packed, "st_conv3x3_pack_weights: null pointer");
  ST_REQUIRE(cout > 0 && cin > 0, "st_conv3x3_pack_weights: bad dims cout=%d cin=%d", cout, cin);
  // (cout, cin) describe the OIHW tensor; the packed filter swaps them when transpose_flip.
  const int pc_out = transpose_flip ? cin : cout;
  const int pc_in = transpose_flip ? cout : cin;
  const int cout_pad = st::cdiv(pc_out, CPAD) * CPAD;
  const long long total = (long long)st_conv3x3_packed_elems(pc_out, pc_in);
  const int blocks = (int)std::min<long long>(st::cdivll(total, 256), 4096);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     w_oihw, w_packed, cout, cin, transpose_flip ? 1 : 0, pc_out, pc_in,
                     cout_pad, total);
  return st::check_launch("st_conv3x3_pack_weights");
}

static int conv3x3_fwd_impl(const float* x, const float* w_packed, const float* bias, float* y,
                            int n, int cin, int hin, int win, int cout, int upsample2x,
                            int pad_mode, int relu, int pool, st_stream_t stream) {
  ST_REQUIRE(x && w_packed && y, "st_conv3x3_fwd: null pointer");
  ST_REQUIRE(n > 0 && cin > 0 && cout > 0 && hin > 0 && win > 0,
             "st_conv3x3_fwd: bad dims n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, hin, win);
  ST_REQUIRE(pad_mode == ST_PAD_REPLICATE || pad_mode == ST_PAD_ZERO,
             "st_conv3x3_fwd: bad pad_mode %d", pad_mode);
  ST_REQUIRE((long long)hin * win * (upsample2x ? 4 : 1) < (1LL << 31),
             "st_conv3x3_fwd: plane too large");
  if (!upsample2x && !pool) {
    // the 3-channel layers (conv1_1, last decoder conv) are HBM-bound: vector-ALU kernels, no MFMA padding
    static const bool no_small = getenv("ST_NO_SMALL_CONV") != nullptr;     // A/B switch for benchmarking
    int rc = ST_OK;
    if (!no_small && st::conv3x3_small_fwd(x, w_packed, bias, y, n, cin, hin, win, cout, st::cdiv(cout, CPAD) * CPAD,
                                           pad_mode == ST_PAD_ZERO, relu ? 1 : 0, (hipStream_t)stream, &rc))
      return rc;
  }
  ConvParams p;
  p.x = x; p.gate = nullptr; p.wp = w_packed; p.bias = bias; p.y = y;
  p.n = n; p.cin = cin; p.hs = hin; p.ws = win; p.cout = cout;
  p.up = upsample2x ? 1 : 0;
  p.H = hin << p.up; p.W = win << p.up;
  p.Ho = p.H; p.Wo = p.W; p.ldo = p.W; p.shift = 0; p.pool = pool;
  p.cout_pad = st::cdiv(cout, CPAD) * CPAD;
  p.nchunks = st::cdiv(cin, KC);
  p.pad_zero = pad_mode == ST_PAD_ZERO;
  p.relu = relu ? 1 : 0;
  return dispatch_conv<false>(p, (hipStream_t)stream);
}

extern "C" int st_conv3x3_fwd(const float* x, const float* w_packed, const float* bias, float* y,
                              int n, int cin, int hin, int win, int cout, int upsample2x,
                              int pad_mode, int relu, st_stream_t stream) {
  return conv3x3_fwd_impl(x, w_packed, bias, y, n, cin, hin, win, cout, upsample2x, pad_mode, relu, 0, stream);
}

extern "C" size_t st_conv3x3_up2_packed_elems(int cout, int cin) {
  if (cout <= 0 || cin <= 0) return 0;
  const size_t cin_pad = size_t(st::cdiv(cin, KC)) * KC;
  const size_t cout_pad = size_t(st::cdiv(cout, CPAD)) * CPAD;
  return cin_pad * 16 * cout_pad;
}

extern "C" int st_conv3x3_pack_weights_up2(const float* w_oihw, float* w_packed, int cout, int cin,
                                           st_stream_t stream) {
  ST_REQUIRE(w_oihw && w_packed, "st_conv3x3_pack_weights_up2: null pointer");
  ST_REQUIRE(cout > 0 && cin > 0, "st_conv3x3_pack_weights_up2: bad dims cout=%d cin=%d", cout, cin);
  const int cout_pad = st::cdiv(cout, CPAD) * CPAD;
  const long long total = (long long)st_conv3x3_up2_packed_elems(cout, cin);
  const int blocks = (int)std::min<long long>(st::cdivll(total, 256), 4096);
  hipLaunchKernelGGL(pack_weights_up2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_oihw, w_packed,
                     cout, cin, cout_pad, total);
  return st::check_launch("st_conv3x3_pack_weights_up2");
}

extern "C" int st_conv3x3_up2_fwd(const float* x, const float* w_packed_up2, const float* bias, float* y, int n,
                                  int cin, int hin, int win, int cout, int relu, st_stream_t stream) {
  ST_REQUIRE(x && w_packed_up2 && y, "st_conv3x3_up2_fwd: null pointer");
  ST_REQUIRE(n > 0 && cin > 0 && cout > 0 && hin > 0 && win > 0,
             "st_conv3x3_up2_fwd: bad dims n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, hin, win);
  ST_REQUIRE((long long)hin * win * 4 < (1LL << 31), "st_conv3x3_up2_fwd: plane too large");
  ConvParams p;
  p.x = x; p.gate = nullptr; p.wp = w_packed_up2; p.bias = bias; p.y = y;
  p.n = n; p.cin = cin; p.hs = hin; p.ws = win; p.cout = cout;
  p.up = 0; p.H = hin; p.W = win;                      // the staged halo is the SOURCE tile
  p.Ho = 2 * hin; p.Wo = 2 * win; p.ldo = 2 * win; p.shift = 0; p.pool = 0;
  p.cout_pad = st::cdiv(cout, CPAD) * CPAD;
  p.nchunks = st::cdiv(cin, KC);
  p.pad_zero = 0; p.relu = relu ? 1 : 0;
  const char* ve = getenv("ST_UP2_VARIANT");        // A/B switch: 1 = the 4-row source tile at two blocks per CU
  const int variant = ve ? atoi(ve) : 0;
  if (variant == 1) return launch_conv_sub<2, 4, 2>(p, (hipStream_t)stream);   // 64 ch x (4 x 32) source pixels
  return launch_conv_sub<2, 2, 3>(p, (hipStream_t)stream);                     // 64 ch x (2 x 32) source pixels
}

extern "C" int st_conv3x3_pool_fwd(const float* x, const float* w_packed, const float* bias, float* ypool,
                                   int n, int cin, int hin, int win, int cout, int pad_mode, int relu,
                                   st_stream_t stream) {
  return conv3x3_fwd_impl(x, w_packed, bias, ypool, n, cin, hin, win, cout, 0, pad_mode, relu, 1, stream);
}

extern "C" size_t st_conv3x3_dgrad_workspace_bytes(int n, int cin, int h, int w, int pad_mode) {
  if (pad_mode != ST_PAD_REPLICATE) return 0;
  return size_t(n) * cin * (h + 2) * st::padded_row(w) * sizeof(float);
}

extern "C" int st_conv3x3_dgrad(const float* dy, const float* y_gate, const float* w_packed_t,
                                float* dx, float* workspace, const float* dx_gate, int n, int cout,
                                int h, int w, int cin, int upsample2x, int pad_mode,
                                st_stream_t stream) {
  ST_REQUIRE(dy && w_packed_t && dx, "st_conv3x3_dgrad: null pointer");
  ST_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0,
             "st_conv3x3_dgrad: bad dims n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, h, w);
  ST_REQUIRE(pad_mode == ST_PAD_REPLICATE || pad_mode == ST_PAD_ZERO,
             "st_conv3x3_dgrad: bad pad_mode %d", pad_mode);
  ST_REQUIRE(!upsample2x || (h % 2 == 0 && w % 2 == 0),
             "st_conv3x3_dgrad: upsampled layer needs even h, w (got %dx%d)", h, w);
  ST_REQUIRE(!(upsample2x && pad_mode == ST_PAD_ZERO),
             "st_conv3x3_dgrad: upsample2x is only supported with replicate padding");
  ST_REQUIRE((long long)(h + 2) * (w + 2) < (1LL << 31), "st_conv3x3_dgrad: plane too large");
  const bool rep = pad_mode == ST_PAD_REPLICATE;
  ST_REQUIRE(!rep || workspace, "st_conv3x3_dgrad: replicate mode needs a workspace of "
             "st_conv3x3_dgrad_workspace_bytes()");
  ST_REQUIRE(rep || !dx_gate, "st_conv3x3_dgrad: dx_gate is only supported with replicate padding");
  hipStream_t s = (hipStream_t)stream;
  // The data gradient is itself a 3x3 correlation of (gated) dy with the transposed, rotated
  // filter, zero padded.  With edge-replicate forward padding it is evaluated on the padded
  // (h+2) x (w+2) domain and folded back onto the edges (and 2x2-summed for an upsample).
  ConvParams p;
  p.x = dy; p.gate = y_gate; p.wp = w_packed_t; p.bias = nullptr;
  p.y = rep ? workspace : dx;
  p.n = n; p.cin = cout; p.hs = h; p.ws = w; p.cout = cin;
  p.up = 0; p.H = h; p.W = w;
  p.Ho = rep ? h + 2 : h; p.Wo = rep ? w + 2 : w; p.shift = rep ? 1 : 0;
  p.ldo = rep ? st::padded_row(w) : w; p.pool = 0;
  p.cout_pad = st::cdiv(cin, CPAD) * CPAD;
  p.nchunks = st::cdiv(cout, KC);
  p.pad_zero = 1; p.relu = 0;
  int rc = y_gate ? dispatch_conv<true>(p, s) : dispatch_conv<false>(p, s);
  if (rc != ST_OK || !rep) return rc;
  return st::launch_dgrad_fold(workspace, dx, dx_gate, n, cin, h, w, upsample2x ? 1 : 0, s);
}
