// Border ring of the decoder chain's data gradient in bf16 mode (generator.py:85-130 backward; the adjoint of
// SymmetricPadding2D((1,1,1,1)) = edge replicate, generator.py:19-39, and of the nearest x2 upsample).
//
// The main launch (conv3x3_bf16_kernel, EPI = 1 / 2) evaluates the data gradient as a zero-padded conv on the
// unpadded grid and finishes every INTERIOR source pixel in its epilogue.  A source pixel on the image border also
// receives what the forward's replicated ring pixels collected: rows -1 and H, columns -1 and W of the padded domain.
// Two small launches redo exactly those pixels, bit-identical to the padded-domain GEMM + dgrad_fold8_kernel path:
//
//   dgrad_band_gemm_kernel   padded-domain values P[py][px] = sum_{chunk, tap} Wt[tap] * dy0[py + ky - 1][px + kx - 1]
//                            (dy0 = dy, zero outside the image) on the BAND: the ring and the rows / columns of the
//                            image next to it (1 deep, 2 under the upsample) -- a gather GEMM with the operands read
//                            straight from global memory, the same v_mfma_f32_32x32x16_bf16 sequence per pixel
//                            (chunks outermost, taps 0..8) as the full GEMM, so every value is the same fp32 number;
//   dgrad_band_fold_kernel   per border source pixel: the sum over its padded positions in dgrad_fold8_kernel's order,
//                            the input-activation gate, and the hand-over forms (G8, pixel-major bf16, fp32, bias
//                            partial sums), written OVER what the main launch left at the border.
//
// Band size is 2 (R+1) (W + 2) + 2 (R+1) (H - 2R) pixels per image (R = 1, or 2 under the upsample): ~0.8 % of a
// 512 x 512 image, 6 % at 64 x 64 -- against a whole padded fp32 image written and re-read before.
#include <algorithm>

#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 16;        // channels per K chunk, as in csrc/conv3x3_bf16.hip
constexpr int CPAD = 128;     // packed weights: output channels padded to a multiple of this
constexpr int NTB = 2;        // 32-pixel band tiles per wave (they share the weight operand)

struct Band {
  int H, W, R;                // conv-input dims (after the upsample), band depth inside the image
  __host__ __device__ int rows_block() const { return (R + 1) * (W + 2); }          // top or bottom block
  __host__ __device__ int cols_block() const { return H > 2 * R ? (R + 1) * (H - 2 * R) : 0; }   // left or right block
  __host__ __device__ int size() const { return 2 * rows_block() + 2 * cols_block(); }
  // padded coordinates (py in [-1, H], px in [-1, W]) of a band pixel -> its index; the caller only asks for
  // pixels that are in the band
  __host__ __device__ int index(int py, int px) const {
    if (py < R) return (py + 1) * (W + 2) + (px + 1);
    if (py >= H - R) return rows_block() + (py - (H - R)) * (W + 2) + (px + 1);
    if (px < R) return 2 * rows_block() + (py - R) * (R + 1) + (px + 1);
    return 2 * rows_block() + cols_block() + (py - R) * (R + 1) + (px - (W - R));
  }
  __host__ __device__ void coords(int idx, int& py, int& px) const {
    const int rb = rows_block(), cb = cols_block();
    if (idx < rb) { py = idx / (W + 2) - 1; px = idx % (W + 2) - 1; return; }
    idx -= rb;
    if (idx < rb) { py = idx / (W + 2) + (H - R); px = idx % (W + 2) - 1; return; }
    idx -= rb;
    if (idx < cb) { py = idx / (R + 1) + R; px = idx % (R + 1) - 1; return; }
    idx -= cb;
    py = idx / (R + 1) + R;
    px = idx % (R + 1) + (W - R);
  }
};

struct BandParams {
  const void* dy;             // G8 [n][k/8][H][W][8] bf16, or fp32 NCHW [n][k][H][W]
  const u32x4* wp;            // packed transposed/flipped bf16 weights [chunk][tap][half][c_pad][8]
  float* scratch;             // [n][c][band]
  const float* gate_f32;      // fp32 NCHW [n][c][hs][ws] or null
  const u32x4* gate_g8;       // G8 [n][c/8][hs][ws][8] or null
  float* dx;                  // optional outputs, as in dgrad_fold8_kernel
  u32x4* g8;
  unsigned short* g16;
  float* bias;                // [n * blocks per image][c]
  int n, k, c, c_pad, nchunks, hs, ws, up;
  Band band;
};

// One wave (= one block: many small blocks spread over the CUs) = NTB band tiles (32 pixels each) x 64 output
// channels of one image.  Operands come straight from global memory / L2, one 16-B load per lane and MFMA operand; the
// 9 x (2 + NTB) loads of chunk c + 1 are all requested before the 9 x 2 x NTB MFMAs of chunk c run, so a chunk costs
// one memory round trip, not nine.
template <bool DYG8>
__global__ __launch_bounds__(64) void dgrad_band_gemm_kernel(BandParams p) {
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.y * 64, b = blockIdx.z;
  const Band bd = p.band;
  const int H = bd.H, W = bd.W, bsz = bd.size();
  const size_t plane = size_t(H) * W;
  // per tile and tap: element offset of the source pixel inside one channel(-group) plane, or -1 (zero padding)
  int soff[NTB][9];
#pragma unroll
  for (int t = 0; t < NTB; ++t) {
    const int idx0 = (blockIdx.x * NTB + t) * 32 + li;
    const int idx = idx0 < bsz ? idx0 : bsz - 1;
    int py, px;
    bd.coords(idx, py, px);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int sy = py + tap / 3 - 1, sx = px + tap % 3 - 1;
      soff[t][tap] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? sy * W + sx : -1;
    }
  }
  f32x16 acc[2][NTB];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int t = 0; t < NTB; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][t][r] = 0.f;

  u32x4 an[9][2], bn[9][NTB];
  auto load = [&](int chunk) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        an[tap][mt] = p.wp[(size_t(chunk * 9 + tap) * 2 + lh) * p.c_pad + m0 + mt * 32 + li];
#pragma unroll
      for (int t = 0; t < NTB; ++t) {
        const int so = soff[t][tap];
        u32x4 bv = {0u, 0u, 0u, 0u};
        if (DYG8) {
          if (so >= 0) bv = reinterpret_cast<const u32x4*>(p.dy)[(size_t(b) * (p.k / 8) + 2 * chunk + lh) * plane + so];
        } else {
          bf16x8 h;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int ch = chunk * KC + lh * 8 + j;
            const float v = (so >= 0 && ch < p.k) ? reinterpret_cast<const float*>(p.dy)[(size_t(b) * p.k + ch) * plane + so] : 0.f;
            h[j] = (__bf16)v;          // round-to-nearest-even, as the main kernel's staging
          }
          bv = __builtin_bit_cast(u32x4, h);
        }
        bn[tap][t] = bv;
      }
    }
  };
  load(0);
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    u32x4 a[9][2], bb[9][NTB];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) a[tap][mt] = an[tap][mt];
#pragma unroll
      for (int t = 0; t < NTB; ++t) bb[tap][t] = bn[tap][t];
    }
    if (chunk + 1 < p.nchunks) load(chunk + 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)          // chunks outermost, taps 0..8: the full GEMM's order per pixel
#pragma unroll
      for (int t = 0; t < NTB; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[tap][mt]),
                                                               __builtin_bit_cast(bf16x8, bb[tap][t]), acc[mt][t], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < NTB; ++t) {
    const int idx = (blockIdx.x * NTB + t) * 32 + li;
    if (idx >= bsz) continue;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < p.c) p.scratch[(size_t(b) * p.c + co) * bsz + idx] = acc[mt][t][r];
      }
  }
}

// One thread = one border source pixel x 8 channels.  Border pixels of an [hs][ws] plane in the order: top row,
// bottom row, left column (rows 1 .. hs-2), right column.
template <bool UP>
__global__ __launch_bounds__(256) void dgrad_band_fold_kernel(BandParams p) {
  __shared__ float red[4][8];
  const Band bd = p.band;
  const int H = bd.H, W = bd.W, bsz = bd.size();
  const int hs = p.hs, ws = p.ws;
  const int nborder = 2 * ws + 2 * (hs - 2);
  const int cg = blockIdx.y, b = blockIdx.z, cgs = p.c >> 3;
  const int e = blockIdx.x * 256 + threadIdx.x;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  const bool live = e < nborder;
  int i = 0, jx = 0;
  if (live) {
    if (e < ws) { i = 0; jx = e; }
    else if (e < 2 * ws) { i = hs - 1; jx = e - ws; }
    else if (e < 2 * ws + hs - 2) { i = e - 2 * ws + 1; jx = 0; }
    else { i = e - 2 * ws - (hs - 2) + 1; jx = ws - 1; }
    int u0 = UP ? 2 * i : i, u1 = UP ? 2 * i + 1 : i;
    if (u0 == 0) u0 = -1;
    if (u1 == H - 1) u1 = H;
    const bool left = jx == 0, right = jx == ws - 1;
    const int v0 = UP ? 2 * jx : jx;
    const float* src = p.scratch + (size_t(b) * p.c + cg * 8) * bsz;
    for (int u = u0; u <= u1; ++u) {
      const int i0 = bd.index(u, v0);
      const int i1 = UP ? bd.index(u, v0 + 1) : 0;
      const int ie = left ? bd.index(u, -1) : (right ? bd.index(u, W) : -1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* sj = src + size_t(j) * bsz;
        // dgrad_fold8_kernel's order: (first [+ second]) + edge, rows accumulated top to bottom
        float t = UP ? sj[i0] + sj[i1] : sj[i0];
        t = t + (ie >= 0 ? sj[ie] : 0.f);
        s[j] += t;
      }
    }
    const size_t pix = size_t(i) * ws + jx;
    const size_t plane_s = size_t(hs) * ws;
    if (p.gate_g8) {
      const u32x4 g = p.gate_g8[(size_t(b) * cgs + cg) * plane_s + pix];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned hb = (j & 1) ? (g[j >> 1] >> 16) : (g[j >> 1] & 0xffffu);
        if (!((hb - 1u) < 0x7f80u)) s[j] = 0.f;
      }
    } else if (p.gate_f32) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        s[j] = p.gate_f32[(size_t(b) * p.c + cg * 8 + j) * plane_s + pix] > 0.f ? s[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const size_t o = (size_t(b) * p.c + cg * 8 + j) * plane_s + pix;
      if (p.dx) p.dx[o] = s[j];
      if (p.g16) p.g16[o] = __builtin_bit_cast(unsigned short, (__bf16)s[j]);
    }
    if (p.g8) {
      bf16x8 h;
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = (__bf16)s[j];
      p.g8[(size_t(b) * cgs + cg) * plane_s + pix] = __builtin_bit_cast(u32x4, h);
    }
  }
  if (p.bias) {      // fixed order: lanes (shuffles), then the 4 waves; dead threads add zeros
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = st::wave_sum(s[j]);
      if ((threadIdx.x & 63) == 0) red[wave][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
      const int j = threadIdx.x;
      p.bias[(size_t(b) * gridDim.x + blockIdx.x) * p.c + cg * 8 + j] = (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]);
    }
  }
}

Band make_band(int h, int w, int up) {
  Band bd;
  bd.H = h; bd.W = w; bd.R = up ? 2 : 1;
  return bd;
}

}  // namespace

namespace st {

size_t dgrad_band_scratch_bytes(int n, int c, int h, int w, int up) {
  return size_t(n) * c * make_band(h, w, up).size() * sizeof(float);
}

// bias partial-sum rows per image written by the fold kernel
int dgrad_band_slabs(int h, int w, int up) {
  const int hs = h >> up, ws = w >> up;
  return cdiv(2 * ws + 2 * (hs - 2), 256);
}

int launch_dgrad_band(const void* dy, int dy_g8, const void* w_packed_t, float* scratch, const float* gate_f32,
                      const void* gate_g8, float* dx_f32, void* g8, void* g16, float* bias, int n, int k, int c, int h,
                      int w, int up, hipStream_t s) {
  ST_REQUIRE(c % 8 == 0 && (h >> up) >= 4 && (w >> up) >= 4, "dgrad band: c=%d, %dx%d", c, h, w);
  ST_REQUIRE(!dy_g8 || k % KC == 0, "dgrad band: a G8 dy needs k %% 16 == 0 (got %d)", k);
  ST_REQUIRE(n <= 65535 && c / 8 <= 65535, "dgrad band: n=%d, c=%d", n, c);
  BandParams p;
  p.dy = dy; p.wp = (const u32x4*)w_packed_t; p.scratch = scratch; p.gate_f32 = gate_f32;
  p.gate_g8 = (const u32x4*)gate_g8; p.dx = dx_f32; p.g8 = (u32x4*)g8; p.g16 = (unsigned short*)g16; p.bias = bias;
  p.n = n; p.k = k; p.c = c; p.c_pad = cdiv(c, CPAD) * CPAD; p.nchunks = cdiv(k, KC);
  p.hs = h >> up; p.ws = w >> up; p.up = up;
  p.band = make_band(h, w, up);
  const int bsz = p.band.size();
  dim3 ggrid(cdiv(bsz, 32 * NTB), cdiv(c, 64), n);
  if (dy_g8)
    hipLaunchKernelGGL(dgrad_band_gemm_kernel<true>, ggrid, dim3(64), 0, s, p);
  else
    hipLaunchKernelGGL(dgrad_band_gemm_kernel<false>, ggrid, dim3(64), 0, s, p);
  int rc = check_launch("st_conv3x3_dgrad_bf16_g8_chain(band gemm)");
  if (rc != ST_OK) return rc;
  dim3 fgrid(dgrad_band_slabs(h, w, up), c / 8, n);
  if (up)
    hipLaunchKernelGGL(dgrad_band_fold_kernel<true>, fgrid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(dgrad_band_fold_kernel<false>, fgrid, dim3(256), 0, s, p);
  return check_launch("st_conv3x3_dgrad_bf16_g8_chain(band fold)");
}

}  // namespace st
