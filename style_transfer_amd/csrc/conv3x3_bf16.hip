// 3x3 stride-1 convolution with bf16 MFMA operands and fp32 accumulation for gfx950 (MI355X):
// the bf16 compute mode of the encoder / decoder / loss-VGG convolutions (BASELINE config 4,
// "bf16 compute / fp32 master weights").
//
// Same layer semantics as csrc/conv3x3.hip (SymmetricPadding2D((1,1,1,1)) -> Conv2d(3x3) -> ReLU of
// generator.py:47-76, 89-117, fused nearest x2 upsample of generator.py:92,105,112, zero-padded
// loss-VGG convs of trainer.py:165-199) and the same boundary: activations are fp32 NCHW in HBM,
// weights are the fp32 OIHW master copy.  Only the MFMA operands are bf16: activations are rounded
// (round-to-nearest-even, v_cvt_pk_bf16_f32) while they are staged into LDS, weights when they are
// packed; products accumulate in fp32 (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate).
//
// GEMM view (per image):  D[cout][pixel] = sum_k W[cout][k] * X[k][pixel],  k = (tap, cin).
//   * M (accumulator registers) = output channels, N (lanes) = 32 consecutive x of one output row:
//     an accumulator register is an NCHW row segment, stores are full 128-B lines.
//   * K step of one MFMA = 16 input channels of one tap.  A lane supplies 8 consecutive channels
//     of its row/column (lanes 0-31: channels 0-7, lanes 32-63: channels 8-15), so both LDS images
//     keep 8 channels innermost with the channel half as the OUTER index: X halo
//     [2 halves][(TH+2)*34 positions][8 ch] and W slab [9 taps][2 halves][BM cout][8 ch].  Every
//     operand is ONE ds_read_b128 whose two 32-lane halves each read 512 contiguous bytes: the
//     16-lane service groups of ds_read_b128 never mix the halves, so the read is conflict-free
//     (a [position][16 ch] image, lanes at 32-B stride, is a 2-way conflict in every group and
//     made the LDS array, not the matrix pipe, the limiter: 0.75 reads per MFMA x 8 cycles x 4 SIMDs).
//   * Staging transposes NCHW -> channel-innermost for free: a thread loads the same halo position
//     of 8 channels (8 coalesced dword loads across the wave), packs them to 8 bf16 and issues one
//     ds_write_b128.  Edge replicate / zero pad / upsample source index are resolved in the
//     per-thread offsets computed once; zero padding is a bit mask at the LDS store.
//   * The weight slab is copied global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPRs, no
//     ds_write) into a double buffer: the packed global layout already is the LDS image.
//   * Both LDS images are double-buffered, one workgroup barrier per chunk; the two waves of a
//     SIMD run staging and MFMA phases in opposite order (see the schedule comment in the kernel).
//
// SPLIT = true is the "bf16x3" mode: fp32-grade results from the bf16 matrix pipe.  Every fp32
// operand v is split exactly into hi = bf16(v) and lo = bf16(v - hi) (the pair is the same 32 bits
// as v and carries 16 significand bits), both halves get their own LDS image, and each product is
// evaluated as hi*hi + hi*lo + lo*hi in the fp32 accumulator (the dropped lo*lo term and the split
// residuals are < 2^-16 relative).  Three MFMAs per product at 16x the fp32 MFMA rate.
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "common.h"

namespace {

constexpr int KC = 16;       // input channels per K chunk (one MFMA K step per tap)
constexpr int TW = 32;       // pixel-tile width == one MFMA N tile
constexpr int HC = TW + 2;   // halo row length
constexpr int CPAD = 128;    // packed weights: cout padded to a multiple of this
constexpr int NTHREADS = 512;   // 8 waves: 2 per SIMD, one block per CU

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct ConvParams {
  const float* x;
  const float* gate;       // GATE kernels: same shape as x; x is read as 0 where gate <= 0
  const u32x4* wp;         // packed bf16 weights, 16-B units: [chunk][part][tap][half][cout_pad]
  const float* bias;
  float* y;
  int n, cin, hs, ws;      // source tensor dims
  int cout, H, W;          // conv-input spatial dims, after the optional upsample
  int Ho, Wo;              // output dims: (H, W) forward; (H+2, W+2) for the padded data gradient
  int pool;                // 1: the epilogue writes maxpool2x2(ceil_mode) of the activation instead of it
  // G8 pooled output only (st_conv3x3_pool_fwd_bf16_g8_codes; even Ho, Wo): one byte per pooled pixel and channel,
  // [n][c/8][Ho/2][Wo/2][8] -- bits 0-1 = the window position (row-major scan order) that holds the maximum under the
  // max-pool backward's tie rule (the first maximum wins), bit 2 = that maximum is > 0 (the ReLU gate)
  unsigned* pcodes = nullptr;
  void* y2 = nullptr;      // fp32-output G8-input form only: the 2 x 2 max pool of the activation, in G8 (see the epilogue)
  int ldo;                 // output row stride in floats (Wo, or Wo rounded up to 4 in the padded workspace)
  int shift;               // output pixel (oy, ox) is centred on input pixel (oy - shift, ox - shift)
  int cout_pad, nchunks;
  int up, pad_zero, relu;
  int tiles_x, tiles_y, ctiles, xcd_swizzle;
  int ntiles;              // PERSIST kernels: total (pixel tile, channel tile) pairs, walked with stride gridDim.x
  int vec_store;           // ldo % 4 == 0 and y 16-B aligned: 16-B-per-lane epilogue stores
  int ablate;              // debug only (ST_BF16_ABLATE): 1 skip MFMA, 2 skip X staging, 4 skip W DMA
  // G8 output only (st_conv3x3_dgrad_bf16_g8): a G8 tensor of the OUTPUT's shape; the result is stored as 0 where
  // it is <= 0 -- the ReLU gate of the layer that produced this kernel's output tensor's forward twin
  const u32x4* ogate = nullptr;
  // EPI kernels only (decoder chain data gradient with the pad / upsample adjoint and the hand-over forms in the
  // epilogue, st_conv3x3_dgrad_bf16_g8_chain): y = the G8 result [n][cout/8][hs][ws][8] on the SOURCE grid
  // (hs, ws = Ho, Wo or their halves under the upsample); the gate is the layer's input activation on that grid
  const u32x4* egate_g8 = nullptr;   // G8, or
  const float* egate_f32 = nullptr;  // fp32 NCHW, or neither
  unsigned short* g16 = nullptr;     // optional: the same result as pixel-major bf16 [n][cout][hs][ws]
  float* dxf = nullptr;              // optional: ... as fp32 NCHW [n][cout][hs][ws] (the bottom of the chain); y may be
                                     // null then
  float* bslabs = nullptr;           // optional: [n * tiles per image * WP][cout] per-wave sums of the result over its
                                     // INTERIOR source pixels (the border ring is summed by the band kernels)
};

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// lane exchanges / reductions on the DPP path (no LDS traffic)
__device__ __forceinline__ float dpp_xor1(float v) {      // quad_perm [1, 0, 3, 2]
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {      // quad_perm [2, 3, 0, 1]
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
// sum over the 32 lanes of each wave half, fixed order: inclusive scan inside the 16-lane rows (row_shr 1, 2, 4, 8,
// zeros shifted in), then lane 15 of row 0 / 2 is added to row 1 / 3 (row_bcast:15): lanes 31 and 63 hold the totals
__device__ __forceinline__ float half_wave_sum_to_lane31(float v) {
#define ST_DPP_ADD(ctrl_, rmask_)                                                                          \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl_, rmask_, 0xF, true))
  ST_DPP_ADD(0x111, 0xF);
  ST_DPP_ADD(0x112, 0xF);
  ST_DPP_ADD(0x114, 0xF);
  ST_DPP_ADD(0x118, 0xF);
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
#undef ST_DPP_ADD
  return v;
}

// 8 bf16 lanes of `v` kept where the same lane of `g` is > 0 (bit test: 0x0001 .. 0x7f80 = positive, incl. +inf)
__device__ __forceinline__ u32x4 gate_g8(u32x4 v, u32x4 g) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned lo = g[k] & 0xffffu, hi = g[k] >> 16;
    const unsigned m = ((lo - 1u) < 0x7f80u ? 0x0000ffffu : 0u) | ((hi - 1u) < 0x7f80u ? 0xffff0000u : 0u);
    v[k] &= m;
  }
  return v;
}

template <int MT, int NT, int WM, int WP, bool SPLIT = false>
struct Cfg {
  static constexpr int BM = 32 * MT * WM;
  static constexpr int TH = NT * WP;
  static constexpr int HR = TH + 2;
  static constexpr int PLANE = HR * HC;          // halo positions
  static constexpr int NITEM = 2 * PLANE;        // (position, channel half) staging items
  static constexpr int NP = (NITEM + NTHREADS - 1) / NTHREADS;
  static constexpr int NPART = SPLIT ? 2 : 1;    // hi (and lo) images
  static constexpr int W_UNITS = NPART * 9 * BM * 2;   // 16-B units of the weight slab: [part][tap][half][BM]
  static constexpr int NWI = W_UNITS / 64;        // 1-KiB LDS-DMA wave instructions per weight slab
  static constexpr int NW = (NWI + 7) / 8;        // ... per wave
  // 16-B units of the halo image: [part][half][position], rounded up to whole 1-KiB LDS-DMA pieces
  static constexpr int X_UNITS = (NPART * NITEM + 63) / 64 * 64;
  static constexpr int NXI = (NPART * NITEM + 63) / 64;   // XG: 1-KiB LDS-DMA wave instructions per halo image
  static constexpr int NXW = (NXI + 7) / 8;       // ... per wave
  static constexpr size_t LDS_BYTES = size_t(2 * W_UNITS + 2 * X_UNITS) * 16;   // both double-buffered
};

__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
  bf16x8 b;
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[j];
  return __builtin_bit_cast(u32x4, b);
}

// TAIL: cin is not a multiple of 16 (the 3-channel layers): channel indices are clamped per load
// and the missing channels masked to zero.
//
// XG / YG: the input / output activation is in the channel-group layout "G8" -- bf16 [n][c/8][h][w][8], the form
// the layers of one network hand to each other in bf16 mode (st_conv3x3_fwd_bf16_g8).  A G8 pixel of one channel
// group is the 16-B unit of the LDS halo image, so an XG kernel stages its input by LDS-DMA like the weights (no
// registers, no conversion, half the bytes of the fp32 form); a YG kernel rounds its result to bf16 (the same
// round-to-nearest-even the next layer's staging would apply to an fp32 result: the chain is bit-identical to the
// fp32-activation form) and writes 16 B per pixel and channel group.  With SPLIT (bf16x3) the layout is the hi/lo
// pair form bf16 [n][c/8][2][h][w][8] (part 0 = bf16(v), part 1 = bf16(v - hi): the same split the staging of an
// fp32 activation performs, done once by the producer): as many bytes as fp32, but both images arrive by LDS-DMA.
//
// EPI != 0 (with YG): the data gradient of a replicate-padded decoder layer inside the G8 chain, computed on the
// UNPADDED grid as a zero-padded conv (exactly the padded-domain GEMM's accumulators at the same pixels) with the
// rest of the old padded workspace -> dgrad_fold8_kernel round trip in the epilogue: [(EPI & 3) == 2: the 2 x 2 sum of
// the nearest-upsample adjoint, row pairs in one wave, column pairs by DPP] -> gate by the input activation (G8, or
// fp32 NCHW when EPI & 4) -> the G8 form, the pixel-major bf16 form and per-wave bias partial sums.  Source pixels on
// the image border also receive the padded ring's gradient: those are recomputed and OVERWRITTEN by the band kernels
// (csrc/dgrad_band.hip) that run after this launch, and are left out of the bias sums here.
template <int MT, int NT, int WM, int WP, bool GATE, bool TAIL, bool SPLIT, bool PIPE, bool XG = false,
          bool YG = false, int EPI = 0>
__global__ __launch_bounds__(NTHREADS, 2) void conv3x3_bf16_kernel(ConvParams p) {
  using C = Cfg<MT, NT, WM, WP, SPLIT>;
  static_assert(WM * WP == 8, "eight waves per block");
  static_assert(EPI == 0 || (YG && !SPLIT && !GATE && NT % 4 == 0), "chain epilogue: G8 output, plain bf16");
  static_assert(!XG || (!GATE && !TAIL), "G8 input: plain forward form only");
  extern __shared__ __attribute__((aligned(16))) u32x4 smem16[];
  u32x4* Ws = smem16;                     // 2 x [part][tap][half][BM]   (filled by LDS-DMA)
  u32x4* Xs = smem16 + 2 * C::W_UNITS;    // 2 x [part][half][position]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave / WP, wp = wave % WP;

  // Block -> (pixel tile, channel tile).  Blocks are dealt round-robin over the 8 XCDs (each with
  // its own L2), so ids that differ by 8 share an XCD: the channel tiles of one pixel tile get
  // consecutive slots on ONE XCD and the 2..4 reads of the same activation tile hit its L2.
  // PERSIST (G8 in and out, bf16): the grid is one block per CU and a block walks tiles id, id + gridDim.x, ...;
  // the LDS-DMA of the next tile's first chunk is issued before the epilogue of the current one, so neither the
  // first-chunk latency nor the block launch sits between two tiles' MFMA loops.
  constexpr bool PERSIST = XG && YG && !SPLIT;
  int b, m0, x0, y0;
#define ST_DECODE_TILE(id_, b_, m0_, x0_, y0_)                                             \
  {                                                                                        \
    int bt_, ct_;                                                                          \
    if (p.xcd_swizzle) {                                                                   \
      const int k_ = (id_) >> 3;                                                           \
      ct_ = k_ % p.ctiles;                                                                 \
      bt_ = (k_ / p.ctiles) * 8 + ((id_) & 7);                                             \
    } else {                                                                               \
      ct_ = (id_) % p.ctiles;                                                              \
      bt_ = (id_) / p.ctiles;                                                              \
    }                                                                                      \
    const int tx_ = bt_ % p.tiles_x;                                                       \
    bt_ /= p.tiles_x;                                                                      \
    const int ty_ = bt_ % p.tiles_y;                                                       \
    b_ = bt_ / p.tiles_y;                                                                  \
    m0_ = ct_ * C::BM;                                                                     \
    x0_ = tx_ * TW;                                                                        \
    y0_ = ty_ * C::TH;                                                                     \
  }
  int tile = blockIdx.x;
  ST_DECODE_TILE(tile, b, m0, x0, y0);

  const size_t plane_src = size_t(p.hs) * p.ws;
  const float* xb = p.x + size_t(b) * p.cin * plane_src;
  const float* gb = GATE ? p.gate + size_t(b) * p.cin * plane_src : nullptr;

  // ---- per-thread staging items: (halo position, channel half) ---------------------------
  unsigned soff[C::NP];    // element offset of the item's first channel inside the chunk
  unsigned smask[C::NP];   // all-ones: keep; 0: zero padding
  int half_[C::NP];
  // XG: wave w copies 1-KiB pieces u = i*8 + w of the halo image; lane -> 16-B unit (half, position) u*64 + lane,
  // source = the (clamped, optionally upsampled) pixel of channel group 2*chunk + half
  unsigned xoff[XG ? C::NXW : 1];
  unsigned xoob = 0;     // bit i: piece i of this lane is a zero-padding position (pad_zero; patched after the DMA)
  const u32x4* xg = nullptr;
#define ST_XOFF(b_, x0_, y0_, xg_, xoff_, xoob_)                                           \
  {                                                                                        \
    xg_ = reinterpret_cast<const u32x4*>(p.x) + size_t(b_) * (p.cin / 8) * C::NPART * plane_src; \
    xoob_ = 0;                                                                             \
    _Pragma("unroll") for (int i = 0; i < C::NXW; ++i) {                                   \
      int idx = min((i * 8 + wave) * 64 + lane, C::NPART * C::NITEM - 1);                  \
      const int part = idx / C::NITEM;                                                     \
      idx -= part * C::NITEM;                                                              \
      const int half = idx / C::PLANE, pidx = idx - half * C::PLANE;                       \
      const int r = pidx / HC, cc = pidx - r * HC;                                         \
      const int gy = (y0_) + r - 1 - p.shift, gx = (x0_) + cc - 1 - p.shift;               \
      if (p.pad_zero && i * 8 + wave < C::NXI && (gy < 0 || gy >= p.H || gx < 0 || gx >= p.W)) \
        xoob_ |= 1u << i;                                                                  \
      int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);                    \
      if (p.up) { cy >>= 1; cx >>= 1; }                                                    \
      xoff_[i] = unsigned(half * C::NPART + part) * unsigned(plane_src) + unsigned(cy * p.ws + cx); \
    }                                                                                      \
  }
  if constexpr (XG) ST_XOFF(b, x0, y0, xg, xoff, xoob);
#pragma unroll
  for (int i = 0; i < (XG ? 0 : C::NP); ++i) {
    const int idx = min(tid + i * NTHREADS, C::NITEM - 1);
    const int half = idx / C::PLANE, pidx = idx - half * C::PLANE;
    const int r = pidx / HC, cc = pidx - r * HC;
    const int gy = y0 + r - 1 - p.shift, gx = x0 + cc - 1 - p.shift;
    const bool inb = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    int cy = min(max(gy, 0), p.H - 1), cx = min(max(gx, 0), p.W - 1);
    if (p.up) { cy >>= 1; cx >>= 1; }
    half_[i] = half;
    soff[i] = unsigned(cy * p.ws + cx) + (TAIL ? 0u : unsigned(half * 8) * unsigned(plane_src));
    smask[i] = (inb || !p.pad_zero) ? 0xffffffffu : 0u;
  }

  float xr[C::NP][8];
  float gr[GATE ? C::NP : 1][8];
  constexpr int NLOADS = C::NP * 8 * (GATE ? 2 : 1);   // dword loads per thread per chunk
  // weight slab: wave w copies 1-KiB pieces u = i*8 + w with global_load_lds_dwordx4 (no VGPR
  // destination); the packed global layout already is the LDS image
  unsigned woff[C::NW];
#pragma unroll
  for (int i = 0; i < C::NW; ++i) {
    const int e = min(i * 8 + wave, C::NWI - 1) * 64 + lane;
    const int th = e / C::BM, r = e - th * C::BM;      // th = (part * 9 + tap) * 2 + half
    woff[i] = unsigned(th) * unsigned(p.cout_pad) + unsigned(r);
  }

  // ---- staging macros ----------------------------------------------------------------------
  // weights of chunk_ -> W buffer (chunk_ & 1) by LDS-DMA
#define ST_ISSUE_W(chunk_) ST_ISSUE_W_AT(chunk_, m0)
#define ST_ISSUE_W_AT(chunk_, m0_)                                                         \
  {                                                                                        \
    const u32x4* wsrc = p.wp + size_t(chunk_) * (18 * C::NPART) * p.cout_pad + (m0_);      \
    u32x4* wdst = Ws + ((chunk_) & 1) * C::W_UNITS;                                        \
    _Pragma("unroll") for (int i = 0; i < C::NW; ++i) {                                    \
      const int u = i * 8 + wave;                                                          \
      if (C::NWI % 8 == 0 || u < C::NWI)                                                   \
        __builtin_amdgcn_global_load_lds(                                                  \
            (const __attribute__((address_space(1))) void*)(wsrc + woff[i]),               \
            (__attribute__((address_space(3))) void*)(wdst + u * 64), 16, 0, 0);           \
    }                                                                                      \
  }

  // activations of chunk_ -> registers (exactly NLOADS dword loads per thread)
#define ST_LOAD_X(chunk_)                                                                  \
  {                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                        \
      if (!TAIL) {                                                                         \
        const float* src = xb + size_t((chunk_) * KC + j) * plane_src;                     \
        _Pragma("unroll") for (int i = 0; i < C::NP; ++i) xr[i][j] = src[soff[i]];         \
        if (GATE) {                                                                        \
          const float* gsrc = gb + size_t((chunk_) * KC + j) * plane_src;                  \
          _Pragma("unroll") for (int i = 0; i < C::NP; ++i) gr[i][j] = gsrc[soff[i]];      \
        }                                                                                  \
      } else {                                                                             \
        _Pragma("unroll") for (int i = 0; i < C::NP; ++i) {                                \
          const int cg = (chunk_) * KC + half_[i] * 8 + j;                                 \
          /* channels past the end are not requested at all (3-channel layers: 13 of 16) */ \
          xr[i][j] = cg < p.cin ? xb[size_t(cg) * plane_src + soff[i]] : 0.f;              \
          if (GATE) gr[i][j] = cg < p.cin ? gb[size_t(cg) * plane_src + soff[i]] : 0.f;    \
        }                                                                                  \
      }                                                                                    \
    }                                                                                      \
  }

  // registers -> bf16 -> X buffer (chunk_ & 1)
#define ST_STORE_X(chunk_)                                                                 \
  {                                                                                        \
    u32x4* xdst = Xs + ((chunk_) & 1) * C::X_UNITS;                                        \
    _Pragma("unroll") for (int i = 0; i < C::NP; ++i) {                                    \
      const int idx = tid + i * NTHREADS;                                                  \
      float v[8];                                                                          \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                      \
        unsigned m = smask[i];                                                             \
        if (TAIL) m &= ((chunk_) * KC + half_[i] * 8 + j) < p.cin ? 0xffffffffu : 0u;      \
        if (GATE) m &= gr[i][j] > 0.f ? 0xffffffffu : 0u;                                  \
        v[j] = __uint_as_float(__float_as_uint(xr[i][j]) & m);                             \
      }                                                                                    \
      if (C::NITEM % NTHREADS == 0 || idx < C::NITEM) {                                    \
        const u32x4 hi = pack8(v);                                                         \
        xdst[idx] = hi;               /* idx = half * PLANE + position */                  \
        if (SPLIT) {                                                                       \
          const bf16x8 hb = __builtin_bit_cast(bf16x8, hi);                                \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) v[j] -= float(hb[j]);              \
          xdst[C::NITEM + idx] = pack8(v);                                                 \
        }                                                                                  \
      }                                                                                    \
    }                                                                                      \
  }

  // 9 taps x MT x NT MFMAs on chunk_'s buffers
#define ST_MFMA(chunk_)                                                                    \
  {                                                                                        \
    const u32x4* aP = aPtr + ((chunk_) & 1) * C::W_UNITS;                                  \
    const u32x4* bP = bPtr + ((chunk_) & 1) * C::X_UNITS;                                  \
    _Pragma("unroll") for (int tap = 0; tap < 9; ++tap) {                                  \
      const int dy = tap / 3, dx = tap % 3;                                                \
      bf16x8 a[MT], bb[NT];                                                                \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                    \
          a[mt] = __builtin_bit_cast(bf16x8, aP[tap * 2 * C::BM + mt * 32]);               \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                    \
          bb[nt] = __builtin_bit_cast(bf16x8, bP[(nt + dy) * HC + dx]);                    \
      if (SPLIT) {                                                                         \
        bf16x8 al[MT], bl[NT];                                                             \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                  \
            al[mt] = __builtin_bit_cast(bf16x8, aP[(9 + tap) * 2 * C::BM + mt * 32]);      \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                  \
            bl[nt] = __builtin_bit_cast(bf16x8, bP[C::NITEM + (nt + dy) * HC + dx]);       \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                  \
          _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bb[nt],          \
                                                                  acc[mt][nt], 0, 0, 0);   \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                  \
          _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                \
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], bl[nt],           \
                                                                  acc[mt][nt], 0, 0, 0);   \
      }                                                                                    \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                    \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                  \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mt], bb[nt], acc[mt][nt], \
                                                                0, 0, 0);                  \
    }                                                                                      \
  }

  // ---- PIPE schedule building blocks: one staging item / one tap at a time ------------------
#define ST_LOAD_ITEM(chunk_, i_)                                                           \
  {                                                                                        \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                        \
      if (!TAIL) {                                                                         \
        xr[i_][j] = (xb + size_t((chunk_) * KC + j) * plane_src)[soff[i_]];                \
        if (GATE) gr[i_][j] = (gb + size_t((chunk_) * KC + j) * plane_src)[soff[i_]];      \
      } else {                                                                             \
        const int cg = min((chunk_) * KC + half_[i_] * 8 + j, p.cin - 1);                  \
        xr[i_][j] = xb[size_t(cg) * plane_src + soff[i_]];                                 \
        if (GATE) gr[i_][j] = gb[size_t(cg) * plane_src + soff[i_]];                       \
      }                                                                                    \
    }                                                                                      \
  }
#define ST_STORE_ITEM(chunk_, i_)                                                          \
  {                                                                                        \
    u32x4* xdst = Xs + ((chunk_) & 1) * C::X_UNITS;                                        \
    const int idx = tid + (i_) * NTHREADS;                                                 \
    float v[8];                                                                            \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                        \
      unsigned m = smask[i_];                                                              \
      if (TAIL) m &= ((chunk_) * KC + half_[i_] * 8 + j) < p.cin ? 0xffffffffu : 0u;       \
      if (GATE) m &= gr[i_][j] > 0.f ? 0xffffffffu : 0u;                                   \
      v[j] = __uint_as_float(__float_as_uint(xr[i_][j]) & m);                              \
    }                                                                                      \
    if (C::NITEM % NTHREADS == 0 || idx < C::NITEM) {                                      \
      const u32x4 hi = pack8(v);                                                           \
      xdst[idx] = hi;                                                                      \
      if (SPLIT) {                                                                         \
        const bf16x8 hb = __builtin_bit_cast(bf16x8, hi);                                  \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) v[j] -= float(hb[j]);                \
        xdst[C::NITEM + idx] = pack8(v);                                                   \
      }                                                                                    \
    }                                                                                      \
  }
  // operands of one tap -> register set s_
#define ST_READ_TAP(chunk_, tap_, s_)                                                      \
  {                                                                                        \
    const u32x4* aP = aPtr + ((chunk_) & 1) * C::W_UNITS;                                  \
    const u32x4* bP = bPtr + ((chunk_) & 1) * C::X_UNITS;                                  \
    const int dy = (tap_) / 3, dx = (tap_) % 3;                                            \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                      \
        pa[s_][mt] = __builtin_bit_cast(bf16x8, aP[(tap_) * 2 * C::BM + mt * 32]);         \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                      \
        pb[s_][nt] = __builtin_bit_cast(bf16x8, bP[(nt + dy) * HC + dx]);                  \
    if (SPLIT) {                                                                           \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                    \
          pal[s_][mt] = __builtin_bit_cast(bf16x8, aP[(9 + (tap_)) * 2 * C::BM + mt * 32]); \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                    \
          pbl[s_][nt] = __builtin_bit_cast(bf16x8, bP[C::NITEM + (nt + dy) * HC + dx]);    \
    }                                                                                      \
  }
#define ST_MFMA_TAP(s_)                                                                    \
  {                                                                                        \
    if (SPLIT) {                                                                           \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                    \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                  \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pal[s_][mt], pb[s_][nt],   \
                                                                acc[mt][nt], 0, 0, 0);     \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                    \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                  \
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s_][mt], pbl[s_][nt],   \
                                                                acc[mt][nt], 0, 0, 0);     \
    }                                                                                      \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                      \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                    \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s_][mt], pb[s_][nt],      \
                                                              acc[mt][nt], 0, 0, 0);       \
  }
  // One chunk, PIPE schedule, as ONE basic block (MORE1_/MORE2_ are compile-time): the LDS reads of
  // tap t+1 are issued before the MFMAs of tap t (counted lgkmcnt waits instead of a drain per tap),
  // and the staging of chunk c+1 / c+2 is cut into 2*NP slices -- [convert+store item i of X(c+1)],
  // [8 global loads of item i of X(c+2)] -- dealt one per tap, so every wave keeps the matrix pipe
  // fed while its own VALU / VMEM / LDS-store work runs in the shadow of the 32-cycle MFMAs.
  // sched_barrier(0) pins the stage boundaries.
#define ST_SLICE(chunk_, k_, MORE1_, MORE2_)                                               \
  {                                                                                        \
    if ((k_) < 2 * C::NP) {                                                                \
      if (((k_) & 1) == 0) { if (MORE1_) ST_STORE_ITEM((chunk_) + 1, (k_) / 2); }          \
      else                 { if (MORE2_) ST_LOAD_ITEM((chunk_) + 2, (k_) / 2); }           \
    }                                                                                      \
  }
#define ST_CHUNK_PIPE(chunk_, MORE1_, MORE2_)                                              \
  {                                                                                        \
    ST_READ_TAP(chunk_, 0, 0);                                                             \
    if (MORE1_) ST_ISSUE_W((chunk_) + 1);                                                  \
    asm volatile("" ::: "memory");                                                         \
    ST_SLICE(chunk_, 0, MORE1_, MORE2_);                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                        \
      if (t < 8) ST_READ_TAP(chunk_, t + 1, (t + 1) & 1);                                  \
      ST_MFMA_TAP(t & 1);                                                                  \
      ST_SLICE(chunk_, t + 1, MORE1_, MORE2_);                                             \
      __builtin_amdgcn_sched_barrier(0);                                                   \
    }                                                                                      \
    if (MORE2_) { ST_BARRIER_KEEP_LOADS(); } else { ST_BARRIER_DRAIN(); }                  \
  }

  // XG: halo image of chunk_ -> X buffer (chunk_ & 1) by LDS-DMA
#define ST_ISSUE_XG(chunk_) ST_ISSUE_XG_AT(chunk_, xg, xoff)
#define ST_ISSUE_XG_AT(chunk_, xg_, xoff_)                                                 \
  {                                                                                        \
    const u32x4* xsrc = (xg_) + size_t(chunk_) * (2 * C::NPART) * plane_src;               \
    u32x4* xdst = Xs + ((chunk_) & 1) * C::X_UNITS;                                        \
    _Pragma("unroll") for (int i = 0; i < C::NXW; ++i) {                                   \
      const int u = i * 8 + wave;                                                          \
      if (C::NXI % 8 == 0 || u < C::NXI)                                                   \
        __builtin_amdgcn_global_load_lds(                                                  \
            (const __attribute__((address_space(1))) void*)(xsrc + (xoff_)[i]),            \
            (__attribute__((address_space(3))) void*)(xdst + u * 64), 16, 0, 0);           \
    }                                                                                      \
  }
  // One chunk of an XG kernel: both operands of chunk c+1 arrive by LDS-DMA while the 9 taps of chunk c run with
  // the operand reads of tap t+1 issued before the MFMAs of tap t.
  //
  // Plain bf16 (ROWS): the pixel operand of tap (dy, dx) and row nt is halo row nt + dy at column shift dx -- the taps
  // of one dx re-read the same NT + 2 halo rows three times.  They are read ONCE per chunk instead, into a register
  // window pbr[dx][row] (taps of dy = 0 read rows 0 .. NT-1 of their dx, taps of dy > 0 only the new row NT-1 + dy):
  // 18 instead of 36 pixel-operand reads per chunk, a third of the LDS read traffic of the loop (the LDS pipe was
  // ~75 % busy under the MFMAs and the leading wave of a SIMD still waited on it).  Same operands in the same MFMA
  // order: bit-identical.
#define ST_READ_TAP_R(chunk_, tap_, s_)                                                    \
  {                                                                                        \
    const u32x4* aP = aPtr + ((chunk_) & 1) * C::W_UNITS;                                  \
    const u32x4* bP = bPtr + ((chunk_) & 1) * C::X_UNITS;                                  \
    const int dy = (tap_) / 3, dx = (tap_) % 3;                                            \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                      \
        pa[s_][mt] = __builtin_bit_cast(bf16x8, aP[(tap_) * 2 * C::BM + mt * 32]);         \
    if (dy == 0) {                                                                         \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                    \
          pbr[dx][nt] = __builtin_bit_cast(bf16x8, bP[nt * HC + dx]);                      \
    } else {                                                                               \
      pbr[dx][NT - 1 + dy] = __builtin_bit_cast(bf16x8, bP[(NT - 1 + dy) * HC + dx]);      \
    }                                                                                      \
  }
#define ST_MFMA_TAP_R(tap_, s_)                                                            \
  {                                                                                        \
    const int dy = (tap_) / 3, dx = (tap_) % 3;                                            \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                      \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                    \
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[s_][mt], pbr[dx][nt + dy], \
                                                              acc[mt][nt], 0, 0, 0);       \
  }
#define ST_CHUNK_DMA_R(chunk_, MORE1_)                                                     \
  {                                                                                        \
    ST_READ_TAP_R(chunk_, 0, 0);                                                           \
    if (MORE1_) {                                                                          \
      if (!(p.ablate & 4)) ST_ISSUE_W((chunk_) + 1);                                       \
      if (!(p.ablate & 2)) ST_ISSUE_XG((chunk_) + 1);                                      \
    }                                                                                      \
    asm volatile("" ::: "memory");                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                        \
      if (t < 8) ST_READ_TAP_R(chunk_, t + 1, (t + 1) & 1);                                \
      ST_MFMA_TAP_R(t, t & 1);                                                             \
      __builtin_amdgcn_sched_barrier(0);                                                   \
    }                                                                                      \
    if (MORE1_) { ST_BARRIER_DRAIN_XG(((chunk_) + 1) & 1); } else { ST_BARRIER_DRAIN(); }  \
  }
#define ST_CHUNK_PIPE_R(chunk_, MORE1_, MORE2_)                                            \
  {                                                                                        \
    ST_READ_TAP_R(chunk_, 0, 0);                                                           \
    if (MORE1_) ST_ISSUE_W((chunk_) + 1);                                                  \
    asm volatile("" ::: "memory");                                                         \
    ST_SLICE(chunk_, 0, MORE1_, MORE2_);                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                        \
      if (t < 8) ST_READ_TAP_R(chunk_, t + 1, (t + 1) & 1);                                \
      ST_MFMA_TAP_R(t, t & 1);                                                             \
      ST_SLICE(chunk_, t + 1, MORE1_, MORE2_);                                             \
      __builtin_amdgcn_sched_barrier(0);                                                   \
    }                                                                                      \
    if (MORE2_) { ST_BARRIER_KEEP_LOADS(); } else { ST_BARRIER_DRAIN(); }                  \
  }
#define ST_CHUNK_DMA(chunk_, MORE1_)                                                       \
  {                                                                                        \
    ST_READ_TAP(chunk_, 0, 0);                                                             \
    if (MORE1_) {                                                                          \
      if (!(p.ablate & 4)) ST_ISSUE_W((chunk_) + 1);                                       \
      if (!(p.ablate & 2)) ST_ISSUE_XG((chunk_) + 1);                                      \
    }                                                                                      \
    asm volatile("" ::: "memory");                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                        \
      if (t < 8) ST_READ_TAP(chunk_, t + 1, (t + 1) & 1);                                  \
      ST_MFMA_TAP(t & 1);                                                                  \
      __builtin_amdgcn_sched_barrier(0);                                                   \
    }                                                                                      \
    if (MORE1_) { ST_BARRIER_DRAIN_XG(((chunk_) + 1) & 1); } else { ST_BARRIER_DRAIN(); }  \
  }

  // Workgroup barrier that does NOT drain the register loads issued after the LDS-DMA pieces:
  // vmcnt is in order, so allowing the NLOADS youngest operations to stay in flight still
  // retires every older LDS-DMA (the counter saturates at 63: a larger NLOADS only waits for a
  // few of the oldest register loads as well); lgkmcnt(0) retires this wave's ds_writes.  The "memory" clobber
  // keeps the compiler from moving LDS accesses across it.
#define ST_BARRIER_KEEP_LOADS()                                                            \
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(NLOADS < 63 ? NLOADS : 63) : "memory")
#define ST_BARRIER_DRAIN() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
  // XG with zero padding: LDS-DMA cannot mask, so it copies the clamped pixel and the lane that copied a padding
  // position overwrites its own 16-B unit with zeros once its DMA has landed (vmcnt(0)), before the barrier that
  // publishes the buffer.  Interior tiles have no such lane (xoob == 0).
#define ST_BARRIER_DRAIN_XG(buf_)                                                          \
  {                                                                                        \
    if (xoob != 0) {                                                                       \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     \
      u32x4* xfix = Xs + (buf_) * C::X_UNITS;                                              \
      _Pragma("unroll") for (int i = 0; i < C::NXW; ++i) {                                 \
        const int u = i * 8 + wave;                                                        \
        if ((xoob >> i) & 1u) xfix[u * 64 + lane] = u32x4{0u, 0u, 0u, 0u};                 \
      }                                                                                    \
    }                                                                                      \
    ST_BARRIER_DRAIN();                                                                    \
  }

  const u32x4* aPtr = Ws + lh * C::BM + wm * (MT * 32) + li;
  const u32x4* bPtr = Xs + lh * C::PLANE + (wp * NT) * HC + li;
  bool primed = false;   // PERSIST: the first chunk of this tile was issued under the previous tile's epilogue
  for (;;) {             // one pass unless PERSIST
  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // Phase schedule (PIPE = false; X and W double-buffered in LDS, ONE barrier per chunk).  In iteration c every wave
  // issues the LDS-DMA of W(c+1), converts/stores X(c+1) from the registers loaded one iteration
  // earlier, issues the register loads of X(c+2) and runs the MFMAs of chunk c.  The two waves
  // that share a SIMD do these in opposite order (waves 0-3 stage first, waves 4-7 compute
  // first), so one wave's staging runs under its partner's MFMAs instead of both idling the
  // matrix pipe together.
  const bool lead = __builtin_amdgcn_readfirstlane(wave) < 4;
  const int nch = p.nchunks;
  if (!PERSIST || !primed) ST_ISSUE_W(0);
  if constexpr (XG) {
    if (!PERSIST || !primed) ST_ISSUE_XG(0);
  } else {
    ST_LOAD_X(0);
    ST_STORE_X(0);
    if (nch > 1) ST_LOAD_X(1);
  }
  if constexpr (XG) { ST_BARRIER_DRAIN_XG(0); } else { ST_BARRIER_DRAIN(); }
  if constexpr (XG) {
    if constexpr (!SPLIT) {
      bf16x8 pa[2][MT], pbr[3][NT + 2];
      int chunk = 0;
      for (; chunk + 1 < nch; ++chunk) ST_CHUNK_DMA_R(chunk, true);
      ST_CHUNK_DMA_R(chunk, false);
    } else {
      bf16x8 pa[2][MT], pb[2][NT], pal[SPLIT ? 2 : 1][MT], pbl[SPLIT ? 2 : 1][NT];
      int chunk = 0;
      for (; chunk + 1 < nch; ++chunk) ST_CHUNK_DMA(chunk, true);
      ST_CHUNK_DMA(chunk, false);
    }
  } else if constexpr (PIPE && !SPLIT) {
    bf16x8 pa[2][MT], pbr[3][NT + 2];
    int chunk = 0;
    for (; chunk + 2 < nch; ++chunk) ST_CHUNK_PIPE_R(chunk, true, true);
    if (chunk + 1 < nch) { ST_CHUNK_PIPE_R(chunk, true, false); ++chunk; }
    ST_CHUNK_PIPE_R(chunk, false, false);
  } else if constexpr (PIPE) {
    bf16x8 pa[2][MT], pb[2][NT], pal[SPLIT ? 2 : 1][MT], pbl[SPLIT ? 2 : 1][NT];
    int chunk = 0;
    for (; chunk + 2 < nch; ++chunk) ST_CHUNK_PIPE(chunk, true, true);
    if (chunk + 1 < nch) { ST_CHUNK_PIPE(chunk, true, false); ++chunk; }
    ST_CHUNK_PIPE(chunk, false, false);
  } else
  for (int chunk = 0; chunk < nch; ++chunk) {
    const bool more1 = chunk + 1 < nch, more2 = chunk + 2 < nch;
    if (more1 && !(p.ablate & 4)) ST_ISSUE_W(chunk + 1);
    asm volatile("" ::: "memory");   // keep the LDS-DMA ahead of the register loads (vmcnt order)
    if (lead && !(p.ablate & 2)) {
      if (more1) ST_STORE_X(chunk + 1);
      if (more2) ST_LOAD_X(chunk + 2);
    }
    if (!(p.ablate & 1)) ST_MFMA(chunk);
    if (!lead && !(p.ablate & 2)) {
      if (more1) ST_STORE_X(chunk + 1);
      if (more2) ST_LOAD_X(chunk + 2);
    }
    if (more2) { ST_BARRIER_KEEP_LOADS(); } else { ST_BARRIER_DRAIN(); }
  }

  // PERSIST: decode the next tile and start the LDS-DMA of its first chunk (buffer 0 of both images is free after
  // the last chunk's barrier; the epilogue's patches live behind the staging buffers)
  int nb = 0, nm0 = 0, nx0 = 0, ny0 = 0;
  unsigned nxoff[XG ? C::NXW : 1];
  unsigned nxoob = 0;
  const u32x4* nxg = nullptr;
  bool has_next = false;
  if constexpr (PERSIST) {
    has_next = tile + int(gridDim.x) < p.ntiles;
    if (has_next) {
      ST_DECODE_TILE(tile + int(gridDim.x), nb, nm0, nx0, ny0);
      ST_XOFF(nb, nx0, ny0, nxg, nxoff, nxoob);
      ST_ISSUE_W_AT(0, nm0);
      ST_ISSUE_XG_AT(0, nxg, nxoff);
    }
  }

  // ---- epilogue: bias + ReLU, NCHW store ---------------------------------------------------
  // An accumulator register is 32 consecutive x of one channel row (one dword per lane).  Stored
  // as-is that is 16 dword-store instructions per 32x32 tile, and a block with the CU to itself
  // becomes store-ISSUE-bound.  Each wave therefore transposes a tile through a private 4 KB LDS
  // patch (the staging buffers are free after the last barrier) and writes 16 B per lane: 8 lanes
  // cover one 128-B row segment, a wave instruction writes 8 channel rows.
  if constexpr (EPI != 0) {
    constexpr bool UPF = (EPI & 3) == 2;
    constexpr bool GF32 = (EPI & 4) != 0;                   // the gate is fp32 NCHW (else G8, or none)
    constexpr bool F32OUT = (EPI & 8) != 0;                 // the bottom of the chain: fp32 NCHW result (p.dxf)
    unsigned* patch = reinterpret_cast<unsigned*>(smem16 + (PERSIST ? 2 * C::W_UNITS + 2 * C::X_UNITS : 0) +
                                                  wave * 128);
    const u32x4* patch16 = reinterpret_cast<const u32x4*>(patch);
    u32x4* yg = reinterpret_cast<u32x4*>(p.y);
    const int cgroups = p.cout >> 3;
    const int hs = UPF ? p.Ho >> 1 : p.Ho, ws = UPF ? p.Wo >> 1 : p.Wo;
    const size_t plane_s = size_t(hs) * ws;
    const int sx = UPF ? ((x0 + li) >> 1) : x0 + li;
    const int sxc = min(sx, ws - 1);                        // clamped twin for the (unconditional) gate loads
    const bool lane_on = !UPF || !(li & 1);                 // under the upsample the even lanes carry the 16 source pixels
    const bool odd = UPF ? (li & 2) != 0 : (li & 1) != 0;   // second pixel of a pixel pair (pixel-major bf16 stores)
    constexpr int NROW = UPF ? NT / 2 : NT;                // output rows of this wave
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int cbase = m0 + wm * (MT * 32) + mt * 32;
      float bsum[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bsum[r] = 0.f;
      // The gate values of several rows are requested before the first one is used (with one request per row in flight
      // the epilogue of the wide, shallow layers -- 64 channels at 512 x 512, K = 27 or 576 -- ran at the latency of
      // its loads).  The loads are UNCONDITIONAL on clamped addresses (rows / columns / channels past the end read a
      // valid neighbour whose result is never stored): a predicated load puts a wait in front of the next one.
      constexpr int GROWS = GF32 ? 2 : NROW;               // fp32 gate: two rows at a time (32 registers)
      u32x2 gq[GF32 ? 1 : NROW][4];
      float gf[GF32 ? GROWS : 1][16];
      auto load_gates = [&](int first) {
        if constexpr (GF32) {
#pragma unroll
          for (int i = 0; i < GROWS; ++i) {
            const int row = min(first + i, NROW - 1);
            const int sy = min(UPF ? (y0 + wp * NT + 2 * row) >> 1 : y0 + wp * NT + row, hs - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int co = min(cbase + (r & 3) + 8 * (r >> 2) + 4 * lh, p.cout - 1);
              gf[i][r] = __builtin_nontemporal_load(p.egate_f32 + (size_t(b) * p.cout + co) * plane_s + size_t(sy) * ws + sxc);
            }
          }
        } else if (p.egate_g8) {
#pragma unroll
          for (int i = 0; i < NROW; ++i) {
            const int sy = min(UPF ? (y0 + wp * NT + 2 * i) >> 1 : y0 + wp * NT + i, hs - 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int cg = min((cbase >> 3) + q, cgroups - 1);
              gq[i][q] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned*>(
                             p.egate_g8 + (size_t(b) * cgroups + cg) * plane_s + size_t(sy) * ws + sxc) + lh * 2);
            }
          }
        }
      };
      if constexpr (!GF32) load_gates(0);
#pragma unroll
      for (int nt = 0; nt < NT; nt += (UPF ? 2 : 1)) {
        const int irow = UPF ? nt / 2 : nt;
        if constexpr (GF32) {
          if (irow % GROWS == 0 && !(p.ablate & 16)) load_gates(irow);
        }
        const int yy = y0 + wp * NT + nt;
        const int sy = UPF ? yy >> 1 : yy;
        const bool ok = lane_on && sy < hs && sx < ws;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (UPF) {      // (p[2i][2j] + p[2i][2j+1]) + (p[2i+1][2j] + p[2i+1][2j+1]): dgrad_fold8_kernel's order
            float a = acc[mt][nt][r], c2 = acc[mt][nt + 1 < NT ? nt + 1 : nt][r];
            a += dpp_xor1(a);
            c2 += dpp_xor1(c2);
            v[r] = a + c2;
          } else {
            v[r] = acc[mt][nt][r] + 0.f;
          }
        }
        if constexpr (GF32) {
          if (!(p.ablate & 16)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = gf[irow % GROWS][r] > 0.f ? v[r] : 0.f;
          }
        } else if (p.egate_g8) {      // keep where the bf16 input activation is > 0 (0x0001 .. 0x7f80)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned hb = (j & 1) ? (gq[irow][q][j >> 1] >> 16) : (gq[irow][q][j >> 1] & 0xffffu);
              if (!((hb - 1u) < 0x7f80u)) v[q * 4 + j] = 0.f;
            }
        }
        const bool inner = ok && sy > 0 && sy < hs - 1 && sx > 0 && sx < ws - 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) bsum[r] += inner ? v[r] : 0.f;
        // G8: [pixel][4 groups][8 ch] through the wave's 2 KB patch, 16 B per lane and pixel
        const int slot = UPF ? (li >> 1) : li;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bf16x4 h;
#pragma unroll
          for (int j = 0; j < 4; ++j) h[j] = (__bf16)v[q * 4 + j];
          if (lane_on) {
            const u32x2 hv = __builtin_bit_cast(u32x2, h);
            patch[(slot * 4 + q) * 4 + lh * 2 + 0] = hv[0];
            patch[(slot * 4 + q) * 4 + lh * 2 + 1] = hv[1];
          }
        }
        asm volatile("" ::: "memory");
        if (!UPF) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int q = k * 2 + lh;
            const int cg = (cbase >> 3) + q;
            const u32x4 v4 = patch16[li * 4 + q];
            if (yg && cg < cgroups && sy < hs && sx < ws && !(p.ablate & 32))
              yg[(size_t(b) * cgroups + cg) * plane_s + size_t(sy) * ws + sx] = v4;
          }
        } else {
          const int q = lane >> 4, px = lane & 15;
          const int cg = (cbase >> 3) + q;
          const u32x4 v4 = patch16[px * 4 + q];
          if (yg && cg < cgroups && sy < hs && (x0 >> 1) + px < ws)
            yg[(size_t(b) * cgroups + cg) * plane_s + size_t(sy) * ws + (x0 >> 1) + px] = v4;
        }
        asm volatile("" ::: "memory");
        if constexpr (F32OUT) {      // fp32 NCHW: an accumulator register is a row segment of one channel
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (ok && co < p.cout) p.dxf[(size_t(b) * p.cout + co) * plane_s + size_t(sy) * ws + sx] = v[r];
          }
        }
        // pixel-major bf16: lanes exchange one register of each pair so that a lane stores two neighbouring pixels
        // of ONE channel (4 B); 16 lanes cover a 64-B row segment (32 B under the upsample)
        if (p.g16) {
          const int px0 = UPF ? (sx & ~1) : (x0 + (li & ~1));
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float send = odd ? v[2 * k] : v[2 * k + 1];
            const float recv = UPF ? dpp_xor2(send) : dpp_xor1(send);
            bf16x2 h2;
            h2[0] = (__bf16)(odd ? recv : v[2 * k]);
            h2[1] = (__bf16)(odd ? v[2 * k + 1] : recv);
            const int r = 2 * k + (odd ? 1 : 0);
            const int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (ok && co < p.cout)
              *reinterpret_cast<unsigned*>(p.g16 + (size_t(b) * p.cout + co) * plane_s + size_t(sy) * ws + px0) =
                  __builtin_bit_cast(unsigned, h2);
          }
        }
      }
      if (p.bslabs) {
        const int tile_img = (y0 / C::TH) * p.tiles_x + x0 / TW;
        float* brow = p.bslabs + ((size_t(b) * p.tiles_x * p.tiles_y + tile_img) * WP + wp) * p.cout;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float t = half_wave_sum_to_lane31(bsum[r]);
          const int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (li == 31 && co < p.cout) brow[co] = t;
        }
      }
    }
  } else if constexpr (YG) {
    // G8 bf16 output: a wave turns each 32 ch x 32 px accumulator tile into [32 px][4 groups][8 ch] bf16 through a
    // private 2 KB LDS patch per image (lanes l and l+32 hold the two halves of a pixel's 8-channel group) and
    // stores 16 B per lane: 32 lanes cover 512 contiguous bytes of one channel group's row.  SPLIT: a second
    // patch / store for the lo image.
    constexpr int NPT = C::NPART;
    // plain bf16: TWO patches per wave, used by alternate rows -- row r + 1 is packed and written while row r's patch
    // reads are still in flight (with one patch and a fence after every row the epilogue ran write -> wait -> read ->
    // wait -> store per row: 1600 cycles per row and wave, 30 % of a 64-channel tile's time by in-kernel stamps)
    constexpr int NPATCH = SPLIT ? 1 : 2;
    unsigned* patch0 = reinterpret_cast<unsigned*>(smem16 + (PERSIST ? 2 * C::W_UNITS + 2 * C::X_UNITS : 0) +
                                                   wave * (128 * NPT * NPATCH));
    u32x4* yg = reinterpret_cast<u32x4*>(p.y);
    const int cgroups = p.cout >> 3;
    const bool pool = p.pool != 0;
    const int Hy = pool ? (p.Ho + 1) >> 1 : p.Ho, Wy = pool ? (p.Wo + 1) >> 1 : p.Wo;
    const size_t plane_y = size_t(Hy) * Wy;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int cbase = m0 + wm * (MT * 32) + mt * 32;
      float bv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
        bv[r] = (p.bias && co < p.cout) ? p.bias[co] : 0.f;
      }
      const short relu_floor = p.relu ? short(0) : short(-32768);      // max with -32768 is the identity
      const s16x4 relu_floor4 = {relu_floor, relu_floor, relu_floor, relu_floor};
      // one output row (POOL_ = false) or one pooled row from the row pair (nt_, nt_ + 1) (POOL_ = true; fused
      // MaxPool2d((2,2),(2,2),ceil_mode=True): rounding is monotonic, so max-then-round equals the pool of the
      // rounded activation; even lanes hold the 16 pooled pixels)
#define ST_G8_ROW(POOL_, nt_)                                                                         \
  {                                                                                                   \
    const int yy = y0 + wp * NT + (nt_);                                                              \
    const int slot = (POOL_) ? (li >> 1) : li;                                                        \
    unsigned* patch = patch0 + (NPATCH == 2 ? (((nt_) >> ((POOL_) ? 1 : 0)) & 1) * (128 * 4) : 0);    \
    const u32x4* patch16 = reinterpret_cast<const u32x4*>(patch);                                     \
    u32x4 gnxt[2] = {gcur[0], gcur[1]};                                                               \
    if (!(POOL_) && !SPLIT && p.ogate && (nt_) + 1 < NT) ST_G8_GATES(gnxt, (nt_) + 1);                \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                   \
      bf16x4 h, l;                                                                                    \
      unsigned cw = 0;                                                                                \
      if constexpr (!(POOL_) && !SPLIT) {                                                             \
        /* packed form: two-wide bias add, one conversion per pair, and the ReLU AFTER the rounding as a packed */ \
        /* signed 16-bit max with 0 (a bf16 with its sign bit set is a negative int16; -0 -> +0 like fmaxf;   */ \
        /* rounding to nearest even is monotonic and keeps 0, so it commutes with the ReLU: same bits) -- the */ \
        /* epilogue of the wide, shallow layers is bound by its VALU instruction count                        */ \
        _Pragma("unroll") for (int j2 = 0; j2 < 2; ++j2) {                                            \
          f32x2 v2 = {acc[mt][nt_][q * 4 + 2 * j2], acc[mt][nt_][q * 4 + 2 * j2 + 1]};                \
          const f32x2 b2 = {bv[q * 4 + 2 * j2], bv[q * 4 + 2 * j2 + 1]};                              \
          v2 += b2;                                                                                   \
          h[2 * j2] = (__bf16)v2[0];                                                                  \
          h[2 * j2 + 1] = (__bf16)v2[1];                                                              \
        }                                                                                             \
        s16x4 hi_ = __builtin_bit_cast(s16x4, h);                                                     \
        hi_ = __builtin_elementwise_max(hi_, relu_floor4);                                            \
        h = __builtin_bit_cast(bf16x4, hi_);                                                          \
      } else                                                                                          \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                 \
        float v = acc[mt][nt_][q * 4 + j];                                                            \
        if ((POOL_) && !SPLIT && p.pcodes) {                                                          \
          /* window (a an / c cn) of the even lane, biased first: the comparisons see the activation's own values */ \
          const float a = v + bv[q * 4 + j];                                                          \
          const float c = acc[mt][(nt_) + 1 < NT ? (nt_) + 1 : (nt_)][q * 4 + j] + bv[q * 4 + j];     \
          const float an = __shfl_xor(a, 1, 64), cn = __shfl_xor(c, 1, 64);                           \
          float m = a;                                                                                \
          unsigned arg = 0;                                                                           \
          if (an > m || an != an) { m = an; arg = 1; }                                                \
          if (c > m || c != c) { m = c; arg = 2; }                                                    \
          if (cn > m || cn != cn) { m = cn; arg = 3; }                                                \
          cw |= (arg | (m > 0.f ? 4u : 0u)) << (8 * j);                                               \
          v = m;                                                                                      \
        } else {                                                                                      \
        if (POOL_) {                                                                                  \
          if (yy + 1 < p.Ho) v = fmaxf(v, acc[mt][(nt_) + 1 < NT ? (nt_) + 1 : (nt_)][q * 4 + j]);    \
          const float vn = __shfl_xor(v, 1, 64);                                                      \
          if (x0 + li + 1 < p.Wo) v = fmaxf(v, vn);                                                   \
        }                                                                                             \
        v += bv[q * 4 + j];                                                                           \
        }                                                                                             \
        if (p.relu) v = fmaxf(v, 0.f);                                                                \
        h[j] = (__bf16)v;                                                                             \
        if (SPLIT) l[j] = (__bf16)(v - float(h[j]));                                                  \
      }                                                                                               \
      if ((POOL_) && !SPLIT && p.pcodes && !(li & 1)) {                                               \
        const int cg_ = (cbase >> 3) + q;                                                             \
        if (cg_ < cgroups && yy < p.Ho && x0 + li < p.Wo)                                             \
          p.pcodes[(((size_t(b) * cgroups + cg_) * Hy + (yy >> 1)) * Wy + ((x0 + li) >> 1)) * 2 + lh] = cw; \
      }                                                                                               \
      if (!(POOL_) || !(li & 1)) {                                                                    \
        const u32x2 hv = __builtin_bit_cast(u32x2, h);                                                \
        patch[(slot * 4 + q) * 4 + lh * 2 + 0] = hv[0];                                               \
        patch[(slot * 4 + q) * 4 + lh * 2 + 1] = hv[1];                                               \
        if (SPLIT) {                                                                                  \
          const u32x2 lv = __builtin_bit_cast(u32x2, l);                                              \
          patch[512 + (slot * 4 + q) * 4 + lh * 2 + 0] = lv[0];                                       \
          patch[512 + (slot * 4 + q) * 4 + lh * 2 + 1] = lv[1];                                       \
        }                                                                                             \
      }                                                                                               \
    }                                                                                                 \
    asm volatile("" ::: "memory");                                                                    \
    if (!(POOL_)) {                                                                                   \
      _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                 \
        const int q = k * 2 + lh;                                                                     \
        const int cg = (cbase >> 3) + q, xs = x0 + li;                                                \
        const bool ok = cg < cgroups && yy < p.Ho && xs < p.Wo;                                       \
        const size_t o = (size_t(b) * cgroups + cg) * NPT * plane_y + size_t(yy) * Wy + xs;           \
        _Pragma("unroll") for (int part = 0; part < NPT; ++part) {                                    \
          u32x4 v = patch16[part * 128 + li * 4 + q];                                                 \
          if (!SPLIT && p.ogate) v = gate_g8(v, gcur[k]);                                             \
          if (ok) yg[o + part * plane_y] = v;                                                         \
        }                                                                                             \
      }                                                                                               \
    } else {                                                                                          \
      const int q = lane >> 4, px = lane & 15;                                                        \
      const int cg = (cbase >> 3) + q;                                                                \
      const bool ok = cg < cgroups && yy < p.Ho && x0 + 2 * px < p.Wo;                                \
      const size_t o =                                                                                \
          (size_t(b) * cgroups + cg) * NPT * plane_y + size_t(yy >> 1) * Wy + (x0 >> 1) + px;         \
      _Pragma("unroll") for (int part = 0; part < NPT; ++part) {                                      \
        const u32x4 v = patch16[part * 128 + px * 4 + q];                                             \
        if (ok) yg[o + part * plane_y] = v;                                                           \
      }                                                                                               \
    }                                                                                                 \
    if (NPATCH == 1) asm volatile("" ::: "memory");                                                   \
    gcur[0] = gnxt[0];                                                                                \
    gcur[1] = gnxt[1];                                                                                \
  }
      // the output gate of row nt_ (st_conv3x3_dgrad_bf16_g8), in the layout of the stores: requested one row AHEAD
      // and UNCONDITIONALLY on clamped coordinates (a load inside the `ok` branch put a full memory latency in front
      // of every store: 20 000 of a gated tile's 32 000 epilogue cycles)
#define ST_G8_GATES(dst_, nt_)                                                                        \
  {                                                                                                   \
    const int gy_ = min(y0 + wp * NT + (nt_), p.Ho - 1), gx_ = min(x0 + li, p.Wo - 1);                \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                   \
      const int cg_ = min((cbase >> 3) + k * 2 + lh, cgroups - 1);                                    \
      dst_[k] = p.ogate[(size_t(b) * cgroups + cg_) * plane_y + size_t(gy_) * Wy + gx_];              \
    }                                                                                                 \
  }
      // Plain bf16, no pool: the rows as a two-stage pipeline.  PACK(r): bias / ReLU / rounding, the row's patch
      // written, its two 16-byte patch reads ISSUED; STORE(r - 1): the previous row's reads (in flight under PACK(r)'s
      // ~45 VALU instructions) gated and stored.  In-kernel stamps had every row paying an exposed LDS round trip
      // (2 x ~250 cycles) between its pack and its stores: 4 000 of the epilogue's 12 800 cycles per tile.
#define ST_G8_PACK(nt_)                                                                               \
  {                                                                                                   \
    unsigned* patch = patch0 + ((nt_) & 1) * (128 * 4);                                               \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                   \
      bf16x4 h;                                                                                       \
      _Pragma("unroll") for (int j2 = 0; j2 < 2; ++j2) {                                              \
        f32x2 v2 = {acc[mt][nt_][q * 4 + 2 * j2], acc[mt][nt_][q * 4 + 2 * j2 + 1]};                  \
        const f32x2 b2 = {bv[q * 4 + 2 * j2], bv[q * 4 + 2 * j2 + 1]};                                \
        v2 += b2;                                                                                     \
        h[2 * j2] = (__bf16)v2[0];                                                                    \
        h[2 * j2 + 1] = (__bf16)v2[1];                                                                \
      }                                                                                               \
      s16x4 hi_ = __builtin_bit_cast(s16x4, h);                                                       \
      hi_ = __builtin_elementwise_max(hi_, relu_floor4);                                              \
      const u32x2 hv = __builtin_bit_cast(u32x2, hi_);                                                \
      patch[(li * 4 + q) * 4 + lh * 2 + 0] = hv[0];                                                   \
      patch[(li * 4 + q) * 4 + lh * 2 + 1] = hv[1];                                                   \
    }                                                                                                 \
    asm volatile("" ::: "memory");                                                                    \
    _Pragma("unroll") for (int k = 0; k < 2; ++k)                                                     \
      vq[(nt_) & 1][k] = reinterpret_cast<const u32x4*>(patch)[li * 4 + k * 2 + lh];                  \
  }
#define ST_G8_STORE(nt_)                                                                              \
  {                                                                                                   \
    const int yy = y0 + wp * NT + (nt_);                                                              \
    _Pragma("unroll") for (int k = 0; k < 2; ++k) {                                                   \
      const int cg = (cbase >> 3) + k * 2 + lh, xs = x0 + li;                                         \
      const bool ok = cg < cgroups && yy < p.Ho && xs < p.Wo;                                         \
      const size_t o = (size_t(b) * cgroups + cg) * plane_y + size_t(yy) * Wy + xs;                   \
      u32x4 v = vq[(nt_) & 1][k];                                                                     \
      if (p.ogate) v = gate_g8(v, gq[(nt_) & 1][k]);                                                  \
      if (ok) yg[o] = v;                                                                              \
    }                                                                                                 \
  }
      if constexpr (!SPLIT) {
        if (!pool) {
          u32x4 vq[2][2], gq[2][2];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int k = 0; k < 2; ++k) gq[i][k] = u32x4{0u, 0u, 0u, 0u};
          if (p.ogate) {
            ST_G8_GATES(gq[0], 0);
            if (NT > 1) ST_G8_GATES(gq[1], 1);
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            ST_G8_PACK(nt);
            if (nt > 0) {
              ST_G8_STORE(nt - 1);
              if (p.ogate && nt + 1 < NT) ST_G8_GATES(gq[(nt + 1) & 1], nt + 1);
            }
          }
          ST_G8_STORE(NT - 1);
          asm volatile("" ::: "memory");     // the next channel tile's PACK(0) writes patch 0 again
          continue;
        }
      }
#undef ST_G8_PACK
#undef ST_G8_STORE
      u32x4 gcur[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
      if (!SPLIT && !pool && p.ogate) ST_G8_GATES(gcur, 0);
      if (!pool) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ST_G8_ROW(false, nt)
      } else {
#pragma unroll
        for (int nt = 0; nt + 1 < NT; nt += 2) ST_G8_ROW(true, nt)
      }
#undef ST_G8_ROW
#undef ST_G8_GATES
    }
  } else if (p.pool) {
    // fused MaxPool2d((2,2),(2,2),ceil_mode=True) (generator.py:53,60,73), as in csrc/conv3x3.hip
    static_assert(NT % 2 == 0, "row pairs live in one wave");
    const int Hp = (p.Ho + 1) >> 1, Wp = (p.Wo + 1) >> 1;
    const int xx = x0 + li;
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
          if (yy + 1 < p.Ho) v = fmaxf(v, acc[mt][nt + 1][r]);
          const float vn = __shfl_xor(v, 1, 64);
          if (xx + 1 < p.Wo) v = fmaxf(v, vn);
          v += bv;
          if (p.relu) v = fmaxf(v, 0.f);
          if (!(li & 1) && co < p.cout && yy < p.Ho && xx < p.Wo)
            p.y[((size_t(b) * p.cout + co) * Hp + (yy >> 1)) * Wp + (xx >> 1)] = v;
        }
      }
    }
  } else if (p.vec_store) {
    float* patch = reinterpret_cast<float*>(smem16) + wave * (32 * 32);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int cbase = m0 + wm * (MT * 32) + mt * 32;
      float bv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
        bv[r] = (p.bias && co < p.cout) ? p.bias[co] : 0.f;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int yy = y0 + wp * NT + nt;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[mt][nt][r] + bv[r];
          if (p.relu) v = fmaxf(v, 0.f);
          patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int idx = q * 64 + lane;
          const int ch = idx >> 3, x4 = (idx & 7) * 4;
          const f32x4 v = *reinterpret_cast<const f32x4*>(patch + ch * 32 + x4);
          const int co = cbase + ch, xs = x0 + x4;
          if (co < p.cout && yy < p.Ho && xs < p.Wo)
            *reinterpret_cast<f32x4*>(p.y + ((size_t(b) * p.cout + co) * p.Ho + yy) * p.ldo + xs) = v;
        }
      }
    }
  } else {
    const int xx = x0 + li;
    if (xx < p.Wo) {
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
  // Second output of a loss tap that feeds a pool (st_conv3x3_fwd_bf16_g8_tap_pool): the 2 x 2 max pool of the
  // activation just written, in G8 -- the next conv's input form -- straight from the accumulators (row pairs are two
  // accumulator tiles of this wave, column pairs neighbouring lanes; rounding is monotonic, so max-then-round equals
  // the pool of the rounded activation).  Even Ho, Wo (host check).
  if constexpr (XG && !YG && !SPLIT && EPI == 0 && NT % 2 == 0) {
    if (p.y2) {
      asm volatile("" ::: "memory");         // the fp32 epilogue's patch reads are done (same wave-private region)
      unsigned* pp = reinterpret_cast<unsigned*>(smem16 + wave * 256);
      const u32x4* pp16 = reinterpret_cast<const u32x4*>(pp);
      u32x4* yg2 = reinterpret_cast<u32x4*>(p.y2);
      const int cgroups = p.cout >> 3, Hy = p.Ho >> 1, Wy = p.Wo >> 1;
      const size_t plane_y = size_t(Hy) * Wy;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int cbase = m0 + wm * (MT * 32) + mt * 32;
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
          bv[r] = (p.bias && co < p.cout) ? p.bias[co] : 0.f;
        }
#pragma unroll
        for (int nt = 0; nt + 1 < NT; nt += 2) {
          const int yy = y0 + wp * NT + nt;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bf16x4 h;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float v = fmaxf(acc[mt][nt][q * 4 + j], acc[mt][nt + 1][q * 4 + j]);
              v = fmaxf(v, __shfl_xor(v, 1, 64));
              v += bv[q * 4 + j];
              if (p.relu) v = fmaxf(v, 0.f);
              h[j] = (__bf16)v;
            }
            if (!(li & 1)) {
              const u32x2 hv = __builtin_bit_cast(u32x2, h);
              pp[((li >> 1) * 4 + q) * 4 + lh * 2 + 0] = hv[0];
              pp[((li >> 1) * 4 + q) * 4 + lh * 2 + 1] = hv[1];
            }
          }
          asm volatile("" ::: "memory");
          {
            const int q = lane >> 4, px = lane & 15;
            const int cg = (cbase >> 3) + q;
            const bool ok = cg < cgroups && yy < p.Ho && x0 + 2 * px < p.Wo;
            const size_t o = (size_t(b) * cgroups + cg) * plane_y + size_t(yy >> 1) * Wy + (x0 >> 1) + px;
            const u32x4 v = pp16[px * 4 + q];
            if (ok) yg2[o] = v;
          }
          asm volatile("" ::: "memory");
        }
      }
    }
  }
  if (!PERSIST || !has_next) break;
  tile += int(gridDim.x);
  b = nb; m0 = nm0; x0 = nx0; y0 = ny0; xg = nxg; xoob = nxoob;
#pragma unroll
  for (int i = 0; i < (XG ? C::NXW : 1); ++i) xoff[i] = nxoff[i];
  primed = true;
  }   // tile loop
}

#undef ST_DECODE_TILE
#undef ST_XOFF
#undef ST_ISSUE_W_AT
#undef ST_ISSUE_XG_AT
#undef ST_LOAD_ITEM
#undef ST_STORE_ITEM
#undef ST_READ_TAP
#undef ST_MFMA_TAP
#undef ST_SLICE
#undef ST_CHUNK_PIPE
#undef ST_CHUNK_PIPE_R
#undef ST_CHUNK_DMA_R
#undef ST_READ_TAP_R
#undef ST_MFMA_TAP_R
#undef ST_ISSUE_W
#undef ST_ISSUE_XG
#undef ST_CHUNK_DMA
#undef ST_LOAD_X
#undef ST_STORE_X
#undef ST_MFMA
#undef ST_BARRIER_KEEP_LOADS
#undef ST_BARRIER_DRAIN
#undef ST_BARRIER_DRAIN_XG

// OIHW fp32 -> packed bf16 [cin_pad/16][nparts][9][2 halves][cout_pad][8] (8 channels innermost: one
// lane's 8 K values are 16 contiguous bytes; the channel half outside cout: the LDS image of a block).  nparts = 1: the rounded weight; nparts = 2 (bf16x3 mode): part 0 =
// hi = bf16(w), part 1 = lo = bf16(w - hi).
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wp,
                                         int O, int I, int flip, int cout, int cin, int cout_pad,
                                         int nparts, long long total) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const int c = int(e % 8);
    long long k = e / 8;
    const int o = int(k % cout_pad);
    k /= cout_pad;
    const int half = int(k % 2);
    k /= 2;
    const int tap = int(k % 9);
    k /= 9;
    const int part = int(k % nparts);
    const int cg = int(k / nparts) * KC + half * 8 + c;
    float v = 0.f;
    if (o < cout && cg < cin) {
      v = flip ? w[(size_t(cg) * I + o) * 9 + (8 - tap)] : w[(size_t(o) * I + cg) * 9 + tap];
    }
    const __bf16 hi = (__bf16)v;
    wp[e] = part ? (__bf16)(v - float(hi)) : hi;
  }
}

template <int MT, int NT, int WM, int WP, bool GATE, bool SPLIT = false>
int launch_conv(ConvParams& p, hipStream_t stream) {
  using C = Cfg<MT, NT, WM, WP, SPLIT>;
  static_assert(C::LDS_BYTES <= 160 * 1024, "LDS image exceeds the CU's 160 KB");
  p.tiles_x = st::cdiv(p.Wo, TW);
  p.tiles_y = st::cdiv(p.Ho, C::TH);
  { const char* e = getenv("ST_BF16_ABLATE"); p.ablate = e ? atoi(e) : 0; }
  const long long ptiles = (long long)p.n * p.tiles_x * p.tiles_y;
  p.ctiles = st::cdiv(p.cout, C::BM);
  ST_REQUIRE(ptiles * p.ctiles <= 0x7fffffffLL, "st_conv3x3_bf16: too many tiles (%lld)", ptiles * p.ctiles);
  p.xcd_swizzle = (ptiles % 8 == 0) ? 1 : 0;
  p.vec_store = (p.ldo % 4 == 0 && reinterpret_cast<uintptr_t>(p.y) % 16 == 0) ? 1 : 0;
  dim3 grid((unsigned)(ptiles * p.ctiles));
  p.ntiles = (int)(ptiles * p.ctiles);
  // PIPE schedule for the forward form (measured +2.7 % bf16, +1.3 % bf16x3 over the layer set); the gated
  // data-gradient form keeps the phase schedule (its doubled register loads make the interleave a small loss)
  if (p.cin % KC != 0)
    ST_LAUNCH_DYN_LDS("st_conv3x3_bf16", (conv3x3_bf16_kernel<MT, NT, WM, WP, GATE, true, SPLIT, false>), grid,
                      dim3(NTHREADS), C::LDS_BYTES, stream, p);
  else
    ST_LAUNCH_DYN_LDS("st_conv3x3_bf16", (conv3x3_bf16_kernel<MT, NT, WM, WP, GATE, false, SPLIT, !GATE>), grid,
                      dim3(NTHREADS), C::LDS_BYTES, stream, p);
  return st::check_launch("st_conv3x3_bf16");
}

// G8 forms (st_conv3x3_fwd_bf16_g8): xg / yg select the activation layout on either side
template <int MT, int NT, int WM, int WP, bool SPLIT = false>
int launch_conv_g8(ConvParams& p, bool xg, bool yg, hipStream_t stream) {
  using C = Cfg<MT, NT, WM, WP, SPLIT>;
  static_assert(C::LDS_BYTES <= 160 * 1024, "LDS image exceeds the CU's 160 KB");
  p.tiles_x = st::cdiv(p.Wo, TW);
  p.tiles_y = st::cdiv(p.Ho, C::TH);
  { const char* e = getenv("ST_BF16_ABLATE"); p.ablate = e ? atoi(e) : 0; }
  const long long ptiles = (long long)p.n * p.tiles_x * p.tiles_y;
  p.ctiles = st::cdiv(p.cout, C::BM);
  ST_REQUIRE(ptiles * p.ctiles <= 0x7fffffffLL, "st_conv3x3_fwd_bf16_g8: too many tiles (%lld)", ptiles * p.ctiles);
  p.xcd_swizzle = (ptiles % 8 == 0) ? 1 : 0;
  p.vec_store = (p.ldo % 4 == 0 && reinterpret_cast<uintptr_t>(p.y) % 16 == 0) ? 1 : 0;
  dim3 grid((unsigned)(ptiles * p.ctiles));
  p.ntiles = (int)(ptiles * p.ctiles);
  if (xg && yg && !SPLIT) {
    // persistent form: one block per CU (a multiple of 8, so a block's tiles stay on its XCD's slots)
    const bool persist = !getenv("ST_G8_NO_PERSIST");
    dim3 pgrid((unsigned)std::min<long long>(p.ntiles, persist ? 256 : p.ntiles));
    ST_LAUNCH_DYN_LDS("st_conv3x3_fwd_bf16_g8",
                      (conv3x3_bf16_kernel<MT, NT, WM, WP, false, false, SPLIT, false, true, true>), pgrid,
                      dim3(NTHREADS), C::LDS_BYTES + 2 * 8 * 128 * 16, stream, p);
  } else if (xg && yg)
    ST_LAUNCH_DYN_LDS("st_conv3x3_fwd_bf16_g8",
                      (conv3x3_bf16_kernel<MT, NT, WM, WP, false, false, SPLIT, false, true, true>), grid,
                      dim3(NTHREADS), C::LDS_BYTES, stream, p);
  else if (xg)
    ST_LAUNCH_DYN_LDS("st_conv3x3_fwd_bf16_g8",
                      (conv3x3_bf16_kernel<MT, NT, WM, WP, false, false, SPLIT, false, true, false>), grid,
                      dim3(NTHREADS), C::LDS_BYTES, stream, p);
  else if (p.cin % KC != 0)
    ST_LAUNCH_DYN_LDS("st_conv3x3_fwd_bf16_g8",
                      (conv3x3_bf16_kernel<MT, NT, WM, WP, false, true, SPLIT, false, false, true>), grid,
                      dim3(NTHREADS), C::LDS_BYTES, stream, p);
  else
    ST_LAUNCH_DYN_LDS("st_conv3x3_fwd_bf16_g8",
                      (conv3x3_bf16_kernel<MT, NT, WM, WP, false, false, SPLIT, true, false, true>), grid,
                      dim3(NTHREADS), C::LDS_BYTES, stream, p);
  return st::check_launch("st_conv3x3_fwd_bf16_g8");
}

// Chain data gradient with the fused epilogue (EPI: see the kernel); xg: dy arrives as G8 (persistent form), else
// fp32 NCHW (the 3-channel top layer: TAIL staging).
template <int MT, int NT, int WM, int WP, int EPI>
int launch_conv_chain(ConvParams& p, bool xg, hipStream_t stream) {
  using C = Cfg<MT, NT, WM, WP, false>;
  p.tiles_x = st::cdiv(p.Wo, TW);
  p.tiles_y = st::cdiv(p.Ho, C::TH);
  { const char* e = getenv("ST_BF16_ABLATE"); p.ablate = e ? atoi(e) : 0; }   // debug: 8 no g16, 16 no gate, 32 no G8, 64 no bias
  if (p.ablate & 8) p.g16 = nullptr;
  if (p.ablate & 16) p.egate_g8 = nullptr;
  if (p.ablate & 64) p.bslabs = nullptr;
  const long long ptiles = (long long)p.n * p.tiles_x * p.tiles_y;
  p.ctiles = st::cdiv(p.cout, C::BM);
  ST_REQUIRE(ptiles * p.ctiles <= 0x7fffffffLL, "st_conv3x3_dgrad_bf16_g8_chain: too many tiles (%lld)", ptiles * p.ctiles);
  p.xcd_swizzle = (ptiles % 8 == 0) ? 1 : 0;
  p.vec_store = 0;
  p.ntiles = (int)(ptiles * p.ctiles);
  if (xg) {
    dim3 pgrid((unsigned)std::min<long long>(p.ntiles, 256));
    ST_LAUNCH_DYN_LDS("st_conv3x3_dgrad_bf16_g8_chain",
                      (conv3x3_bf16_kernel<MT, NT, WM, WP, false, false, false, false, true, true, EPI>), pgrid,
                      dim3(NTHREADS), C::LDS_BYTES + 8 * 128 * 16, stream, p);
  } else if (p.cin % KC != 0) {
    ST_LAUNCH_DYN_LDS("st_conv3x3_dgrad_bf16_g8_chain",
                      (conv3x3_bf16_kernel<MT, NT, WM, WP, false, true, false, false, false, true, EPI>),
                      dim3((unsigned)p.ntiles), dim3(NTHREADS), C::LDS_BYTES, stream, p);
  } else {
    ST_LAUNCH_DYN_LDS("st_conv3x3_dgrad_bf16_g8_chain",
                      (conv3x3_bf16_kernel<MT, NT, WM, WP, false, false, false, false, false, true, EPI>),
                      dim3((unsigned)p.ntiles), dim3(NTHREADS), C::LDS_BYTES, stream, p);
  }
  return st::check_launch("st_conv3x3_dgrad_bf16_g8_chain");
}

// rows of the bias partial-sum buffer the fused chain epilogue writes per image: (pixel tiles) x (waves along y)
inline int chain_main_slabs(int cin, int h, int w) {
  const int th = cin > 64 ? 16 : 32, wpn = cin > 64 ? 4 : 8;
  return st::cdiv(w, TW) * st::cdiv(h, th) * wpn;
}

template <bool GATE>
int dispatch_conv(ConvParams& p, hipStream_t s, bool split = false) {
  // bf16x3: hi and lo images of both operands -> 64 ch x (16 rows x 32) is the tile that fits LDS
  if (split) return launch_conv<1, 4, 2, 4, GATE, true>(p, s);
  if (p.cout > 64) return launch_conv<2, 4, 2, 4, GATE>(p, s);   // 128 ch x (16 rows x 32)
  if (p.cout > 32) {
    // the gated (data-gradient) form prefetches twice the registers per staging item: use the
    // 64 ch x (16 rows x 32) tile there so nothing spills
    if constexpr (GATE) return launch_conv<1, 4, 2, 4, GATE>(p, s);
    else return launch_conv<2, 4, 1, 8, GATE>(p, s);             //  64 ch x (32 rows x 32)
  }
  return launch_conv<1, 4, 1, 8, GATE>(p, s);                    //  32 ch x (32 rows x 32)
}

}  // namespace

namespace st {
// csrc/conv3x3.hip: adjoint of the replicate pad (+ upsample) on the padded data gradient
int launch_dgrad_fold(const float* dxp, float* dx, const float* gate, int n, int cin, int h, int w, int up,
                      hipStream_t s);
// csrc/dgrad_fold8.hip: the same, eight channels per item, with the G8 / G16 / bias-sum outputs of the G8 chain
int fold8_blocks_per_plane(int hs, int ws, int* iters_out);
int launch_dgrad_fold8(const float* dxp, const float* gate_f32, const void* gate_g8, float* dx, void* g8, void* g16,
                       float* bias, int n, int c, int h, int w, int up, hipStream_t s);
// csrc/dgrad_band.hip: the border ring of the chain data gradient (padded-domain values on the band of pixels the
// border source pixels need, then their fold / gate / hand-over forms; overwrites the main launch's border pixels)
size_t dgrad_band_scratch_bytes(int n, int c, int h, int w, int up);
int dgrad_band_slabs(int h, int w, int up);
int launch_dgrad_band(const void* dy, int dy_g8, const void* w_packed_t, float* scratch, const float* gate_f32,
                      const void* gate_g8, float* dx_f32, void* g8, void* g16, float* bias, int n, int k, int c, int h,
                      int w, int up, hipStream_t s);
}  // namespace st

extern "C" size_t st_conv3x3_packed_elems_bf16(int cout, int cin) {
  if (cout <= 0 || cin <= 0) return 0;
  const size_t cin_pad = size_t(st::cdiv(cin, KC)) * KC;
  const size_t cout_pad = size_t(st::cdiv(cout, CPAD)) * CPAD;
  return cin_pad * 9 * cout_pad;
}

extern "C" size_t st_conv3x3_packed_elems_bf16x3(int cout, int cin) {
  return 2 * st_conv3x3_packed_elems_bf16(cout, cin);
}

static int pack_weights_bf16_impl(const float* w_oihw, void* w_packed, int cout, int cin, int transpose_flip,
                                  int nparts, st_stream_t stream) {
  ST_REQUIRE(w_oihw && w_packed, "st_conv3x3_pack_weights_bf16: null pointer");
  ST_REQUIRE(cout > 0 && cin > 0, "st_conv3x3_pack_weights_bf16: bad dims cout=%d cin=%d", cout, cin);
  ST_REQUIRE(reinterpret_cast<uintptr_t>(w_packed) % 16 == 0,
             "st_conv3x3_pack_weights_bf16: w_packed must be 16-byte aligned");
  const int pc_out = transpose_flip ? cin : cout;
  const int pc_in = transpose_flip ? cout : cin;
  const int cout_pad = st::cdiv(pc_out, CPAD) * CPAD;
  const long long total = (long long)st_conv3x3_packed_elems_bf16(pc_out, pc_in) * nparts;
  const int blocks = (int)std::min<long long>(st::cdivll(total, 256), 4096);
  hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     w_oihw, (__bf16*)w_packed, cout, cin, transpose_flip ? 1 : 0, pc_out, pc_in,
                     cout_pad, nparts, total);
  return st::check_launch("st_conv3x3_pack_weights_bf16");
}

extern "C" int st_conv3x3_pack_weights_bf16(const float* w_oihw, void* w_packed, int cout, int cin,
                                            int transpose_flip, st_stream_t stream) {
  return pack_weights_bf16_impl(w_oihw, w_packed, cout, cin, transpose_flip, 1, stream);
}

extern "C" int st_conv3x3_pack_weights_bf16x3(const float* w_oihw, void* w_packed, int cout, int cin,
                                              int transpose_flip, st_stream_t stream) {
  return pack_weights_bf16_impl(w_oihw, w_packed, cout, cin, transpose_flip, 2, stream);
}

namespace st {
// csrc/conv3x3_bf16_cin3.hip
bool conv3x3_bf16_cin3_fwd(const float* x, const void* w_packed, const float* bias, void* y, bool y_g8, int n, int cin,
                           int h, int w, int cout, int cout_pad, int pad_zero, int relu, hipStream_t s, int* rc);
}  // namespace st

static int conv3x3_fwd_bf16_impl(const float* x, const void* w_packed, const float* bias, float* y,
                                 int n, int cin, int hin, int win, int cout, int upsample2x,
                                 int pad_mode, int relu, int pool, st_stream_t stream, bool split = false) {
  ST_REQUIRE(x && w_packed && y, "st_conv3x3_fwd_bf16: null pointer");
  ST_REQUIRE(n > 0 && cin > 0 && cout > 0 && hin > 0 && win > 0,
             "st_conv3x3_fwd_bf16: bad dims n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, hin, win);
  ST_REQUIRE(pad_mode == ST_PAD_REPLICATE || pad_mode == ST_PAD_ZERO,
             "st_conv3x3_fwd_bf16: bad pad_mode %d", pad_mode);
  // per-thread source offsets are 32-bit element offsets that include up to 8 channel planes
  ST_REQUIRE((long long)hin * win * 16 < (1LL << 31), "st_conv3x3_fwd_bf16: plane too large");
  ConvParams p;
  p.x = x; p.gate = nullptr; p.wp = (const u32x4*)w_packed; p.bias = bias; p.y = y;
  p.n = n; p.cin = cin; p.hs = hin; p.ws = win; p.cout = cout;
  p.up = upsample2x ? 1 : 0;
  p.H = hin << p.up; p.W = win << p.up;
  p.Ho = p.H; p.Wo = p.W; p.ldo = p.W; p.shift = 0; p.pool = pool;
  p.cout_pad = st::cdiv(cout, CPAD) * CPAD;
  p.nchunks = st::cdiv(cin, KC);
  p.pad_zero = pad_mode == ST_PAD_ZERO;
  p.relu = relu ? 1 : 0;
  if (!split && !pool && !upsample2x) {          // the 3-channel first layer: csrc/conv3x3_bf16_cin3.hip
    int rc = ST_OK;
    if (st::conv3x3_bf16_cin3_fwd(x, w_packed, bias, y, false, n, cin, hin, win, cout, p.cout_pad, p.pad_zero, p.relu,
                                  (hipStream_t)stream, &rc))
      return rc;
  }
  return dispatch_conv<false>(p, (hipStream_t)stream, split);
}

extern "C" int st_conv3x3_fwd_bf16(const float* x, const void* w_packed, const float* bias, float* y,
                                   int n, int cin, int hin, int win, int cout, int upsample2x,
                                   int pad_mode, int relu, st_stream_t stream) {
  return conv3x3_fwd_bf16_impl(x, w_packed, bias, y, n, cin, hin, win, cout, upsample2x, pad_mode, relu, 0, stream);
}

extern "C" int st_conv3x3_pool_fwd_bf16(const float* x, const void* w_packed, const float* bias, float* ypool,
                                        int n, int cin, int hin, int win, int cout, int pad_mode, int relu,
                                        st_stream_t stream) {
  return conv3x3_fwd_bf16_impl(x, w_packed, bias, ypool, n, cin, hin, win, cout, 0, pad_mode, relu, 1, stream);
}

static int conv3x3_fwd_g8_impl(const void* x, int x_g8, const void* w_packed, const float* bias, void* y,
                               int y_g8, int n, int cin, int hin, int win, int cout, int upsample2x,
                               int pad_mode, int relu, int pool, st_stream_t stream, bool split,
                               const void* ogate = nullptr, void* pool_codes = nullptr, void* y2_pooled = nullptr) {
  ST_REQUIRE(x && w_packed && y, "st_conv3x3_fwd_bf16_g8: null pointer");
  ST_REQUIRE(n > 0 && cin > 0 && cout > 0 && hin > 0 && win > 0,
             "st_conv3x3_fwd_bf16_g8: bad dims n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, hin, win);
  ST_REQUIRE(pad_mode == ST_PAD_REPLICATE || pad_mode == ST_PAD_ZERO, "st_conv3x3_fwd_bf16_g8: bad pad_mode %d",
             pad_mode);
  ST_REQUIRE(!(pool && upsample2x), "st_conv3x3_fwd_bf16_g8: pool and upsample2x exclude each other");
  ST_REQUIRE(!ogate || (y_g8 && !pool && !split && reinterpret_cast<uintptr_t>(ogate) % 16 == 0),
             "st_conv3x3_dgrad_bf16_g8: an output gate needs a 16-byte aligned G8 gate and a G8 output");
  if (!x_g8 && !y_g8)
    return conv3x3_fwd_bf16_impl((const float*)x, w_packed, bias, (float*)y, n, cin, hin, win, cout, upsample2x,
                                 pad_mode, relu, pool, stream, split);
  ST_REQUIRE(!x_g8 || cin % KC == 0, "st_conv3x3_fwd_bf16_g8: a G8 input needs cin %% 16 == 0 (got %d)", cin);
  ST_REQUIRE(!x_g8 || reinterpret_cast<uintptr_t>(x) % 16 == 0, "st_conv3x3_fwd_bf16_g8: x must be 16-byte aligned");
  ST_REQUIRE(!y_g8 || cout % 8 == 0, "st_conv3x3_fwd_bf16_g8: a G8 output needs cout %% 8 == 0 (got %d)", cout);
  ST_REQUIRE(!y_g8 || reinterpret_cast<uintptr_t>(y) % 16 == 0, "st_conv3x3_fwd_bf16_g8: y must be 16-byte aligned");
  ST_REQUIRE((long long)hin * win * 16 < (1LL << 31), "st_conv3x3_fwd_bf16_g8: plane too large");
  ConvParams p;
  p.x = (const float*)x; p.gate = nullptr; p.wp = (const u32x4*)w_packed; p.bias = bias; p.y = (float*)y;
  p.ogate = (const u32x4*)ogate;
  p.pcodes = (unsigned*)pool_codes;
  p.y2 = y2_pooled;
  p.n = n; p.cin = cin; p.hs = hin; p.ws = win; p.cout = cout;
  p.up = upsample2x ? 1 : 0;
  p.H = hin << p.up; p.W = win << p.up;
  p.Ho = p.H; p.Wo = p.W; p.ldo = p.W; p.shift = 0; p.pool = pool ? 1 : 0;
  p.cout_pad = st::cdiv(cout, CPAD) * CPAD;
  p.nchunks = st::cdiv(cin, KC);
  p.pad_zero = pad_mode == ST_PAD_ZERO;
  p.relu = relu ? 1 : 0;
  hipStream_t s = (hipStream_t)stream;
  if (!x_g8 && y_g8 && !split && !pool && !upsample2x && !ogate && !pool_codes) {
    int rc = ST_OK;
    if (st::conv3x3_bf16_cin3_fwd((const float*)x, w_packed, bias, y, true, n, cin, hin, win, cout, p.cout_pad,
                                  p.pad_zero, p.relu, s, &rc))
      return rc;
  }
  if (split) return launch_conv_g8<1, 4, 2, 4, true>(p, x_g8 != 0, y_g8 != 0, s);   // 64 ch x (16 rows x 32)
  if (cout > 64) return launch_conv_g8<2, 4, 2, 4>(p, x_g8 != 0, y_g8 != 0, s);   // 128 ch x (16 rows x 32)
  if (cout > 32) return launch_conv_g8<2, 4, 1, 8>(p, x_g8 != 0, y_g8 != 0, s);   //  64 ch x (32 rows x 32)
  return launch_conv_g8<1, 4, 1, 8>(p, x_g8 != 0, y_g8 != 0, s);                  //  32 ch x (32 rows x 32)
}

extern "C" int st_conv3x3_fwd_bf16_g8(const void* x, int x_g8, const void* w_packed, const float* bias, void* y,
                                      int y_g8, int n, int cin, int hin, int win, int cout, int upsample2x,
                                      int pad_mode, int relu, int pool, st_stream_t stream) {
  return conv3x3_fwd_g8_impl(x, x_g8, w_packed, bias, y, y_g8, n, cin, hin, win, cout, upsample2x, pad_mode, relu,
                             pool, stream, false);
}

// The pool-input layers of a G8 chain under autograd (the loss VGG's conv1_2 / conv2_2 / conv3_4, trainer.py:165-199):
// conv + ReLU + 2 x 2 max pool with the POOLED activation written in G8 and, for the backward, one byte per pooled
// pixel and channel (ConvParams::pcodes) -- the full-resolution activation, which the backward only ever used to find
// the window's winner and its sign, is never written.
extern "C" int st_conv3x3_pool_fwd_bf16_g8_codes(const void* x, int x_g8, const void* w_packed, const float* bias,
                                                 void* y_pooled_g8, void* codes, int n, int cin, int hin, int win,
                                                 int cout, int pad_mode, int relu, st_stream_t stream) {
  ST_REQUIRE(codes && reinterpret_cast<uintptr_t>(codes) % 8 == 0, "st_conv3x3_pool_fwd_bf16_g8_codes: codes must be 8-byte aligned");
  ST_REQUIRE(hin > 0 && win > 0 && hin % 2 == 0 && win % 2 == 0,
             "st_conv3x3_pool_fwd_bf16_g8_codes: even h and w expected (got %dx%d)", hin, win);
  return conv3x3_fwd_g8_impl(x, x_g8, w_packed, bias, y_pooled_g8, 1, n, cin, hin, win, cout, 0, pad_mode, relu, 1,
                             stream, false, nullptr, codes);
}

// A loss tap that feeds a pool (the loss VGG's conv1_2 / conv2_2 at the default taps, trainer.py:446-457): the fp32
// NCHW activation the loss reads AND its 2 x 2 max pool in G8 for the next conv, from one launch -- instead of
// st_maxpool2x2_fwd_g8 reading the fp32 activation back (805 MB per loss-VGG pass at B=8, 512 x 512).
extern "C" int st_conv3x3_fwd_bf16_g8_tap_pool(const void* x_g8, const void* w_packed, const float* bias, float* y,
                                               void* y_pooled_g8, int n, int cin, int hin, int win, int cout,
                                               int pad_mode, int relu, st_stream_t stream) {
  ST_REQUIRE(y_pooled_g8 && reinterpret_cast<uintptr_t>(y_pooled_g8) % 16 == 0,
             "st_conv3x3_fwd_bf16_g8_tap_pool: y_pooled_g8 must be 16-byte aligned");
  ST_REQUIRE(hin > 0 && win > 0 && hin % 2 == 0 && win % 2 == 0 && cout % 8 == 0,
             "st_conv3x3_fwd_bf16_g8_tap_pool: even h and w and cout %% 8 == 0 expected (got %dx%d, cout=%d)", hin, win, cout);
  return conv3x3_fwd_g8_impl(x_g8, 1, w_packed, bias, y, 0, n, cin, hin, win, cout, 0, pad_mode, relu, 0, stream, false,
                             nullptr, nullptr, y_pooled_g8);
}

extern "C" int st_conv3x3_fwd_bf16x3_g8(const void* x, int x_g8, const void* w_packed, const float* bias, void* y,
                                        int y_g8, int n, int cin, int hin, int win, int cout, int upsample2x,
                                        int pad_mode, int relu, int pool, st_stream_t stream) {
  return conv3x3_fwd_g8_impl(x, x_g8, w_packed, bias, y, y_g8, n, cin, hin, win, cout, upsample2x, pad_mode, relu,
                             pool, stream, true);
}

// Data gradient of a ZERO-padded layer inside a G8 chain (the loss VGG's backward, trainer.py:472 through the
// extractors of :165-199): the same implicit GEMM on the transposed / flipped filter (st_conv3x3_pack_weights_bf16
// with transpose_flip) with no bias and no activation.  A G8 `dy` must already carry the layer's own ReLU gate (its
// producer applied it: `dx_gate_g8` of the launch before, or st_grad_join_g8); `dx_gate_g8` zeroes the result
// where the layer's INPUT activation (G8, the shape of dx) is <= 0, so the G8 result feeds the next launch as is.
extern "C" int st_conv3x3_dgrad_bf16_g8(const void* dy, int dy_g8, const void* w_packed_t, void* dx, int dx_g8,
                                        const void* dx_gate_g8, int n, int cout, int h, int w, int cin,
                                        st_stream_t stream) {
  ST_REQUIRE(dy && w_packed_t && dx, "st_conv3x3_dgrad_bf16_g8: null pointer");
  ST_REQUIRE(!dx_gate_g8 || dx_g8, "st_conv3x3_dgrad_bf16_g8: dx_gate_g8 needs a G8 dx");
  return conv3x3_fwd_g8_impl(dy, dy_g8, w_packed_t, nullptr, dx, dx_g8, n, cout, h, w, cin, 0, ST_PAD_ZERO, 0, 0,
                             stream, false, dx_gate_g8);
}

// The fused chain form (EPI kernels + band kernels) covers the decoder's training shapes: more than 32 channels (the
// 64- / 128-channel tiles), pixel pairs whole (pixel-major bf16 stores), at least a 4 x 4 source grid so that the four
// border sides are distinct.  ST_NO_FUSED_FOLD=1 keeps the padded workspace + dgrad_fold8_kernel path (A/B runs).
static bool chain_fused_ok(int cin, int h, int w, int up) {
  if (getenv("ST_NO_FUSED_FOLD") != nullptr) return false;      // read per call: tests flip it inside one process
  const int hs = h >> up, ws = w >> up;
  return cin > 32 && cin % 8 == 0 && (h & ((2 << up) - 1)) == 0 && (w & ((2 << up) - 1)) == 0 && hs >= 4 && ws >= 4;
}

extern "C" int st_conv3x3_dgrad_chain_bias_slabs(int n, int cin, int h, int w, int upsample2x) {
  const int up = upsample2x ? 1 : 0;
  if (n <= 0 || cin <= 0 || h <= 0 || w <= 0 || (w >> up) % 4 != 0) return 0;
  const int fold = n * st::fold8_blocks_per_plane(h >> up, w >> up, nullptr);
  if (!chain_fused_ok(cin, h, w, up)) return fold;
  return std::max(fold, n * (chain_main_slabs(cin, h, w) + st::dgrad_band_slabs(h, w, up)));
}

// Data gradient of a REPLICATE-padded (decoder) layer inside a G8 chain: dy fp32 NCHW or G8 (already gated), the
// padded-domain result goes to `workspace` (st_conv3x3_dgrad_workspace_bytes) and st::launch_dgrad_fold8 folds it,
// applies the gate of the layer below and writes the requested forms (see include/st_hip.h).
extern "C" int st_conv3x3_dgrad_bf16_g8_chain(const void* dy, int dy_g8, const void* w_packed_t, float* workspace,
                                              const void* dx_gate, int dx_gate_g8, float* dx_f32, void* dx_g8,
                                              void* dx_g16, float* bias_slabs, int n, int cout, int h, int w, int cin,
                                              int upsample2x, st_stream_t stream) {
  ST_REQUIRE(dy && w_packed_t && workspace, "st_conv3x3_dgrad_bf16_g8_chain: null pointer");
  ST_REQUIRE(dx_f32 || dx_g8 || dx_g16, "st_conv3x3_dgrad_bf16_g8_chain: no output requested");
  ST_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0,
             "st_conv3x3_dgrad_bf16_g8_chain: bad dims n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, h, w);
  ST_REQUIRE(!upsample2x || (h % 2 == 0 && w % 2 == 0),
             "st_conv3x3_dgrad_bf16_g8_chain: upsampled layer needs even h, w (got %dx%d)", h, w);
  ST_REQUIRE(cin % 8 == 0 && (w >> (upsample2x ? 1 : 0)) % 4 == 0,
             "st_conv3x3_dgrad_bf16_g8_chain: needs cin %% 8 == 0 and a source width that is a multiple of 4");
  ST_REQUIRE(!dy_g8 || (cout % KC == 0 && reinterpret_cast<uintptr_t>(dy) % 16 == 0),
             "st_conv3x3_dgrad_bf16_g8_chain: a G8 dy needs cout %% 16 == 0 and 16-byte alignment");
  for (const void* q : {(const void*)workspace, dx_gate, (const void*)dx_f32, (const void*)dx_g8, (const void*)dx_g16})
    ST_REQUIRE(reinterpret_cast<uintptr_t>(q) % 16 == 0, "st_conv3x3_dgrad_bf16_g8_chain: 16-byte aligned buffers expected");
  ST_REQUIRE((long long)(h + 2) * (w + 2) * 16 < (1LL << 31), "st_conv3x3_dgrad_bf16_g8_chain: plane too large");
  hipStream_t s = (hipStream_t)stream;
  const int up = upsample2x ? 1 : 0;
  ConvParams p;
  p.x = (const float*)dy; p.gate = nullptr; p.wp = (const u32x4*)w_packed_t; p.bias = nullptr; p.y = workspace;
  p.n = n; p.cin = cout; p.hs = h; p.ws = w; p.cout = cin;
  p.up = 0; p.H = h; p.W = w;
  p.cout_pad = st::cdiv(cin, CPAD) * CPAD;
  p.nchunks = st::cdiv(cout, KC);
  p.pad_zero = 1; p.relu = 0; p.pool = 0;
  if (chain_fused_ok(cin, h, w, up) && !(dx_f32 && dx_gate && !dx_gate_g8)) {     // (fp32 result AND fp32 gate: form 2)
    // Fused form: the zero-padded conv on the unpadded grid with fold / gate / hand-over forms in its epilogue
    // (interior source pixels are final), then the band kernels recompute the border source pixels from the
    // padded-domain values of the ring and its neighbours and overwrite them.  No padded fp32 workspace round trip.
    p.y = (float*)dx_g8; p.g16 = (unsigned short*)dx_g16; p.bslabs = bias_slabs; p.dxf = dx_f32;
    p.egate_g8 = dx_gate_g8 ? (const u32x4*)dx_gate : nullptr;
    p.egate_f32 = dx_gate_g8 ? nullptr : (const float*)dx_gate;
    p.Ho = h; p.Wo = w; p.shift = 0; p.ldo = w;
    int rc;
    const bool xg = dy_g8 != 0;
    if (dx_f32 && !p.egate_f32) {       // the bottom of the chain hands fp32 back (with any bf16 form asked for as well)
      if (cin > 64) rc = up ? launch_conv_chain<2, 4, 2, 4, 10>(p, xg, s) : launch_conv_chain<2, 4, 2, 4, 9>(p, xg, s);
      else rc = up ? launch_conv_chain<2, 4, 1, 8, 10>(p, xg, s) : launch_conv_chain<2, 4, 1, 8, 9>(p, xg, s);
    } else if (p.egate_f32) {  // the top of the chain (64 <- 3 at full resolution): the one layer whose input is kept fp32
      if (cin > 64) rc = up ? launch_conv_chain<2, 4, 2, 4, 6>(p, xg, s) : launch_conv_chain<2, 4, 2, 4, 5>(p, xg, s);
      else rc = up ? launch_conv_chain<2, 4, 1, 8, 6>(p, xg, s) : launch_conv_chain<2, 4, 1, 8, 5>(p, xg, s);
    } else {
      if (cin > 64) rc = up ? launch_conv_chain<2, 4, 2, 4, 2>(p, xg, s) : launch_conv_chain<2, 4, 2, 4, 1>(p, xg, s);
      else rc = up ? launch_conv_chain<2, 4, 1, 8, 2>(p, xg, s) : launch_conv_chain<2, 4, 1, 8, 1>(p, xg, s);
    }
    if (rc != ST_OK) return rc;
    float* band_bias = bias_slabs ? bias_slabs + size_t(n) * chain_main_slabs(cin, h, w) * cin : nullptr;
    if (bias_slabs) {     // rows of the (either-path) buffer beyond the ones written here read as zero
      const int used = n * (chain_main_slabs(cin, h, w) + st::dgrad_band_slabs(h, w, up));
      const int all = st_conv3x3_dgrad_chain_bias_slabs(n, cin, h, w, upsample2x);
      if (all > used) {
        hipError_t e = hipMemsetAsync(bias_slabs + size_t(used) * cin, 0, size_t(all - used) * cin * sizeof(float), s);
        if (e != hipSuccess) return st::fail(ST_ERR_LAUNCH, "st_conv3x3_dgrad_bf16_g8_chain: memset: %s", hipGetErrorString(e));
      }
    }
    return st::launch_dgrad_band(dy, dy_g8, w_packed_t, workspace, p.egate_f32, p.egate_g8, dx_f32, dx_g8, dx_g16,
                                 band_bias, n, cout, cin, h, w, up, s);
  }
  p.Ho = h + 2; p.Wo = w + 2; p.shift = 1;
  p.ldo = st::padded_row(w);
  int rc;
  if (dy_g8) {
    if (cin > 64) rc = launch_conv_g8<2, 4, 2, 4>(p, true, false, s);
    else if (cin > 32) rc = launch_conv_g8<2, 4, 1, 8>(p, true, false, s);
    else rc = launch_conv_g8<1, 4, 1, 8>(p, true, false, s);
  } else {
    rc = dispatch_conv<false>(p, s, false);
  }
  if (rc != ST_OK) return rc;
  if (bias_slabs) {       // the buffer is sized for either path: rows this path does not write must read as zero
    const int mine = n * st::fold8_blocks_per_plane(h >> up, w >> up, nullptr);
    const int all = st_conv3x3_dgrad_chain_bias_slabs(n, cin, h, w, upsample2x);
    if (all > mine) {
      hipError_t e = hipMemsetAsync(bias_slabs + size_t(mine) * cin, 0, size_t(all - mine) * cin * sizeof(float), s);
      if (e != hipSuccess) return st::fail(ST_ERR_LAUNCH, "st_conv3x3_dgrad_bf16_g8_chain: memset: %s", hipGetErrorString(e));
    }
  }
  return st::launch_dgrad_fold8(workspace, dx_gate_g8 ? nullptr : (const float*)dx_gate, dx_gate_g8 ? dx_gate : nullptr,
                                dx_f32, dx_g8, dx_g16, bias_slabs, n, cin, h, w, up, s);
}

extern "C" int st_conv3x3_dgrad_fold8_bias_slabs(int h, int w, int upsample2x) {
  const int up = upsample2x ? 1 : 0;
  if (h <= 0 || w <= 0 || (w >> up) % 4 != 0) return 0;
  return st::fold8_blocks_per_plane(h >> up, w >> up, nullptr);
}

extern "C" int st_conv3x3_fwd_bf16x3(const float* x, const void* w_packed, const float* bias, float* y,
                                     int n, int cin, int hin, int win, int cout, int upsample2x,
                                     int pad_mode, int relu, st_stream_t stream) {
  return conv3x3_fwd_bf16_impl(x, w_packed, bias, y, n, cin, hin, win, cout, upsample2x, pad_mode, relu, 0, stream,
                               true);
}

extern "C" int st_conv3x3_pool_fwd_bf16x3(const float* x, const void* w_packed, const float* bias, float* ypool,
                                          int n, int cin, int hin, int win, int cout, int pad_mode, int relu,
                                          st_stream_t stream) {
  return conv3x3_fwd_bf16_impl(x, w_packed, bias, ypool, n, cin, hin, win, cout, 0, pad_mode, relu, 1, stream, true);
}

static int conv3x3_dgrad_bf16_impl(const float* dy, const float* y_gate, const void* w_packed_t,
                                   float* dx, float* workspace, const float* dx_gate, int n,
                                   int cout, int h, int w, int cin, int upsample2x, int pad_mode,
                                   st_stream_t stream, bool split) {
  ST_REQUIRE(dy && w_packed_t && dx, "st_conv3x3_dgrad_bf16: null pointer");
  ST_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0,
             "st_conv3x3_dgrad_bf16: bad dims n=%d cin=%d cout=%d h=%d w=%d", n, cin, cout, h, w);
  ST_REQUIRE(pad_mode == ST_PAD_REPLICATE || pad_mode == ST_PAD_ZERO,
             "st_conv3x3_dgrad_bf16: bad pad_mode %d", pad_mode);
  ST_REQUIRE(!upsample2x || (h % 2 == 0 && w % 2 == 0),
             "st_conv3x3_dgrad_bf16: upsampled layer needs even h, w (got %dx%d)", h, w);
  ST_REQUIRE(!(upsample2x && pad_mode == ST_PAD_ZERO),
             "st_conv3x3_dgrad_bf16: upsample2x is only supported with replicate padding");
  ST_REQUIRE((long long)(h + 2) * (w + 2) * 16 < (1LL << 31), "st_conv3x3_dgrad_bf16: plane too large");
  const bool rep = pad_mode == ST_PAD_REPLICATE;
  ST_REQUIRE(!rep || workspace, "st_conv3x3_dgrad_bf16: replicate mode needs a workspace of "
             "st_conv3x3_dgrad_workspace_bytes()");
  ST_REQUIRE(rep || !dx_gate, "st_conv3x3_dgrad_bf16: dx_gate is only supported with replicate padding");
  hipStream_t s = (hipStream_t)stream;
  ConvParams p;
  p.x = dy; p.gate = y_gate; p.wp = (const u32x4*)w_packed_t; p.bias = nullptr;
  p.y = rep ? workspace : dx;
  p.n = n; p.cin = cout; p.hs = h; p.ws = w; p.cout = cin;
  p.up = 0; p.H = h; p.W = w;
  p.Ho = rep ? h + 2 : h; p.Wo = rep ? w + 2 : w; p.shift = rep ? 1 : 0;
  p.ldo = rep ? st::padded_row(w) : w; p.pool = 0;
  p.cout_pad = st::cdiv(cin, CPAD) * CPAD;
  p.nchunks = st::cdiv(cout, KC);
  p.pad_zero = 1; p.relu = 0;
  int rc = y_gate ? dispatch_conv<true>(p, s, split) : dispatch_conv<false>(p, s, split);
  if (rc != ST_OK || !rep) return rc;
  return st::launch_dgrad_fold(workspace, dx, dx_gate, n, cin, h, w, upsample2x ? 1 : 0, s);
}

extern "C" int st_conv3x3_dgrad_bf16(const float* dy, const float* y_gate, const void* w_packed_t,
                                     float* dx, float* workspace, const float* dx_gate, int n,
                                     int cout, int h, int w, int cin, int upsample2x, int pad_mode,
                                     st_stream_t stream) {
  return conv3x3_dgrad_bf16_impl(dy, y_gate, w_packed_t, dx, workspace, dx_gate, n, cout, h, w, cin, upsample2x,
                                 pad_mode, stream, false);
}

extern "C" int st_conv3x3_dgrad_bf16x3(const float* dy, const float* y_gate, const void* w_packed_t,
                                       float* dx, float* workspace, const float* dx_gate, int n,
                                       int cout, int h, int w, int cin, int upsample2x, int pad_mode,
                                       st_stream_t stream) {
  return conv3x3_dgrad_bf16_impl(dy, y_gate, w_packed_t, dx, workspace, dx_gate, n, cout, h, w, cin, upsample2x,
                                 pad_mode, stream, true);
}
