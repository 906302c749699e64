// The two 3-channel layers of the generator on the vector ALU (gfx950): conv1_1 (3 -> 64,
// generator.py:47-49; also the loss VGG's first conv, trainer.py:165-199) and the decoder's last
// conv (64 -> 3, generator.py:115-117).
//
// On the MFMA kernels of csrc/conv3x3.hip these layers pad K (3 -> 8 input channels) or M
// (3 -> 32 output channels) and run MFMA-bound on the padding: 0.64 ms and 1.29 ms at batch 16,
// 512x512 for 14.5 GFLOP each.  Their real bound is HBM (1.07 GB written / read, ~0.2 ms), and the
// 7.2 G FMAs fit the vector ALU in about the same time, so they are plain fp32 FMA kernels:
//   * weights are staged once per block into LDS from the packed layout of st_conv3x3_pack_weights
//     ([cin/8][9 taps][8][cout_pad]) and read with broadcast 16-B LDS loads;
//   * cin_small: one thread = two output pixels, their 2 x 27 inputs in registers, output channels in
//     groups of 16 accumulators per pixel; stores are 128-B row segments;
//   * cout_small: one thread = 4 consecutive pixels x 3 output channels; per chunk of 8 input
//     channels the (16+2) x (64+2) halo is staged in LDS (edge replicate / zero pad resolved in
//     per-thread offsets computed once), rows are read as one 16-B + one 8-B LDS load.
// Same semantics and boundary as st_conv3x3_fwd (fp32 NCHW in / out, bias, optional ReLU); used by
// it transparently when the shape fits (no upsample, no fused pool).
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "common.h"

namespace {

constexpr int KC = 8;
constexpr size_t WS_LDS_MAX = size_t(512) * 9 * 4 * sizeof(float);   // LDS weight image of the cout <= 4 kernels at cin = 512

// ---- cin == 3, cout <= 64 ---------------------------------------------------------------
// Weights go through LDS ([27 steps][64 couts], 6.9 KB, staged once per block) and are read with
// broadcast ds_read_b128 (every lane the same address); a thread owns TWO pixels (rows r and r+8 of a
// 16 x 32 tile) so that each weight read feeds two FMAs and the LDS pipe stays at half load.
// (Wave-uniform global loads would be the natural form, but the compiler turns 27 x 16 of them into
// vector loads + v_readlane and spills SGPRs.)  The kernel is bound by those LDS reads (a broadcast ds_read_b128
// still returns 1 KB to the wave; 432 per thread, the LDS pipe is shared by four SIMDs): 3.5-4.0 TB/s of output.
// A four-pixels-per-thread form with packed FMAs (half the LDS reads per FMA, 1 KB stores) was tried in round 2:
// hipcc spills it (hoisted operand splats), so this form stays.
constexpr int ACO = 64;      // output channels held in the LDS weight image

template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_cin_small_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ y, int H, int W, int cout, int cout_pad, int pad_zero, int relu,
    int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) float ws[CIN * 9 * ACO];
  __shared__ float bs[ACO];
  int bt = blockIdx.x;
  const int tx = bt % tiles_x;
  bt /= tiles_x;
  const int ty = bt % tiles_y;
  const int b = bt / tiles_y;
  const int tid = threadIdx.x;
  for (int i = tid; i < CIN * 9 * ACO; i += 256) {
    const int step = i / ACO, co = i - step * ACO;       // step = c * 9 + tap
    const int c = step / 9, tap = step - c * 9;
    ws[i] = co < cout ? wp[size_t(tap * KC + c) * cout_pad + co] : 0.f;
  }
  if (tid < ACO) bs[tid] = (bias && tid < cout) ? bias[tid] : 0.f;

  // Pixels past the right / bottom edge are clamped onto the edge pixel: such a thread recomputes that
  // pixel and stores the identical value (no branch around the stores: LLVM would sink the FMA chains into it).
  const int ox = min(tx * 32 + (tid & 31), W - 1);
  const size_t plane = size_t(H) * W;
  const float* xb = x + size_t(b) * CIN * plane;
  float in[2][CIN * 9];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int oy = min(ty * 16 + (tid >> 5) + 8 * p, H - 1);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int gy = oy + tap / 3 - 1, gx = ox + tap % 3 - 1;
      const bool keep = (gy >= 0 && gy < H && gx >= 0 && gx < W) || !pad_zero;
      const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        const float v = xb[c * plane + size_t(cy) * W + cx];
        in[p][c * 9 + tap] = keep ? v : 0.f;
      }
    }
  }
  __syncthreads();

  for (int co0 = 0; co0 < cout; co0 += 16) {
    float acc[2][16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[0][k] = acc[1][k] = bs[co0 + k];
#pragma unroll
    for (int step = 0; step < CIN * 9; ++step) {
      float wv[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4*>(ws + step * ACO + co0 + q * 4);
        wv[q * 4 + 0] = t.x; wv[q * 4 + 1] = t.y; wv[q * 4 + 2] = t.z; wv[q * 4 + 3] = t.w;
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        acc[0][k] = fmaf(wv[k], in[0][step], acc[0][k]);
        acc[1][k] = fmaf(wv[k], in[1][step], acc[1][k]);
      }
      // compiler fence: without it all 108 weight reads of the group are hoisted above the FMAs and spill
      if (step % 3 == 2) asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int oy = min(ty * 16 + (tid >> 5) + 8 * p, H - 1);
      float* yb = y + size_t(b) * cout * plane + size_t(oy) * W + ox;
#pragma unroll
      for (int k = 0; k < 16; ++k) {        // cout % 16 == 0 (host check): no per-channel branch either
        float v = acc[p][k];
        if (relu) v = fmaxf(v, 0.f);
        yb[size_t(co0 + k) * plane] = v;
      }
    }
  }
}

// ---- cin == 3, cout <= 64 on the fp32 matrix pipe ------------------------------------------------
// The same layer as conv3x3_cin_small_kernel as a GEMM with K = 27 (padded to 28 = 14 steps of
// v_mfma_f32_32x32x2_f32, exact fp32 products accumulated in k order): D[cout][pixel] = sum_k W[cout][k] X[k][pixel],
// k = channel * 9 + tap.  The vector-ALU form is bound by its weight reads (432 broadcast ds_read_b128 per thread);
// here the 2 x 14 weight fragments of a lane stay in registers and the only operand traffic is one conflict-free
// ds_read_b32 per MFMA.
//   * persistent blocks of 4 waves (4 per CU), tile = 4 rows x 128 columns x 64 channels; wave w owns row w and walks
//     its eight 32-channel x 32-column accumulator tiles ONE at a time (see the loop): 16 accumulator registers;
//   * the (4 + 2) x (128 + 2) halo of the 3 input channels is staged per tile (9.5 KB; padding resolved there), the
//     next tile's halo requested before the current tile's MFMAs.
// B=16, 512 x 512: 310 -> 266 us (45 -> 53 % of the HBM peak for its 1.07 GB of output); grid sweep 512 / 768 / 1024 /
// 2048 / one block per tile: 295 / 277 / 266 / 282 / 283 us.  ST_CIN3_VALU=1 selects the vector-ALU kernel.
constexpr int MTW = 128, MTH = 4;                  // tile: 4 rows x 128 columns (wave = row, accumulator tile = 32 columns)
constexpr int MHC = MTW + 4;                       // halo row stride (MTW + 2 used)
constexpr int MPLANE = (MTH + 2) * MHC;

__global__ __launch_bounds__(256, 4) void conv3x3_cin3_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ y, int H, int W, int cout, int cout_pad, int pad_zero, int relu,
    int tiles_x, int tiles_y, int ntiles) {
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  constexpr int NH = 3 * (MTH + 2) * (MTW + 2), NHP = (NH + 255) / 256;      // halo elements, per thread
  __shared__ float xs[3 * MPLANE];
  __shared__ __attribute__((aligned(16))) float patch[4 * 32 * 32];
  __shared__ float bsh[64];
  if (threadIdx.x < 64) bsh[threadIdx.x] = (bias && int(threadIdx.x) < cout) ? bias[threadIdx.x] : 0.f;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const size_t plane = size_t(H) * W;

  // halo element i of this thread: position (r, cc) of channel c <- pixel (y0 + r - 1, x0 + cc - 1)
  int hpk[NHP];       // (channel << 16) | (row << 8) | column
#pragma unroll
  for (int j = 0; j < NHP; ++j) {
    const int i = min(tid + j * 256, NH - 1);
    const int c = i / ((MTH + 2) * (MTW + 2)), rem = i - c * ((MTH + 2) * (MTW + 2));
    const int r = rem / (MTW + 2);
    hpk[j] = (c << 16) | (r << 8) | (rem - r * (MTW + 2));
  }
  float hv[NHP];
  // a persistent block walks tiles blockIdx.x, + gridDim.x, ...: the halo of the next tile is requested before the
  // MFMAs and the stores of the current one
  auto load_halo = [&](int t) {
    const int tx = t % tiles_x;
    const int rest = t / tiles_x;
    const int ty = rest % tiles_y, b = rest / tiles_y;
    const float* xb = x + size_t(b) * 3 * plane;
#pragma unroll
    for (int j = 0; j < NHP; ++j) {
      const int gy = ty * MTH + ((hpk[j] >> 8) & 255) - 1, gx = tx * MTW + (hpk[j] & 255) - 1;
      const bool keep = (gy >= 0 && gy < H && gx >= 0 && gx < W) || !pad_zero;
      const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
      const float v = xb[(hpk[j] >> 16) * plane + size_t(cy) * W + cx];
      hv[j] = keep ? v : 0.f;
    }
  };
  int tile = blockIdx.x;
  load_halo(tile);

  // weight fragments: step s of lane (li, lh) = W[cout = mt * 32 + li][k = 2 s + lh]; k = 27 and channels >= cout are zero
  float wa[2][14];
#pragma unroll
  for (int s2 = 0; s2 < 14; ++s2) {
    const int k = 2 * s2 + lh;
    const int c = min(k / 9, 2), tap = k - (k / 9) * 9;
    const bool real = k < 27;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int co = mt * 32 + li;
      wa[mt][s2] = (real && co < cout) ? wp[size_t((real ? tap : 0) * KC + c) * cout_pad + co] : 0.f;
    }
  }
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const float* bp = xs + wave * MHC + li;
  for (; tile < ntiles; tile += gridDim.x) {
    const int tx = tile % tiles_x;
    const int rest = tile / tiles_x;
    const int ty = rest % tiles_y, b = rest / tiles_y;
    const int x0 = tx * MTW, y0 = ty * MTH;
    __syncthreads();                       // the previous tile's operand reads are done
#pragma unroll
    for (int j = 0; j < NHP; ++j)
      if (NH % 256 == 0 || tid + j * 256 < NH)
        xs[(hpk[j] >> 16) * MPLANE + ((hpk[j] >> 8) & 255) * MHC + (hpk[j] & 255)] = hv[j];
    __syncthreads();
    if (tile + int(gridDim.x) < ntiles) load_halo(tile + int(gridDim.x));

    // One 32-channel x 32-column accumulator tile at a time (a chain of 14 MFMAs), its LDS-patch transpose and its four
    // 16-byte-per-lane stores right behind it: 16 accumulator registers instead of 128, so four blocks share a CU and
    // one wave's stores drain under the other waves' MFMAs.  (With all eight tiles of a wave held in registers -- 229
    // registers, two blocks per CU -- the phases simply added up: 313 us = 107 of MFMAs + 125 of stores + 35 of halo
    // loads + the rest, measured by leaving each out.)
    // An accumulator register is 32 consecutive x of one channel row; stored as is that is 16 dword-store
    // instructions per tile and the store ISSUE binds, hence the transpose: 8 lanes cover one 128-B row segment, an
    // instruction writes 8 channel rows.  (w % 4 == 0, y 16-B aligned: host check)
    // operand offsets of the two k of a step are compile-time constants selected by lh
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int yw = y0 + wv;
    float* pt = patch + wave * (32 * 32);
    // lh made opaque per tile: the 14 operand-offset selects (and nothing else) are redone per tile instead of being
    // hoisted into 14 registers; the stores are [scalar base] + [one per-lane byte offset] for the same reason
    int lho = lh;
    asm volatile("" : "+v"(lho));
    const unsigned st_off = unsigned((size_t(lane >> 3) * plane + (lane & 7) * 4) * sizeof(float));
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 14; ++s2) {
          const int k0 = 2 * s2, k1 = 2 * s2 + 1;
          const int o0 = (k0 / 9) * MPLANE + ((k0 % 9) / 3) * MHC + (k0 % 9) % 3;
          const int o1 = k1 < 27 ? (k1 / 9) * MPLANE + ((k1 % 9) / 3) * MHC + (k1 % 9) % 3 : 0;
          const float bv = bp[(lho ? o1 : o0) + nt * 32];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[mt][s2], bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[r] + bsh[mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
          if (relu) v = fmaxf(v, 0.f);
          pt[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(pt + (q * 8 + (lane >> 3)) * 32 + (lane & 7) * 4);
          const int cg = mt * 32 + q * 8;                      // + lane / 8: the same side of cout (a multiple of 16)
          unsigned char* base = reinterpret_cast<unsigned char*>(y + (size_t(b) * cout + cg) * plane + size_t(yw) * W +
                                                                 x0 + nt * 32);
          if (cg < cout && yw < H && x0 + nt * 32 + (lane & 7) * 4 < W)
            *reinterpret_cast<f32x4*>(base + st_off) = v;
        }
        __builtin_amdgcn_sched_barrier(0);      // one tile's registers at a time
      }
    }
  }
}

// ---- cout <= 4 (cin % 8 == 0) -----------------------------------------------------------
constexpr int BTH = 16, BTW = 64;                 // output tile of a block
constexpr int BHR = BTH + 2, BHC = BTW + 2;       // halo
constexpr int BLD = 68;                           // LDS row stride (floats): 16-B aligned rows
constexpr int BNP = (BHR * BHC + 255) / 256;      // halo positions per thread

// forward tile (the weight-gradient kernel below keeps the 16 x 64 tile of BTH / BTW)
constexpr int FTW = 128;                          // tile width
constexpr int RPT = 1;                            // output rows per thread (a thread = RPT rows x 4 pixels x CO channels)
constexpr int FKC = 8;                            // input channels staged per chunk
constexpr int FTH = RPT * (1024 / FTW);
constexpr int FHR = FTH + 2, FHC = FTW + 2;
constexpr int FLD = FTW + 4;
constexpr int FNP = (FHR * FHC + 255) / 256;

template <int CO>
__global__ __launch_bounds__(256) void conv3x3_cout_small_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ y, int cin, int H, int W, int cout_pad, int pad_zero, int relu,
    int tiles_x, int tiles_y) {
  __shared__ __attribute__((aligned(16))) float xs[FKC * FHR * FLD];
  extern __shared__ __attribute__((aligned(16))) float wsm[];     // [cin * 9][4]: (co 0..2, pad) per (channel, tap)
  int bt = blockIdx.x;
  const int tx = bt % tiles_x;
  bt /= tiles_x;
  const int ty = bt % tiles_y;
  const int b = bt / tiles_y;
  const int tid = threadIdx.x;
  const int x0 = tx * FTW, y0 = ty * FTH;
  const size_t plane = size_t(H) * W;
  const float* xb = x + size_t(b) * cin * plane;

  // staging positions of this thread (source offset inside a plane, LDS offset, zero-pad mask)
  unsigned soff[FNP], loff[FNP], smask[FNP];
#pragma unroll
  for (int i = 0; i < FNP; ++i) {
    const int idx = min(tid + i * 256, FHR * FHC - 1);
    const int r = idx / FHC, cc = idx - r * FHC;
    const int gy = y0 + r - 1, gx = x0 + cc - 1;
    const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
    soff[i] = unsigned(cy) * unsigned(W) + unsigned(cx);
    loff[i] = unsigned(r * FLD + cc);
    smask[i] = (inb || !pad_zero) ? 0xffffffffu : 0u;
  }

  for (int i = tid; i < cin * 9 * 4; i += 256) {
    const int o = i & 3, ct = i >> 2;                  // ct = channel * 9 + tap
    const int c = ct / 9, tap = ct - c * 9;
    wsm[i] = o < CO ? wp[size_t(((c / KC) * 9 + tap) * KC + (c % KC)) * cout_pad + o] : 0.f;
  }
  const int row = (tid / (FTW / 4)) * RPT, col4 = (tid % (FTW / 4)) * 4;
  float acc[RPT][CO][4];
#pragma unroll
  for (int r = 0; r < RPT; ++r)
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[r][o][q] = bias ? bias[o] : 0.f;

  // Software pipeline: the global loads of chunk k+1 are issued right after chunk k has been written to LDS and
  // stay in flight under chunk k's FMAs (the staging registers are free by then).
  const int nchunks = cin / FKC;
  float st_[FKC][FNP];
#pragma unroll
  for (int c = 0; c < FKC; ++c)
#pragma unroll
    for (int i = 0; i < FNP; ++i) st_[c][i] = xb[size_t(c) * plane + soff[i]];
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    __syncthreads();                                  // previous chunk's reads are done
    // positions past the halo's end were clamped onto its last element: those threads store that element again
#pragma unroll
    for (int c = 0; c < FKC; ++c)
#pragma unroll
      for (int i = 0; i < FNP; ++i)
        xs[c * (FHR * FLD) + loff[i]] = __uint_as_float(__float_as_uint(st_[c][i]) & smask[i]);
    __syncthreads();
    if (chunk + 1 < nchunks) {
#pragma unroll
      for (int c = 0; c < FKC; ++c) {
        const float* src = xb + size_t((chunk + 1) * FKC + c) * plane;
#pragma unroll
        for (int i = 0; i < FNP; ++i) st_[c][i] = src[soff[i]];
      }
    }
#pragma unroll
    for (int c = 0; c < FKC; ++c) {
      float v[RPT + 2][6];
#pragma unroll
      for (int dy = 0; dy < RPT + 2; ++dy) {
        const float* rp = xs + c * (FHR * FLD) + (row + dy) * FLD + col4;
        const float4 a = *reinterpret_cast<const float4*>(rp);
        const float2 e = *reinterpret_cast<const float2*>(rp + 4);
        v[dy][0] = a.x; v[dy][1] = a.y; v[dy][2] = a.z; v[dy][3] = a.w; v[dy][4] = e.x; v[dy][5] = e.y;
      }
      const int ch = chunk * FKC + c;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float4 w4 = *reinterpret_cast<const float4*>(wsm + (ch * 9 + tap) * 4);   // broadcast
        const float wr[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int r = 0; r < RPT; ++r)
#pragma unroll
          for (int o = 0; o < CO; ++o) {
            const float wv = wr[o];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[r][o][q] = fmaf(wv, v[r + tap / 3][q + tap % 3], acc[r][o][q]);
          }
      }
    }
  }

  const int ox = x0 + col4;
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int oy = y0 + row + r;
    if (oy < H && ox < W) {
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        float* dst = y + (size_t(b) * CO + o) * plane + size_t(oy) * W + ox;
        float rr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rr[q] = relu ? fmaxf(acc[r][o][q], 0.f) : acc[r][o][q];
        if (ox + 3 < W && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
          *reinterpret_cast<float4*>(dst) = make_float4(rr[0], rr[1], rr[2], rr[3]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (ox + q < W) dst[q] = rr[q];
        }
      }
    }
  }
}

// ---- cout <= 4, row-strip form (W % 4 == 0, 16-B aligned planes) -------------------------------------------
// No LDS staging of the input at all: a group of LPR lanes owns a strip of SR output rows x LPR*4 pixels and walks
// over the input channels; per channel it loads the SR + 2 input rows of its strip ONCE as aligned 16-B vectors
// (the column to the left / right comes from the neighbouring lane; only the two lanes at the ends of a group load
// a halo dword), and every loaded row feeds the (up to) three output rows it belongs to from registers.  The rows of
// channel c + 1 are in flight while channel c is multiplied.  Weights: LDS, 9 broadcast reads per channel.
typedef float sf4 __attribute__((ext_vector_type(4)));

template <int SR>
struct StripRows {
  sf4 a[SR + 2];
  float e[SR + 2];
};

template <int CO, int LPR, int SR, bool ZP>
__global__ __launch_bounds__(256, (SR > 4 ? 2 : 3)) void conv3x3_cout_strip_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ y, int cin, int H, int W, int cout_pad, int relu,
    int tiles_x, int tiles_y) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];     // [cin * 9][4]: (co 0..2, pad) per (channel, tap)
  constexpr int GROUPS = 256 / LPR;
  int bt = blockIdx.x;
  const int tx = bt % tiles_x;
  bt /= tiles_x;
  const int ty = bt % tiles_y;
  const int b = bt / tiles_y;
  const int tid = threadIdx.x;
  const int g = tid / LPR, t = tid % LPR;
  const size_t plane = size_t(H) * W;
  const float* xb = x + size_t(b) * cin * plane;

  for (int i = tid; i < cin * 9 * 4; i += 256) {
    const int o = i & 3, ct = i >> 2;                  // ct = channel * 9 + tap
    const int c = ct / 9, tap = ct - c * 9;
    wsm[i] = o < CO ? wp[size_t(((c / KC) * 9 + tap) * KC + (c % KC)) * cout_pad + o] : 0.f;
  }

  const int y0 = (ty * GROUPS + g) * SR;               // first output row of this strip
  const int colraw = (tx * LPR + t) * 4;
  const bool live = colraw < W;
  const int col = min(colraw, W - 4);                 // dead lanes shadow the last column group
  const bool has_l = col > 0, has_r = col + 4 < W;
  const bool edge_l = t == 0, edge_r = t == LPR - 1 || (tid & 63) == 63;
  const bool eload = (edge_l && has_l) || (edge_r && has_r && !(edge_l && has_l));
  const int eoff = (edge_l && has_l) ? col - 1 : col + 4;
  // LPR == 1 would need both halos from memory; the host never picks it
  unsigned roff[SR + 2];
  float rmask[SR + 2];
#pragma unroll
  for (int i = 0; i < SR + 2; ++i) {
    const int gy = y0 - 1 + i;
    roff[i] = unsigned(min(max(gy, 0), H - 1)) * unsigned(W);
    rmask[i] = (ZP && (gy < 0 || gy >= H)) ? 0.f : 1.f;
  }

  float acc[SR][CO][4];
  float bv[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) bv[o] = bias ? bias[o] : 0.f;
#pragma unroll
  for (int r = 0; r < SR; ++r)
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[r][o][q] = bv[o];
  __syncthreads();

  // Rolling prefetch: as soon as row i of channel c has been multiplied, the same row of channel c + 1 is requested
  // (the scheduling barriers keep the request there; hipcc otherwise sinks all of them below the FMAs).  hipcc loads
  // into fresh registers and copies them at the loop's end behind a wait for everything, so a wave stalls once per
  // channel for the tail of its requests; the other waves of the SIMD cover that.  (Tying the request to the row's
  // own registers with "+v" asm operands does not survive register allocation: it copies registers with loads in
  // flight.  Tried and removed.)  Buffer loads: 32-bit per-lane offsets, the channel offset is the scalar offset.
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, int(unsigned(size_t(cin) * plane * sizeof(float))), 0x00020000);
  const unsigned plane_bytes = unsigned(plane * sizeof(float));
  const int edelta = eload ? (eoff - col) * 4 : 0;     // lanes without a halo column re-read their own first element
  StripRows<SR> rw;
#pragma unroll
  for (int i = 0; i < SR + 2; ++i) {
    const int vo = int((roff[i] + unsigned(col)) * 4u);
    rw.a[i] = __builtin_bit_cast(sf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, 0, 0));
    rw.e[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vo + edelta, 0, 0));
  }
  for (int ch = 0; ch < cin; ++ch) {
    float wv[9][CO];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float4 w4 = *reinterpret_cast<const float4*>(wsm + (ch * 9 + tap) * 4);   // broadcast
      const float wr[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
      for (int o = 0; o < CO; ++o) wv[tap][o] = wr[o];
    }
    const int so = int(unsigned(min(ch + 1, cin - 1)) * plane_bytes);    // the last channel re-reads itself (unused)
#pragma unroll
    for (int i = 0; i < SR + 2; ++i) {
      sf4 a = rw.a[i];
      float e = rw.e[i];
      if (ZP) { a = a * rmask[i]; e = e * rmask[i]; }
      const float nl = __shfl_up(a[3], 1, 64), nr = __shfl_down(a[0], 1, 64);
      float v[6];
      v[0] = !has_l ? (ZP ? 0.f : a[0]) : (edge_l ? e : nl);
      v[1] = a[0]; v[2] = a[1]; v[3] = a[2]; v[4] = a[3];
      v[5] = !has_r ? (ZP ? 0.f : a[3]) : (edge_r ? e : nr);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int r = i - dy;
        if (r < 0 || r >= SR) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int o = 0; o < CO; ++o)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[r][o][q] = fmaf(wv[dy * 3 + dx][o], v[q + dx], acc[r][o][q]);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int vo = int((roff[i] + unsigned(col)) * 4u);
      rw.a[i] = __builtin_bit_cast(sf4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, so, 0));
      rw.e[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vo + edelta, so, 0));
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if (live) {
#pragma unroll
    for (int r = 0; r < SR; ++r) {
      const int oy = y0 + r;
      if (oy < H) {
#pragma unroll
        for (int o = 0; o < CO; ++o) {
          sf4 out;
#pragma unroll
          for (int q = 0; q < 4; ++q) out[q] = relu ? fmaxf(acc[r][o][q], 0.f) : acc[r][o][q];
          *reinterpret_cast<sf4*>(y + (size_t(b) * CO + o) * plane + size_t(oy) * W + col) = out;
        }
      }
    }
  }
}

template <int LPR, int SR>
int launch_cout_strip(const float* x, const float* wp, const float* bias, float* y, int n, int cin, int h, int w,
                      int cout_pad, int pad_zero, int relu, hipStream_t s) {
  const int tiles_x = st::cdiv(w, LPR * 4), tiles_y = st::cdiv(h, (256 / LPR) * SR);
  const dim3 grid((unsigned)(n * tiles_x * tiles_y));
  const size_t lds = size_t(cin) * 9 * 4 * sizeof(float);
  // cin <= 512 (host check): up to 72 KB of dynamic LDS, opted in for that maximum once per device
  if (pad_zero) {
    static st::DynLdsOptIn optin;
    const int rc = st::allow_dynamic_lds(conv3x3_cout_strip_kernel<3, LPR, SR, true>, WS_LDS_MAX, optin,
                                         "st_conv3x3_fwd(cout=3, strips)");
    if (rc != ST_OK) return rc;
    hipLaunchKernelGGL((conv3x3_cout_strip_kernel<3, LPR, SR, true>), grid, dim3(256), lds, s, x, wp, bias, y, cin, h,
                       w, cout_pad, relu, tiles_x, tiles_y);
  } else {
    static st::DynLdsOptIn optin;
    const int rc = st::allow_dynamic_lds(conv3x3_cout_strip_kernel<3, LPR, SR, false>, WS_LDS_MAX, optin,
                                         "st_conv3x3_fwd(cout=3, strips)");
    if (rc != ST_OK) return rc;
    hipLaunchKernelGGL((conv3x3_cout_strip_kernel<3, LPR, SR, false>), grid, dim3(256), lds, s, x, wp, bias, y, cin, h,
                       w, cout_pad, relu, tiles_x, tiles_y);
  }
  return st::check_launch("st_conv3x3_fwd(cout=3, strips)");
}

template <int SR>
int launch_cout_strip_w(const float* x, const float* wp, const float* bias, float* y, int n, int cin, int h, int w,
                        int cout_pad, int pad_zero, int relu, hipStream_t s) {
  if (w >= 256) return launch_cout_strip<64, SR>(x, wp, bias, y, n, cin, h, w, cout_pad, pad_zero, relu, s);
  if (w >= 128) return launch_cout_strip<32, SR>(x, wp, bias, y, n, cin, h, w, cout_pad, pad_zero, relu, s);
  return launch_cout_strip<16, SR>(x, wp, bias, y, n, cin, h, w, cout_pad, pad_zero, relu, s);
}

// ---- weight / bias gradient of a layer with cout == 3 (the decoder's last conv) ------------
//   dW[o][c][tap] = sum_{n,y,x} G[n][o][y][x] * Xpad[n][c][y+ky-1][x+kx-1],  G = dy (* (y > 0))
// 1728 outputs, each a reduction over every pixel: on the MFMA weight-gradient kernel the 3 output channels
// are padded to 64 rows (0.51 ms at batch 8, 256 x 256 for 0.9 GFLOP).  Here a block owns WC = 4 input
// channels and a range of 16 x 64 pixel tiles; a thread owns 4 consecutive pixels, keeps the 4 x 3 x 9
// partial sums of its pixels in registers over the whole range (432 FMAs per tile), and the block reduces
// them once at the end (wave shuffles, then LDS) into a partial slab [split][3][cin * 9], the layout the
// fixed-order slab reduction of csrc/conv3x3_wgrad.hip sums.  Deterministic, no atomics.
constexpr int WC = 4;

// XG8: x is the G8 activation (bf16 [n][cin/8][hs][ws][8], the hand-over form of the bf16 decoder chain): the block's
// WC = 4 channels are one half of a pixel's 16-byte unit, read as 8 bytes and widened to fp32 (exact).
template <int CO, bool XG8 = false>
__global__ __launch_bounds__(256) void wgrad_cout_small_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ ygate,
    float* __restrict__ slabs, float* __restrict__ bias_slabs, int n, int cin, int hs, int ws, int H, int W,
    int up, int pad_zero, int tiles_x, int tiles_y, int ntiles, int tiles_per_split) {
  __shared__ __attribute__((aligned(16))) float xs[WC * BHR * BLD];
  __shared__ float red[4][WC * CO * 9 + CO];
  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * WC, split = blockIdx.y;
  const int t_begin = split * tiles_per_split, t_end = min(t_begin + tiles_per_split, ntiles);
  const int row = tid >> 4, col4 = (tid & 15) * 4;
  const size_t plane_src = size_t(hs) * ws, plane_out = size_t(H) * W;

  float acc[WC][CO][9];
  float bacc[CO];
#pragma unroll
  for (int c = 0; c < WC; ++c)
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[c][o][t] = 0.f;
#pragma unroll
  for (int o = 0; o < CO; ++o) bacc[o] = 0.f;

  for (int t = t_begin; t < t_end; ++t) {
    int rem = t;
    const int tx = rem % tiles_x;
    rem /= tiles_x;
    const int ty = rem % tiles_y;
    const int b = rem / tiles_y;
    const int x0 = tx * BTW, y0 = ty * BTH;
    // halo of WC channels -> LDS (edge replicate / zero pad / x2-upsample source index in the address)
    float st_[WC][BNP];
    unsigned loff[BNP], smask[BNP];
#pragma unroll
    for (int i = 0; i < BNP; ++i) {
      const int idx = min(tid + i * 256, BHR * BHC - 1);
      const int r = idx / BHC, cc = idx - r * BHC;
      const int gy = y0 + r - 1, gx = x0 + cc - 1;
      const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
      int cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
      if (up) { cy >>= 1; cx >>= 1; }
      loff[i] = unsigned(r * BLD + cc);
      smask[i] = (inb || !pad_zero) ? 0xffffffffu : 0u;
      if constexpr (XG8) {
        static_assert(WC == 4, "half a G8 unit");
        const unsigned short* xg = reinterpret_cast<const unsigned short*>(x);
        const size_t at = ((size_t(b) * (cin >> 3) + (c0 >> 3)) * plane_src + size_t(cy) * ws + cx) * 8 + (c0 & 7);
        const uint2 v = *reinterpret_cast<const uint2*>(xg + at);
        st_[0][i] = __uint_as_float(v.x << 16);
        st_[1][i] = __uint_as_float(v.x & 0xffff0000u);
        st_[2][i] = __uint_as_float(v.y << 16);
        st_[3][i] = __uint_as_float(v.y & 0xffff0000u);
      } else {
#pragma unroll
        for (int c = 0; c < WC; ++c)
          st_[c][i] = x[(size_t(b) * cin + min(c0 + c, cin - 1)) * plane_src + size_t(cy) * ws + cx];
      }
    }
    // this thread's 4 pixels of the CO gradient planes (zero outside the image)
    float g[CO][4];
    const int oy = y0 + row, ox = x0 + col4;
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      const size_t base = (size_t(b) * CO + o) * plane_out + size_t(min(oy, H - 1)) * W;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool ok = oy < H && ox + q < W;
        const size_t at = base + min(ox + q, W - 1);
        float v = dy[at];
        if (ygate) v = ygate[at] > 0.f ? v : 0.f;
        g[o][q] = ok ? v : 0.f;
      }
    }
    // the bias gradient sums the gradient as it is; the weight gradient of the bf16 mode multiplies bf16 operands
    // (XG8 = that mode's entry point): dy is rounded to nearest-even here, x already is bf16
#pragma unroll
    for (int o = 0; o < CO; ++o) bacc[o] += (g[o][0] + g[o][1]) + (g[o][2] + g[o][3]);
    if constexpr (XG8) {
#pragma unroll
      for (int o = 0; o < CO; ++o) {
#pragma unroll
        for (int q = 0; q < 4; ++q) g[o][q] = float((__bf16)g[o][q]);
      }
    }
    __syncthreads();                                  // previous tile's reads are done
#pragma unroll
    for (int c = 0; c < WC; ++c)
#pragma unroll
      for (int i = 0; i < BNP; ++i)
        if (BHR * BHC % 256 == 0 || tid + i * 256 < BHR * BHC)
          xs[c * (BHR * BLD) + loff[i]] = __uint_as_float(__float_as_uint(st_[c][i]) & smask[i]);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < WC; ++c) {
      float v[3][6];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float* rp = xs + c * (BHR * BLD) + (row + r) * BLD + col4;
        const float4 a = *reinterpret_cast<const float4*>(rp);
        const float2 e = *reinterpret_cast<const float2*>(rp + 4);
        v[r][0] = a.x; v[r][1] = a.y; v[r][2] = a.z; v[r][3] = a.w; v[r][4] = e.x; v[r][5] = e.y;
      }
#pragma unroll
      for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            acc[c][o][tap] = fmaf(g[o][q], v[tap / 3][q + tap % 3], acc[c][o][tap]);
    }
  }

  // block reduction: 64-lane shuffle tree per value, then the 4 wave sums through LDS (fixed order)
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int c = 0; c < WC; ++c)
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float v = st::wave_sum(acc[c][o][tap]);
        if (lane == 0) red[wave][(c * CO + o) * 9 + tap] = v;
      }
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    const float v = st::wave_sum(bacc[o]);
    if (lane == 0) red[wave][WC * CO * 9 + o] = v;
  }
  __syncthreads();
  if (tid < WC * CO * 9) {
    const int c = tid / (CO * 9), o = (tid / 9) % CO, tap = tid % 9;
    const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    if (c0 + c < cin) slabs[(size_t(split) * CO + o) * cin * 9 + size_t(c0 + c) * 9 + tap] = v;
  } else if (tid < WC * CO * 9 + CO && bias_slabs && blockIdx.x == 0) {
    bias_slabs[size_t(split) * CO + (tid - WC * CO * 9)] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
  }
}

}  // namespace

namespace st {

// true when one of the small-channel kernels takes the layer (then *rc is its launch status)
bool conv3x3_small_fwd(const float* x, const float* w_packed, const float* bias, float* y, int n, int cin,
                       int h, int w, int cout, int cout_pad, int pad_zero, int relu, hipStream_t s, int* rc) {
  if (cin == 3 && cout <= ACO && cout % 16 == 0) {
    const int tiles_x = cdiv(w, 32), tiles_y = cdiv(h, 16);
    const long long blocks = (long long)n * tiles_x * tiles_y;
    const int mtx = cdiv(w, MTW), mty = cdiv(h, MTH);
    const long long mblocks = (long long)n * mtx * mty;
    if (blocks > 0x7fffffffLL || mblocks > 0x7fffffffLL) return false;
    // ST_CIN3_VALU=1: the vector-ALU form (A/B measurements)
    if (getenv("ST_CIN3_VALU") == nullptr && w % 4 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0)
      hipLaunchKernelGGL(conv3x3_cin3_mfma_kernel, dim3((unsigned)std::min<long long>(mblocks, 1024)), dim3(256), 0, s, x,
                         w_packed, bias, y, h, w, cout, cout_pad, pad_zero, relu, mtx, mty, (int)mblocks);
    else
      hipLaunchKernelGGL(conv3x3_cin_small_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, s, x, w_packed, bias, y,
                         h, w, cout, cout_pad, pad_zero, relu, tiles_x, tiles_y);
    *rc = check_launch("st_conv3x3_fwd(cin=3)");
    return true;
  }
  // Row-strip kernel (w % 4 == 0, 16-B aligned, one image below 4 GB): 8-row strips when they make at least two
  // waves per SIMD (2 M pixels), 2-row strips down to 256 K pixels (B=8 at 256 x 256: 55 us against 58 us), the
  // LDS-tiled kernel below that and for every other shape.
  const long long pixels = (long long)n * h * w;
  if (cout == 3 && cin % KC == 0 && cin <= 512 && cout_pad >= 4 && w % 4 == 0 && w >= 64 &&
      reinterpret_cast<uintptr_t>(x) % 16 == 0 && reinterpret_cast<uintptr_t>(y) % 16 == 0 &&
      (long long)cin * h * w * 4 < (1LL << 32) && (long long)n * cdiv(w, 64) * cdiv(h, 8) < 0x7fffffffLL &&
      pixels >= (1 << 18)) {
    if (pixels >= (1 << 21)) *rc = launch_cout_strip_w<8>(x, w_packed, bias, y, n, cin, h, w, cout_pad, pad_zero, relu, s);
    else *rc = launch_cout_strip_w<2>(x, w_packed, bias, y, n, cin, h, w, cout_pad, pad_zero, relu, s);
    return true;
  }
  if (cout == 3 && cin % KC == 0 && cin <= 512 && cout_pad >= 4) {
    const int tiles_x = cdiv(w, FTW), tiles_y = cdiv(h, FTH);
    const long long blocks = (long long)n * tiles_x * tiles_y;
    if (blocks > 0x7fffffffLL) return false;
    static st::DynLdsOptIn optin;      // weights of up to 512 channels: 72 KB of dynamic LDS
    *rc = st::allow_dynamic_lds(conv3x3_cout_small_kernel<3>, WS_LDS_MAX, optin, "st_conv3x3_fwd(cout=3)");
    if (*rc != ST_OK) return true;
    hipLaunchKernelGGL(conv3x3_cout_small_kernel<3>, dim3((unsigned)blocks), dim3(256), size_t(cin) * 9 * 4 * sizeof(float),
                       s, x, w_packed, bias, y,
                       cin, h, w, cout_pad, pad_zero, relu, tiles_x, tiles_y);
    *rc = check_launch("st_conv3x3_fwd(cout=3)");
    return true;
  }
  return false;
}

// Weight / bias gradient of a cout == 3 layer.  Plan: *nsplit partial slabs [nsplit][3][cin*9] (+ [nsplit][3] bias
// slabs) that the caller sums with its fixed-order slab reduction.
int wgrad_small_nsplit(int n, int cout, int cin, int H, int W) {
  if (cout != 3) return 0;
  const long long ntiles = (long long)n * cdiv(W, BTW) * cdiv(H, BTH);
  if (ntiles > 0x7fffffffLL) return 0;
  const int chunks = cdiv(cin, WC);
  int nsplit = (int)std::max<long long>(1, std::min<long long>(ntiles, cdiv(1024, chunks)));
  const int per = cdiv((int)ntiles, nsplit);
  return cdiv((int)ntiles, per);
}

int wgrad_small_launch(const float* x, const float* dy, const float* ygate, float* slabs, float* bias_slabs, int n,
                       int cin, int hs, int ws, int H, int W, int up, int pad_zero, hipStream_t s, bool x_g8) {
  const int tiles_x = cdiv(W, BTW), tiles_y = cdiv(H, BTH);
  const int ntiles = n * tiles_x * tiles_y;
  const int nsplit = wgrad_small_nsplit(n, 3, cin, H, W);
  const int per = cdiv(ntiles, nsplit);
  if (x_g8)     // cin % 8 == 0 checked by the caller
    hipLaunchKernelGGL((wgrad_cout_small_kernel<3, true>), dim3(cdiv(cin, WC), nsplit), dim3(256), 0, s, x, dy, ygate,
                       slabs, bias_slabs, n, cin, hs, ws, H, W, up, pad_zero, tiles_x, tiles_y, ntiles, per);
  else
    hipLaunchKernelGGL((wgrad_cout_small_kernel<3, false>), dim3(cdiv(cin, WC), nsplit), dim3(256), 0, s, x, dy, ygate,
                       slabs, bias_slabs, n, cin, hs, ws, H, W, up, pad_zero, tiles_x, tiles_y, ntiles, per);
  return check_launch("st_conv3x3_wgrad(cout=3)");
}

}  // namespace st
