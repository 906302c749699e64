// Un-normalised channel covariance ("Gram" term) of VincentStyleLoss.calc_2_moments
// (losses.py:49-54) and its gradient, as fp32 MFMA GEMMs for gfx950.
//
//   forward   cov[b] = Xc Xc^T,  Xc = X[b] - mean (C x n, n = H*W)   -> C x C,   K = pixels
//   backward  dX[b]  = (G + G^T) Xc + gmean / n                      -> C x n,   K = channels
//
// Forward: a block owns a T x T tile of cov (T = 128, or 64 when C <= 64) and a contiguous pixel
// range (split-K; partial slabs are summed in a fixed order: deterministic, no atomics).  Rows are
// mean-centred while they are staged into LDS (exactly the reference's `x - mu` operand, no
// E[xy] - E[x]E[y] cancellation); both MFMA operands are rows of the same staged slab, read with an
// odd row stride so the 32 channel lanes hit 32 banks.
#include <algorithm>
#include <cstdint>

#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int KN = 64;        // pixels per forward chunk
constexpr int LSF = KN + 1;   // odd LDS row stride (forward)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Tile pair (ti <= tj) of the upper triangle for a linear index: cov is symmetric, the strictly lower tiles are
// never computed (C = 512: 10 of 16 tiles) -- cov_reduce_kernel mirrors them.
__device__ __forceinline__ void tri_tile(int idx, int tiles, int& ti, int& tj) {
  int row = 0, left = idx;
  while (left >= tiles - row) { left -= tiles - row; ++row; }
  ti = row;
  tj = row + left;
}

// One T x T tile over the block's pixel range.  DIAG (both operands are the same rows: staged once) is a template
// parameter: as a run-time condition around the second operand's loads it made hipcc keep both staging arrays in
// SCRATCH behind a dynamic offset, with a wait and a scratch store after every global load.
template <int T, bool DIAG>
__device__ __forceinline__ void cov_tile(const float* __restrict__ xb, const float* __restrict__ mb,
                                         float* __restrict__ slab, float* smem, int C, int n, int i0, int j0,
                                         int c_begin, int c_end) {
  constexpr int MT = T / 64;
  constexpr int R = T / 4;      // rows each wave stages per operand
  float* Xi = smem;
  float* Xj = smem + T * LSF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const float* Xjr = DIAG ? Xi : Xj;

  f32x16 acc[MT][MT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int c = 0; c < MT; ++c)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[a][c][k] = 0.f;

  // per-row constants of this wave's staging rows (wave w stages rows w, w+4, ...; lane = pixel of the chunk).
  // Row addresses are rebuilt from 32-bit element offsets at every load (two integer ops against 64 MFMA cycles):
  // 2 R row POINTERS took 128 registers and left one wave per SIMD.
  // the rows' means wait in LDS (2 T floats): as registers they cost 2 R = 64 of the 256 a wave may have here
  float* Mi = smem + 2 * T * LSF;
  float* Mj = Mi + T;
  if (tid < T) {
    Mi[tid] = mb[min(i0 + tid, C - 1)];
    Mj[tid] = mb[min(j0 + tid, C - 1)];
  }
  // Buffer loads: ONE per-lane offset (the chunk's pixel, clamped to n - 1) and the row (clamped to C - 1, wave-uniform)
  // in the scalar offset, so no per-row address registers exist (as hoisted loop invariants they cost 64-128
  // registers and spilled).  The hardware range check covers the per-lane offset only, NOT the scalar offset, so
  // both clamps are explicit: every address lies inside this image's C x n floats; the clamped duplicates are
  // masked at the LDS store.
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, int(unsigned(size_t(C) * n * sizeof(float))), 0x00020000);
  const unsigned row_bytes = unsigned(n) * 4u;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  float vi[R], vj[DIAG ? 1 : R];
  auto load = [&](int ch) {                       // coalesced 256-B row segments
    const int vo = int(unsigned(min(ch * KN + lane, n - 1)) * 4u);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int ri = min(i0 + wave_u + 4 * r, C - 1), rj = min(j0 + wave_u + 4 * r, C - 1);
      vi[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, int(unsigned(ri) * row_bytes), 0));
      if (!DIAG)
        vj[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, int(unsigned(rj) * row_bytes), 0));
    }
  };
  if (c_begin < c_end) load(c_begin);
  for (int ch = c_begin; ch < c_end; ++ch) {
    const bool pok = ch * KN + lane < n;
    __syncthreads();                               // the previous chunk's MFMA reads are done
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = wave + 4 * r;
      Xi[row * LSF + lane] = (pok && i0 + row < C) ? vi[r] - Mi[row] : 0.f;    // exactly the reference's x - mu
      if (!DIAG) Xj[row * LSF + lane] = (pok && j0 + row < C) ? vj[r] - Mj[row] : 0.f;
    }
    __syncthreads();
    if (ch + 1 < c_end) load(ch + 1);              // next chunk's global loads fly under this chunk's MFMAs
#pragma unroll 8
    for (int s = 0; s < KN / 2; ++s) {
      float a[MT], bb[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        a[m] = Xi[(wm * MT * 32 + m * 32 + li) * LSF + 2 * s + lh];
        bb[m] = Xjr[(wn * MT * 32 + m * 32 + li) * LSF + 2 * s + lh];
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < MT; ++q)
          acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bb[q], acc[m][q], 0, 0, 0);
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < MT; ++q) {
      const int j = j0 + wn * MT * 32 + q * 32 + li;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int i = i0 + wm * MT * 32 + m * 32 + (k & 3) + 8 * (k >> 2) + 4 * lh;
        if (i < C && j < C) slab[(size_t)i * C + j] = acc[m][q][k];
      }
    }
}

template <int T>
__global__ __launch_bounds__(NT, 2) void cov_fwd_kernel(const float* __restrict__ x,
                                                       const float* __restrict__ mean,
                                                       float* __restrict__ slabs, int B, int C, int n,
                                                       int chunks_per_split) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tiles = (C + T - 1) / T;
  int ti, tj;
  tri_tile(blockIdx.x, tiles, ti, tj);
  const int split = blockIdx.y, b = blockIdx.z;
  const float* xb = x + (size_t)b * C * n;
  const float* mb = mean + (size_t)b * C;
  const int nchunks = (n + KN - 1) / KN;
  const int c_begin = split * chunks_per_split, c_end = min(c_begin + chunks_per_split, nchunks);
  float* slab = slabs + ((size_t)split * B + b) * C * C;
  if (ti == tj) cov_tile<T, true>(xb, mb, slab, smem, C, n, ti * T, tj * T, c_begin, c_end);
  else cov_tile<T, false>(xb, mb, slab, smem, C, n, ti * T, tj * T, c_begin, c_end);
}

// Fixed-order sum of the split-K slabs; element (i, j) of a strictly-lower TILE is read from (j, i) of the upper
// tile that was computed (within a diagonal tile both triangles exist and are bit-identical: the same products in
// the same order with the factors swapped).  cov comes out exactly symmetric.
__global__ __launch_bounds__(256) void cov_reduce_kernel(const float* __restrict__ slabs,
                                                        float* __restrict__ out, int B, int C, int T, int nsplit) {
  const long long cc = (long long)C * C, total = cc * B;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long bb = e / cc;
    const int r = int(e - bb * cc);
    const int i = r / C, j = r - i * C;
    const long long src = (i / T <= j / T) ? (long long)i * C + j : (long long)j * C + i;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += slabs[((size_t)k * B + bb) * cc + src];
    out[e] = s;
  }
}

// ---- backward: dX = (G + G^T) Xc + gmean / n ---------------------------------------------------
constexpr int BI = 128;       // output channel rows per block
constexpr int BN = 128;       // pixels per block
constexpr int KJ = 32;        // reduction channels per chunk
constexpr int LSB = KJ + 1;

__global__ __launch_bounds__(NT) void cov_bwd_kernel(const float* __restrict__ x,
                                                    const float* __restrict__ mean,
                                                    const float* __restrict__ gcov,
                                                    const float* __restrict__ gmean,
                                                    float* __restrict__ dx, int C, int n) {
  __shared__ float Ss[BI * LSB];   // [i][j chunk]  S = G + G^T
  __shared__ float Xs[KJ * BN];    // [j chunk][pixel]  centred
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * BN, i0 = blockIdx.y * BI, b = blockIdx.z;
  const float* xb = x + (size_t)b * C * n;
  const float* mb = mean + (size_t)b * C;
  const float* gb = gcov + (size_t)b * C * C;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[a][c][k] = 0.f;

  for (int j0 = 0; j0 < C; j0 += KJ) {
    // S tile: thread -> (row = tid/2 + {0}, 16 columns)   128 rows x 32 cols = 4096 / 256 = 16 each
    float sv[16];
    {
      const int row = tid >> 1, cb = (tid & 1) * 16;
      const int gi = min(i0 + row, C - 1);
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int gj = min(j0 + cb + k, C - 1);
        sv[k] = gb[(size_t)gi * C + gj] + gb[(size_t)gj * C + gi];
      }
    }
    // X tile: wave w -> rows w, w+4, ... (8 rows), lanes -> 2 x 64 pixels
    float xv[KJ / 4][2];
#pragma unroll
    for (int r = 0; r < KJ / 4; ++r) {
      const int gj = min(j0 + wave + 4 * r, C - 1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int p = min(n0 + h * 64 + lane, n - 1);
        xv[r][h] = xb[(size_t)gj * n + p] - mb[gj];
      }
    }
    __syncthreads();
    {
      const int row = tid >> 1, cb = (tid & 1) * 16;
#pragma unroll
      for (int k = 0; k < 16; ++k)
        Ss[row * LSB + cb + k] = (i0 + row < C && j0 + cb + k < C) ? sv[k] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < KJ / 4; ++r) {
      const int jr = wave + 4 * r;
#pragma unroll
      for (int h = 0; h < 2; ++h) Xs[jr * BN + h * 64 + lane] = (j0 + jr < C) ? xv[r][h] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KJ / 2; ++s) {
      float a[2], bb[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        a[m] = Ss[(wm * 64 + m * 32 + li) * LSB + 2 * s + lh];
        bb[m] = Xs[(2 * s + lh) * BN + wn * 64 + m * 32 + li];
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bb[q], acc[m][q], 0, 0, 0);
    }
  }
  const float inv_n = 1.f / float(n);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = i0 + wm * 64 + m * 32 + (k & 3) + 8 * (k >> 2) + 4 * lh;
      if (i >= C) continue;
      const float gm = gmean ? gmean[(size_t)b * C + i] * inv_n : 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int p = n0 + wn * 64 + q * 32 + li;
        if (p < n) dx[((size_t)b * C + i) * n + p] = acc[m][q][k] + gm;
      }
    }
}

struct Plan {
  int T, tiles, nsplit, chunks_per_split;
};
Plan plan_for(int B, int C, int n) {
  Plan p;
  p.T = C <= 64 ? 64 : 128;
  p.tiles = st::cdiv(C, p.T);
  const int nchunks = st::cdiv(n, KN);
  const int base = B * (p.tiles * (p.tiles + 1) / 2);      // upper-triangular tile pairs only
  int want = std::max(1, 1024 / std::max(1, base));
  p.nsplit = std::max(1, std::min(want, nchunks));
  p.chunks_per_split = st::cdiv(nchunks, p.nsplit);
  p.nsplit = st::cdiv(nchunks, p.chunks_per_split);
  return p;
}

}  // namespace

extern "C" size_t st_cov_workspace_bytes(int b, int c, int n) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  const Plan p = plan_for(b, c, n);
  return size_t(p.nsplit) * b * c * c * sizeof(float);
}

extern "C" int st_cov_fwd(const float* x, const float* mean, float* cov, float* workspace, int b, int c,
                          int n, st_stream_t stream) {
  ST_REQUIRE(x && mean && cov && workspace, "st_cov_fwd: null pointer");
  ST_REQUIRE(b > 0 && c > 0 && n > 0, "st_cov_fwd: bad dims b=%d c=%d n=%d", b, c, n);
  ST_REQUIRE((long long)c * n * 4 < (1LL << 31), "st_cov_fwd: one image of %d x %d floats exceeds the 2 GB a buffer offset spans", c, n);
  const Plan p = plan_for(b, c, n);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(p.tiles * (p.tiles + 1) / 2, p.nsplit, b);
  if (p.T == 64) {
    hipLaunchKernelGGL(cov_fwd_kernel<64>, grid, dim3(NT), (2 * 64 * LSF + 2 * 64) * sizeof(float), s, x, mean,
                       workspace, b, c, n, p.chunks_per_split);
  } else {
    constexpr int lds = (2 * 128 * LSF + 2 * 128) * sizeof(float);
    ST_LAUNCH_DYN_LDS("st_cov_fwd", cov_fwd_kernel<128>, grid, dim3(NT), lds, s, x, mean, workspace, b, c, n,
                      p.chunks_per_split);
  }
  const long long total = (long long)b * c * c;
  hipLaunchKernelGGL(cov_reduce_kernel, dim3((int)std::min<long long>(st::cdivll(total, 256), 2048)),
                     dim3(256), 0, s, workspace, cov, b, c, p.T, p.nsplit);
  return st::check_launch("st_cov_fwd");
}

extern "C" int st_cov_bwd(const float* x, const float* mean, const float* gcov, const float* gmean,
                          float* dx, int b, int c, int n, st_stream_t stream) {
  ST_REQUIRE(x && mean && gcov && dx, "st_cov_bwd: null pointer");
  ST_REQUIRE(b > 0 && c > 0 && n > 0, "st_cov_bwd: bad dims b=%d c=%d n=%d", b, c, n);
  dim3 grid(st::cdiv(n, BN), st::cdiv(c, BI), b);
  hipLaunchKernelGGL(cov_bwd_kernel, grid, dim3(NT), 0, (hipStream_t)stream, x, mean, gcov, gmean, dx, c, n);
  return st::check_launch("st_cov_bwd");
}
