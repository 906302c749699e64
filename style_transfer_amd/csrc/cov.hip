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

template <int T>
__global__ __launch_bounds__(NT) void cov_fwd_kernel(const float* __restrict__ x,
                                                    const float* __restrict__ mean,
                                                    float* __restrict__ slabs, int B, int C, int n,
                                                    int chunks_per_split) {
  constexpr int MT = T / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xi = smem;
  float* Xj = smem + T * LSF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles = (C + T - 1) / T;
  const int i0 = (blockIdx.x / tiles) * T, j0 = (blockIdx.x % tiles) * T;
  const int split = blockIdx.y, b = blockIdx.z;
  const float* xb = x + (size_t)b * C * n;
  const float* mb = mean + (size_t)b * C;
  const int nchunks = (n + KN - 1) / KN;
  const int c_begin = split * chunks_per_split, c_end = min(c_begin + chunks_per_split, nchunks);

  f32x16 acc[MT][MT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int c = 0; c < MT; ++c)
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[a][c][k] = 0.f;

  // staging: wave w loads rows w, w+4, ...; lane = pixel of the chunk (coalesced 256-B rows)
  for (int ch = c_begin; ch < c_end; ++ch) {
    const int p = ch * KN + lane;
    const bool pok = p < n;
    const int pc = min(p, n - 1);
    float vi[T / 4], vj[T / 4];
#pragma unroll
    for (int r = 0; r < T / 4; ++r) {
      const int row = wave + 4 * r;
      const int gi = min(i0 + row, C - 1), gj = min(j0 + row, C - 1);
      vi[r] = xb[(size_t)gi * n + pc] - mb[gi];
      vj[r] = xb[(size_t)gj * n + pc] - mb[gj];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < T / 4; ++r) {
      const int row = wave + 4 * r;
      Xi[row * LSF + lane] = (pok && i0 + row < C) ? vi[r] : 0.f;
      Xj[row * LSF + lane] = (pok && j0 + row < C) ? vj[r] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int s = 0; s < KN / 2; ++s) {
      float a[MT], bb[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        a[m] = Xi[(wm * MT * 32 + m * 32 + li) * LSF + 2 * s + lh];
        bb[m] = Xj[(wn * MT * 32 + m * 32 + li) * LSF + 2 * s + lh];
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < MT; ++q)
          acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bb[q], acc[m][q], 0, 0, 0);
    }
  }
  float* slab = slabs + ((size_t)split * B + b) * C * C;
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

__global__ __launch_bounds__(256) void cov_reduce_kernel(const float* __restrict__ slabs,
                                                        float* __restrict__ out, long long n, int nsplit) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n;
       e += (long long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += slabs[(size_t)k * n + e];
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
  const int base = B * p.tiles * p.tiles;
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
  const Plan p = plan_for(b, c, n);
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(p.tiles * p.tiles, p.nsplit, b);
  if (p.T == 64) {
    hipLaunchKernelGGL(cov_fwd_kernel<64>, grid, dim3(NT), 2 * 64 * LSF * sizeof(float), s, x, mean,
                       workspace, b, c, n, p.chunks_per_split);
  } else {
    constexpr int lds = 2 * 128 * LSF * sizeof(float);
    ST_LAUNCH_DYN_LDS("st_cov_fwd", cov_fwd_kernel<128>, grid, dim3(NT), lds, s, x, mean, workspace, b, c, n,
                      p.chunks_per_split);
  }
  const long long total = (long long)b * c * c;
  hipLaunchKernelGGL(cov_reduce_kernel, dim3((int)std::min<long long>(st::cdivll(total, 256), 2048)),
                     dim3(256), 0, s, workspace, cov, total, p.nsplit);
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
