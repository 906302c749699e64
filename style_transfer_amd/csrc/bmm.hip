// Batched square fp32 matrix products on the fp32 matrix pipe for gfx950 (MI355X): the C x C side of
// VincentStyleLoss.l2wass_dist (src/models/losses.py:67-81 -- root_cov_stl = V sqrt(L) V^T, root @ cov_synth @ root,
// their gradients, V diag(g) V^T of the eigenvalue backward) and the coupled Newton-Schulz iteration that replaces
// eigh(cov_stl) (style_transfer_amd/losses.py).  SURVEY.md K7b: "matmuls -> MFMA".
//
//   C[b] = alpha * op(A[b]) . op(B[b]) + diag * I          n x n, row-major, n % 64 == 0, op = identity | transpose
//
//   * v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation in k order (an fmaf chain per output element).
//   * A block (4 waves) owns a BT x BT tile of one matrix (BT = 128, or 64 when 128-tiles would leave CUs idle);
//     wave w owns the (BT/2) x (BT/2) quadrant (w / 2, w % 2): (BT/64)^2 accumulator tiles.
//   * Both operands are staged k-major in LDS -- As[k][m], Bs[k][n], row stride BT + 4 floats -- so that every
//     operand fragment is one ds_read_b32 with 32 consecutive floats per half-wave (conflict-free; the +4 keeps the
//     16-byte vector stores of the straight copy aligned and moves the two k rows a wave reads to different banks).
//     An operand whose contiguous dimension is k (A not transposed, B transposed) is transposed on the way in: a
//     thread loads groups of 4 consecutive k of one row (16-byte loads) and writes scalars; the other form is a straight
//     copy of 16-byte vectors.
//   * K chunks of 32, the next chunk's global loads in flight during the current chunk's MFMAs.
#include <cstdint>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 32;        // k per chunk
constexpr int NT = 256;       // 4 waves

struct BmmParams {
  const float* a;
  const float* b;
  float* c;
  int n;
  int ta, tb;                 // operand stored transposed
  float alpha, diag;
};

// one operand tile (BT rows of m, KC of k) -> registers.  kmaj: the stored matrix has k as its contiguous dimension
// (element (m, k) at src[m * n + k]); else m is contiguous (element at src[k * n + m]).
template <int BT>
struct Stage {
  static constexpr int NV = BT * KC / 4 / NT;    // 16-byte vectors per thread (4 for BT = 128, 2 for BT = 64)
  f32x4 v[NV];
};

template <int BT>
__device__ __forceinline__ void load_tile(Stage<BT>& s, const float* __restrict__ src, int n, bool kmaj, int m0, int k0,
                                          int tid) {
#pragma unroll
  for (int i = 0; i < Stage<BT>::NV; ++i) {
    const int e = tid + i * NT;                   // vector index
    if (kmaj) {
      const int m = e % BT, kq = e / BT;          // kq: which group of 4 k
      s.v[i] = *reinterpret_cast<const f32x4*>(src + size_t(m0 + m) * n + k0 + kq * 4);
    } else {
      const int mq = e % (BT / 4), k = e / (BT / 4);
      s.v[i] = *reinterpret_cast<const f32x4*>(src + size_t(k0 + k) * n + m0 + mq * 4);
    }
  }
}

template <int BT>
__device__ __forceinline__ void store_tile(const Stage<BT>& s, float* __restrict__ dst, bool kmaj, int tid) {
  constexpr int LD = BT + 4;
#pragma unroll
  for (int i = 0; i < Stage<BT>::NV; ++i) {
    const int e = tid + i * NT;
    if (kmaj) {
      const int m = e % BT, kq = e / BT;
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[(kq * 4 + j) * LD + m] = s.v[i][j];
    } else {
      const int mq = e % (BT / 4), k = e / (BT / 4);
      *reinterpret_cast<f32x4*>(dst + k * LD + mq * 4) = s.v[i];
    }
  }
}

template <int BT>
__global__ __launch_bounds__(NT) void bmm_f32_kernel(BmmParams p) {
  constexpr int LD = BT + 4;
  constexpr int WT = BT / 64;                     // accumulator tiles per wave and dimension
  __shared__ __attribute__((aligned(16))) float As[KC * LD];
  __shared__ __attribute__((aligned(16))) float Bs[KC * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int tiles = p.n / BT;
  const int tn = blockIdx.x % tiles, tm = (blockIdx.x / tiles) % tiles, b = blockIdx.x / (tiles * tiles);
  const int m0 = tm * BT, n0 = tn * BT;
  const size_t mat = size_t(p.n) * p.n;
  const float* A = p.a + b * mat;
  const float* B = p.b + b * mat;
  // op(A)[m][k]: stored [m][k] (k contiguous) unless ta; op(B)[k][n]: stored [k][n] (n contiguous) unless tb
  const bool a_kmaj = !p.ta, b_kmaj = p.tb != 0;

  f32x16 acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Stage<BT> sa, sb;
  load_tile<BT>(sa, A, p.n, a_kmaj, m0, 0, tid);
  load_tile<BT>(sb, B, p.n, b_kmaj, n0, 0, tid);
  const int wm = (wave >> 1) * (BT / 2), wn = (wave & 1) * (BT / 2);
  for (int k0 = 0; k0 < p.n; k0 += KC) {
    __syncthreads();                              // the previous chunk's reads are done
    store_tile<BT>(sa, As, a_kmaj, tid);
    store_tile<BT>(sb, Bs, b_kmaj, tid);
    __syncthreads();
    if (k0 + KC < p.n) {
      load_tile<BT>(sa, A, p.n, a_kmaj, m0, k0 + KC, tid);
      load_tile<BT>(sb, B, p.n, b_kmaj, n0, k0 + KC, tid);
    }
#pragma unroll
    for (int s = 0; s < KC / 2; ++s) {
      float av[WT], bv[WT];
#pragma unroll
      for (int i = 0; i < WT; ++i) av[i] = As[(2 * s + lh) * LD + wm + i * 32 + li];
#pragma unroll
      for (int j = 0; j < WT; ++j) bv[j] = Bs[(2 * s + lh) * LD + wn + j * 32 + li];
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }

  // accumulator register r of lane (li, lh): row 8 * (r / 4) + 4 * lh + r % 4, column li of the 32 x 32 tile
  float* C = p.c + b * mat;
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const int col = n0 + wn + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + i * 32 + 8 * (r >> 2) + 4 * lh + (r & 3);
        float v = p.alpha * acc[i][j][r];
        if (row == col) v += p.diag;
        C[size_t(row) * p.n + col] = v;
      }
    }
}

}  // namespace

extern "C" int st_bmm_f32(const float* a, const float* b, float* c, int batch, int n, int trans_a, int trans_b,
                          float alpha, float diag, st_stream_t stream) {
  ST_REQUIRE(a && b && c, "st_bmm_f32: null pointer");
  ST_REQUIRE(batch > 0 && n > 0 && n % 64 == 0, "st_bmm_f32: batch=%d n=%d (n must be a multiple of 64)", batch, n);
  ST_REQUIRE(c != a && c != b, "st_bmm_f32: the result may not alias an operand");
  ST_REQUIRE(reinterpret_cast<uintptr_t>(a) % 16 == 0 && reinterpret_cast<uintptr_t>(b) % 16 == 0,
             "st_bmm_f32: operands must be 16-byte aligned");
  ST_REQUIRE((long long)batch * (n / 64) * (n / 64) <= 0x7fffffffLL, "st_bmm_f32: too many tiles");
  BmmParams p{a, b, c, n, trans_a ? 1 : 0, trans_b ? 1 : 0, alpha, diag};
  hipStream_t s = (hipStream_t)stream;
  // 128-tiles when they give every CU a block (or n is large), else 64-tiles: [16, 512, 512] -> 256 blocks of 128,
  // [16, 256, 256] -> 256 blocks of 64
  const long long big = (long long)batch * (n / 128) * (n / 128);
  if (n % 128 == 0 && big >= 192)
    hipLaunchKernelGGL(bmm_f32_kernel<128>, dim3((unsigned)big), dim3(NT), 0, s, p);
  else
    hipLaunchKernelGGL(bmm_f32_kernel<64>, dim3((unsigned)(batch * (n / 64) * (n / 64))), dim3(NT), 0, s, p);
  return st::check_launch("st_bmm_f32");
}
