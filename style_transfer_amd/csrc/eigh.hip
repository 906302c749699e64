// Batched symmetric eigendecomposition of SMALL matrices (even C <= 128), one workgroup per matrix, everything in LDS:
// the eigh / eigvalsh calls of VincentStyleLoss.l2wass_dist (losses.py:69,73,78) for the 64- and 128-channel style
// layers (relu1_2, relu2_2).  rocSOLVER's syevd runs those sizes as a chain of hundreds of tiny launches (4.3 ms for
// [8,64,64], 2.0 ms for [8,128,128]); a 64x64 matrix and its eigenvector matrix are 33 KB -- they fit one CU's LDS.
//
// Algorithm: two-sided cyclic Jacobi with the round-robin ("chess tournament") parallel ordering: a sweep is C-1
// rounds, each round rotates C/2 DISJOINT index pairs at once --
//   (1) one thread per pair computes the rotation (c, s) that zeroes a_pq (Rutishauser's stable formulas);
//   (2) A <- A J and V <- V J   (column pairs; every pair touches its own two columns);
//   (3) A <- J^T A              (row pairs);
// three barriers per round, no global memory traffic until the end.  Converged when a whole sweep rotates nothing
// (|a_pq| negligible against both diagonal entries), at most `max_sweeps` sweeps.  Eigenvalues are sorted ascending
// (rank by counting) and the eigenvector columns permuted with them, like LAPACK's syevd.
#include <cstdlib>

#include "common.h"

namespace {

template <int CMAX>
struct EighCfg {
  static constexpr int LD = CMAX + 1;                 // odd stride: a column walk hits all banks
  static constexpr int NT = CMAX >= 128 ? 1024 : 256;
  static constexpr size_t LDS_BYTES = size_t(2 * CMAX * LD + 4 * CMAX) * sizeof(float) + CMAX * sizeof(int);
};

template <int CMAX>
__global__ __launch_bounds__(EighCfg<CMAX>::NT) void syevj_small_kernel(const float* __restrict__ a_in,
                                                                        float* __restrict__ w_out,
                                                                        float* __restrict__ v_out,
                                                                        int* __restrict__ sweeps_out, int c,
                                                                        int max_sweeps, float abs_tol) {
  using Cfg = EighCfg<CMAX>;
  constexpr int LD = Cfg::LD, NT = Cfg::NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* A = smem;                       // [c][LD]
  float* V = A + CMAX * LD;              // [c][LD]
  float* cs = V + CMAX * LD;             // [c/2] cosines
  float* sn = cs + CMAX;                 // [c/2] sines
  float* wv = sn + CMAX;                 // [c] eigenvalues (final)
  float* flag = wv + CMAX;               // [0]: rotations in this sweep
  int* rank = reinterpret_cast<int*>(flag + CMAX);  // [c] final ordering
  const int tid = threadIdx.x;
  const float* ab = a_in + (size_t)blockIdx.x * c * c;

  // load, symmetrised (the caller's matrix may differ in the last ulp between the triangles); V = I
  for (int e = tid; e < c * c; e += NT) {
    const int i = e / c, j = e - i * c;
    A[i * LD + j] = 0.5f * (ab[e] + ab[j * c + i]);
    V[i * LD + j] = i == j ? 1.f : 0.f;
  }
  __syncthreads();
  // Absolute floor of the rotation test: abs_tol * ||A||_F.  The relative test alone (|a_pq| negligible against BOTH
  // diagonal entries) never ends on a matrix with a block of numerically-zero eigenvalues -- there the entries are
  // rounding noise of the large part and "small against a diagonal entry of 1e-3" is out of reach in fp32 -- the
  // kernel then burns all max_sweeps sweeps (measured on the Gram loss's root.cov.root: 2.7 ms instead of 1 ms, and
  // a "not converged" report for a decomposition that is as accurate as fp32 allows).
  {
    float ssq = 0.f;
    for (int e = tid; e < c * c; e += NT) {
      const int i = e / c, j = e - i * c;
      const float v = A[i * LD + j];
      ssq += v * v;
    }
    ssq = st::wave_sum(ssq);
    if ((tid & 63) == 0) cs[tid >> 6] = ssq;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int k = 0; k < NT / 64; ++k) t += cs[k];
      flag[1] = abs_tol * sqrtf(t);
    }
    __syncthreads();
  }
  const float floor_abs = flag[1];
  const int half = c >> 1;
  int* pp = rank;                        // [c/2] pair indices of the current round (rank is only used at the end)
  int* qq = reinterpret_cast<int*>(wv);  // [c/2]              "                    (wv likewise)
  // thread -> (slot, line): `line` = the row (step 2) / column (step 3) this thread updates, `slot` = which of the
  // NT / CMAX pairs of an iteration; no integer divisions in the loops
  constexpr int SLOTS = NT / CMAX;
  const int line = tid & (CMAX - 1), slot = tid / CMAX;
  int sweep = 0;
  bool converged = false;
  for (; sweep < max_sweeps; ++sweep) {
    if (tid == 0) flag[0] = 0.f;
    __syncthreads();
    for (int round = 0; round < c - 1; ++round) {
      // (1) Round-robin seating: seat 0 keeps player 0, the other c-1 players move one seat round the table every
      // round, so seat i >= 1 holds player ((i - 1 - round) mod (c - 1)) + 1; seat k plays seat c-1-k.
      if (tid < half) {
        int m = tid - 1 - round;                        // seat tid (>= 1 unless tid == 0)
        m += (m < 0) ? (c - 1) : 0;
        int p = tid == 0 ? 0 : m + 1;
        int m2 = c - 2 - tid - round;                   // seat c-1-tid
        m2 += (m2 < 0) ? (c - 1) : 0;
        int q = m2 + 1;
        if (p > q) { const int t = p; p = q; q = t; }
        const float app = A[p * LD + p], aqq = A[q * LD + q], apq = A[p * LD + q];
        float cc = 1.f, ss = 0.f;
        const float g = 100.f * fabsf(apq);
        const bool negligible = ((fabsf(app) + g == fabsf(app)) && (fabsf(aqq) + g == fabsf(aqq))) ||
                                fabsf(apq) <= floor_abs;
        if (apq != 0.f && !negligible) {
          const float theta = 0.5f * (aqq - app) / apq;
          const float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
          cc = 1.f / sqrtf(t * t + 1.f);
          ss = t * cc;
          flag[0] = 1.f;                  // benign race: every writer stores the same value
        }
        cs[tid] = cc;
        sn[tid] = ss;
        pp[tid] = p;
        qq[tid] = q;
      }
      __syncthreads();
      // (2) columns: A <- A J, V <- V J   (this thread: row `line` of the pairs slot, slot + SLOTS, ...)
      if (line < c) {
        for (int k = slot; k < half; k += SLOTS) {
          const float ss = sn[k];
          if (ss != 0.f) {
            const float cc = cs[k];
            const int p = pp[k], q = qq[k];
            const float x = A[line * LD + p], y = A[line * LD + q];
            const float vx = V[line * LD + p], vy = V[line * LD + q];
            A[line * LD + p] = cc * x - ss * y;
            A[line * LD + q] = ss * x + cc * y;
            V[line * LD + p] = cc * vx - ss * vy;
            V[line * LD + q] = ss * vx + cc * vy;
          }
        }
      }
      __syncthreads();
      // (3) rows: A <- J^T A   (this thread: column `line`)
      if (line < c) {
        for (int k = slot; k < half; k += SLOTS) {
          const float ss = sn[k];
          if (ss != 0.f) {
            const float cc = cs[k];
            const int p = pp[k], q = qq[k];
            const float x = A[p * LD + line], y = A[q * LD + line];
            // the rotated pair itself: exact zero off the diagonal
            A[p * LD + line] = (line == q) ? 0.f : cc * x - ss * y;
            A[q * LD + line] = (line == p) ? 0.f : ss * x + cc * y;
          }
        }
      }
      __syncthreads();
    }
    if (flag[0] == 0.f) { ++sweep; converged = true; break; }   // uniform: read after the round's barriers
    __syncthreads();
  }
  __syncthreads();
  // eigenvalues = diagonal; ascending order by rank counting (ties broken by index)
  if (tid < c) wv[tid] = A[tid * LD + tid];
  __syncthreads();
  if (tid < c) {
    const float mine = wv[tid];
    int r = 0;
    for (int j = 0; j < c; ++j) {
      const float o = wv[j];
      r += (o < mine || (o == mine && j < tid)) ? 1 : 0;
    }
    rank[tid] = r;
  }
  __syncthreads();
  if (tid < c) w_out[(size_t)blockIdx.x * c + rank[tid]] = wv[tid];
  float* vb = v_out + (size_t)blockIdx.x * c * c;
  for (int e = tid; e < c * c; e += NT) {           // eigenvector column j goes to column rank[j]
    const int i = e / c, j = e - i * c;
    vb[(size_t)i * c + rank[j]] = V[i * LD + j];
  }
  // sweeps used (the last, rotation-free one included), or max_sweeps + 1 if the loop ran out without one: a
  // decomposition that converges in its last permitted sweep reports max_sweeps and is NOT a failure
  if (sweeps_out && tid == 0) sweeps_out[blockIdx.x] = converged ? sweep : max_sweeps + 1;
}

}  // namespace

extern "C" int st_syevj_small_max_dim(void) { return 128; }

// off-diagonal entries at or below this fraction of ||A||_F are left alone (see the kernel)
constexpr float SYEVJ_ABS_TOL = 1e-8f;

extern "C" int st_syevj_small_batched(const float* a, float* w, float* v, int* sweeps, int batch, int c,
                                      int max_sweeps, st_stream_t stream) {
  ST_REQUIRE(a && w && v, "st_syevj_small_batched: null pointer");
  ST_REQUIRE(batch > 0 && c >= 2 && c <= 128 && c % 2 == 0,
             "st_syevj_small_batched: batch=%d, c=%d (supported: even c in [2, 128])", batch, c);
  ST_REQUIRE(max_sweeps > 0, "st_syevj_small_batched: max_sweeps=%d", max_sweeps);
  hipStream_t s = (hipStream_t)stream;
  static const float abs_tol = getenv("ST_SYEVJ_ABS_TOL") ? (float)atof(getenv("ST_SYEVJ_ABS_TOL")) : SYEVJ_ABS_TOL;
  if (c <= 64) {
    using Cfg = EighCfg<64>;
    ST_LAUNCH_DYN_LDS("st_syevj_small_batched", syevj_small_kernel<64>, dim3(batch), dim3(Cfg::NT), Cfg::LDS_BYTES, s,
                      a, w, v, sweeps, c, max_sweeps, abs_tol);
  } else {
    using Cfg = EighCfg<128>;
    ST_LAUNCH_DYN_LDS("st_syevj_small_batched", syevj_small_kernel<128>, dim3(batch), dim3(Cfg::NT), Cfg::LDS_BYTES,
                      s, a, w, v, sweeps, c, max_sweeps, abs_tol);
  }
  return st::check_launch("st_syevj_small_batched");
}
