// RGB-uv soft histogram (RGBuvHistBlock.forward, losses.py:169-322) and the Hellinger distance of
// HistLoss (losses.py:338-347), forward and backward, for gfx950.
//
// Per image the reference builds, for each of the three channel pairings, hist = (Iy * Ku)^T @ Kv
// with Ku/Kv = kernel(|log-chroma - bin|) evaluated in float64 (the bin centres are a float64
// np.linspace) and cast to float32.  Here:
//   prep      one pass over the image: clamp, bilinear resize to insz x insz (when larger), Iy and
//             the three logs -> a small planar record per pixel;
//   accumulate one block per (pixel range, pairing, image): 64-pixel chunks; the kernel weights are
//             evaluated in fp64 into an LDS tile private to the block, and the 64x64 histogram is
//             accumulated across the chunk by v_mfma_f32_32x32x2_f32 (K = pixels);
//   finalize  fixed-order sum of the per-range slabs, image total, normalisation.
// No atomics anywhere: bin values are run-to-run deterministic, and in 'thresholding' mode with
// intensity_scale = 0 they are exact integer counts (each product is 0 or 1, sums < 2^24).
// The backward pass is the same structure with two 64x64x64 MFMA products per chunk.
#include <algorithm>
#include <atomic>
#include <cstdint>

#include <type_traits>

#include "common.h"

namespace {

constexpr int HB = 64;        // histogram bins per axis (HistLoss default, losses.py:327)
constexpr int NT = 256;
constexpr int CH = 64;        // pixels per chunk
constexpr int LS = 65;        // padded LDS row stride (odd: rows walk across lanes)
constexpr int NREC = 7;       // per-pixel record planes: Iy, logR, logG, logB, R, G, B
constexpr float EPSF = 1e-6f; // losses.py:119

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct HistParams {
  double bins[HB];
  double thr;         // eps / 2 of the thresholding method
  double inv_sigma2;  // 1 / sigma^2
  int method;
  int intensity_scale;
};

// ---- torch.nn.functional.interpolate(mode='bilinear', align_corners=False) source mapping -----
struct Lerp {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Lerp lerp_for(int dst, float scale, int in_size) {
  float src = scale * (float(dst) + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  Lerp r;
  r.i0 = min(int(src), in_size - 1);
  r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
  r.l1 = src - float(r.i0);
  r.l0 = 1.f - r.l1;
  return r;
}

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

// ---- prep ------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void hist_prep_kernel(const float* __restrict__ x, float* __restrict__ rec,
                                                      int cstride_c, int H, int W, int hs, int ws,
                                                      int resize, int intensity_scale, long long total) {
  const int N = hs * ws;
  const float sh = float(H) / float(hs), sw = float(W) / float(ws);
  for (long long e = blockIdx.x * (long long)NT + threadIdx.x; e < total; e += (long long)gridDim.x * NT) {
    const int p = int(e % N);
    const long long b = e / N;
    const int py = p / ws, px = p - py * ws;
    const float* xb = x + b * cstride_c * (long long)H * W;
    float rgb[3];
    if (resize) {
      const Lerp ly = lerp_for(py, sh, H), lx = lerp_for(px, sw, W);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* xp = xb + (long long)c * H * W;
        const float v00 = clamp01(xp[(size_t)ly.i0 * W + lx.i0]), v01 = clamp01(xp[(size_t)ly.i0 * W + lx.i1]);
        const float v10 = clamp01(xp[(size_t)ly.i1 * W + lx.i0]), v11 = clamp01(xp[(size_t)ly.i1 * W + lx.i1]);
        rgb[c] = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb[c] = clamp01(xb[(long long)c * H * W + p]);
    }
    float* r = rec + b * NREC * (long long)N + p;
    const float iy = intensity_scale
                         ? sqrtf(rgb[0] * rgb[0] + rgb[1] * rgb[1] + rgb[2] * rgb[2] + EPSF)
                         : 1.f;
    r[0] = iy;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      r[(size_t)(1 + c) * N] = logf(rgb[c] + EPSF);
      r[(size_t)(4 + c) * N] = rgb[c];
    }
  }
}

// kernel weight of one (log-chroma, bin) pair.  The reference subtracts a float64 bin centre from the float32
// log-chroma (np.linspace is float64, losses.py:210-216) and carries on in float64; here the DIFFERENCE -- the
// cancelling step, and the comparison of the thresholding method, whose counts are exact integers -- is formed in fp64
// and the rational / exponential of it in fp32 (1 / (1 + q) by v_rcp_f32, 1 ulp): <= 3e-7 relative per weight, the
// size of the error the fp32 MFMA accumulation of the 22 500 weights has anyway.  (All-fp64 weights with their fp64
// division made hist_accum_kernel arithmetic-bound: 88 us for 0.55 GFLOP of matrix work at B=8.)
template <int METHOD>
__device__ __forceinline__ float kweight(float u, double bin, const HistParams& hp) {
  const double d = fabs(double(u) - bin);
  if constexpr (METHOD == ST_HIST_THRESHOLDING) return d <= hp.thr ? 1.f : 0.f;
  const float df = float(d);
  const float q = df * df * float(hp.inv_sigma2);
  if constexpr (METHOD == ST_HIST_RBF) return expf(-q);
  return __builtin_amdgcn_rcpf(1.0f + q);
}
// the 16 bins x 2 weights of a lane's pixel for a chunk, with the method resolved ONCE per chunk: with the (uniform)
// method test inside every weight the fill was a chain of 32 x 3 scalar branches with one weight's latency exposed
// between each pair
#define ST_HIST_FILL(DST_A, DST_B, SCALE_A)                                                       \
  {                                                                                               \
    auto fill_ = [&](auto m_) {                                                                   \
      constexpr int M_ = decltype(m_)::value;                                                     \
      _Pragma("unroll") for (int k = 0; k < 16; ++k) {                                            \
        const int bin = wave * 16 + k;                                                            \
        const double bc = hp.bins[bin];                                                           \
        DST_A[lane * LS + bin] = ok ? (SCALE_A) * kweight<M_>(u, bc, hp) : 0.f;                   \
        DST_B[lane * LS + bin] = ok ? kweight<M_>(v, bc, hp) : 0.f;                               \
      }                                                                                           \
    };                                                                                            \
    if (hp.method == ST_HIST_THRESHOLDING) fill_(std::integral_constant<int, ST_HIST_THRESHOLDING>());   \
    else if (hp.method == ST_HIST_RBF) fill_(std::integral_constant<int, ST_HIST_RBF>());         \
    else fill_(std::integral_constant<int, ST_HIST_INVERSE_QUADRATIC>());                         \
  }

__device__ __forceinline__ void uv_of(const float* __restrict__ rec, int N, int p, int ch, float& u,
                                      float& v) {
  const float l0 = rec[(size_t)1 * N + p], l1 = rec[(size_t)2 * N + p], l2 = rec[(size_t)3 * N + p];
  // pairings of losses.py:206-209, 244-247, 284-287
  if (ch == 0) { u = l0 - l1; v = l0 - l2; }
  else if (ch == 1) { u = l1 - l0; v = l1 - l2; }
  else { u = l2 - l0; v = l2 - l1; }
}

// ---- accumulate --------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void hist_accum_kernel(const float* __restrict__ rec,
                                                       float* __restrict__ slabs, int N, int B,
                                                       int chunks_per_split, HistParams hp) {
  __shared__ float As[CH * LS];   // [pixel][bin]  Iy * Ku
  __shared__ float Bs[CH * LS];   // [pixel][bin]  Kv
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x, ch = blockIdx.y, b = blockIdx.z;
  const float* r = rec + (size_t)b * NREC * N;
  const int nchunks = (N + CH - 1) / CH;
  const int c_begin = split * chunks_per_split, c_end = min(c_begin + chunks_per_split, nchunks);

  f32x16 acc;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = 0.f;
  const int it = wave >> 1, jt = wave & 1;

  for (int c = c_begin; c < c_end; ++c) {
    const int p = c * CH + lane;          // this lane's pixel; the wave covers bins wave*16..+15
    const bool ok = p < N;
    float u = 0.f, v = 0.f, iy = 0.f;
    if (ok) {
      uv_of(r, N, p, ch, u, v);
      iy = r[p];
    }
    __syncthreads();   // previous chunk's MFMA reads are done
    ST_HIST_FILL(As, Bs, iy)
    __syncthreads();
#pragma unroll
    for (int s = 0; s < CH / 2; ++s) {
      const float a = As[(2 * s + lh) * LS + it * 32 + li];
      const float bb = Bs[(2 * s + lh) * LS + jt * 32 + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
    }
  }
  float* slab = slabs + (((size_t)split * B + b) * 3 + ch) * (HB * HB);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = it * 32 + (k & 3) + 8 * (k >> 2) + 4 * lh;
    slab[i * HB + jt * 32 + li] = acc[k];
  }
}

// ---- finalize: sum slabs, normalise ------------------------------------------------------------
// Stage 1 (grid E/NT x B): one element per thread, fixed-order sum over the split slabs (coalesced
// across the block), un-normalised value -> hist, per-block partial total -> part[b][blockIdx.x].
// Stage 2 (same grid): image total = fixed-order sum of its E/NT partials, then the true division of
// losses.py:319-320.  Deterministic; bins stay exact integers in thresholding mode.
constexpr int HIST_E = 3 * HB * HB;
constexpr int HIST_PARTS = HIST_E / NT;

__global__ __launch_bounds__(NT) void hist_sum_slabs_kernel(const float* __restrict__ slabs, int nsplit, int B,
                                                           float* __restrict__ hist, float* __restrict__ part) {
  __shared__ float red[NT / 64];
  const int b = blockIdx.y, e = blockIdx.x * NT + threadIdx.x;
  float t = 0.f;
  for (int sp = 0; sp < nsplit; ++sp) t += slabs[((size_t)sp * B + b) * HIST_E + e];
  hist[(size_t)b * HIST_E + e] = t;
  const float s = st::block_sum<NT>(t, red);
  if (threadIdx.x == 0) part[b * HIST_PARTS + blockIdx.x] = s;
}

__global__ __launch_bounds__(NT) void hist_normalize_kernel(float* __restrict__ hist, const float* __restrict__ part,
                                                           float* __restrict__ totals) {
  const int b = blockIdx.y, e = blockIdx.x * NT + threadIdx.x;
  float total = 0.f;
  for (int k = 0; k < HIST_PARTS; ++k) total += part[b * HIST_PARTS + k];   // same order in every thread
  const float den = total + EPSF;   // losses.py:319-320 (a true division, like the reference)
  hist[(size_t)b * HIST_E + e] = hist[(size_t)b * HIST_E + e] / den;
  if (blockIdx.x == 0 && threadIdx.x == 0) totals[b] = total;
}

// ---- Hellinger pieces --------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void hellinger_sq_partial_kernel(const float* __restrict__ ht,
                                                                 const float* __restrict__ hi, long long n,
                                                                 float* __restrict__ partials) {
  __shared__ float red[NT / 64];
  float s = 0.f;
  for (long long e = blockIdx.x * (long long)NT + threadIdx.x; e < n; e += (long long)gridDim.x * NT) {
    const float d = sqrtf(ht[e]) - sqrtf(hi[e]);
    s += d * d;
  }
  s = st::block_sum<NT>(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(NT) void sum_partials_kernel(const float* __restrict__ partials, int n,
                                                         float* __restrict__ out) {
  __shared__ float red[NT / 64];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += NT) s += partials[i];
  s = st::block_sum<NT>(s, red);
  if (threadIdx.x == 0) out[0] = s;
}

__global__ __launch_bounds__(NT) void hellinger_sq_bwd_kernel(const float* __restrict__ ht,
                                                             const float* __restrict__ hi,
                                                             const float* __restrict__ gq, float scale,
                                                             float* __restrict__ dhi, long long n) {
  const float g = scale * (gq ? gq[0] : 1.f);
  for (long long e = blockIdx.x * (long long)NT + threadIdx.x; e < n; e += (long long)gridDim.x * NT) {
    const float sh = sqrtf(hi[e]);
    dhi[e] = -g * (sqrtf(ht[e]) - sh) / sh;    // d/dhi (sqrt(ht) - sqrt(hi))^2
  }
}

// ---- backward: d(unnormalised hist) --------------------------------------------------------------
__global__ __launch_bounds__(NT) void hist_bwd_norm_kernel(const float* __restrict__ ghist,
                                                          const float* __restrict__ hist,
                                                          const float* __restrict__ totals,
                                                          float* __restrict__ draw) {
  __shared__ float red[NT / 64];
  constexpr int E = 3 * HB * HB;
  const int b = blockIdx.x;
  float s = 0.f;
  for (int e = threadIdx.x; e < E; e += NT) s += ghist[(size_t)b * E + e] * hist[(size_t)b * E + e];
  const float dot = st::block_sum<NT>(s, red);
  const float inv = 1.f / (totals[b] + EPSF);
  for (int e = threadIdx.x; e < E; e += NT) draw[(size_t)b * E + e] = (ghist[(size_t)b * E + e] - dot) * inv;
}

// d kernel / d(log-chroma), from the stored fp32 weight k (no intensity factor)
__device__ __forceinline__ float kderiv(float u, float binf, float k, float inv_sigma2, int method) {
  if (method == ST_HIST_THRESHOLDING) return 0.f;
  const float t = -2.f * (u - binf) * inv_sigma2 * k;
  return method == ST_HIST_RBF ? t : t * k;
}

// Per (pixel range, pairing, image): gradient w.r.t. Iy, u, v of every pixel of the range.
//   t1[i][p] = sum_j dH[i][j] Kv[p][j];  gIy = sum_i Ku[p][i] t1[i][p];  gu = Iy sum_i Ku'[p][i] t1[i][p]
//   t2[j][p] = sum_i dH[i][j] Ku[p][i];  gv = Iy sum_j Kv'[p][j] t2[j][p]
__global__ __launch_bounds__(NT) void hist_bwd_pixel_kernel(const float* __restrict__ rec,
                                                           const float* __restrict__ draw,
                                                           float* __restrict__ gpix, int N, int B,
                                                           int chunks_per_split, HistParams hp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dH = smem;                 // [i][j]   stride LS
  float* dHT = dH + HB * LS;        // [j][i]
  float* Ku = dHT + HB * LS;        // [pixel][bin]
  float* Kv = Ku + CH * LS;
  float* binf = Kv + CH * LS;       // [64]
  float* part = binf + HB;          // [3 kinds][2 tiles][64 pixels]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int split = blockIdx.x, ch = blockIdx.y, b = blockIdx.z;
  const float* r = rec + (size_t)b * NREC * N;
  const float* dsrc = draw + ((size_t)b * 3 + ch) * (HB * HB);
  for (int e = tid; e < HB * HB; e += NT) {
    const int i = e >> 6, j = e & 63;
    const float v = dsrc[e];
    dH[i * LS + j] = v;
    dHT[j * LS + i] = v;
  }
  if (tid < HB) binf[tid] = float(hp.bins[tid]);
  const float inv_s2 = float(hp.inv_sigma2);
  const int nchunks = (N + CH - 1) / CH;
  const int c_begin = split * chunks_per_split, c_end = min(c_begin + chunks_per_split, nchunks);
  const int tt = wave >> 1, pt = wave & 1;   // MFMA tile: rows tt*32.. (bins), cols pt*32.. (pixels)
  float* gout = gpix + (((size_t)b * 3 + ch) * 3) * N;

  for (int c = c_begin; c < c_end; ++c) {
    const int p = c * CH + lane;
    const bool ok = p < N;
    float u = 0.f, v = 0.f;
    if (ok) uv_of(r, N, p, ch, u, v);
    __syncthreads();   // previous chunk fully consumed (also orders the dH fill on the first pass)
    ST_HIST_FILL(Ku, Kv, 1.f)
    __syncthreads();
    // this lane's MFMA column = pixel pt*32 + li of the chunk
    const int pc = pt * 32 + li;
    const int pg = c * CH + pc;
    float uc = 0.f, vc = 0.f;
    if (pg < N) uv_of(r, N, pg, ch, uc, vc);

    // ---- phase 1: t1 = dH x Kv^T ----
    f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
    for (int s = 0; s < HB / 2; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dH[(tt * 32 + li) * LS + 2 * s + lh],
                                                 Kv[(pt * 32 + li) * LS + 2 * s + lh], acc, 0, 0, 0);
    float sy = 0.f, su = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = tt * 32 + (k & 3) + 8 * (k >> 2) + 4 * lh;
      const float kk = Ku[pc * LS + i];
      sy += kk * acc[k];
      su += kderiv(uc, binf[i], kk, inv_s2, hp.method) * acc[k];
    }
    sy += __shfl_xor(sy, 32, 64);
    su += __shfl_xor(su, 32, 64);
    // ---- phase 2: t2 = dH^T x Ku^T ----
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
    for (int s = 0; s < HB / 2; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dHT[(tt * 32 + li) * LS + 2 * s + lh],
                                                 Ku[(pt * 32 + li) * LS + 2 * s + lh], acc, 0, 0, 0);
    float sv = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int j = tt * 32 + (k & 3) + 8 * (k >> 2) + 4 * lh;
      sv += kderiv(vc, binf[j], Kv[pc * LS + j], inv_s2, hp.method) * acc[k];
    }
    sv += __shfl_xor(sv, 32, 64);
    if (lh == 0) {
      part[(0 * 2 + tt) * CH + pc] = sy;
      part[(1 * 2 + tt) * CH + pc] = su;
      part[(2 * 2 + tt) * CH + pc] = sv;
    }
    __syncthreads();
    if (tid < CH && c * CH + tid < N) {
      const int q = c * CH + tid;
      const float iy = r[q];
      gout[(size_t)0 * N + q] = part[0 * CH + tid] + part[1 * CH + tid];
      gout[(size_t)1 * N + q] = iy * (part[2 * CH + tid] + part[3 * CH + tid]);
      gout[(size_t)2 * N + q] = iy * (part[4 * CH + tid] + part[5 * CH + tid]);
    }
  }
}

// Per pixel: (gIy, gu, gv) of the three pairings -> gradient w.r.t. the sampled, clamped RGB.
__global__ __launch_bounds__(NT) void hist_bwd_chain_kernel(const float* __restrict__ rec,
                                                           const float* __restrict__ gpix,
                                                           float* __restrict__ gs, int N, int intensity_scale,
                                                           long long total) {
  for (long long e = blockIdx.x * (long long)NT + threadIdx.x; e < total; e += (long long)gridDim.x * NT) {
    const int p = int(e % N);
    const long long b = e / N;
    const float* r = rec + b * NREC * (long long)N;
    const float* g = gpix + b * 9 * (long long)N + p;
    float gy = 0.f, gu[3], gv[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      gy += g[(size_t)(ch * 3 + 0) * N];
      gu[ch] = g[(size_t)(ch * 3 + 1) * N];
      gv[ch] = g[(size_t)(ch * 3 + 2) * N];
    }
    // u0 = lR-lG, v0 = lR-lB; u1 = lG-lR, v1 = lG-lB; u2 = lB-lR, v2 = lB-lG
    float dl[3];
    dl[0] = (gu[0] + gv[0]) - gu[1] - gu[2];
    dl[1] = -gu[0] + (gu[1] + gv[1]) - gv[2];
    dl[2] = -gv[0] - gv[1] + (gu[2] + gv[2]);
    const float iy = r[p];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float s = r[(size_t)(4 + c) * N + p];
      float d = dl[c] / (s + EPSF);
      if (intensity_scale) d += gy * s / iy;
      gs[(b * 3 + c) * (long long)N + p] = d;
    }
  }
}

// Adjoint of clamp + bilinear resize: gather form (every input pixel collects from the few
// destination pixels that may have sampled it), deterministic.
__global__ __launch_bounds__(NT) void hist_bwd_resize_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ gs,
                                                            const float* __restrict__ gout, float scale,
                                                            float* __restrict__ dx, int cstride_c, int H, int W,
                                                            int hs, int ws, int resize, int accumulate,
                                                            long long total) {
  const float g0 = scale * (gout ? gout[0] : 1.f);
  const float sh = float(H) / float(hs), sw = float(W) / float(ws);
  for (long long e = blockIdx.x * (long long)NT + threadIdx.x; e < total; e += (long long)gridDim.x * NT) {
    const int xx = int(e % W);
    long long t = e / W;
    const int yy = int(t % H);
    t /= H;
    const int c = int(t % cstride_c);
    const long long b = t / cstride_c;
    float g = 0.f;
    const float v = x[e];
    if (c < 3 && v >= 0.f && v <= 1.f) {      // torch.clamp passes the gradient on [min, max]
      const float* gp = gs + (b * 3 + c) * (long long)hs * ws;
      if (!resize) {
        g = gp[(size_t)yy * ws + xx];
      } else {
        // destination rows whose source coordinate lies in [yy-1, yy+1): (+-1 row of slack)
        const int dy_lo = max(int(ceilf((float(yy) - 0.5f) / sh - 0.5f)) - 1, 0);
        const int dy_hi = min(int(floorf((float(yy) + 1.5f) / sh - 0.5f)) + 1, hs - 1);
        const int dx_lo = max(int(ceilf((float(xx) - 0.5f) / sw - 0.5f)) - 1, 0);
        const int dx_hi = min(int(floorf((float(xx) + 1.5f) / sw - 0.5f)) + 1, ws - 1);
        for (int dy = dy_lo; dy <= dy_hi; ++dy) {
          const Lerp ly = lerp_for(dy, sh, H);
          const float wy = (ly.i0 == yy ? ly.l0 : 0.f) + (ly.i1 == yy ? ly.l1 : 0.f);
          if (wy == 0.f) continue;
          for (int dxx = dx_lo; dxx <= dx_hi; ++dxx) {
            const Lerp lx = lerp_for(dxx, sw, W);
            const float wx = (lx.i0 == xx ? lx.l0 : 0.f) + (lx.i1 == xx ? lx.l1 : 0.f);
            if (wx != 0.f) g += wy * wx * gp[(size_t)dy * ws + dxx];
          }
        }
      }
    }
    g *= g0;
    dx[e] = accumulate ? dx[e] + g : g;
  }
}

struct Geo {
  int hs, ws, N, resize, nsplit, chunks_per_split;
};

Geo geometry(int n, int h, int w, int insz) {
  Geo g;
  g.resize = (h > insz || w > insz) ? 1 : 0;     // losses.py:171
  g.hs = g.resize ? insz : h;
  g.ws = g.resize ? insz : w;
  g.N = g.hs * g.ws;
  const int nchunks = st::cdiv(g.N, CH);
  int want = std::max(1, 1024 / std::max(1, n * 3));
  g.nsplit = std::max(1, std::min(want, nchunks));
  g.chunks_per_split = st::cdiv(nchunks, g.nsplit);
  g.nsplit = st::cdiv(nchunks, g.chunks_per_split);
  return g;
}

// floats of the pixel-record region (its head doubles as the finalize partial-total buffer)
size_t rec_elems(int n, int N) { return std::max(size_t(n) * NREC * N, size_t(n) * HIST_PARTS); }

int fill_params(HistParams& hp, int method, float sigma, const double* bins_host, int nbins,
                int intensity_scale, const char* who) {
  ST_REQUIRE(bins_host, "%s: null bins", who);
  ST_REQUIRE(nbins >= 1 && nbins <= HB, "%s: 1 <= h <= %d bins are supported (HistLoss uses %d), got %d", who, HB, HB,
             nbins);
  ST_REQUIRE(method >= ST_HIST_INVERSE_QUADRATIC && method <= ST_HIST_THRESHOLDING, "%s: bad method %d", who,
             method);
  // h < 64: the kernels always work on 64 x 64 bins; the unused centres sit at 1e30, where every kernel weight and
  // its derivative evaluate to exactly 0, so rows / columns >= h of the output are zero and the totals see nothing
  for (int i = 0; i < HB; ++i) hp.bins[i] = i < nbins ? bins_host[i] : 1e30;
  // losses.py:164-165: eps = (|lo| + |hi|) / h, membership is |d| <= eps / 2
  hp.thr = (fabs(bins_host[0]) + fabs(bins_host[nbins - 1])) / double(nbins) / 2.0;
  hp.inv_sigma2 = 1.0 / (double(sigma) * double(sigma));
  hp.method = method;
  hp.intensity_scale = intensity_scale ? 1 : 0;
  return ST_OK;
}

constexpr size_t BWD_LDS = size_t(2 * HB * LS + 2 * CH * LS + HB + 6 * CH) * sizeof(float);

// floats of the forward-only workspace: pixel records + split slabs
size_t fwd_elems(int n, const Geo& g) { return rec_elems(n, g.N) + size_t(g.nsplit) * n * 3 * HB * HB; }

// ---- the library's own scratch for st_rgbuv_hist_fwd_static (callers that cannot allocate) ------------
// One copy per GPU the library runs on, held until the process ends.  In HOST code the array's name is only a
// stand-in for the symbol: the device address comes from hipGetSymbolAddress, once per device.
constexpr size_t STATIC_SCRATCH_FLOATS = size_t(ST_HIST_STATIC_SCRATCH_BYTES) / sizeof(float);
constexpr int MAX_DEVICES = 64;
static __device__ __align__(256) float scratch[STATIC_SCRATCH_FLOATS];
std::atomic<float*> scratch_ptr[MAX_DEVICES];

// device address of `scratch` on the CURRENT device (resolved on the first call per device, then cached)
int static_scratch(float** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return st::fail(ST_ERR_LAUNCH, "st_rgbuv_hist_fwd_static: hipGetDevice: %s", hipGetErrorString(e));
  ST_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "st_rgbuv_hist_fwd_static: device %d out of range", dev);
  float* p = scratch_ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p)
      return st::fail(ST_ERR_LAUNCH, "st_rgbuv_hist_fwd_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
    scratch_ptr[dev].store(p, std::memory_order_release);   // every thread resolves the same address
  }
  *out = p;
  return ST_OK;
}

}  // namespace

extern "C" size_t st_rgbuv_hist_workspace_bytes(int n, int h, int w, int insz) {
  if (n <= 0 || h <= 0 || w <= 0 || insz <= 0) return 0;
  const Geo g = geometry(n, h, w, insz);
  // records + forward slabs (also reused as the per-pixel gradient buffers in backward)
  const size_t rec = rec_elems(n, g.N);
  const size_t slabs = size_t(g.nsplit) * n * 3 * HB * HB;
  const size_t bwd = size_t(n) * 3 * HB * HB + size_t(n) * 9 * g.N + size_t(n) * 3 * g.N;
  return (rec + std::max(slabs, bwd)) * sizeof(float);
}

extern "C" int st_rgbuv_hist_fwd(const float* x, float* hist, float* totals, float* workspace, int n,
                                 int c, int h, int w, int insz, int method, float sigma,
                                 const double* bins_host, int nbins, int intensity_scale,
                                 st_stream_t stream) {
  ST_REQUIRE(x && hist && totals && workspace, "st_rgbuv_hist_fwd: null pointer");
  ST_REQUIRE(n > 0 && c >= 3 && h > 0 && w > 0 && insz > 0, "st_rgbuv_hist_fwd: bad dims n=%d c=%d h=%d w=%d",
             n, c, h, w);
  HistParams hp;
  int rc = fill_params(hp, method, sigma, bins_host, nbins, intensity_scale, "st_rgbuv_hist_fwd");
  if (rc != ST_OK) return rc;
  const Geo g = geometry(n, h, w, insz);
  hipStream_t s = (hipStream_t)stream;
  float* rec = workspace;
  float* slabs = workspace + rec_elems(n, g.N);
  const long long total = (long long)n * g.N;
  hipLaunchKernelGGL(hist_prep_kernel, dim3((int)std::min<long long>(st::cdivll(total, NT), 2048)), dim3(NT),
                     0, s, x, rec, c, h, w, g.hs, g.ws, g.resize, hp.intensity_scale, total);
  hipLaunchKernelGGL(hist_accum_kernel, dim3(g.nsplit, 3, n), dim3(NT), 0, s, rec, slabs, g.N, n,
                     g.chunks_per_split, hp);
  // the pixel records are dead once the slabs exist: their head holds the per-block partial totals
  static_assert(HIST_E % NT == 0, "one histogram element per thread");
  hipLaunchKernelGGL(hist_sum_slabs_kernel, dim3(HIST_PARTS, n), dim3(NT), 0, s, slabs, g.nsplit, n, hist, rec);
  hipLaunchKernelGGL(hist_normalize_kernel, dim3(HIST_PARTS, n), dim3(NT), 0, s, hist, rec, totals);
  return st::check_launch("st_rgbuv_hist_fwd");
}

extern "C" size_t st_rgbuv_hist_static_scratch_bytes(void) { return ST_HIST_STATIC_SCRATCH_BYTES; }

extern "C" int st_rgbuv_hist_fwd_static(const float* x, float* hist, float* totals, int n, int c, int h, int w,
                                        int insz, int method, float sigma, const double* bins_host, int nbins,
                                        int intensity_scale, st_stream_t stream) {
  ST_REQUIRE(n > 0 && c >= 3 && h > 0 && w > 0 && insz > 0, "st_rgbuv_hist_fwd_static: bad dims n=%d c=%d h=%d w=%d",
             n, c, h, w);
  const size_t need = fwd_elems(n, geometry(n, h, w, insz)) * sizeof(float);
  ST_REQUIRE(need <= size_t(ST_HIST_STATIC_SCRATCH_BYTES),
             "st_rgbuv_hist_fwd_static: n=%d images of %dx%d need %zu bytes of scratch, the library's static buffer "
             "holds %zu (use st_rgbuv_hist_fwd with a caller workspace)", n, h, w, need,
             size_t(ST_HIST_STATIC_SCRATCH_BYTES));
  float* ws = nullptr;
  const int rc = static_scratch(&ws);
  if (rc != ST_OK) return rc;
  // same kernels as st_rgbuv_hist_fwd; every float of the scratch that is read has been written by this call
  // (prep fills the records, every accumulate block stores its whole slab, the slab sum writes the partial totals),
  // so nothing left over from an earlier, larger call can leak into a result
  return st_rgbuv_hist_fwd(x, hist, totals, ws, n, c, h, w, insz, method, sigma, bins_host, nbins, intensity_scale,
                           stream);
}

extern "C" int st_rgbuv_hist_bwd(const float* x, const float* hist, const float* totals,
                                 const float* ghist, const float* gout, float scale, float* dx,
                                 float* workspace, int n, int c, int h, int w, int insz, int method,
                                 float sigma, const double* bins_host, int nbins, int intensity_scale,
                                 int accumulate, st_stream_t stream) {
  ST_REQUIRE(x && hist && totals && ghist && dx && workspace, "st_rgbuv_hist_bwd: null pointer");
  ST_REQUIRE(n > 0 && c >= 3 && h > 0 && w > 0 && insz > 0, "st_rgbuv_hist_bwd: bad dims n=%d c=%d h=%d w=%d",
             n, c, h, w);
  HistParams hp;
  int rc = fill_params(hp, method, sigma, bins_host, nbins, intensity_scale, "st_rgbuv_hist_bwd");
  if (rc != ST_OK) return rc;
  const Geo g = geometry(n, h, w, insz);
  hipStream_t s = (hipStream_t)stream;
  float* rec = workspace;
  float* draw = workspace + rec_elems(n, g.N);
  float* gpix = draw + size_t(n) * 3 * HB * HB;
  float* gs = gpix + size_t(n) * 9 * g.N;
  const long long total = (long long)n * g.N;
  const int eb = (int)std::min<long long>(st::cdivll(total, NT), 2048);
  hipLaunchKernelGGL(hist_prep_kernel, dim3(eb), dim3(NT), 0, s, x, rec, c, h, w, g.hs, g.ws, g.resize,
                     hp.intensity_scale, total);
  hipLaunchKernelGGL(hist_bwd_norm_kernel, dim3(n), dim3(NT), 0, s, ghist, hist, totals, draw);
  ST_LAUNCH_DYN_LDS("st_rgbuv_hist_bwd", hist_bwd_pixel_kernel, dim3(g.nsplit, 3, n), dim3(NT), BWD_LDS, s, rec,
                    draw, gpix, g.N, n, g.chunks_per_split, hp);
  hipLaunchKernelGGL(hist_bwd_chain_kernel, dim3(eb), dim3(NT), 0, s, rec, gpix, gs, g.N, hp.intensity_scale,
                     total);
  const long long tin = (long long)n * c * h * w;
  hipLaunchKernelGGL(hist_bwd_resize_kernel, dim3((int)std::min<long long>(st::cdivll(tin, NT), 2048)),
                     dim3(NT), 0, s, x, gs, gout, scale, dx, c, h, w, g.hs, g.ws, g.resize, accumulate, tin);
  return st::check_launch("st_rgbuv_hist_bwd");
}

extern "C" int st_hellinger_sq_fwd(const float* h_target, const float* h_input, long long n,
                                   float* workspace, float* out, st_stream_t stream) {
  ST_REQUIRE(h_target && h_input && workspace && out, "st_hellinger_sq_fwd: null pointer");
  ST_REQUIRE(n > 0, "st_hellinger_sq_fwd: n=%lld", n);
  const int parts = (int)std::max<long long>(1, std::min<long long>(st::cdivll(n, NT * 8), 1024));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(hellinger_sq_partial_kernel, dim3(parts), dim3(NT), 0, s, h_target, h_input, n, workspace);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(NT), 0, s, workspace, parts, out);
  return st::check_launch("st_hellinger_sq_fwd");
}

extern "C" int st_hellinger_sq_bwd(const float* h_target, const float* h_input, const float* gq, float scale,
                                   float* dh_input, long long n, st_stream_t stream) {
  ST_REQUIRE(h_target && h_input && dh_input, "st_hellinger_sq_bwd: null pointer");
  ST_REQUIRE(n > 0, "st_hellinger_sq_bwd: n=%lld", n);
  hipLaunchKernelGGL(hellinger_sq_bwd_kernel, dim3((int)std::min<long long>(st::cdivll(n, NT), 2048)), dim3(NT),
                     0, (hipStream_t)stream, h_target, h_input, gq, scale, dh_input, n);
  return st::check_launch("st_hellinger_sq_bwd");
}
