// AdaIN statistics + remap + alpha blend for gfx950 -- HBM-bound wavefront reductions.
//
// Replaces AdaIN.calc_mean_std / adaptive_instance_normalization / forward
// (transformer.py:15-24, 46-59) and the blend `alpha*t + (1-alpha)*fc` (trainer.py:434,702).
//
// One 256-thread block per (n, c) plane.  When both planes fit the register budget
// (<= 256 threads x V4 float4) the content AND style planes are each read from HBM exactly
// once with 16-B coalesced loads, the statistics are the exact two-pass form (mean, then
// centred sum of squares) on registers, and the output is written once: 3 algorithmic
// passes (read content, read style, write out).  Larger planes stream from global memory
// (the re-reads of a 16..256 KB plane are served by L2 / Infinity Cache).
// Reductions are shuffle + fixed-order LDS sums: run-to-run deterministic, no float atomics.
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "common.h"

namespace {

constexpr int NT = 256;

struct Stats {
  float mean, m2;
};

// Read-once streams use non-temporal loads: measured on the configs[1] AdaIN launch (402 MB, nothing
// cached) 74.2 -> 62.6 us, 5.43 -> 6.43 TB/s.  Stores stay temporal: the decoder reads `out` next and
// part of it is still in the Infinity Cache by then (non-temporal stores measured slower in the step).
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg_stream(const float4* p) {
  const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

// ---- register-resident path -----------------------------------------------------------
template <int V4>
__global__ __launch_bounds__(NT) void adain_reg_kernel(const float* __restrict__ content,
                                                      const float* __restrict__ style,
                                                      float* __restrict__ out, int hw_c, int hw_s,
                                                      float eps_c, float eps_s, float alpha,
                                                      float* __restrict__ stats,
    const float* __restrict__ eps_dev) {
  if (eps_dev) { eps_c = eps_dev[0]; eps_s = eps_dev[1]; }   // device-resident eps (no host read)
  __shared__ float red[2 * NT / 64];
  const int tid = threadIdx.x;
  const size_t plane = blockIdx.x;
  const float4* c4 = reinterpret_cast<const float4*>(content + plane * hw_c);
  const float4* s4 = reinterpret_cast<const float4*>(style + plane * hw_s);
  float4* o4 = reinterpret_cast<float4*>(out + plane * hw_c);
  const int n4c = hw_c >> 2, n4s = hw_s >> 2;

  float4 cv[V4], sv[V4];
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    const int idx = tid + i * NT;
    cv[i] = idx < n4c ? ldg_stream(c4 + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    const int idx = tid + i * NT;
    sv[i] = idx < n4s ? ldg_stream(s4 + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float sc = 0.f, ss = 0.f;
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    sc += (cv[i].x + cv[i].y) + (cv[i].z + cv[i].w);
    ss += (sv[i].x + sv[i].y) + (sv[i].z + sv[i].w);
  }
  // two block sums at once
  sc = st::wave_sum(sc);
  ss = st::wave_sum(ss);
  const int wave = tid >> 6;
  if ((tid & 63) == 0) { red[wave] = sc; red[NT / 64 + wave] = ss; }
  __syncthreads();
  float tc = 0.f, ts = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) { tc += red[w]; ts += red[NT / 64 + w]; }
  const float mean_c = tc / float(hw_c), mean_s = ts / float(hw_s);

  float qc = 0.f, qs = 0.f;
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    const int idx = tid + i * NT;
    if (idx < n4c) {
      const float a = cv[i].x - mean_c, b = cv[i].y - mean_c, c = cv[i].z - mean_c, d = cv[i].w - mean_c;
      qc += (a * a + b * b) + (c * c + d * d);
    }
    if (idx < n4s) {
      const float a = sv[i].x - mean_s, b = sv[i].y - mean_s, c = sv[i].z - mean_s, d = sv[i].w - mean_s;
      qs += (a * a + b * b) + (c * c + d * d);
    }
  }
  qc = st::wave_sum(qc);
  qs = st::wave_sum(qs);
  __syncthreads();
  if ((tid & 63) == 0) { red[wave] = qc; red[NT / 64 + wave] = qs; }
  __syncthreads();
  tc = 0.f; ts = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) { tc += red[w]; ts += red[NT / 64 + w]; }
  const float std_c = sqrtf(tc / float(hw_c - 1) + eps_c);
  const float std_s = sqrtf(ts / float(hw_s - 1) + eps_s);
  if (stats && tid == 0) {
    float4 sv4 = make_float4(mean_c, std_c, mean_s, std_s);
    reinterpret_cast<float4*>(stats)[plane] = sv4;
  }
  const float inv_c = 1.f / std_c;
  const float beta = 1.f - alpha;
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    const int idx = tid + i * NT;
    if (idx < n4c) {
      float4 v = cv[i], r;
      r.x = alpha * ((v.x - mean_c) * inv_c * std_s + mean_s) + beta * v.x;
      r.y = alpha * ((v.y - mean_c) * inv_c * std_s + mean_s) + beta * v.y;
      r.z = alpha * ((v.z - mean_c) * inv_c * std_s + mean_s) + beta * v.z;
      r.w = alpha * ((v.w - mean_c) * inv_c * std_s + mean_s) + beta * v.w;
      o4[idx] = r;
    }
  }
}

// ---- large planes (16K < hw <= 64K floats: 1024^2 .. 2048^2 images) -----------------------
// 512 threads hold the whole content plane in registers (32 float4 each = 256 KB per block), so
// it is read from HBM once, its statistics are the exact two-pass form on registers and the
// output is written once.  The style plane is only needed for its two statistics: it is streamed
// ONCE, accumulating d = x - K and d*d in fp64 with K = mean of the plane's first 512 elements
// (shifted one-pass moments: with fp64 accumulators the cancellation in s2 - s1^2/n is far below
// fp32 resolution), instead of being read twice.  Algorithmic traffic: content + style + out.
constexpr int NTB_MAX_HW = 512 * 4 * 32;     // largest content plane adain_big_kernel holds in registers (1024 x 16 float4 = 512 x 32)

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int NTB, int V4>
__global__ __launch_bounds__(NTB) void adain_big_kernel(const float* __restrict__ content,
                                                       const float* __restrict__ style,
                                                       float* __restrict__ out, int hw_c, int hw_s,
                                                       float eps_c, float eps_s, float alpha,
                                                       float* __restrict__ stats,
    const float* __restrict__ eps_dev) {
  if (eps_dev) { eps_c = eps_dev[0]; eps_s = eps_dev[1]; }   // device-resident eps (no host read)
  constexpr int NW = NTB / 64;
  __shared__ float red[NW];
  __shared__ double redd[2 * NW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t plane = blockIdx.x;
  const float4* c4 = reinterpret_cast<const float4*>(content + plane * hw_c);
  const float4* s4 = reinterpret_cast<const float4*>(style + plane * hw_s);
  float4* o4 = reinterpret_cast<float4*>(out + plane * hw_c);
  const int n4c = hw_c >> 2, n4s = hw_s >> 2;

  float4 cv[V4];
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    const int idx = tid + i * NTB;
    cv[i] = idx < n4c ? ldg_stream(c4 + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // style shift K: mean of the first min(1024, hw_s) elements
  const float* sp = style + plane * hw_s;
  const int nk = min(NTB, hw_s);
  float k0 = tid < nk ? sp[tid] : 0.f;
  k0 = st::wave_sum(k0);
  if (lane == 0) red[wave] = k0;
  __syncthreads();
  float K = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) K += red[w];
  K /= float(nk);
  // one pass over the style plane, 8 float4 in flight per thread
  double s1 = 0.0, s2 = 0.0;
  for (int base = tid; base < n4s; base += 8 * NTB) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * NTB;
      v[u] = idx < n4s ? ldg_stream(s4 + idx) : make_float4(K, K, K, K);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float a = v[u].x - K, b = v[u].y - K, c = v[u].z - K, d = v[u].w - K;
      s1 += double((a + b) + (c + d));
      s2 += double(a * a + b * b) + double(c * c + d * d);
    }
  }
  // content sum (registers)
  float sc = 0.f;
#pragma unroll
  for (int i = 0; i < V4; ++i) sc += (cv[i].x + cv[i].y) + (cv[i].z + cv[i].w);
  sc = st::wave_sum(sc);
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  __syncthreads();
  if (lane == 0) { red[wave] = sc; redd[wave] = s1; redd[NW + wave] = s2; }
  __syncthreads();
  float tc = 0.f;
  double t1 = 0.0, t2 = 0.0;
#pragma unroll
  for (int w = 0; w < NW; ++w) { tc += red[w]; t1 += redd[w]; t2 += redd[NW + w]; }
  const float mean_c = tc / float(hw_c);
  const double ns = double(hw_s);
  const float mean_s = float(double(K) + t1 / ns);
  const float var_s = float((t2 - t1 * t1 / ns) / (ns - 1.0));
  float qc = 0.f;
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    const int idx = tid + i * NTB;
    if (idx < n4c) {
      const float a = cv[i].x - mean_c, b = cv[i].y - mean_c, c = cv[i].z - mean_c, d = cv[i].w - mean_c;
      qc += (a * a + b * b) + (c * c + d * d);
    }
  }
  qc = st::wave_sum(qc);
  __syncthreads();
  if (lane == 0) red[wave] = qc;
  __syncthreads();
  tc = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) tc += red[w];
  const float std_c = sqrtf(tc / float(hw_c - 1) + eps_c);
  const float std_s = sqrtf(fmaxf(var_s, 0.f) + eps_s);
  if (stats && tid == 0) reinterpret_cast<float4*>(stats)[plane] = make_float4(mean_c, std_c, mean_s, std_s);
  const float inv_c = 1.f / std_c, beta = 1.f - alpha;
  const float ga = alpha * inv_c * std_s + beta, gb = alpha * (mean_s - mean_c * inv_c * std_s);
#pragma unroll
  for (int i = 0; i < V4; ++i) {
    const int idx = tid + i * NTB;
    if (idx < n4c) {
      // out = ga * x + gb  (in place: no second register set)
      cv[i].x = fmaf(ga, cv[i].x, gb);
      cv[i].y = fmaf(ga, cv[i].y, gb);
      cv[i].z = fmaf(ga, cv[i].z, gb);
      cv[i].w = fmaf(ga, cv[i].w, gb);
      o4[idx] = cv[i];
    }
  }
}

// ---- streaming helpers (any hw, any alignment) ------------------------------------------
__device__ __forceinline__ Stats plane_stats(const float* __restrict__ p, int hw, float* red) {
  // 16-B loads when the plane allows it (every NCHW plane of the model does: hw % 4 == 0)
  const bool vec = (hw & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  const float4* p4 = reinterpret_cast<const float4*>(p);
  const int n4 = hw >> 2;
  float s = 0.f;
  if (vec) {
    for (int i = threadIdx.x; i < n4; i += NT) {
      const float4 v = p4[i];
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (int i = threadIdx.x; i < hw; i += NT) s += p[i];
  }
  const float mean = st::block_sum<NT>(s, red) / float(hw);
  float q = 0.f;
  if (vec) {
    for (int i = threadIdx.x; i < n4; i += NT) {
      const float4 v = p4[i];
      const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
  } else {
    for (int i = threadIdx.x; i < hw; i += NT) {
      const float d = p[i] - mean;
      q += d * d;
    }
  }
  Stats r;
  r.mean = mean;
  r.m2 = st::block_sum<NT>(q, red);
  return r;
}

__global__ __launch_bounds__(NT) void adain_stream_kernel(const float* __restrict__ content,
                                                         const float* __restrict__ style,
                                                         float* __restrict__ out, int hw_c,
                                                         int hw_s, float eps_c, float eps_s,
                                                         float alpha, float* __restrict__ stats,
    const float* __restrict__ eps_dev) {
  if (eps_dev) { eps_c = eps_dev[0]; eps_s = eps_dev[1]; }   // device-resident eps (no host read)
  __shared__ float red[NT / 64];
  const size_t plane = blockIdx.x;
  const float* cp = content + plane * hw_c;
  const float* sp = style + plane * hw_s;
  float* op = out + plane * hw_c;
  const Stats c = plane_stats(cp, hw_c, red);
  const Stats s = plane_stats(sp, hw_s, red);
  const float std_c = sqrtf(c.m2 / float(hw_c - 1) + eps_c);
  const float std_s = sqrtf(s.m2 / float(hw_s - 1) + eps_s);
  if (stats && threadIdx.x == 0) {
    stats[plane * 4 + 0] = c.mean; stats[plane * 4 + 1] = std_c;
    stats[plane * 4 + 2] = s.mean; stats[plane * 4 + 3] = std_s;
  }
  const float inv_c = 1.f / std_c, beta = 1.f - alpha;
  for (int i = threadIdx.x; i < hw_c; i += NT) {
    const float v = cp[i];
    op[i] = alpha * ((v - c.mean) * inv_c * std_s + s.mean) + beta * v;
  }
}

__global__ __launch_bounds__(NT) void meanstd_kernel(const float* __restrict__ x,
                                                    float* __restrict__ mean,
                                                    float* __restrict__ stdv, int hw, float eps) {
  __shared__ float red[NT / 64];
  const size_t plane = blockIdx.x;
  const Stats s = plane_stats(x + plane * hw, hw, red);
  if (threadIdx.x == 0) {
    mean[plane] = s.mean;
    stdv[plane] = sqrtf(s.m2 / float(hw - 1) + eps);
  }
}

// Large planes (>= 128 KB: the loss network's 64/128-channel taps at training sizes, 256 KB - 1 MB per plane).
// One block per plane as above would stream the plane twice with few loads in flight (measured 2.5 TB/s at batch 8,
// 512 x 512).  Here: 1024 threads, four 16-B loads in flight per thread, ONE pass with shifted sums
//   S1 = sum(x - K), S2 = sum((x - K)^2), K = the average of 64 samples spread over the plane,
//   mean = K + S1/n,  m2 = S2 - S1^2/n
// (K lies within a fraction of a standard deviation of the mean, so the cancellation in m2 costs a few ulp whatever
// the data; every wave computes the same K from the same 64 loads, no barrier; per-thread partials are 64-256
// terms, then a fixed-order tree).
constexpr int NTL = 1024;

__global__ __launch_bounds__(NTL) void meanstd_big_kernel(const float* __restrict__ x,
                                                         float* __restrict__ mean,
                                                         float* __restrict__ stdv, int hw, float eps) {
  __shared__ float red[NTL / 64];
  const size_t plane = blockIdx.x;
  const float* p = x + plane * hw;
  const float4* p4 = reinterpret_cast<const float4*>(p);
  const int n4 = hw >> 2;
  const float K = st::wave_sum(p[size_t(threadIdx.x & 63) * size_t(hw >> 6)]) * (1.f / 64.f);
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  int i = threadIdx.x;
  for (; i + 3 * NTL < n4; i += 4 * NTL) {
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ldg_stream(p4 + i + k * NTL);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = v[k].x - K, b = v[k].y - K, c = v[k].z - K, d = v[k].w - K;
      s1[k] += (a + b) + (c + d);
      s2[k] += (a * a + b * b) + (c * c + d * d);
    }
  }
  for (; i < n4; i += NTL) {
    const float4 v = ldg_stream(p4 + i);
    const float a = v.x - K, b = v.y - K, c = v.z - K, d = v.w - K;
    s1[0] += (a + b) + (c + d);
    s2[0] += (a * a + b * b) + (c * c + d * d);
  }
  const float S1 = st::block_sum<NTL>((s1[0] + s1[1]) + (s1[2] + s1[3]), red);
  const float S2 = st::block_sum<NTL>((s2[0] + s2[1]) + (s2[2] + s2[3]), red);
  if (threadIdx.x == 0) {
    const float n = float(hw);
    mean[plane] = K + S1 / n;
    const float m2 = fmaxf(S2 - S1 * S1 / n, 0.f);
    stdv[plane] = sqrtf(m2 / float(hw - 1) + eps);
  }
}

// ---- backward --------------------------------------------------------------------------
// out = alpha * ((x - mu_c)/sg_c * sg_s + mu_s) + (1 - alpha) * x,  sg = sqrt(var_unbiased + eps).
// Per plane: S0 = sum g, S1 = sum g*xhat; then
//   d eps_c  = -alpha * sg_s * S1 / (2 sg_c^2)          d eps_s = alpha * S1 / (2 sg_s)
//   dx_j     = alpha*sg_s/sg_c * (g_j - S0/n - xhat_j*S1/(n-1)) + (1-alpha) g_j
//   dstyle_i = alpha * (S0/n_s + S1 * (s_i - mu_s) / ((n_s-1) sg_s))
__global__ __launch_bounds__(NT) void adain_bwd_kernel(const float* __restrict__ gout,
                                                      const float* __restrict__ content,
                                                      const float* __restrict__ style,
                                                      const float* __restrict__ stats, float alpha,
                                                      float* __restrict__ dcontent,
                                                      float* __restrict__ dstyle,
                                                      float* __restrict__ deps_part, int hw_c, int hw_s) {
  __shared__ float red[NT / 64];
  const size_t plane = blockIdx.x;
  const float mu_c = stats[plane * 4 + 0], sg_c = stats[plane * 4 + 1];
  const float mu_s = stats[plane * 4 + 2], sg_s = stats[plane * 4 + 3];
  const float* gp = gout + plane * hw_c;
  const float* xp = content + plane * hw_c;
  const float inv_c = 1.f / sg_c;
  float s0 = 0.f, s1 = 0.f;
  for (int i = threadIdx.x; i < hw_c; i += NT) {
    const float g = gp[i];
    s0 += g;
    s1 += g * ((xp[i] - mu_c) * inv_c);
  }
  s0 = st::block_sum<NT>(s0, red);
  s1 = st::block_sum<NT>(s1, red);
  if (threadIdx.x == 0) {
    deps_part[plane * 2 + 0] = -alpha * sg_s * s1 * 0.5f * inv_c * inv_c;
    deps_part[plane * 2 + 1] = alpha * s1 * 0.5f / sg_s;
  }
  if (dcontent) {
    float* dp = dcontent + plane * hw_c;
    const float a = alpha * sg_s * inv_c, beta = 1.f - alpha;
    const float m0 = s0 / float(hw_c), m1 = s1 / float(hw_c - 1);
    for (int i = threadIdx.x; i < hw_c; i += NT) {
      const float g = gp[i];
      dp[i] = a * (g - m0 - (xp[i] - mu_c) * inv_c * m1) + beta * g;
    }
  }
  if (dstyle) {
    const float* sp = style + plane * hw_s;
    float* dp = dstyle + plane * hw_s;
    const float k0 = alpha * s0 / float(hw_s), k1 = alpha * s1 / (float(hw_s - 1) * sg_s);
    for (int i = threadIdx.x; i < hw_s; i += NT) dp[i] = k0 + k1 * (sp[i] - mu_s);
  }
}

// deps[k] = sum over planes of deps_part[plane][k], fixed order
__global__ __launch_bounds__(NT) void deps_reduce_kernel(const float* __restrict__ part, int planes,
                                                        float* __restrict__ deps) {
  __shared__ float red[NT / 64];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < planes; i += NT) { a += part[(size_t)i * 2]; b += part[(size_t)i * 2 + 1]; }
  a = st::block_sum<NT>(a, red);
  b = st::block_sum<NT>(b, red);
  if (threadIdx.x == 0) { deps[0] = a; deps[1] = b; }
}

}  // namespace

extern "C" int st_adain_bwd(const float* gout, const float* content, const float* style,
                            const float* stats, float alpha, float* dcontent, float* dstyle,
                            float* deps, float* workspace, int n, int c, int hw_content, int hw_style,
                            st_stream_t stream) {
  ST_REQUIRE(gout && content && style && stats && deps && workspace, "st_adain_bwd: null pointer");
  ST_REQUIRE(n > 0 && c > 0 && hw_content > 0 && hw_style > 0,
             "st_adain_bwd: bad dims n=%d c=%d hw_c=%d hw_s=%d", n, c, hw_content, hw_style);
  const long long planes = (long long)n * c;
  ST_REQUIRE(planes <= 0x7fffffffLL, "st_adain_bwd: too many planes");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(adain_bwd_kernel, dim3((unsigned)planes), dim3(NT), 0, s, gout, content, style, stats,
                     alpha, dcontent, dstyle, workspace, hw_content, hw_style);
  hipLaunchKernelGGL(deps_reduce_kernel, dim3(1), dim3(NT), 0, s, workspace, (int)planes, deps);
  return st::check_launch("st_adain_bwd");
}

static int adain_fwd_impl(const float* content, const float* style, float* out, int n, int c,
                          int hw_content, int hw_style, float eps_content, float eps_style,
                          const float* eps_dev, float alpha, float* stats, st_stream_t stream) {
  ST_REQUIRE(content && style && out, "st_adain_fwd: null pointer");
  ST_REQUIRE(n > 0 && c > 0 && hw_content > 0 && hw_style > 0,
             "st_adain_fwd: bad dims n=%d c=%d hw_c=%d hw_s=%d", n, c, hw_content, hw_style);
  // transformer.py:19 `assert 0 < eps < 1.0`
  ST_REQUIRE(eps_dev || (eps_content > 0.f && eps_content < 1.f && eps_style > 0.f && eps_style < 1.f),
             "st_adain_fwd: eps must be in (0,1): %g %g", eps_content, eps_style);
  const long long planes = (long long)n * c;
  ST_REQUIRE(planes <= 0x7fffffffLL, "st_adain_fwd: too many planes");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)planes), block(NT);
  const bool vec_ok = (hw_content % 4 == 0) && (hw_style % 4 == 0) &&
                      (reinterpret_cast<uintptr_t>(content) % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(style) % 16 == 0) &&
                      (reinterpret_cast<uintptr_t>(out) % 16 == 0) &&
                      (!stats || reinterpret_cast<uintptr_t>(stats) % 16 == 0);
  const int need = std::max(hw_content, hw_style);
  if (vec_ok && need <= NT * 4 * 4) {
    hipLaunchKernelGGL(adain_reg_kernel<4>, grid, block, 0, s, content, style, out, hw_content,
                       hw_style, eps_content, eps_style, alpha, stats, eps_dev);
  } else if (vec_ok && need <= NT * 4 * 16) {
    hipLaunchKernelGGL(adain_reg_kernel<16>, grid, block, 0, s, content, style, out, hw_content,
                       hw_style, eps_content, eps_style, alpha, stats, eps_dev);
  } else if (vec_ok && hw_content <= NTB_MAX_HW && hw_style >= 2) {
    // 512 threads x 32 float4.  The 1024 x 16 form (4 waves per SIMD at <= 128 registers, ST_ADAIN_BIG_1024=1) was
    // measured SLOWER on the 2048 x 2048 case: 107 us against 77 us.
    static const bool narrow = getenv("ST_ADAIN_BIG_1024") == nullptr;
    if (narrow)
      hipLaunchKernelGGL((adain_big_kernel<512, 32>), grid, dim3(512), 0, s, content, style, out, hw_content,
                         hw_style, eps_content, eps_style, alpha, stats, eps_dev);
    else
      hipLaunchKernelGGL((adain_big_kernel<1024, 16>), grid, dim3(1024), 0, s, content, style, out, hw_content,
                         hw_style, eps_content, eps_style, alpha, stats, eps_dev);
  } else {
    hipLaunchKernelGGL(adain_stream_kernel, grid, block, 0, s, content, style, out, hw_content,
                       hw_style, eps_content, eps_style, alpha, stats, eps_dev);
  }
  return st::check_launch("st_adain_fwd");
}

extern "C" int st_adain_fwd(const float* content, const float* style, float* out, int n, int c,
                            int hw_content, int hw_style, float eps_content, float eps_style,
                            float alpha, float* stats, st_stream_t stream) {
  return adain_fwd_impl(content, style, out, n, c, hw_content, hw_style, eps_content, eps_style, nullptr, alpha,
                        stats, stream);
}

extern "C" int st_adain_fwd_dev_eps(const float* content, const float* style, float* out, int n, int c,
                                    int hw_content, int hw_style, const float* eps_dev, float alpha,
                                    float* stats, st_stream_t stream) {
  ST_REQUIRE(eps_dev, "st_adain_fwd_dev_eps: null eps pointer");
  return adain_fwd_impl(content, style, out, n, c, hw_content, hw_style, 0.f, 0.f, eps_dev, alpha, stats, stream);
}

extern "C" int st_meanstd_fwd(const float* x, float* mean, float* stdv, int planes, int hw,
                              float eps, st_stream_t stream) {
  ST_REQUIRE(x && mean && stdv, "st_meanstd_fwd: null pointer");
  ST_REQUIRE(planes > 0 && hw > 0, "st_meanstd_fwd: bad dims planes=%d hw=%d", planes, hw);
  if (hw >= 32768 && hw % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 && !getenv("ST_MEANSTD_TWO_PASS")) {
    hipLaunchKernelGGL(meanstd_big_kernel, dim3(planes), dim3(NTL), 0, (hipStream_t)stream, x, mean, stdv, hw, eps);
    return st::check_launch("st_meanstd_fwd");
  }
  hipLaunchKernelGGL(meanstd_kernel, dim3(planes), dim3(NT), 0, (hipStream_t)stream, x, mean,
                     stdv, hw, eps);
  return st::check_launch("st_meanstd_fwd");
}
