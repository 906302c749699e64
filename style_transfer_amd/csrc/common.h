// Shared host-side helpers for libst_hip.so (error convention of include/st_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "st_hip.h"

namespace st {

char* error_buffer();  // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ST_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return ST_OK;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// row stride (floats) of the padded (w + 2)-wide data-gradient workspace: a multiple of 4 so that rows
// start 16-B aligned and the kernels can store 16 B per lane
__host__ __device__ inline int padded_row(int w) { return (w + 2 + 3) & ~3; }
inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }

// Opt a kernel in to more than the default 64 KB of dynamic LDS: once per (launch site, device), thread-safe.
// A launch site keeps one `static st::DynLdsOptIn` per kernel instantiation and calls allow_dynamic_lds() before
// the launch; the attribute is per device, so a second GPU in the same process gets its own call.
struct DynLdsOptIn {
  std::atomic<uint64_t> done{0};   // bit d: device d has the attribute
};
template <typename K>
inline int allow_dynamic_lds(K kernel, size_t bytes, DynLdsOptIn& state, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  const uint64_t bit = uint64_t(1) << dev;
  if (state.done.load(std::memory_order_acquire) & bit) return ST_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess)
    return fail(ST_ERR_LAUNCH, "%s: hipFuncSetAttribute(%zu bytes of dynamic LDS): %s", what, bytes,
                hipGetErrorString(e));
  state.done.fetch_or(bit, std::memory_order_release);
  return ST_OK;
}

}  // namespace st

// launch `kernel` with `lds` bytes of dynamic LDS after opting it in on the current device (per launch site)
#define ST_LAUNCH_DYN_LDS(what, kernel, grid, block, lds, stream, ...)                          \
  do {                                                                                          \
    static st::DynLdsOptIn st_optin_;                                                           \
    const int st_rc_ = st::allow_dynamic_lds(kernel, (lds), st_optin_, what);                   \
    if (st_rc_ != ST_OK) return st_rc_;                                                         \
    hipLaunchKernelGGL(kernel, grid, block, (lds), stream, __VA_ARGS__);                        \
  } while (0)

#define ST_REQUIRE(cond, ...) \
  do {                        \
    if (!(cond)) return st::fail(ST_ERR_BAD_ARG, __VA_ARGS__); \
  } while (0)

// ---- device helpers ---------------------------------------------------------------------
#ifdef __HIPCC__
namespace st {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Block-wide sum for blockDim.x == NT (multiple of 64); result broadcast to every thread.
// `red` is an LDS array of at least NT/64 floats.  Fixed order => run-to-run deterministic.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6;
  __syncthreads();  // protect `red` from a previous use
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) t += red[w];
  return t;
}

}  // namespace st
#endif
