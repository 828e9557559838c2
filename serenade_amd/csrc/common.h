// Shared host/device helpers for libserenade_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#include "serenade_hip.h"

void srn_set_error(const char* fmt, ...);

#define SRN_CHECK_ARG(cond, ...)      \
  do {                                \
    if (!(cond)) {                    \
      srn_set_error(__VA_ARGS__);     \
      return -1;                      \
    }                                 \
  } while (0)

#define SRN_CHECK_HIP(expr)                                                         \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) {                                                         \
      srn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return -2;                                                                    \
    }                                                                               \
  } while (0)

#define SRN_CHECK_LAUNCH() SRN_CHECK_HIP(hipGetLastError())

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel.  One of these per kernel
// instantiation (function-local static) remembers, per device and thread-safely, that the size has been granted,
// so a process that moves from cuda:0 to cuda:1, or launches from several threads, still gets its LDS.
// `bytes` must be the same on every call for one kernel (pass the kernel's maximum).
struct SrnSmemAttr {
  std::atomic<unsigned long long> done{0};
  int ensure(const void* fn, int bytes) {
    int dev = 0;
    SRN_CHECK_HIP(hipGetDevice(&dev));
    const bool tracked = dev >= 0 && dev < 64;
    const unsigned long long bit = tracked ? 1ull << dev : 0ull;
    if (tracked && (done.load(std::memory_order_acquire) & bit)) return 0;
    SRN_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.fetch_or(bit, std::memory_order_release);
    return 0;
  }
};

// ---- device math --------------------------------------------------------------------------
// Mish(x) = x * tanh(softplus(x)) (decoder.py:72,84).  tanh(log1p(e^x)) = n / (n + 2) with
// n = e^x (e^x + 2): one exp, one divide, no cancellation for very negative x; softplus
// threshold 20 as in torch (x > 20 -> x).
__device__ __forceinline__ float srn_mish(float x) {
  const float e = expf(fminf(x, 20.0f));
  const float n = e * (e + 2.0f);
  const float y = x * (n / (n + 2.0f));
  return x > 20.0f ? x : y;  // branch-free select
}

__device__ __forceinline__ float srn_silu(float x) { return x / (1.0f + expf(-x)); }

__device__ __forceinline__ float srn_gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

__device__ __forceinline__ float srn_act(float v, int act, float slope) {
  switch (act) {
    case SRN_ACT_LEAKY: return v > 0.0f ? v : v * slope;
    case SRN_ACT_SILU: return srn_silu(v);
    case SRN_ACT_MISH: return srn_mish(v);
    default: return v;
  }
}

// 64-lane wavefront all-reduce (sum) through cross-lane shuffles.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
