// Internal helpers shared by the .hip translation units of libtrl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "trl_hip.h"

void trl_set_error(const char *fmt, ...);

#define TRL_REQUIRE(cond, ...)      \
  do {                              \
    if (!(cond)) {                  \
      trl_set_error(__VA_ARGS__);   \
      return TRL_EINVAL;            \
    }                               \
  } while (0)

// Launch errors only: nothing here synchronises, so an asynchronous fault surfaces at
// the caller's next synchronising call, as with any HIP launch.
#define TRL_CHECK_LAUNCH(what)                                                   \
  do {                                                                           \
    hipError_t e__ = hipGetLastError();                                          \
    if (e__ != hipSuccess) {                                                     \
      trl_set_error("%s: launch failed: %s", what, hipGetErrorString(e__));      \
      return TRL_EHIP;                                                           \
    }                                                                            \
  } while (0)

static inline int64_t trl_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// 64-lane wavefront sum by butterfly shuffles: every lane ends with the total, and the
// order of additions is fixed, so reductions built on it are run-to-run reproducible.
__device__ __forceinline__ float trl_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block sum for blockDim.x == 64 * NW; result valid in every thread. `red` holds NW floats.
template <int NW>
__device__ __forceinline__ float trl_block_sum(float v, float *red) {
  v = trl_wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();  // protects `red` against the previous use
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NW; ++i) s += red[i];
  return s;
}
