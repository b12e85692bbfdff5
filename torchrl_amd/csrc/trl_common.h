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

// ---- one element of the TD / critic loss and its derivative: squared error (dqn_controller.py:77-78,
// ddpg_controller.py:112) or Huber (extension)
__device__ __forceinline__ float trl_loss_elem(float td, int kind, float delta, float *g) {
  if (kind == TRL_LOSS_MSE) {
    *g = 2.0f * td;  // pow(2) backward: 2 * td
    return td * td;
  }
  const float a = fabsf(td);
  if (a <= delta) {
    *g = td;
    return 0.5f * td * td;
  }
  *g = td > 0.f ? delta : -delta;
  return delta * (a - 0.5f * delta);
}

// ---- torch.optim.Adam's default single-tensor update on one element, shared by the flat-buffer
// update kernels (trl_loss.hip) and the fused learner step's reduce launch (trl_learn.hip).
struct TrlAdamScalars {
  float w1;        // 1 - beta1            (exp_avg.lerp_ weight)
  float beta2;     //                      (exp_avg_sq.mul_)
  float c2;        // 1 - beta2            (addcmul_ value)
  float bc2_sqrt;  // sqrt(1 - beta2^step)
  float eps;
  float neg_step;  // -(lr / (1 - beta1^step))
  float scale;     // gradient scale (1 = none)
  float clip;      // clamp bound (<= 0 = none)
};

#include <math.h>
static inline __host__ __device__ TrlAdamScalars trl_adam_scalars(double lr, double beta1, double beta2, double eps,
                                                                  int64_t step, double grad_scale, double clip_value) {
  TrlAdamScalars s;
  s.w1 = (float)(1.0 - beta1);
  s.beta2 = (float)beta2;
  s.c2 = (float)(1.0 - beta2);
  s.bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
  s.eps = (float)eps;
  s.neg_step = (float)(-(lr / (1.0 - pow(beta1, (double)step))));
  s.scale = (float)grad_scale;
  s.clip = (float)clip_value;
  return s;
}

// scale + clamp the gradient (written back when changed), then lerp_ / mul_().addcmul_() / addcdiv_
// in torch's op order, one f32 rounding per op
__device__ __forceinline__ void trl_adam_elem(float *p, float *g, float *m, float *v, const TrlAdamScalars &s) {
  float gi = *g;
  if (s.scale != 1.0f) gi = __fmul_rn(gi, s.scale);
  if (s.clip > 0.f) gi = fminf(fmaxf(gi, -s.clip), s.clip);
  if (s.scale != 1.0f || s.clip > 0.f) *g = gi;
  float mi = *m, vi = *v;
  mi = __fadd_rn(mi, __fmul_rn(s.w1, __fsub_rn(gi, mi)));
  vi = __fadd_rn(__fmul_rn(vi, s.beta2), __fmul_rn(__fmul_rn(s.c2, gi), gi));
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vi), s.bc2_sqrt), s.eps);
  *p = __fadd_rn(*p, __fdiv_rn(__fmul_rn(s.neg_step, mi), denom));
  *m = mi;
  *v = vi;
}

// ---- counter-based generator (Philox-4x32-10) shared by the device sampler (trl_replay.hip) and the act kernels
// (trl_act.hip): 64 random bits as a function of (seed, call number, position, attempt) alone
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1;
  c[3] = (uint32_t)p0;
  c[0] = n0;
  c[2] = n2;
}

__device__ __forceinline__ uint64_t philox64(uint64_t seed, uint64_t call, uint32_t pos, uint32_t attempt) {
  uint32_t c[4] = {(uint32_t)call, (uint32_t)(call >> 32), pos, attempt};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return ((uint64_t)c[0] << 32) | c[1];
}

// ---- internal launchers of csrc/trl_linear_u8.hip, used by the DQN step chain (csrc/trl_offpolicy.hip)
struct TrlU8Update {
  float *pw, *mw, *vw;   // weight [N, K]: parameter, exp_avg, exp_avg_sq
  float *pb, *mb, *vb;   // bias [N]
  TrlAdamScalars s;
};
int64_t trl_u8_fwd_ws_floats(int64_t B, int32_t N, int32_t K, int P);
// rec != NULL: the uint8 rows are read in place out of a replay ring -- row b of x starts at x + rec[b] * stride bytes
// (x = ring base + the frame column's offset in a record, rec[b] = the physical record of batch row b)
int trl_u8_fwd_launch(const char *fn, int P, const uint8_t *const *x, const float *const *w, const float *const *b,
                      float *const *y, int64_t B, int32_t K, int32_t N, int32_t act, float *ws, hipStream_t st,
                      const int32_t *rec, int64_t stride);
int trl_u8_wgrad_launch(const char *fn, const float *g, const uint8_t *x, int64_t B, int32_t K, int32_t N, float *dw,
                        float *db, const TrlU8Update *u, hipStream_t st, const int32_t *rec, int64_t stride);

// ---- csrc/trl_replay.hip: the prologue of a learner step that reads its frames in place (csrc/trl_offpolicy.hip).  One
// launch: rec[b] = the physical record of logical row idx[b] (idx: device or pinned host memory), and the three
// one-float columns of a transition (action, reward, done) gathered densely
int trl_ring_rows_launch(const char *fn, const void *ring, int64_t capacity, int64_t stride, int64_t head, int64_t count,
                         const int64_t *idx, int64_t B, const int32_t *f32_col_off /*[3]*/, int32_t *rec, float *action,
                         float *reward, float *done, hipStream_t st);

// ---- device-resident Adam state (csrc/trl_loss.hip), for chains that are captured into a HIP graph
int trl_adam_prep_launch(const char *fn, trl_adam_state *state, hipStream_t st);     // ++step, derive the scalars
int trl_adam_dev_launch(const char *fn, float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                        const trl_adam_state *state, hipStream_t st);                // update with the derived scalars
