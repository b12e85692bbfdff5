// The per-env-step side of the on-policy / off-policy loops, one launch each:
//   * rollout push: the env step's transitions (one packed host row per env, pinned) scattered into the
//     [T, N, .] or blocked [N/E, T, E, .] rollout columns -- replaces the per-env
//     `buffer.extend([transition])` of examples/ppo.py:28-35 / examples/a2c.py:26-28 (a torch.cat of every
//     column per env per step) and, in this package's first version, one H2D copy + five slice copies;
//   * exploration: epsilon-greedy over Q rows (dqn_controller.py:8-24, 57-59), Normal sampling of the PPO
//     policy (ppo_controller.py:24-30), Categorical sampling of the A2C policy (a2c_controller.py:21-27) and
//     the Ornstein-Uhlenbeck noise of DDPG (ddpg_controller.py:7-39, 87-99), drawn ON THE DEVICE from a
//     counter-based generator (Philox-4x32-10): the action batch never leaves the GPU.  The reference draws
//     from numpy's global generator on the host, so parity with it is distributional, not draw for draw;
//     a result here is a function of (seed, call number, row) alone; the caller numbers its calls.
#include <math.h>

#include "trl_common.h"

namespace {

// ---------------------------------------------------------------------------------------------- rollout push
struct PushCols {
  int32_t ncols;
  int32_t width[TRL_MAX_COLS];     // floats per env
  int32_t off[TRL_MAX_COLS + 1];   // prefix sum of the widths inside a packed row
  int32_t benv[TRL_MAX_COLS];      // envs per group of the destination (1: dense [N, width])
  int64_t gstride[TRL_MAX_COLS];   // floats between two groups of the destination
  float *dst[TRL_MAX_COLS];        // already offset to the time step being written
};

// one lane per float of the step's [N, row_w] block; the packed row is read straight from pinned host memory
__global__ __launch_bounds__(256) void k_rollout_push(const float *__restrict__ rows, int64_t N, int row_w, PushCols c) {
  const int64_t total = N * row_w;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t n = e / row_w;
    const int j = (int)(e - n * row_w);
    const float v = rows[e];
#pragma unroll
    for (int i = 0; i < TRL_MAX_COLS; ++i) {
      if (i < c.ncols && j >= c.off[i] && j < c.off[i + 1]) {
        const int64_t g = n / c.benv[i], r = n - g * c.benv[i];
        c.dst[i][g * c.gstride[i] + r * c.width[i] + (j - c.off[i])] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- exploration
// 23 random bits + one half: k + 0.5 is exact for every k < 2^23 (24 significant bits), so the value is strictly inside
// (0, 1) -- with 24 bits the half was rounded away for k >= 2^23 and k = 2^24 - 1 gave exactly 1.0
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }

// two independent standard normals from 64 random bits (Box-Muller)
__device__ __forceinline__ void normal2(uint64_t bits, float *z0, float *z1) {
  const float u = u01((uint32_t)(bits >> 32)), v = u01((uint32_t)bits);
  const float r = sqrtf(-2.0f * logf(u));
  float sn, cs;
  sincosf(6.283185307179586f * v, &sn, &cs);
  *z0 = r * cs;
  *z1 = r * sn;
}

// epsilon-greedy: row b takes argmax_a q[b, a] (first maximum, as torch.argmax) with probability 1 - eps + eps / A
// and every other action with probability eps / A -- the categorical of dqn_controller.py:16-22
__global__ __launch_bounds__(256) void k_act_eps_greedy(const float *__restrict__ q, int64_t n, int A, float eps,
                                                        uint64_t seed, uint64_t call,
                                                        int64_t *__restrict__ out) {
  for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < n; b += (int64_t)gridDim.x * 256) {
    const float *row = q + b * A;
    int best = 0;
    float bv = row[0];
    for (int a = 1; a < A; ++a)
      if (row[a] > bv) { bv = row[a]; best = a; }
    const uint64_t bits = philox64(seed, call, (uint32_t)b, (uint32_t)(b >> 32));
    const bool explore = u01((uint32_t)(bits >> 32)) < eps;
    const int uni = (int)(((uint64_t)(uint32_t)bits * (uint64_t)A) >> 32);     // uniform over [0, A)
    out[b] = explore ? uni : best;
  }
}

// action[b, a] = mean[b, a] + exp(log_std[a]) * N(0, 1): Normal(mean, std).sample() of models.py:139-141
__global__ __launch_bounds__(256) void k_act_normal(const float *__restrict__ mean, const float *__restrict__ log_std,
                                                    int64_t total, int A, uint64_t seed,
                                                    uint64_t call, float *__restrict__ out) {
  const int64_t pairs = (total + 1) >> 1;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < pairs; p += (int64_t)gridDim.x * 256) {
    float z0, z1;
    normal2(philox64(seed, call, (uint32_t)p, (uint32_t)(p >> 32)), &z0, &z1);
    const int64_t e0 = 2 * p, e1 = 2 * p + 1;
    out[e0] = mean[e0] + expf(log_std[e0 % A]) * z0;
    if (e1 < total) out[e1] = mean[e1] + expf(log_std[e1 % A]) * z1;
  }
}

// Categorical(probs).sample(): inverse CDF over the row's probabilities (normalised by their sum, as
// torch.distributions.Categorical does, models.py:105-107)
__global__ __launch_bounds__(256) void k_act_categorical(const float *__restrict__ probs, int64_t n, int A, uint64_t seed,
                                                         uint64_t call,
                                                         int64_t *__restrict__ out) {
  for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < n; b += (int64_t)gridDim.x * 256) {
    const float *row = probs + b * A;
    float s = 0.f;
    for (int a = 0; a < A; ++a) s += row[a];
    const float u = u01((uint32_t)(philox64(seed, call, (uint32_t)b, (uint32_t)(b >> 32)) >> 32)) * s;
    // (u < s always in exact arithmetic; if the running sum's rounding leaves u >= acc to the end, the pick is the
    // last action that HAS probability, never a zero-probability one)
    float acc = 0.f;
    int pick = -1, last = 0;
    for (int a = 0; a < A; ++a) {
      if (row[a] > 0.f) last = a;
      acc += row[a];
      if (pick < 0 && u < acc && row[a] > 0.f) pick = a;
    }
    out[b] = pick < 0 ? last : pick;
  }
}

// OU step + noisy, clipped action.  ONE state of A values is shared by every env (ddpg_controller.py:29-36):
//   x <- x + theta * (mu - x) + sigma * N(0, 1);   action[b, a] = clip(actor[b, a] + x[a], low, high)
// Every workgroup derives the same new state from the old one (same counter, position = a); workgroup 0 writes
// it to state_out (a second buffer: no workgroup may read a half-updated state).
__global__ __launch_bounds__(256) void k_act_ou(const float *__restrict__ actor, int64_t n, int A,
                                                const float *__restrict__ state_in, float *__restrict__ state_out,
                                                float theta, float mu, float sigma, float low, float high, uint64_t seed,
                                                uint64_t call, float *__restrict__ out) {
  __shared__ float x_new[64];
  if ((int)threadIdx.x < A) {
    float z0, z1;
    normal2(philox64(seed, call, threadIdx.x, 0x4f55u), &z0, &z1);
    const float x = state_in[threadIdx.x];
    const float nx = x + (theta * (mu - x) + sigma * z0);
    x_new[threadIdx.x] = nx;
    if (blockIdx.x == 0) state_out[threadIdx.x] = nx;
  }
  __syncthreads();
  const int64_t total = n * A;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const float v = actor[e] + x_new[e % A];
    out[e] = fminf(fmaxf(v, low), high);
  }
}

unsigned act_grid(int64_t work) {
  const int64_t g = trl_cdiv(work, 256);
  return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int trl_rollout_push(const float *rows_pinned, int64_t N, const trl_rollout_cols *cols, void *stream) {
  const char *fn = "trl_rollout_push";
  TRL_REQUIRE(rows_pinned && cols, "%s: null pointer", fn);
  TRL_REQUIRE(N > 0 && N < (1ll << 31), "%s: N=%lld out of range", fn, (long long)N);
  TRL_REQUIRE(cols->ncols >= 1 && cols->ncols <= TRL_MAX_COLS, "%s: ncols=%d out of [1, %d]", fn, cols->ncols, TRL_MAX_COLS);
  PushCols c{};
  c.ncols = cols->ncols;
  for (int i = 0; i < cols->ncols; ++i) {
    TRL_REQUIRE(cols->width[i] > 0 && cols->dst[i] != nullptr, "%s: column %d: empty or null", fn, i);
    const int be = cols->block_envs[i];
    TRL_REQUIRE(be >= 1 && N % be == 0, "%s: column %d: block_envs=%d must divide N=%lld", fn, i, be, (long long)N);
    TRL_REQUIRE(cols->group_stride[i] >= (int64_t)be * cols->width[i], "%s: column %d: group stride %lld overlaps", fn, i,
                (long long)cols->group_stride[i]);
    c.width[i] = cols->width[i];
    c.off[i + 1] = c.off[i] + cols->width[i];
    c.benv[i] = be;
    c.gstride[i] = cols->group_stride[i];
    c.dst[i] = cols->dst[i];
  }
  const int row_w = c.off[c.ncols];
  hipLaunchKernelGGL(k_rollout_push, dim3(act_grid(N * row_w)), dim3(256), 0, (hipStream_t)stream, rows_pinned, N, row_w, c);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

#define TRL_ACT_COMMON(fn)                                                                                          \
  TRL_REQUIRE(n >= 0 && n < (1ll << 40) && A >= 1 && A <= 1024, "%s: n=%lld A=%d out of range", fn, (long long)n, A);  \
  if (n == 0) return TRL_OK;                                                                                        \
  hipStream_t st = (hipStream_t)stream

extern "C" int trl_act_epsilon_greedy(const float *q, int64_t n, int32_t A, double eps, uint64_t seed,
                                      uint64_t call, int64_t *actions, void *stream) {
  const char *fn = "trl_act_epsilon_greedy";
  TRL_REQUIRE(q && actions, "%s: null pointer", fn);
  TRL_ACT_COMMON(fn);
  hipLaunchKernelGGL(k_act_eps_greedy, dim3(act_grid(n)), dim3(256), 0, st, q, n, (int)A, (float)eps, seed,
                     call, actions);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

extern "C" int trl_act_normal(const float *mean, const float *log_std, int64_t n, int32_t A, uint64_t seed,
                              uint64_t call, float *actions, void *stream) {
  const char *fn = "trl_act_normal";
  TRL_REQUIRE(mean && log_std && actions, "%s: null pointer", fn);
  TRL_ACT_COMMON(fn);
  hipLaunchKernelGGL(k_act_normal, dim3(act_grid((n * A + 1) / 2)), dim3(256), 0, st, mean, log_std, n * A, (int)A, seed,
                     call, actions);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

extern "C" int trl_act_categorical(const float *probs, int64_t n, int32_t A, uint64_t seed, uint64_t call,
                                   int64_t *actions, void *stream) {
  const char *fn = "trl_act_categorical";
  TRL_REQUIRE(probs && actions, "%s: null pointer", fn);
  TRL_ACT_COMMON(fn);
  hipLaunchKernelGGL(k_act_categorical, dim3(act_grid(n)), dim3(256), 0, st, probs, n, (int)A, seed,
                     call, actions);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

extern "C" int trl_act_ou_noise(const float *actor_out, int64_t n, int32_t A, const float *state_in, float *state_out,
                                double theta, double mu, double sigma, double low, double high, uint64_t seed,
                                uint64_t call, float *actions, void *stream) {
  const char *fn = "trl_act_ou_noise";
  TRL_REQUIRE(actor_out && state_in && state_out && actions && state_in != state_out, "%s: null pointer or aliased state", fn);
  TRL_REQUIRE(A <= 64, "%s: action size %d > 64", fn, A);
  TRL_ACT_COMMON(fn);
  hipLaunchKernelGGL(k_act_ou, dim3(act_grid(n * A)), dim3(256), 0, st, actor_out, n, (int)A, state_in, state_out,
                     (float)theta, (float)mu, (float)sigma, (float)low, (float)high, seed,
                     call, actions);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}
