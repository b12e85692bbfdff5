// Fused loss kernels with their backward seeds, and the flat-buffer update kernels.
//   TD (DQN / Double-DQN)   dqn_controller.py:64-78
//   DDPG critic             ddpg_controller.py:107-112
//   PPO clipped surrogate   ppo_controller.py:55-68 (+ Normal.log_prob / entropy closed forms)
//   A2C                     a2c_controller.py:48-59 (+ Categorical(probs) closed forms)
//   Adam / clip / Polyak    torch.optim.Adam defaults, ppo_controller.py:72-73,
//                           ddpg_controller.py:42-57, dqn_controller.py:85-86
// All are HBM/latency-bound elementwise + reduction kernels (a few tens of bytes per
// sample); the only matmuls of the learner stay in PyTorch-ROCm.  Reductions: 64-lane
// butterfly per wave -> LDS per block -> per-block partials in `ws` -> one finalize block,
// always in the same order, so a loss is bitwise reproducible run to run.  A batch that
// fits one block (B <= 256 rows per pass, the reference minibatch sizes 32-256) is
// finished in the same launch.
#include <math.h>

#include <mutex>

#include "trl_common.h"

namespace {

constexpr int NT = 256;        // threads per block
constexpr int NW = NT / 64;    // waves per block
constexpr int MAX_BLOCKS = 1024;
constexpr int MAX_PPO_A = 64;  // dlog_std partial columns kept in ws
constexpr int WS_FLOATS = MAX_BLOCKS * (4 + MAX_PPO_A);

inline unsigned loss_grid(int64_t B) {
  int64_t g = trl_cdiv(B, NT);
  return (unsigned)(g < 1 ? 1 : (g > MAX_BLOCKS ? MAX_BLOCKS : g));
}

// Sum column k of the [nb, K] partials with one block; valid in every thread.
__device__ __forceinline__ float sum_partials(const float *ws, int nb, int K, int k, float *red) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += NT) s += ws[i * K + k];
  return trl_block_sum<NW>(s, red);
}

__device__ __forceinline__ float loss_elem(float td, int kind, float delta, float *g) {
  return trl_loss_elem(td, kind, delta, g);
}

// ------------------------------------------------------------------------------ TD
__global__ __launch_bounds__(NT) void k_td_loss(const float *__restrict__ q,
                                                const float *__restrict__ qnt,
                                                const float *__restrict__ qno,
                                                const float *__restrict__ action,
                                                const float *__restrict__ reward,
                                                const float *__restrict__ done, int64_t B, int A,
                                                float gamma, int kind, float hdelta,
                                                float *__restrict__ loss_out, float *__restrict__ dq,
                                                float *__restrict__ ws) {
  __shared__ float red[NW];
  const float invB = 1.0f / (float)B;
  float acc = 0.f;
  for (int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x; b < B; b += (int64_t)gridDim.x * NT) {
    int a = (int)action[b];  // action.long()
    a = a < 0 ? 0 : (a >= A ? A - 1 : a);
    const float *tn = qnt + b * A;
    float nq;
    if (qno) {  // Double-DQN: first argmax of the online net, value from the target net
      const float *on = qno + b * A;
      int am = 0;
      float best = on[0];
      for (int i = 1; i < A; ++i) {
        const float x = on[i];
        if (x > best) {
          best = x;
          am = i;
        }
      }
      nq = tn[am];
    } else {
      nq = tn[0];
      for (int i = 1; i < A; ++i) nq = fmaxf(nq, tn[i]);
    }
    // reward + gamma * next_q * (1 - done)
    const float y = reward[b] + (gamma * nq) * (1.0f - done[b]);
    const float td = y - q[b * A + a];
    float g;
    acc += loss_elem(td, kind, hdelta, &g);
    if (dq) {
      const float gq = -(invB * g);  // d loss / d q[b, a]
      for (int i = 0; i < A; ++i) dq[b * A + i] = (i == a) ? gq : 0.0f;
    }
  }
  const float s = trl_block_sum<NW>(acc, red);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1)
      loss_out[0] = s * invB;
    else
      ws[blockIdx.x] = s;
  }
}

__global__ __launch_bounds__(NT) void k_mean_finalize(const float *__restrict__ ws, int nb, int64_t B,
                                                      float *__restrict__ loss_out) {
  __shared__ float red[NW];
  const float s = sum_partials(ws, nb, 1, 0, red);
  if (threadIdx.x == 0) loss_out[0] = s * (1.0f / (float)B);
}

// ------------------------------------------------------------------------------ DDPG critic
__global__ __launch_bounds__(NT) void k_ddpg_critic(const float *__restrict__ cur,
                                                    const float *__restrict__ nxt,
                                                    const float *__restrict__ reward,
                                                    const float *__restrict__ done, int64_t B,
                                                    float gamma, int kind, float hdelta,
                                                    float *__restrict__ loss_out,
                                                    float *__restrict__ dcur, float *__restrict__ ws) {
  __shared__ float red[NW];
  const float invB = 1.0f / (float)B;
  float acc = 0.f;
  for (int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x; b < B; b += (int64_t)gridDim.x * NT) {
    // reward + (1 - done) * gamma * target_q
    const float y = reward[b] + ((1.0f - done[b]) * gamma) * nxt[b];
    const float td = cur[b] - y;
    float g;
    acc += loss_elem(td, kind, hdelta, &g);
    if (dcur) dcur[b] = invB * g;
  }
  const float s = trl_block_sum<NW>(acc, red);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1)
      loss_out[0] = s * invB;
    else
      ws[blockIdx.x] = s;
  }
}

// ------------------------------------------------------------------------------ PPO
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;   // math.log(math.sqrt(2 * math.pi))
constexpr float ENT_CONST = 1.41893853320467274178f;      // 0.5 + 0.5 * math.log(2 * math.pi)

struct PpoScalars {
  float lo, hi;      // 1 - clip, 1 + clip
  float alpha, beta;
  float dls_ent;     // -beta / A: entropy's share of dloss/dlog_std
};

__device__ __forceinline__ void ppo_write_losses(float s_min, float s_crit, float s_ent, int64_t B, int A,
                                                 const PpoScalars &p, float *out) {
  const float invBA = 1.0f / (float)(B * A);
  const float actor = -(s_min * invBA);
  const float critic = s_crit * (1.0f / (float)B);
  const float ent = s_ent * invBA;
  out[0] = actor;
  out[1] = critic;
  out[2] = ent;
  out[3] = (actor + p.alpha * critic) - p.beta * ent;
}

__global__ __launch_bounds__(NT) void k_ppo_loss(
    const float *__restrict__ mean, const float *__restrict__ log_std, const float *__restrict__ value,
    const float *__restrict__ action, const float *__restrict__ old_logp, const float *__restrict__ adv,
    const float *__restrict__ returns, int64_t B, int A, PpoScalars p, float *__restrict__ losses,
    float *__restrict__ dmean, float *__restrict__ dls, float *__restrict__ dvalue,
    float *__restrict__ ws) {
  __shared__ float red[NW];
  const int K = 3 + A;
  const bool single = gridDim.x == 1;
  const float invB = 1.0f / (float)B;
  const float gmin = -(1.0f / (float)(B * A));  // d loss / d min(surr1, surr2) element
  const int64_t b0 = (int64_t)blockIdx.x * NT + threadIdx.x;
  const int64_t bstep = (int64_t)gridDim.x * NT;

  float a_crit = 0.f, a_min = 0.f, a_ent = 0.f;
  for (int64_t b = b0; b < B; b += bstep) {  // critic: mse_loss(values, returns)
    const float dv = value[b] - returns[b];
    a_crit += dv * dv;
    if (dvalue) dvalue[b] = p.alpha * ((2.0f * invB) * dv);
  }
  for (int a = 0; a < A; ++a) {
    const float sigma = expf(log_std[a]);  // self.log_std.exp()
    const float var = sigma * sigma;       // scale ** 2
    const float logsig = logf(sigma);      // scale.log()
    const float den = 2.0f * var;
    const float ent_a = ENT_CONST + logsig;
    float a_dls = 0.f;
    for (int64_t b = b0; b < B; b += bstep) {
      const int64_t i = b * A + a;
      const float d = action[i] - mean[i];
      const float sq = d * d;
      const float lp = ((-sq) / den - logsig) - LOG_SQRT_2PI;  // Normal.log_prob
      const float ratio = expf(lp - old_logp[i]);
      const float ad = adv[b];
      const float s1 = ratio * ad;
      const float rc = fminf(fmaxf(ratio, p.lo), p.hi);
      const float s2 = rc * ad;
      a_min += fminf(s1, s2);
      a_ent += ent_a;
      if (dmean) {
        // torch.minimum backward: the smaller side takes the gradient, a tie splits it
        const float g1 = s1 < s2 ? gmin : (s1 == s2 ? 0.5f * gmin : 0.f);
        const float g2 = s2 < s1 ? gmin : (s1 == s2 ? 0.5f * gmin : 0.f);
        const bool inr = ratio >= p.lo && ratio <= p.hi;  // clamp backward mask
        const float g_ratio = g1 * ad + (inr ? g2 * ad : 0.f);
        const float g_lp = g_ratio * ratio;  // exp backward
        dmean[i] = (g_lp / den) * (2.0f * d);
        a_dls += g_lp * (sq / var - 1.0f);
      }
    }
    if (dls) {
      const float s = trl_block_sum<NW>(a_dls, red);
      if (threadIdx.x == 0) {
        if (single)
          dls[a] = s + p.dls_ent;
        else
          ws[blockIdx.x * K + 3 + a] = s;
      }
    }
  }
  const float s_min = trl_block_sum<NW>(a_min, red);
  const float s_crit = trl_block_sum<NW>(a_crit, red);
  const float s_ent = trl_block_sum<NW>(a_ent, red);
  if (threadIdx.x == 0) {
    if (single) {
      ppo_write_losses(s_min, s_crit, s_ent, B, A, p, losses);
    } else {
      ws[blockIdx.x * K + 0] = s_min;
      ws[blockIdx.x * K + 1] = s_crit;
      ws[blockIdx.x * K + 2] = s_ent;
    }
  }
}

__global__ __launch_bounds__(NT) void k_ppo_finalize(const float *__restrict__ ws, int nb, int64_t B, int A,
                                                     PpoScalars p, float *__restrict__ losses,
                                                     float *__restrict__ dls) {
  __shared__ float red[NW];
  const int K = 3 + A;
  const float s_min = sum_partials(ws, nb, K, 0, red);
  const float s_crit = sum_partials(ws, nb, K, 1, red);
  const float s_ent = sum_partials(ws, nb, K, 2, red);
  if (threadIdx.x == 0) ppo_write_losses(s_min, s_crit, s_ent, B, A, p, losses);
  if (dls)
    for (int a = 0; a < A; ++a) {
      const float s = sum_partials(ws, nb, K, 3 + a, red);
      if (threadIdx.x == 0) dls[a] = s + p.dls_ent;
    }
}

__global__ __launch_bounds__(NT) void k_normal_logp(const float *__restrict__ mean,
                                                    const float *__restrict__ log_std,
                                                    const float *__restrict__ action, int64_t n, int A,
                                                    float *__restrict__ logp) {
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const int a = (int)(i % A);
    const float sigma = expf(log_std[a]);
    const float var = sigma * sigma;
    const float logsig = logf(sigma);
    const float d = action[i] - mean[i];
    logp[i] = ((-(d * d)) / (2.0f * var) - logsig) - LOG_SQRT_2PI;
  }
}

// ------------------------------------------------------------------------------ A2C
constexpr float F32_EPS = 1.1920928955078125e-07f;  // torch.finfo(torch.float32).eps
constexpr float F32_LOWEST = -3.4028234663852886e+38f;

__device__ __forceinline__ void a2c_write_losses(float s_act, float s_crit, float s_ent, int64_t B,
                                                 float alpha, float beta, float *out) {
  const float invB = 1.0f / (float)B;
  const float actor = -(s_act * invB);
  const float critic = s_crit * invB;
  const float ent = s_ent * invB;
  out[0] = actor;
  out[1] = critic;
  out[2] = ent;
  out[3] = (actor + alpha * critic) - beta * ent;
}

__global__ __launch_bounds__(NT) void k_a2c_loss(const float *__restrict__ probs,
                                                 const float *__restrict__ value,
                                                 const float *__restrict__ action,
                                                 const float *__restrict__ returns, int64_t B, int A,
                                                 float alpha, float beta, float *__restrict__ losses,
                                                 float *__restrict__ dprobs, float *__restrict__ dvalue,
                                                 float *__restrict__ ws) {
  __shared__ float red[NW];
  const float invB = 1.0f / (float)B;
  float a_act = 0.f, a_crit = 0.f, a_ent = 0.f;
  for (int64_t b = (int64_t)blockIdx.x * NT + threadIdx.x; b < B; b += (int64_t)gridDim.x * NT) {
    const float *pr = probs + b * A;
    int act = (int)action[b];
    act = act < 0 ? 0 : (act >= A ? A - 1 : act);
    const float v = value[b], ret = returns[b];
    const float advantage = ret - v;  // detached in the actor term
    const float dv = v - ret;
    a_crit += dv * dv;
    if (dvalue) dvalue[b] = alpha * ((2.0f * invB) * dv);
    float s = 0.f;  // Categorical(probs): probs / probs.sum(-1, keepdim=True)
    for (int i = 0; i < A; ++i) s += pr[i];
    float ent = 0.f, gdot = 0.f, logp = 0.f;
    for (int i = 0; i < A; ++i) {
      const float pn = pr[i] / s;
      const float cl = fminf(fmaxf(pn, F32_EPS), 1.0f - F32_EPS);  // clamp_probs
      const float logit = logf(cl);
      ent += fmaxf(logit, F32_LOWEST) * pn;
      if (i == act) logp = logit;
      if (dprobs) {
        const float dlogit = (pn >= F32_EPS && pn <= 1.0f - F32_EPS) ? 1.0f / cl : 0.f;
        float g = (beta * invB) * (logit + pn * dlogit);        // from -beta * entropy
        if (i == act) g += (-(advantage * invB)) * dlogit;      // from the actor term
        gdot += g * pn;
      }
    }
    a_ent += -ent;
    a_act += advantage * logp;
    if (dprobs) {  // back through the normalisation: (g_j - sum_a g_a p_a) / s
      for (int i = 0; i < A; ++i) {
        const float pn = pr[i] / s;
        const float cl = fminf(fmaxf(pn, F32_EPS), 1.0f - F32_EPS);
        const float logit = logf(cl);
        const float dlogit = (pn >= F32_EPS && pn <= 1.0f - F32_EPS) ? 1.0f / cl : 0.f;
        float g = (beta * invB) * (logit + pn * dlogit);
        if (i == act) g += (-(advantage * invB)) * dlogit;
        dprobs[b * A + i] = (g - gdot) / s;
      }
    }
  }
  const float s_act = trl_block_sum<NW>(a_act, red);
  const float s_crit = trl_block_sum<NW>(a_crit, red);
  const float s_ent = trl_block_sum<NW>(a_ent, red);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1) {
      a2c_write_losses(s_act, s_crit, s_ent, B, alpha, beta, losses);
    } else {
      ws[blockIdx.x * 3 + 0] = s_act;
      ws[blockIdx.x * 3 + 1] = s_crit;
      ws[blockIdx.x * 3 + 2] = s_ent;
    }
  }
}

__global__ __launch_bounds__(NT) void k_a2c_finalize(const float *__restrict__ ws, int nb, int64_t B,
                                                     float alpha, float beta, float *__restrict__ losses) {
  __shared__ float red[NW];
  const float s_act = sum_partials(ws, nb, 3, 0, red);
  const float s_crit = sum_partials(ws, nb, 3, 1, red);
  const float s_ent = sum_partials(ws, nb, 3, 2, red);
  if (threadIdx.x == 0) a2c_write_losses(s_act, s_crit, s_ent, B, alpha, beta, losses);
}

// ------------------------------------------------------------------------------ update
typedef TrlAdamScalars AdamScalars;

__global__ __launch_bounds__(NT) void k_adam(float *__restrict__ p, float *__restrict__ g,
                                             float *__restrict__ m, float *__restrict__ v, int64_t n,
                                             AdamScalars s) {
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
    trl_adam_elem(p + i, g + i, m + i, v + i, s);
}

// graph-replay form: scalars derived on the device from the device-resident step counter
__global__ void k_adam_prep(trl_adam_state *st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int64_t step = st->step + 1;
    st->step = step;
    const AdamScalars s = trl_adam_scalars(st->lr, st->beta1, st->beta2, st->eps, step, st->grad_scale, st->clip_value);
    *(AdamScalars *)st->derived = s;
  }
}

__global__ __launch_bounds__(NT) void k_adam_dev(float *__restrict__ p, float *__restrict__ g,
                                                 float *__restrict__ m, float *__restrict__ v, int64_t n,
                                                 const trl_adam_state *__restrict__ st) {
  const AdamScalars s = *(const AdamScalars *)st->derived;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
    trl_adam_elem(p + i, g + i, m + i, v + i, s);
}

__global__ __launch_bounds__(NT) void k_polyak(float *__restrict__ t, const float *__restrict__ s, int64_t n,
                                               float tau, float omt) {
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT)
    t[i] = __fadd_rn(__fmul_rn(tau, s[i]), __fmul_rn(omt, t[i]));
}

inline unsigned ew_grid(int64_t n) {
  int64_t g = trl_cdiv(n, NT);
  return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int64_t trl_loss_workspace_bytes(void) { return (int64_t)WS_FLOATS * 4; }

extern "C" int trl_td_loss(const float *q, const float *q_next_target, const float *q_next_online,
                           const float *action, const float *reward, const float *done, int64_t B,
                           int32_t A, double gamma, int32_t kind, double huber_delta, float *loss_out,
                           float *dq, void *ws, void *stream) {
  const char *fn = "trl_td_loss";
  TRL_REQUIRE(q && q_next_target && action && reward && done && loss_out && ws, "%s: null pointer", fn);
  TRL_REQUIRE(B > 0 && A > 0 && B * A < (1ll << 40), "%s: B=%lld A=%d out of range", fn, (long long)B, A);
  TRL_REQUIRE(kind == TRL_LOSS_MSE || (kind == TRL_LOSS_HUBER && huber_delta > 0), "%s: bad loss kind", fn);
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = loss_grid(B);
  hipLaunchKernelGGL(k_td_loss, dim3(grid), dim3(NT), 0, st, q, q_next_target, q_next_online, action, reward,
                     done, B, (int)A, (float)gamma, (int)kind, (float)huber_delta, loss_out, dq, (float *)ws);
  TRL_CHECK_LAUNCH(fn);
  if (grid > 1) {
    hipLaunchKernelGGL(k_mean_finalize, dim3(1), dim3(NT), 0, st, (const float *)ws, (int)grid, B, loss_out);
    TRL_CHECK_LAUNCH(fn);
  }
  return TRL_OK;
}

extern "C" int trl_ddpg_critic_loss(const float *cur_q, const float *next_q, const float *reward,
                                    const float *done, int64_t B, double gamma, int32_t kind,
                                    double huber_delta, float *loss_out, float *dcur_q, void *ws,
                                    void *stream) {
  const char *fn = "trl_ddpg_critic_loss";
  TRL_REQUIRE(cur_q && next_q && reward && done && loss_out && ws, "%s: null pointer", fn);
  TRL_REQUIRE(B > 0 && B < (1ll << 40), "%s: B=%lld out of range", fn, (long long)B);
  TRL_REQUIRE(kind == TRL_LOSS_MSE || (kind == TRL_LOSS_HUBER && huber_delta > 0), "%s: bad loss kind", fn);
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = loss_grid(B);
  hipLaunchKernelGGL(k_ddpg_critic, dim3(grid), dim3(NT), 0, st, cur_q, next_q, reward, done, B, (float)gamma,
                     (int)kind, (float)huber_delta, loss_out, dcur_q, (float *)ws);
  TRL_CHECK_LAUNCH(fn);
  if (grid > 1) {
    hipLaunchKernelGGL(k_mean_finalize, dim3(1), dim3(NT), 0, st, (const float *)ws, (int)grid, B, loss_out);
    TRL_CHECK_LAUNCH(fn);
  }
  return TRL_OK;
}

namespace {
int ppo_loss_impl(const char *fn, const float *mean, const float *log_std, const float *value, const float *action,
                  const float *old_logp, const float *adv, const float *returns, int64_t B, int32_t A, double clip,
                  double alpha, double beta, float *losses_out, float *dmean, float *dlog_std, float *dvalue, void *ws,
                  void *stream) {
  TRL_REQUIRE(mean && log_std && value && action && old_logp && adv && returns && losses_out && ws,
              "%s: null pointer", fn);
  TRL_REQUIRE(B > 0 && A > 0 && A <= MAX_PPO_A && B * A < (1ll << 40), "%s: B=%lld A=%d out of range (A <= %d)",
              fn, (long long)B, A, MAX_PPO_A);
  const bool grads = dmean || dlog_std || dvalue;
  TRL_REQUIRE(!grads || (dmean && dlog_std && dvalue), "%s: pass all three gradient buffers or none", fn);
  PpoScalars p;
  p.lo = (float)(1.0 - clip);
  p.hi = (float)(1.0 + clip);
  p.alpha = (float)alpha;
  p.beta = (float)beta;
  p.dls_ent = (float)(-beta / (double)A);
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = loss_grid(B);
  hipLaunchKernelGGL(k_ppo_loss, dim3(grid), dim3(NT), 0, st, mean, log_std, value, action, old_logp, adv,
                     returns, B, (int)A, p, losses_out, dmean, dlog_std, dvalue, (float *)ws);
  TRL_CHECK_LAUNCH(fn);
  if (grid > 1) {
    hipLaunchKernelGGL(k_ppo_finalize, dim3(1), dim3(NT), 0, st, (const float *)ws, (int)grid, B, (int)A, p,
                       losses_out, dlog_std);
    TRL_CHECK_LAUNCH(fn);
  }
  return TRL_OK;
}

// ---- scratch of the no-allocation entry point: the per-block partial sums of ONE loss launch (any batch size: the
// grid is capped at MAX_BLOCKS), WS_FLOATS * 4 = 272 KiB per device, held by every GPU the library runs on until the
// process ends.  In host code the array's name is only a handle: the device address comes from
// hipGetSymbolAddress, once per device with that device current, and is what the kernels are given.
static __device__ __align__(256) float g_static_scratch[WS_FLOATS];
constexpr int MAX_DEVICES = 64;
float *g_static_scratch_ptr[MAX_DEVICES];      // per device: each GPU holds its own copy at its own address
std::mutex g_static_scratch_mu;

int static_scratch(const char *fn, float **out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
    trl_set_error("%s: hipGetDevice failed or device ordinal %d is beyond %d: %s", fn, dev, MAX_DEVICES,
                  hipGetErrorString(e));
    return TRL_EHIP;
  }
  std::lock_guard<std::mutex> lk(g_static_scratch_mu);
  if (!g_static_scratch_ptr[dev]) {
    void *p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch));
    if (e != hipSuccess || !p) {
      trl_set_error("%s: hipGetSymbolAddress(static scratch) failed on device %d: %s", fn, dev, hipGetErrorString(e));
      return TRL_EHIP;
    }
    g_static_scratch_ptr[dev] = (float *)p;
  }
  *out = g_static_scratch_ptr[dev];
  return TRL_OK;
}
}  // namespace

extern "C" int trl_ppo_loss(const float *mean, const float *log_std, const float *value,
                            const float *action, const float *old_logp, const float *adv,
                            const float *returns, int64_t B, int32_t A, double clip, double alpha,
                            double beta, float *losses_out, float *dmean, float *dlog_std, float *dvalue,
                            void *ws, void *stream) {
  return ppo_loss_impl("trl_ppo_loss", mean, log_std, value, action, old_logp, adv, returns, B, A, clip, alpha, beta,
                       losses_out, dmean, dlog_std, dvalue, ws, stream);
}

extern "C" int trl_ppo_loss_static(const float *mean, const float *log_std, const float *value, const float *action,
                                   const float *old_logp, const float *adv, const float *returns, int64_t B,
                                   int32_t A, double clip, double alpha, double beta, float *losses_out, float *dmean,
                                   float *dlog_std, float *dvalue, void *stream) {
  const char *fn = "trl_ppo_loss_static";
  // sizes the fixed scratch cannot serve are refused here, before anything is enqueued
  TRL_REQUIRE(A >= 1 && A <= MAX_PPO_A, "%s: A=%d is beyond the static scratch (1 <= A <= %d partial columns)", fn, A,
              MAX_PPO_A);
  TRL_REQUIRE(B >= 1 && B <= TRL_PPO_LOSS_STATIC_MAX_B, "%s: B=%lld is beyond the limit of this entry point (1 <= B <= %lld)",
              fn, (long long)B, (long long)TRL_PPO_LOSS_STATIC_MAX_B);
  float *ws = nullptr;
  int rc = static_scratch(fn, &ws);
  if (rc) return rc;
  // every partial the finalize block reads is written by this call's first launch: nothing has to start at zero
  return ppo_loss_impl(fn, mean, log_std, value, action, old_logp, adv, returns, B, A, clip, alpha, beta, losses_out,
                       dmean, dlog_std, dvalue, ws, stream);
}

extern "C" int trl_normal_logp(const float *mean, const float *log_std, const float *action, int64_t B,
                               int32_t A, float *logp, void *stream) {
  const char *fn = "trl_normal_logp";
  TRL_REQUIRE(mean && log_std && action && logp, "%s: null pointer", fn);
  TRL_REQUIRE(B >= 0 && A > 0 && B * A < (1ll << 40), "%s: B=%lld A=%d out of range", fn, (long long)B, A);
  if (B == 0) return TRL_OK;
  hipLaunchKernelGGL(k_normal_logp, dim3(ew_grid(B * A)), dim3(NT), 0, (hipStream_t)stream, mean, log_std,
                     action, B * A, (int)A, logp);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

extern "C" int trl_a2c_loss(const float *probs, const float *value, const float *action,
                            const float *returns, int64_t B, int32_t A, double alpha, double beta,
                            float *losses_out, float *dprobs, float *dvalue, void *ws, void *stream) {
  const char *fn = "trl_a2c_loss";
  TRL_REQUIRE(probs && value && action && returns && losses_out && ws, "%s: null pointer", fn);
  TRL_REQUIRE(B > 0 && A > 0 && B * A < (1ll << 40), "%s: B=%lld A=%d out of range", fn, (long long)B, A);
  TRL_REQUIRE((dprobs == nullptr) == (dvalue == nullptr), "%s: pass both gradient buffers or none", fn);
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = loss_grid(B);
  hipLaunchKernelGGL(k_a2c_loss, dim3(grid), dim3(NT), 0, st, probs, value, action, returns, B, (int)A,
                     (float)alpha, (float)beta, losses_out, dprobs, dvalue, (float *)ws);
  TRL_CHECK_LAUNCH(fn);
  if (grid > 1) {
    hipLaunchKernelGGL(k_a2c_finalize, dim3(1), dim3(NT), 0, st, (const float *)ws, (int)grid, B, (float)alpha,
                       (float)beta, losses_out);
    TRL_CHECK_LAUNCH(fn);
  }
  return TRL_OK;
}

extern "C" int trl_adam_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                             double lr, double beta1, double beta2, double eps, int64_t step,
                             double grad_scale, double clip_value, void *stream) {
  const char *fn = "trl_adam_step";
  TRL_REQUIRE(param && grad && exp_avg && exp_avg_sq, "%s: null pointer", fn);
  TRL_REQUIRE(n >= 0 && step >= 1, "%s: need n >= 0 and step >= 1", fn);
  if (n == 0) return TRL_OK;
  const AdamScalars s = trl_adam_scalars(lr, beta1, beta2, eps, step, grad_scale, clip_value);
  hipLaunchKernelGGL(k_adam, dim3(ew_grid(n)), dim3(NT), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, n, s);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

static_assert(sizeof(AdamScalars) == 32, "AdamScalars must fill trl_adam_state::derived");

int trl_adam_prep_launch(const char *fn, trl_adam_state *state, hipStream_t st) {
  TRL_REQUIRE(state && (uintptr_t)state % 8 == 0, "%s: the device Adam state must be a non-null 8 B-aligned pointer", fn);
  hipLaunchKernelGGL(k_adam_prep, dim3(1), dim3(64), 0, st, state);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

int trl_adam_dev_launch(const char *fn, float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                        const trl_adam_state *state, hipStream_t st) {
  TRL_REQUIRE(param && grad && exp_avg && exp_avg_sq && state && n >= 0, "%s: bad Adam arguments", fn);
  if (n == 0) return TRL_OK;
  hipLaunchKernelGGL(k_adam_dev, dim3(ew_grid(n)), dim3(NT), 0, st, param, grad, exp_avg, exp_avg_sq, n, state);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

extern "C" int trl_adam_step_dev(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                                 trl_adam_state *state, void *stream) {
  const char *fn = "trl_adam_step_dev";
  TRL_REQUIRE(param && grad && exp_avg && exp_avg_sq && state, "%s: null pointer", fn);
  TRL_REQUIRE(n >= 0, "%s: n < 0", fn);
  TRL_REQUIRE((uintptr_t)state % 8 == 0, "%s: state must be 8-byte aligned", fn);
  if (n == 0) return TRL_OK;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_adam_prep, dim3(1), dim3(64), 0, st, state);
  TRL_CHECK_LAUNCH(fn);
  hipLaunchKernelGGL(k_adam_dev, dim3(ew_grid(n)), dim3(NT), 0, st, param, grad, exp_avg, exp_avg_sq, n,
                     (const trl_adam_state *)state);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

extern "C" int trl_polyak(float *target, const float *source, int64_t n, double tau, void *stream) {
  const char *fn = "trl_polyak";
  TRL_REQUIRE(target && source, "%s: null pointer", fn);
  TRL_REQUIRE(n >= 0, "%s: n < 0", fn);
  if (n == 0) return TRL_OK;
  hipLaunchKernelGGL(k_polyak, dim3(ew_grid(n)), dim3(NT), 0, (hipStream_t)stream, target, source, n,
                     (float)tau, (float)(1.0 - tau));
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}
