// Discounted return / GAE as a reverse scan over [T, N] with one lane per env.
// Replaces the Python loops of a2c_controller.py:38-43 and ppo_controller.py:41-46
// (one loop per env buffer, examples/ppo.py:49-51) and the caller's
// `adv = returns - values` (examples/ppo.py:54).
//
// Arithmetic (every op rounded to f32 on its own; this file is built with
// -ffp-contract=off and uses the explicit _rn intrinsics so nothing fuses):
//   delta = (r[t] + gamma * v[t+1]) - v[t]
//   gae   = delta + gamma_lambda * gae
//   ret   = gae + v[t];   adv = ret - v[t]
// which is the reference's op order, so returns/advantages are bit-identical to it.
//
// Structure: a workgroup owns EPB consecutive envs and walks T backwards in tiles of TT
// rows.  The recurrence is serial in t, so the work is split by role:
//   - all lanes:   load the reward/value tile (coalesced along the env axis), form delta
//                  off the dependent chain and stage it in LDS;
//   - EPB lanes:   run the chain  gae = delta + c * gae  over the tile out of LDS
//                  (2 dependent VALU ops per step), writing gae back in place;
//   - all lanes:   ret/adv from gae and the staged value tile, coalesced stores.
// HBM traffic is the algorithmic minimum: r and v read once (plus one halo row of v per
// tile), ret and adv written once = 16 B per (t, env) step.
#include "trl_common.h"

namespace {

template <int EPB, bool MASKED>
__global__ __launch_bounds__(256) void k_gae_tiled(const float *__restrict__ reward,
                                                   const float *__restrict__ value,
                                                   const float *__restrict__ done,
                                                   const float *__restrict__ bootstrap, int T, int N,
                                                   float gamma, float gl, int out_nt,
                                                   float *__restrict__ returns,
                                                   float *__restrict__ adv) {
  constexpr int TT = 128;            // rows per tile
  constexpr int RPP = 256 / EPB;     // rows covered by one pass of the block
  __shared__ float s_v[TT + 1][EPB];  // value tile + halo row v[t0 + TT]
  __shared__ float s_d[TT][EPB];      // delta, overwritten in place by gae
  __shared__ float s_m[MASKED ? TT : 1][EPB];  // gamma_lambda * (1 - done)

  const int e = threadIdx.x % EPB;
  const int tl0 = threadIdx.x / EPB;
  const int env = blockIdx.x * EPB + e;
  const bool live = env < N;
  float gae = 0.0f;  // carried by the chain lanes (threadIdx.x < EPB) across tiles

  const int ntiles = (T + TT - 1) / TT;
  for (int tile = ntiles - 1; tile >= 0; --tile) {
    const int t0 = tile * TT;
    const int rows = (T - t0) < TT ? (T - t0) : TT;
    // ---- stage values (+ halo)
    for (int tl = tl0; tl <= rows; tl += RPP) {
      const int t = t0 + tl;
      float v = 0.0f;
      if (live) v = (t < T) ? value[(int64_t)t * N + env] : bootstrap[env];
      s_v[tl][e] = v;
    }
    __syncthreads();
    // ---- delta, off the chain
    for (int tl = tl0; tl < rows; tl += RPP) {
      const int64_t g = (int64_t)(t0 + tl) * N + env;
      float r = 0.0f, nd = 1.0f;
      if (live) {
        r = reward[g];
        if (MASKED) nd = __fsub_rn(1.0f, done[g]);
      }
      float gv = __fmul_rn(gamma, s_v[tl + 1][e]);
      if (MASKED) {
        gv = __fmul_rn(gv, nd);
        s_m[tl][e] = __fmul_rn(gl, nd);
      }
      s_d[tl][e] = __fsub_rn(__fadd_rn(r, gv), s_v[tl][e]);
    }
    __syncthreads();
    // ---- the serial chain: EPB lanes, newest row first
    if (threadIdx.x < EPB) {
      for (int tl = rows - 1; tl >= 0; --tl) {
        const float c = MASKED ? s_m[tl][e] : gl;
        gae = __fadd_rn(s_d[tl][e], __fmul_rn(c, gae));
        s_d[tl][e] = gae;
      }
    }
    __syncthreads();
    // ---- returns / advantages
    for (int tl = tl0; tl < rows; tl += RPP) {
      if (!live) continue;
      const int t = t0 + tl;
      const float v = s_v[tl][e];
      const float ret = __fadd_rn(s_d[tl][e], v);
      const int64_t o = out_nt ? ((int64_t)env * T + t) : ((int64_t)t * N + env);
      returns[o] = ret;
      if (adv) adv[o] = __fsub_rn(ret, v);
    }
    __syncthreads();
  }
}

template <int EPB>
int launch(bool masked, const float *reward, const float *value, const float *done,
           const float *bootstrap, int T, int N, float gamma, float gl, int out_nt, float *returns,
           float *adv, hipStream_t st) {
  const unsigned grid = (unsigned)trl_cdiv(N, EPB);
  if (masked)
    hipLaunchKernelGGL((k_gae_tiled<EPB, true>), dim3(grid), dim3(256), 0, st, reward, value, done,
                       bootstrap, T, N, gamma, gl, out_nt, returns, adv);
  else
    hipLaunchKernelGGL((k_gae_tiled<EPB, false>), dim3(grid), dim3(256), 0, st, reward, value, done,
                       bootstrap, T, N, gamma, gl, out_nt, returns, adv);
  TRL_CHECK_LAUNCH("trl_gae_reverse_scan");
  return TRL_OK;
}

}  // namespace

extern "C" int trl_gae_reverse_scan(const float *reward, const float *value, const float *done,
                                    const float *bootstrap, int64_t T, int64_t N, double gamma_d,
                                    double lmbda_d, int32_t mode, int32_t out_layout, float *returns,
                                    float *adv, void *stream) {
  const char *fn = "trl_gae_reverse_scan";
  const float gamma = (float)gamma_d;                     // self.gamma * values[step + 1]
  const float gamma_lambda = (float)(gamma_d * lmbda_d);  // self.gamma * self.lmbda * gae
  TRL_REQUIRE(reward && value && bootstrap && returns, "%s: null pointer", fn);
  TRL_REQUIRE(mode == TRL_GAE_REFERENCE || mode == TRL_GAE_MASKED, "%s: bad mode %d", fn, mode);
  TRL_REQUIRE(mode != TRL_GAE_MASKED || done != nullptr, "%s: masked mode needs done", fn);
  TRL_REQUIRE(out_layout == TRL_LAYOUT_TN || out_layout == TRL_LAYOUT_NT, "%s: bad out_layout %d", fn,
              out_layout);
  TRL_REQUIRE(T >= 0 && N >= 0 && T < (1ll << 31) && N < (1ll << 31) && T * N < (1ll << 40),
              "%s: T=%lld N=%lld out of range", fn, (long long)T, (long long)N);
  if (T == 0 || N == 0) return TRL_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool masked = mode == TRL_GAE_MASKED;
  // Envs per workgroup: as many as still leave >= 256 workgroups (one per CU), floor 16
  // (a 64 B row segment).
  if (N >= 256 * 64)
    return launch<64>(masked, reward, value, done, bootstrap, (int)T, (int)N, gamma, gamma_lambda,
                      out_layout, returns, adv, st);
  if (N >= 256 * 32)
    return launch<32>(masked, reward, value, done, bootstrap, (int)T, (int)N, gamma, gamma_lambda,
                      out_layout, returns, adv, st);
  return launch<16>(masked, reward, value, done, bootstrap, (int)T, (int)N, gamma, gamma_lambda,
                    out_layout, returns, adv, st);
}
