// One launch for a whole actor-critic learner step's network work: forward, loss and backward
// of BOTH small MLPs of ActorCriticNet (torchrl/contrib/models.py:112-147) with the PPO loss of
// ppo_controller.py:55-68 in between, for one minibatch.  A second small launch sums the
// per-workgroup gradient partials in a fixed order and writes the gradients and the losses.
//
// Why: with <= 64-wide layers a learner step is latency-bound, not HBM- or MFMA-bound.  The
// layer-by-layer kernels (trl_mlp.hip) hand every activation through LDS behind workgroup
// barriers and through HBM between forward and backward.  Here a wave owns a strip of 32
// samples from the observation to the parameter-gradient partials and never meets a barrier:
//
//   * "transposed" products: the MFMA computes H^T[n][row] = W[n][k] X^T[k][row], i.e. the
//     weights are the A operand and the samples sit on the lane axis (row = lane & 31).  The
//     accumulator layout of v_mfma_f32_32x32x2_f32 (lane: column = lane & 31; register r:
//     row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) then IS a valid B operand of the next layer's
//     product once its k index is walked in that same permuted order: register r of lane l
//     is feature feat(r, l >> 5), and the A operand simply reads W[.][feat(r, l >> 5)].  So
//     activations (forward) and activation gradients (backward) go from layer to layer in
//     registers; only the weights are read from LDS.
//   * weight gradients dW[n][k] = sum_row g[n][row] h[k][row] reduce over the lane axis, so g
//     and h are transposed through a wave-private LDS tile ([feature][row], stride 36: the
//     transposing writes and the 16 B operand reads are bank-conflict free) and accumulate in
//     registers over all strips of the wave.
//   * every MFMA operand that comes from LDS is fetched four k-steps at a time with one
//     ds_read_b128 (the k order of a product is free, so it is chosen to make a lane's four
//     consecutive steps contiguous), one group ahead of the MFMAs that consume it: with one
//     wave per SIMD nothing else hides the LDS latency.
//   * the last layer (fan-out <= 4: a value, an action mean per dimension) runs on the VALU:
//     a dot product per sample, and per-lane partials for its weight gradient.
//   * bias gradients and last-layer weight gradients are per-lane partial sums over the strips,
//     reduced across lanes once, at the end.
//
// Three kernels share this plan: k_ac_learn (four waves, any fan-in up to 64), k_ac_learn_pair (two hidden layers over a
// fan-in <= 4 at large batches: the two waves of a SIMD split the step by role) and k_ac_learn_wide (ONE hidden layer
// of up to 256 units over a fan-in <= 4: layer 0 computed once per accumulator layout, nothing through LDS tiles);
// k_ac_infer / k_ac_infer_wide are their forwards alone, bit for bit.
//
// Arithmetic is exact f32 (v_mfma_f32_32x32x2_f32 = k-ordered fma chain); the loss arithmetic
// is the one of k_ppo_loss (trl_loss.hip), op for op.  Sums over k and over the batch are
// taken in a different (fixed) order than torch's GEMMs, like any other GEMM kernel.
#include <stdlib.h>

#include "trl_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int LDT = 36;       // row stride of the [feature][row] transpose tiles (16 B rows: ds_read_b128 operands)
constexpr int LDW = 68;       // row stride of the 64 x 64 weight tiles
constexpr int MAX_WG = 256;   // partial sets per network
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;
constexpr float ENT_CONST = 1.41893853320467274178f;

// feature index held by accumulator register r of a lane in half `half` (= lane >> 5)
__device__ __forceinline__ int feat(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

struct NetArgs {
  int32_t depth;        // Linear layers: 2 or 3
  int32_t dims[4];      // dims[0] fan-in; dims[l+1] fan-out of layer l
  const float *w[3];
  const float *b[3];
  int32_t poff[3];      // offset of dW_l in a partial set; db_l follows dW_l
  int32_t nparams;      // dense parameter count = offset of the extras
};

// extras of a partial set (after the parameters): [0] sum of (v - ret)^2 (critic) or of
// min(surr1, surr2) (actor); [1] sum of entropies (actor); [2 + a] dlog_std[a] (actor)
constexpr int NEXTRA = 8;

struct LearnArgs {
  NetArgs net[2];       // 0 = critic, 1 = actor
  const float *obs, *action, *returns, *old_logp, *adv, *log_std;
  int64_t B;
  int32_t A;
  int32_t kind;         // KIND_PPO: Normal policy, clipped surrogate; KIND_A2C: softmax policy, adv * log-prob
  float lo, hi, alpha, beta;
  float *ws;            // [2][gridDim.x][pstride]
  int32_t pstride;
  int32_t tune_prio;    // (tuning build) wave priorities of k_ac_learn_pair: A in bits 1:0, B in bits 3:2
};
constexpr int KIND_PPO = 0, KIND_A2C = 1;
constexpr float F32_EPS = 1.1920928955078125e-07f;   // torch.finfo(torch.float32).eps
constexpr float F32_LOWEST = -3.4028234663852886e+38f;

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Diagnostic build only (-DTRL_LEARN_STAMP, tools/learn_stamps.py): shader-clock and 100 MHz
// real-time stamps per wave at four points, written to the unused upper half of the workspace.
#ifdef TRL_LEARN_STAMP
#define STAMP(i)                                                                                        \
  do {                                                                                                  \
    if (lane == 0) {                                                                                    \
      unsigned long long *sp = (unsigned long long *)(P.ws + 2ll * 128 * P.pstride) +                   \
                               (((int64_t)net * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + wave) * 8;                \
      sp[2 * (i)] = __builtin_amdgcn_s_memtime();                                                       \
      sp[2 * (i) + 1] = __builtin_amdgcn_s_memrealtime();                                               \
    }                                                                                                   \
  } while (0)
// in-loop phase clocks: PH(k) adds the shader cycles since the previous PH to phase k
#define PH_DECL unsigned long long ph_t = __builtin_amdgcn_s_memtime(), ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PH(k)                                                   \
  do {                                                          \
    const unsigned long long now = __builtin_amdgcn_s_memtime(); \
    ph_acc[k] += now - ph_t;                                    \
    ph_t = now;                                                 \
  } while (0)
#define PH_OUT                                                                                          \
  do {                                                                                                  \
    if (lane == 0) {                                                                                    \
      unsigned long long *sp = (unsigned long long *)(P.ws + 2ll * 128 * P.pstride) + 2 * 128 * 8 * 8 + \
                               (((int64_t)net * gridDim.x + blockIdx.x) * (blockDim.x >> 6) + wave) * 16;               \
      for (int k = 0; k < 12; ++k) sp[k] = ph_acc[k];                                                    \
    }                                                                                                   \
  } while (0)
// absolute shader-clock stamps of four consecutive strips of every pair of workgroup 0 (tools/learn_pair_trace.py)
#define TRACE(slot)                                                                                            \
  do {                                                                                                         \
    if (blockIdx.x == 0 && it >= 8 && it < 12 && lane == 0) {                                                  \
      unsigned long long *tp = (unsigned long long *)(P.ws + 2ll * 128 * P.pstride) + 2 * 128 * 8 * 8 * 3 +   \
                               ((((int64_t)net * 4 + pair) * 4 + (it - 8)) * 2 + (roleA ? 0 : 1)) * 8;        \
      tp[slot] = __builtin_amdgcn_s_memtime();                                                                 \
    }                                                                                                          \
  } while (0)
#else
#define STAMP(i)
#define PH_DECL
#define PH(k)
#define PH_OUT
#define TRACE(slot)
#endif

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave execute in issue order; this only stops the compiler from
  // moving them across the point where other lanes' data is expected
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// acc[nb][r] <- vec[nb * 32 + feat(r, half)] : 16 B LDS reads (4 consecutive features per read)
__device__ __forceinline__ void load_feat_vec(f32x16 (&acc)[2], const float *vec, int half) {
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t = *(const float4 *)(vec + nb * 32 + 8 * q + 4 * half);
      acc[nb][4 * q + 0] = t.x; acc[nb][4 * q + 1] = t.y; acc[nb][4 * q + 2] = t.z; acc[nb][4 * q + 3] = t.w;
    }
}

__device__ __forceinline__ void relu2(f32x16 (&h)[2]) {
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      // max(x, 0) as ONE instruction the compiler can see: a signed-integer max on the bit pattern (negative floats,
      // -0 included, are negative integers).  fmaxf adds a canonicalize; an inline-asm v_max_f32 is opaque to the
      // hazard recognizer, which then leaves out the wait states between an MFMA and this read of its result.
      const float x = h[nb][r];      // (a copy: bit-casting the vector-element lvalue itself reads element 0)
      const int b = __float_as_int(x);
      h[nb][r] = __int_as_float(b > 0 ? b : 0);
    }
}

// a wave-uniform float moved to a scalar register (the compiler keeps what the VALU computed in a vector register,
// live across the whole strip loop)
__device__ __forceinline__ float uni(float x) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}

__device__ __forceinline__ float4 ld4(const float *p) { return *(const float4 *)p; }
__device__ __forceinline__ float f4(const float4 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }
// keeps LDS reads issued above this point above it (the operand prefetch)
__device__ __forceinline__ void issue_fence() { asm volatile("" ::: "memory"); }

// tile[(nb * 32 + feat(r, half)) * LDT + row] <- v[nb][r]   (row = lane & 31)
__device__ __forceinline__ void put_tile(float *tile, const f32x16 (&v)[2], int half, int r32) {
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[(nb * 32 + feat(r, half)) * LDT + r32] = v[nb][r];
}

// sum of the 32 floats of one tile row (16 B aligned), four interleaved partial sums, fixed order
__device__ __forceinline__ float row_sum32(const float *row) {
  float4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = ld4(row + 4 * i);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { s0 += v[i].x; s1 += v[i].y; s2 += v[i].z; s3 += v[i].w; }
  return (s0 + s1) + (s2 + s3);
}

// ---- forward pieces, shared by the learner kernel and the inference kernel
// layer 0: h1^T = relu(W0 x^T + b0); x^T from the strip's LDS tile, k walked in natural order
template <int LDW0>
__device__ __forceinline__ void fwd_layer0(f32x16 (&h1)[2], const float *sX, const float *sW0, const float *sB,
                                           int ksteps0, int half, int r32) {
  load_feat_vec(h1, sB, half);
  for (int s = 0; s < ksteps0; ++s) {
    const int k = 2 * s + half;
    const float bv = sX[k * LDT + r32];
    const float a0 = sW0[r32 * LDW0 + k];
    const float a1 = sW0[(32 + r32) * LDW0 + k];
    h1[0] = MFMA(a0, bv, h1[0]);
    h1[1] = MFMA(a1, bv, h1[1]);
  }
  relu2(h1);
}

// layer 1: h2^T = relu(W1 h1^T + b1), the B operand chained in registers.  Group gq = (kb, q) is
// steps s = 4q .. 4q+3 of k-block kb, i.e. features kb*32 + 8q + 4half + i: one 16 B read per
// fan-out block, fetched one group ahead
__device__ __forceinline__ void fwd_layer1(f32x16 (&h2)[2], const f32x16 (&h1)[2], const float *sW1,
                                           const float *sB1, int half, int r32) {
  load_feat_vec(h2, sB1, half);
  const float *wl = sW1 + r32 * LDW + 4 * half;
  float4 a0 = ld4(wl), a1 = ld4(wl + 32 * LDW);
#pragma unroll
  for (int gq = 0; gq < 8; ++gq) {
    const int kb = gq >> 2, q = gq & 3;
    float4 n0 = a0, n1 = a1;
    if (gq + 1 < 8) {
      const int c = ((gq + 1) >> 2) * 32 + 8 * ((gq + 1) & 3);
      n0 = ld4(wl + c);
      n1 = ld4(wl + 32 * LDW + c);
    }
    issue_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      h2[0] = MFMA(f4(a0, i), h1[kb][4 * q + i], h2[0]);
      h2[1] = MFMA(f4(a1, i), h1[kb][4 * q + i], h2[1]);
    }
    a0 = n0; a1 = n1;
  }
  relu2(h2);
}

// last layer on the VALU: out[n] = WL[n] . hl + bL[n] (both lane halves end with the row's value).  Two features
// at a time (v_pk_fma_f32: a register pair of the activations times a pair of the 16 B weight read);
// p[n] = (sum over even registers, sum over odd registers)
template <int NO>
__device__ __forceinline__ void fwd_last_acc(f32x2 (&p)[NO], const f32x16 (&hl)[2], const float *sWL, int half) {
#pragma unroll
  for (int n = 0; n < NO; ++n) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 w4 = *(const float4 *)(sWL + n * 64 + kb * 32 + 8 * q + 4 * half);
        p[n] = __builtin_elementwise_fma(f32x2{w4.x, w4.y}, f32x2{hl[kb][4 * q + 0], hl[kb][4 * q + 1]}, p[n]);
        p[n] = __builtin_elementwise_fma(f32x2{w4.z, w4.w}, f32x2{hl[kb][4 * q + 2], hl[kb][4 * q + 3]}, p[n]);
      }
  }
}

template <int NO>
__device__ __forceinline__ void fwd_last_finish(float (&out)[NO], const f32x2 (&p)[NO], const float *sBL) {
#pragma unroll
  for (int n = 0; n < NO; ++n) {
    const float s = p[n].x + p[n].y;
    const float t = s + __shfl_xor(s, 32, 64);
    out[n] = t + sBL[n];
  }
}

template <int NO>
__device__ __forceinline__ void fwd_last(float (&out)[NO], const f32x16 (&hl)[2], const float *sWL,
                                         const float *sBL, int half) {
  f32x2 p[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) p[n] = f32x2{0.f, 0.f};
  fwd_last_acc<NO>(p, hl, sWL, half);
  fwd_last_finish<NO>(out, p, sBL);
}

// layer 1 backward, input side: t^T[k][row] = sum_n W1[n][k] g^T[n][row]; A = W1^T[k = kb*32 + r32][n], four n per
// read, one group ahead
__device__ __forceinline__ void bwd_dx1(f32x16 (&t)[2], const f32x16 (&g)[2], const float *sW1T, int half, int r32) {
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) t[kb][r] = 0.f;
  const float *wt = sW1T + r32 * LDW + 4 * half;
  float4 a0 = ld4(wt), a1 = ld4(wt + 32 * LDW);
#pragma unroll
  for (int gq = 0; gq < 8; ++gq) {
    const int nb = gq >> 2, q = gq & 3;
    float4 n0 = a0, n1 = a1;
    if (gq + 1 < 8) {
      const int c = ((gq + 1) >> 2) * 32 + 8 * ((gq + 1) & 3);
      n0 = ld4(wt + c);
      n1 = ld4(wt + 32 * LDW + c);
    }
    issue_fence();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      t[0] = MFMA(f4(a0, i), g[nb][4 * q + i], t[0]);
      t[1] = MFMA(f4(a1, i), g[nb][4 * q + i], t[1]);
    }
    a0 = n0; a1 = n1;
  }
}

// weights of one network, zero-padded, into LDS, once per workgroup.  Loads first, stores after:
// every load of a tile is in flight at once (a load -> store loop pays the L2 latency per iteration)
template <int D, int LDW0, int NO, bool WITH_T, int NT = 256>
__device__ __forceinline__ void stage_weights(const NetArgs &a, float *sW0, float *sW1, float *sW1T, float *sWL,
                                              float *sB, int tid) {
  const int K0 = a.dims[0], H1 = a.dims[1], HL = a.dims[D - 1], NOr = a.dims[D];
  {
    constexpr int N0 = (64 * LDW0 + NT - 1) / NT;
    float v[N0];
#pragma unroll
    for (int i = 0; i < N0; ++i) {
      const int e = tid + NT * i, n = e / LDW0, k = e % LDW0;
      v[i] = (n < H1 && k < K0 && e < 64 * LDW0) ? a.w[0][n * K0 + k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < N0; ++i)
      if (tid + NT * i < 64 * LDW0) sW0[tid + NT * i] = v[i];
  }
  if (D == 3) {
    const int H2 = a.dims[2];
    constexpr int N1 = 64 * 64 / NT;
    float v[N1];
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      const int e = tid + NT * i, n = e >> 6, k = e & 63;
      v[i] = (n < H2 && k < H1) ? a.w[1][n * H1 + k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      const int e = tid + NT * i, n = e >> 6, k = e & 63;
      sW1[n * LDW + k] = v[i];
      if (WITH_T) sW1T[k * LDW + n] = v[i];
    }
  }
  // last layer: [n][64]
  for (int e = tid; e < NO * 64; e += NT) {
    const int n = e >> 6, k = e & 63;
    sWL[e] = (n < NOr && k < HL) ? a.w[D - 1][n * HL + k] : 0.f;
  }
  // biases: b0 (64), b1 (64), bL (4)
  for (int e = tid; e < 64 + 64 + 4; e += NT) {
    float v = 0.f;
    if (e < 64) v = e < H1 ? a.b[0][e] : 0.f;
    else if (e < 128) v = (D == 3 && e - 64 < a.dims[2]) ? a.b[1][e - 64] : 0.f;
    else v = e - 128 < NOr ? a.b[D - 1][e - 128] : 0.f;
    sB[e] = v;
  }
}

// ---- loss of one strip row and its gradient w.r.t. the network output (arithmetic of k_ppo_loss / k_a2c_loss,
// trl_loss.hip).  `valid`: the row exists; each row's loss terms are summed once, by lane half 0.
struct LossConst {
  float invB, gmin;
};
template <int NO>
__device__ __forceinline__ void loss_grad(const LearnArgs &P, int net, int A, bool valid, int half, const LossConst &lc,
                                          const float (&out)[NO], float ret, float ad, const float (&act)[NO],
                                          const float (&olp)[NO], const float (&den)[NO], const float (&var)[NO],
                                          const float (&logsig)[NO], float (&gout)[NO], float &s_main, float &s_ent,
                                          float (&s_dls)[NO]) {
  const float invB = lc.invB, gmin = lc.gmin;
  const bool count = valid && half == 0;    // each row's loss terms are summed once
#pragma unroll
  for (int n = 0; n < NO; ++n) gout[n] = 0.f;
  if (net == 0) {
    const float dv = out[0] - ret;            // mse_loss(values, returns)
    if (count) s_main += dv * dv;
    if (valid) gout[0] = P.alpha * ((2.0f * invB) * dv);
  } else if (P.kind == KIND_A2C) {
    // out = the actor's logits; nn.Softmax(dim=1) (models.py:99-103), then Categorical(probs) and
    // the loss of a2c_controller.py:48-59 in the arithmetic of k_a2c_loss; `ad` is the detached
    // advantage returns - V(obs) (the caller's forward-only pass)
    float zmax = out[0];
#pragma unroll
    for (int n = 1; n < NO; ++n)
      if (n < A) zmax = fmaxf(zmax, out[n]);
    float pr[NO], esum = 0.f;
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      pr[n] = n < A ? expf(out[n] - zmax) : 0.f;
      esum += pr[n];
    }
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      pr[n] = pr[n] / esum;
      s += pr[n];                             // Categorical: probs / probs.sum(-1, keepdim=True)
    }
    int ai = (int)act[0];                     // action.long()
    ai = ai < 0 ? 0 : (ai >= A ? A - 1 : ai);
    float ent = 0.f, gdot = 0.f, logp = 0.f, gg[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      gg[n] = 0.f;
      if (n < A) {
        const float pn = pr[n] / s;
        const float cl = fminf(fmaxf(pn, F32_EPS), 1.0f - F32_EPS);   // clamp_probs
        const float logit = logf(cl);
        ent += fmaxf(logit, F32_LOWEST) * pn;
        if (n == ai) logp = logit;
        const float dlogit = (pn >= F32_EPS && pn <= 1.0f - F32_EPS) ? 1.0f / cl : 0.f;
        float g1 = (P.beta * invB) * (logit + pn * dlogit);          // from -beta * entropy
        if (n == ai) g1 += (-(ad * invB)) * dlogit;                   // from the actor term
        gg[n] = g1;
        gdot += g1 * pn;
      }
    }
    float dp[NO], dsum = 0.f;
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      dp[n] = n < A ? (gg[n] - gdot) / s : 0.f;                       // back through the normalisation
      dsum += dp[n] * pr[n];
    }
#pragma unroll
    for (int n = 0; n < NO; ++n)
      if (valid && n < A) gout[n] = (dp[n] - dsum) * pr[n];            // softmax backward
    if (count) {
      s_main += ad * logp;
      s_ent += -ent;
    }
  } else {
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      if (n < A) {
        const float d = act[n] - out[n];
        const float sq = d * d;
        const float lp = ((-sq) / den[n] - logsig[n]) - LOG_SQRT_2PI;   // Normal.log_prob
        const float ratio = expf(lp - olp[n]);
        const float s1 = ratio * ad;
        const float rc = fminf(fmaxf(ratio, P.lo), P.hi);
        const float s2 = rc * ad;
        // torch.minimum backward: the smaller side takes the gradient, a tie splits it
        const float g1 = s1 < s2 ? gmin : (s1 == s2 ? 0.5f * gmin : 0.f);
        const float g2 = s2 < s1 ? gmin : (s1 == s2 ? 0.5f * gmin : 0.f);
        const bool inr = ratio >= P.lo && ratio <= P.hi;   // clamp backward mask
        const float g_ratio = g1 * ad + (inr ? g2 * ad : 0.f);
        const float g_lp = g_ratio * ratio;                // exp backward
        if (count) {
          s_main += fminf(s1, s2);
          s_ent += ENT_CONST + logsig[n];
          s_dls[n] += g_lp * (sq / var[n] - 1.0f);
        }
        if (valid) gout[n] = (g_lp / den[n]) * (2.0f * d);
      }
    }
  }
}

// D = depth (2: in -> H -> out, 3: in -> H -> H -> out); KB0 = 32-blocks of the fan-in;
// NO = padded fan-out of the last layer (actual fan-outs <= NO)
// XP: the strip's observations are prefetched into registers (fan-in <= 12)
// (hidden layers of at most 64 units; one wider hidden layer over a small fan-in is k_ac_learn_wide's)
template <int D, int KB0, int NO, bool XP>
__global__ __launch_bounds__(256) void k_ac_learn(LearnArgs P) {
  constexpr int NW = 4, NT = 64 * NW;                // one wave per SIMD, with the whole register file
  constexpr int LDW0 = 32 * KB0 + 1;
  constexpr int XT = 32 * KB0 * LDT + 4;             // floats of the x tile (+ a dummy slot for idle prefetch lanes)
  constexpr int WT = XT + 64 * LDT + (D == 3 ? 64 * LDT : 0);   // per-wave scratch
  constexpr int WTP = (WT + 3) & ~3;
  // One LDS array, carved by hand with the weights first: everything every strip reads (biases, last layer, W0, W1,
  // W1^T) then sits below 64 KB, within reach of a ds instruction's 16-bit offset from a lane's base register.
  // As separate arrays the compiler placed them behind the 120+ KB of tiles and kept one address register per read.
  constexpr int OFF_B = 0;                                          // b0, b1 (padded with zeros), bL
  constexpr int OFF_WL = OFF_B + ((64 + 64 + 4 + 3) & ~3);
  constexpr int OFF_W0 = OFF_WL + NO * 64;
  constexpr int OFF_W1 = OFF_W0 + ((64 * LDW0 + 3) & ~3);           // W1[n][k]
  constexpr int OFF_W1T = OFF_W1 + (D == 3 ? 64 * LDW : 0);         // W1^T: [k][n], the backward's A operand
  constexpr int OFF_T = OFF_W1T + (D == 3 ? 64 * LDW : 0);
  __shared__ __align__(16) float smem[OFF_T + NW * WTP];
  float *const sB = smem + OFF_B, *const sWL = smem + OFF_WL, *const sW0 = smem + OFF_W0, *const sW1 = smem + OFF_W1,
               *const sW1T = smem + OFF_W1T, *const sT = smem + OFF_T;
  static_assert(WTP >= 32 * 32 * KB0 && (D == 2 || WTP >= 32 * 64) && NW * WTP >= NW * 12 * 64 + NW * 16,
                "the end-of-kernel reduction re-uses the per-wave scratch");

  const int net = blockIdx.y;
  const NetArgs &a = P.net[net];
  // (the wave index as a scalar: the strip number, the row base and every input address derived from it then stay
  // in scalar registers, and only the lane's own 32-bit offsets take vector registers)
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5,
            r32 = lane & 31;
  const int K0 = a.dims[0], H1 = a.dims[1];
  const int HL = a.dims[D - 1];      // fan-in of the last layer
  const int NOr = a.dims[D];         // its actual fan-out
  const int64_t B = P.B;
  const int A = P.A;

  STAMP(0);
  stage_weights<D, LDW0, NO, true, NT>(a, sW0, sW1, sW1T, sWL, sB, tid);
  float *sX = sT + wave * WTP;      // [32 KB0][LDT]  x^T of the strip
  float *sG = sX + XT;              // [64][LDT]      gradient w.r.t. a layer's pre-activation
  float *sH = sG + 64 * LDT;        // [64][LDT]      (D == 3) h1^T
  for (int e = lane; e < XT; e += 64) sX[e] = 0.f;   // rows >= K0 stay zero for good
  __syncthreads();
  STAMP(1);

  // ---- per-wave persistent state
  f32x16 accW0[2][KB0];
  f32x16 accW1[2][2];
  float vdb0 = 0.f, vdb1 = 0.f;      // bias gradients of feature `lane`: row sums of the g tiles
  // fan-in <= 4: dW0 is 32 one-step products of v_mfma_f32_4x4x1 per strip (16 blocks of 4 features x 4
  // inputs: lane = feature on the A side, lane % 4 = input on the B side; 8 cycles each) instead of 32 padded
  // 32x32x2 MFMAs into accW0; element i of dw0m is dW0[4 (lane / 4) + i][lane % 4].  With a fan-in < 4 row 3 of the
  // x tile holds ones (W0's column 3 is zero, so the forward does not see them) and column 3 is the bias gradient.
  const bool w0v = KB0 == 1 && XP && a.dims[0] <= 4;
  const bool ones3 = w0v && K0 < 4;
  if (ones3 && lane < 32) sX[3 * LDT + lane] = 1.f;
  f32x4 dw0m = {0.f, 0.f, 0.f, 0.f};
  f32x16 dWL[NO][2];
  float dbL[NO];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
    for (int kb = 0; kb < KB0; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) accW0[nb][kb][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) accW1[nb][kb][r] = 0.f;
  }
#pragma unroll
  for (int n = 0; n < NO; ++n) {
    dbL[n] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) dWL[n][kb][r] = 0.f;
  }
  float s_main = 0.f, s_ent = 0.f;
  float s_dls[NO];
  float den[NO], var[NO], logsig[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) {
    s_dls[n] = 0.f;
    den[n] = var[n] = 1.f; logsig[n] = 0.f;
    if (net == 1 && P.kind == KIND_PPO && n < A) {
      const float sigma = expf(P.log_std[n]);   // self.log_std.exp()
      var[n] = uni(sigma * sigma);              // scale ** 2
      logsig[n] = uni(logf(sigma));             // scale.log()
      den[n] = uni(2.0f * (sigma * sigma));
    }
  }
  const float invB = uni(1.0f / (float)B);
  const float gmin = uni(-(1.0f / (float)(B * A)));   // d loss / d min(surr1, surr2) element
  const int q64 = 64 / K0, m64 = 64 % K0;
  const int ksteps0 = (K0 + 1) >> 1;

  // ---- inputs of a strip are fetched one strip ahead, into registers (nothing else would hide
  // the HBM latency): XPF observation elements per lane (XP: fan-in <= 12; wider observations are
  // read in place) and the per-sample loss inputs (both lane halves hold their row's values).
  // The fetch is branch-free -- clamped addresses, idle lanes write a dummy LDS slot -- so that
  // the loads of strip i+1 stay in flight across strip i (a load under a branch makes the
  // compiler drain the vector-memory counter at the next join).
  constexpr int XPF = 6;
  const int nelem = 32 * K0;
  int xoff[XPF];
  {
    int rr = lane / K0, kk = lane % K0;
#pragma unroll
    for (int j = 0; j < XPF; ++j) {
      xoff[j] = lane + 64 * j < nelem ? kk * LDT + rr : XT - 4;
      rr += q64; kk += m64;
      if (kk >= K0) { kk -= K0; ++rr; }
    }
  }
  // slot 0: returns (critic) / advantages (actor); action and old log-prob per dimension (the
  // critic and the padded dimensions re-read slot 0's array: always a valid address)
  const float *in0 = net == 0 ? P.returns : P.adv;
  const float *in_act[NO], *in_olp[NO];
  const bool ppo_actor = net == 1 && P.kind == KIND_PPO;
  const int astr = ppo_actor ? A : 1;
#pragma unroll
  for (int n = 0; n < NO; ++n) {      // A2C: the action index of the row, in every slot
    in_act[n] = net == 0 ? in0 : (ppo_actor ? P.action + (n < A ? n : 0) : P.action);
    in_olp[n] = ppo_actor ? P.old_logp + (n < A ? n : 0) : in0;
  }
  float xr[XPF], p_in0 = 0.f, p_act[NO], p_olp[NO];
#pragma unroll
  for (int j = 0; j < XPF; ++j) xr[j] = 0.f;
  const int64_t nstrips = (B + 31) >> 5;
  auto fetch = [&](int64_t strip) {      // (strip is wave-uniform: scalar bases, 32-bit lane offsets)
    strip = strip < nstrips ? strip : nstrips - 1;
    const int64_t row0 = strip << 5;
    const int64_t left = B - row0;                    // rows from row0 to the end (>= 1)
    const int rows = (int)(left < 32 ? left : 32);
    const int rl = r32 < rows ? r32 : rows - 1;       // the lane's row within the strip, clamped to the last valid one
    if (XP) {
      const float *base = P.obs + row0 * K0;
      const int last = rows * K0 - 1;                 // last element of the strip's rows
#pragma unroll
      for (int j = 0; j < XPF; ++j) {
        const int e = lane + 64 * j;
        xr[j] = base[e < last ? e : last];
      }
    }
    p_in0 = (in0 + row0)[rl];
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      p_act[n] = (in_act[n] + row0 * astr)[rl * astr];
      p_olp[n] = (in_olp[n] + row0 * astr)[rl * astr];
    }
  };

  const int64_t sstride = (int64_t)gridDim.x * NW;
  int64_t strip = (int64_t)blockIdx.x * NW + wave;
  fetch(strip);
  PH_DECL
  for (; strip < nstrips; strip += sstride) {
    const int64_t row0 = strip << 5;
    const bool valid = row0 + r32 < B;
    PH(5);

    // ---- x^T of the strip -> sX[k][row] (the strip's rows are contiguous in obs)
    if (XP) {
#pragma unroll
      for (int j = 0; j < XPF; ++j) sX[xoff[j]] = xr[j];
    } else {
      const float *base = P.obs + row0 * K0;
      const int64_t avail = (B - row0) * K0;
      int rr = lane / K0, kk = lane % K0;
      for (int e = lane; e < nelem; e += 64) {
        sX[kk * LDT + rr] = e < avail ? base[e] : 0.f;
        rr += q64; kk += m64;
        if (kk >= K0) { kk -= K0; ++rr; }
      }
    }
    const float ret = p_in0, ad = p_in0;
    float act[NO], olp[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) { act[n] = p_act[n]; olp[n] = p_olp[n]; }
    fetch(strip + sstride);
    wave_lds_fence();
    PH(6);

    // ---- forward: h1, (h2), out
    f32x16 h1[2], h2[2];
    float out[NO];
    fwd_layer0<LDW0>(h1, sX, sW0, sB, ksteps0, half, r32);
    PH(7);
    if (D == 3) fwd_layer1(h2, h1, sW1, sB + 64, half, r32);
    PH(8);
    const f32x16 (&hl)[2] = D == 3 ? h2 : h1;
    fwd_last<NO>(out, hl, sWL, sB + 128, half);

    PH(0);
    // ---- loss and its gradient w.r.t. out (arithmetic of k_ppo_loss)
    float gout[NO];
    loss_grad<NO>(P, net, A, valid, half, LossConst{invB, gmin}, out, ret, ad, act, olp, den, var, logsig, gout, s_main,
                  s_ent, s_dls);

#pragma unroll
    for (int n = 0; n < NO; ++n)
      if (half == 0) dbL[n] += gout[n];
    // ---- last layer backward (VALU): dWL, dbL partials; g = (WL^T gout) * (hl > 0)
    f32x16 g[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NO; ++n) {
          const float4 w4 = *(const float4 *)(sWL + n * 64 + kb * 32 + 8 * q + 4 * half);
          t[0] = fmaf(w4.x, gout[n], t[0]);
          t[1] = fmaf(w4.y, gout[n], t[1]);
          t[2] = fmaf(w4.z, gout[n], t[2]);
          t[3] = fmaf(w4.w, gout[n], t[3]);
#pragma unroll
          for (int i = 0; i < 4; ++i) dWL[n][kb][4 * q + i] = fmaf(gout[n], hl[kb][4 * q + i], dWL[n][kb][4 * q + i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) g[kb][4 * q + i] = hl[kb][4 * q + i] > 0.f ? t[i] : 0.f;
      }

    PH(1);
    // ---- layer 1 backward (D == 3): dW1 += g h1^T (rows through LDS), g <- (W1^T g) * (h1 > 0)
    if (D == 3) {
      put_tile(sG, g, half, r32);
      // rows in the order (j, half, i) -> row 8j + 4half + i: a lane's four steps are one 16 B read
      const float *ga = sG + r32 * LDT + 4 * half, *ha = sH + r32 * LDT + 4 * half;
      {
        put_tile(sH, h1, half, r32);
        wave_lds_fence();
        {
          float4 g0 = ld4(ga), g1 = ld4(ga + 32 * LDT), x0 = ld4(ha), x1 = ld4(ha + 32 * LDT);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float4 ng0 = g0, ng1 = g1, nx0 = x0, nx1 = x1;
            if (j + 1 < 4) {
              ng0 = ld4(ga + 8 * (j + 1)); ng1 = ld4(ga + 32 * LDT + 8 * (j + 1));
              nx0 = ld4(ha + 8 * (j + 1)); nx1 = ld4(ha + 32 * LDT + 8 * (j + 1));
            }
            issue_fence();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              accW1[0][0] = MFMA(f4(g0, i), f4(x0, i), accW1[0][0]);
              accW1[0][1] = MFMA(f4(g0, i), f4(x1, i), accW1[0][1]);
              accW1[1][0] = MFMA(f4(g1, i), f4(x0, i), accW1[1][0]);
              accW1[1][1] = MFMA(f4(g1, i), f4(x1, i), accW1[1][1]);
            }
            g0 = ng0; g1 = ng1; x0 = nx0; x1 = nx1;
          }
        }
        vdb1 += row_sum32(sG + lane * LDT);     // db1[lane] += sum over the strip's rows of g[lane][row] (beside the MFMAs)
        f32x16 t[2];
        bwd_dx1(t, g, sW1T, half, r32);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) g[kb][r] = h1[kb][r] > 0.f ? t[kb][r] : 0.f;
      }
    }

    PH(2);
    // ---- layer 0 backward: dW0 += g x^T
    put_tile(sG, g, half, r32);
    wave_lds_fence();
    PH(9);
    if (w0v) {
      const float *grow = sG + lane * LDT, *xrow = sX + (lane & 3) * LDT;
      float4 gv[8], xv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { gv[i] = ld4(grow + 4 * i); xv[i] = ld4(xrow + 4 * i); }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          dw0m = __builtin_amdgcn_mfma_f32_4x4x1f32(f4(gv[i], c), f4(xv[i], c), dw0m, 0, 0, 0);
      if (!ones3) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s0 += gv[i].x; s1 += gv[i].y; s2 += gv[i].z; s3 += gv[i].w; }
        vdb0 += (s0 + s1) + (s2 + s3);
      }
    } else {
      const float *ga = sG + r32 * LDT + 4 * half, *xa = sX + r32 * LDT + 4 * half;
      float4 g0 = ld4(ga), g1 = ld4(ga + 32 * LDT), xv[KB0];
#pragma unroll
      for (int kb = 0; kb < KB0; ++kb) xv[kb] = ld4(xa + kb * 32 * LDT);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 ng0 = g0, ng1 = g1, nx[KB0];
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb) nx[kb] = xv[kb];
        if (j + 1 < 4) {
          ng0 = ld4(ga + 8 * (j + 1)); ng1 = ld4(ga + 32 * LDT + 8 * (j + 1));
#pragma unroll
          for (int kb = 0; kb < KB0; ++kb) nx[kb] = ld4(xa + kb * 32 * LDT + 8 * (j + 1));
        }
        issue_fence();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int kb = 0; kb < KB0; ++kb) {
            accW0[0][kb] = MFMA(f4(g0, i), f4(xv[kb], i), accW0[0][kb]);
            accW0[1][kb] = MFMA(f4(g1, i), f4(xv[kb], i), accW0[1][kb]);
          }
        g0 = ng0; g1 = ng1;
#pragma unroll
        for (int kb = 0; kb < KB0; ++kb) xv[kb] = nx[kb];
      }
      vdb0 += row_sum32(sG + lane * LDT);     // beside the dW0 MFMAs
    }
    wave_lds_fence();     // the next strip overwrites sG and sX
    PH(3);
  }
  PH_OUT;

  STAMP(2);
  // =========================================================================== partials out
  // Everything below runs once per workgroup and re-uses the per-wave scratch regions of sT.
  float *part = P.ws + ((int64_t)net * gridDim.x + blockIdx.x) * P.pstride;

  // per-lane partial vectors dWL[n] -> per-feature sums over the wave's 32 rows: transposed
  // through the wave's own tile (the bias gradients vdb0 / vdb1 are per feature already)
  const float vec_db1 = vdb1;
  float vec_wl[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) {
    put_tile(sG, dWL[n], half, r32);
    wave_lds_fence();
    float s = 0.f;
    for (int r = 0; r < 32; ++r) s += sG[lane * LDT + r];
    vec_wl[n] = s;
    wave_lds_fence();
  }
  // scalars: wave sums (fixed butterfly order)
  float sc[2 + 2 * NO];
  sc[0] = trl_wave_sum(s_main);
  sc[1] = trl_wave_sum(s_ent);
#pragma unroll
  for (int n = 0; n < NO; ++n) {
    sc[2 + n] = trl_wave_sum(s_dls[n]);
    sc[2 + NO + n] = trl_wave_sum(dbL[n]);
  }

  // ---- weight tiles: every wave puts its accumulators into its own scratch region, then the
  // workgroup adds the four copies in wave order
  float *mine = sT + wave * WTP;
  auto flush_tile = [&](auto &acc, int KBn, int N, int K, int off) {
    const int ld = 32 * KBn;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {      // one 32-row half of the tile per round (a half fits every wave's region)
      __syncthreads();     // the regions are free (previous round / the per-wave transposes)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        if (kb < KBn) {
#pragma unroll
          for (int r = 0; r < 16; ++r) mine[feat(r, half) * ld + kb * 32 + r32] = acc(nb, kb)[r];
        }
      }
      __syncthreads();
      const int rows = N - 32 * nb < 32 ? N - 32 * nb : 32;
      for (int e = tid; e < rows * K; e += NT) {
        const int i = (e / K) * ld + (e % K);
        float v = sT[i];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += sT[w * WTP + i];
        part[off + 32 * nb * K + e] = v;
      }
    }
  };
  {
    auto acc0 = [&](int nb, int kb) -> f32x16 & { return accW0[nb][kb < KB0 ? kb : 0]; };
    if (!w0v) flush_tile(acc0, KB0, H1, K0, a.poff[0]);     // (workgroup-uniform)
  }
  if (D == 3) {
    auto acc1 = [&](int nb, int kb) -> f32x16 & { return accW1[nb][kb]; };
    flush_tile(acc1, 2, a.dims[2], H1, a.poff[1]);
  }
  __syncthreads();
  // ---- per-feature vectors [wave][slot][64] and scalars [wave][16], summed in wave order
  constexpr int VS = 12;     // slots per wave: db0, db1, NO last-layer rows, 4 dW0 columns (w0v)
  static_assert(2 + NO + 4 <= VS, "vector slots");
  float *vsum = sT, *scal = sT + NW * VS * 64;
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 2 + 2 * NO; ++j) scal[wave * 16 + j] = sc[j];
  }
  if (!ones3) {
    vsum[(wave * VS + 0) * 64 + lane] = vdb0;
  } else if ((lane & 3) == 3) {              // column 3 of the 4x4 products: the ones row
#pragma unroll
    for (int i = 0; i < 4; ++i) vsum[(wave * VS + 0) * 64 + 4 * (lane >> 2) + i] = dw0m[i];
  }
  vsum[(wave * VS + 1) * 64 + lane] = vec_db1;
#pragma unroll
  for (int n = 0; n < NO; ++n) vsum[(wave * VS + 2 + n) * 64 + lane] = vec_wl[n];
#pragma unroll
  for (int i = 0; i < 4; ++i) vsum[(wave * VS + 2 + NO + (lane & 3)) * 64 + 4 * (lane >> 2) + i] = dw0m[i];
  __syncthreads();
  for (int sl = tid >> 6; sl < 2 + NO + (w0v ? 4 : 0); sl += NW) {
    const int f = tid & 63;
    float v = vsum[(0 * VS + sl) * 64 + f];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += vsum[(w * VS + sl) * 64 + f];
    if (sl >= 2 + NO) {
      const int k = sl - 2 - NO;
      if (f < H1 && k < K0) part[a.poff[0] + f * K0 + k] = v;
    } else if (sl == 0) {
      if (f < H1) part[a.poff[0] + H1 * K0 + f] = v;
    } else if (sl == 1) {
      if (D == 3 && f < a.dims[2]) part[a.poff[1] + a.dims[2] * H1 + f] = v;
    } else {
      const int n = sl - 2;
      if (f < HL && n < NOr) part[a.poff[D - 1] + n * HL + f] = v;
    }
  }
  if (tid < 2 + 2 * NO) {
    float v = scal[tid];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += scal[w * 16 + tid];
    if (tid < 2 + NO) {                       // s_main, s_ent, dls[n]
      if (tid < 2 + A) part[a.nparams + tid] = v;
    } else {                                  // dbL[n]
      const int n = tid - 2 - NO;
      if (n < NOr) part[a.poff[D - 1] + NOr * HL + n] = v;
    }
  }
  STAMP(3);
}

// ---------------------------------------------------------------------------------------------------------------
// One WIDE hidden layer (65 .. 256 units: the reference's own ActorCriticNet / A2CNet width, ppo_controller.py:14)
// over a fan-in of at most 9.  That shape has no square product: per 64-feature block only layer 0 (K <= 9), the
// last layer's dot product and their transposes, so the step is VALU- and LDS-bound, not MFMA-bound, and what counts
// is instructions per strip.  k_ac_learn's plan (activations held from forward to backward, every gradient tile
// transposed through LDS) needs 128 + 128 registers for 4 blocks and ran out of them.  Here nothing is held and
// nothing is transposed: layer 0 is computed TWICE, in the two accumulator layouts the MFMA offers for free --
//   forward:  H^T = W0' X'^T  (A = weights, B = samples): lane = sample, register = feature.  The last layer's dot
//             product is then a per-lane FMA chain, as in k_ac_learn; the activations are dropped.
//   backward: H   = X' W0'^T  (the SAME two operand registers, swapped): lane = feature, register = sample.  The
//             ReLU mask, g = (WL^T gout) * mask, dWL[f] += sum_row h g_out, and the 4x4x1 MFMA outer products
//             dW0[f][k] += g[f][row] x[k][row] all reduce over REGISTERS of the lane that owns feature f: one
//             accumulator per feature, no LDS tile.  The recompute is 2-5 MFMAs per 32 features.
// W0' = [W0 | b0] and X' = [x | 1] (the bias is one more k column, so the products start from C = 0, and column K0 of
// the dW0 products is the bias gradient wherever a group of four columns has room for it).  W0's operands and the
// lane's own last-layer weights are loop-invariant and live in registers; per strip the wave reads only the x tile,
// the last layer's rows (forward) and the strip's output gradients from LDS.  One output (NO == 1): WL[f] is a
// factor of the whole of feature f's layer-0 gradient and is applied once, at the end.
// (amdgpu_waves_per_eu(2): with a budget of 256 registers the compiler takes the MFMA results in VGPRs; with the 512
// of one wave per SIMD it puts them in AGPRs and pays a v_accvgpr_read per element ahead of every VALU use)
// KS = k steps of two of the layer-0 products: the fan-in and the bias column, K0 + 1 <= 2 KS (a compile-time count: a
// branch around an MFMA costs copies of its 16 accumulators).  NG = groups of four columns of the layer-0 gradient (one
// 4x4x1 product per row and group): the K0 columns of dW0 and, where a group has room, the ones column = db0.
// BV: a fan-in of 4 or 8 fills its groups, and db0 is summed on the VALU instead of opening a group for one column.
//   fan-in <= 3: KS 2, NG 1;  4: 3, 1, BV;  5: 3, 2;  6-7: 4, 2;  8: 5, 2, BV;  9: 5, 3   (wide_class below)
template <int NO, int HB, int KS, int NG, bool BV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_ac_learn_wide(LearnArgs P) {
  constexpr int NW = 4, NT = 64 * NW, F = HB * 64;
  constexpr int XR = 4 * NG > 2 * KS ? 4 * NG : 2 * KS;   // rows of the x tile: the inputs, the ones row, zeros
  constexpr int XT = XR * LDT + 4;                    // (+ a dummy slot for idle prefetch lanes)
  constexpr int WT = XT + NO * LDT;                   // per-wave scratch: x^T tile, output-gradient rows
  constexpr int NS = NO + 4 * NG + 1;                 // per-feature sums: NO rows of dWL, 4 NG columns of [dW0 | db0], db0 (BV)
  constexpr int OFF_WL = 0, OFF_BL = OFF_WL + HB * NO * 64, OFF_S = OFF_BL + 4, OFF_T = OFF_S + NW * 16,
                OFF_R = OFF_T + NW * WT;
  static_assert(XT % 4 == 0 && WT % 4 == 0 && OFF_T % 4 == 0, "16 B rows");
  __shared__ __align__(16) float smem[OFF_R + NW * NS * F];
  float *const sWL = smem + OFF_WL, *const sBL = smem + OFF_BL, *const scal = smem + OFF_S, *const sR = smem + OFF_R;

  const int net = blockIdx.y;
  const NetArgs &a = P.net[net];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5,
            r32 = lane & 31;
  const int K0 = a.dims[0], H1 = a.dims[1], NOr = a.dims[2];
  const int64_t B = P.B;
  const int A = P.A;

  STAMP(0);
  for (int e = tid; e < HB * NO * 64; e += NT) {      // block q of the hidden features at sWL + q * NO * 64, [n][64]
    const int q = e / (NO * 64), r = e - q * NO * 64, n = r >> 6, k = q * 64 + (r & 63);
    sWL[e] = (n < NOr && k < H1) ? a.w[1][n * H1 + k] : 0.f;
  }
  if (tid < 4) sBL[tid] = tid < NOr ? a.b[1][tid] : 0.f;
  float *sX = smem + OFF_T + wave * WT;     // [XR][LDT]  x'^T of the strip: rows < K0 the observation, row K0 ones
  float *sGo = sX + XT;                     // [NO][LDT]  gradient w.r.t. the outputs
  for (int e = lane; e < WT; e += 64) sX[e] = 0.f;
  if (lane < 32) sX[K0 * LDT + lane] = 1.f;
  // layer 0's weight operand, the same register for both products: W0'[f][k = 2 s + half] of feature
  // f = 64 q + 32 nb + (lane & 31); and the lane's own last-layer weights
  float wop[HB][2][KS];
#pragma unroll
  for (int q = 0; q < HB; ++q)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int f = q * 64 + nb * 32 + r32;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + half;
        float v = 0.f;
        if (f < H1 && k < K0) v = a.w[0][f * K0 + k];
        if (f < H1 && k == K0) v = a.b[0][f];
        wop[q][nb][s] = v;
      }
    }
  __syncthreads();
  STAMP(1);

  // ---- per-wave persistent state: everything per FEATURE (lane & 31 of block q, half nb), summed over the rows
  // 8 j + 4 half + i of every strip
  f32x2 dwl[HB][2][NO], vdb0[HB][2];    // (even rows, odd rows)
  f32x4 dw0m[HB][2][NG];   // element i of lane 4 b + j, group g: [dW0 | db0][feature 4 (b & 7) + i][4 g + j], rows of half b >> 3
#pragma unroll
  for (int q = 0; q < HB; ++q)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int n = 0; n < NO; ++n) dwl[q][nb][n] = f32x2{0.f, 0.f};
      vdb0[q][nb] = f32x2{0.f, 0.f};
#pragma unroll
      for (int g = 0; g < NG; ++g) dw0m[q][nb][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  float dbL[NO], s_main = 0.f, s_ent = 0.f, s_dls[NO], den[NO], var[NO], logsig[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) {
    dbL[n] = 0.f; s_dls[n] = 0.f;
    den[n] = var[n] = 1.f; logsig[n] = 0.f;
    if (net == 1 && P.kind == KIND_PPO && n < A) {
      const float sigma = expf(P.log_std[n]);   // self.log_std.exp()
      var[n] = uni(sigma * sigma);              // scale ** 2
      logsig[n] = uni(logf(sigma));             // scale.log()
      den[n] = uni(2.0f * (sigma * sigma));
    }
  }
  const float invB = uni(1.0f / (float)B);
  const float gmin = uni(-(1.0f / (float)(B * A)));   // d loss / d min(surr1, surr2) element

  // ---- a strip's inputs are fetched one strip ahead, into registers, branch-free (as in k_ac_learn)
  constexpr int XPF = KS;      // 32 K0 <= 64 KS elements
  const int nelem = 32 * K0;
  int xoff[XPF];
#pragma unroll
  for (int j = 0; j < XPF; ++j) {
    const int e = lane + 64 * j;
    xoff[j] = e < nelem ? (e % K0) * LDT + e / K0 : XT - 4;
  }
  const float *in0 = net == 0 ? P.returns : P.adv;
  const float *in_act[NO], *in_olp[NO];
  const bool ppo_actor = net == 1 && P.kind == KIND_PPO;
  const int astr = ppo_actor ? A : 1;
#pragma unroll
  for (int n = 0; n < NO; ++n) {      // A2C: the action index of the row, in every slot
    in_act[n] = net == 0 ? in0 : (ppo_actor ? P.action + (n < A ? n : 0) : P.action);
    in_olp[n] = ppo_actor ? P.old_logp + (n < A ? n : 0) : in0;
  }
  float xr[XPF], p_in0 = 0.f, p_act[NO], p_olp[NO];
  const int64_t nstrips = (B + 31) >> 5;
  auto fetch = [&](int64_t strip) {      // (strip is wave-uniform: scalar bases, 32-bit lane offsets)
    strip = strip < nstrips ? strip : nstrips - 1;
    const int64_t row0 = strip << 5;
    const int64_t left = B - row0;
    const int rows = (int)(left < 32 ? left : 32);
    const int rl = r32 < rows ? r32 : rows - 1;
    const float *base = P.obs + row0 * K0;
    const int last = rows * K0 - 1;
#pragma unroll
    for (int j = 0; j < XPF; ++j) {
      const int e = lane + 64 * j;
      xr[j] = base[e < last ? e : last];
    }
    p_in0 = (in0 + row0)[rl];
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      p_act[n] = (in_act[n] + row0 * astr)[rl * astr];
      p_olp[n] = (in_olp[n] + row0 * astr)[rl * astr];
    }
  };

  const int64_t sstride = (int64_t)gridDim.x * NW;
  int64_t strip = (int64_t)blockIdx.x * NW + wave;
  fetch(strip);
  PH_DECL
  for (; strip < nstrips; strip += sstride) {
    const bool valid = (strip << 5) + r32 < B;
    PH(5);
#pragma unroll
    for (int j = 0; j < XPF; ++j) sX[xoff[j]] = xr[j];
    const float ret = p_in0, ad = p_in0;
    float act[NO], olp[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) { act[n] = p_act[n]; olp[n] = p_olp[n]; }
    fetch(strip + sstride);
    wave_lds_fence();
    float xop[KS];           // the sample operand, the same register for both products: x'[row = lane & 31][k = 2 s + half]
#pragma unroll
    for (int s = 0; s < KS; ++s) xop[s] = sX[(2 * s + half) * LDT + r32];
    PH(6);

    // ---- forward (lane = sample): out = WL relu(W0' x') + bL, block by block; the activations are not kept
    float out[NO];
    // (an offset the compiler cannot see through: it may otherwise prove the last layer's rows loop-invariant and
    // hold all of them in registers across the strip loop)
    int wl_off = 0;
    asm volatile("" : "+v"(wl_off));
    {
      f32x2 pacc2[NO];
#pragma unroll
      for (int n = 0; n < NO; ++n) pacc2[n] = f32x2{0.f, 0.f};
#pragma unroll
      for (int q = 0; q < HB; ++q) {
        f32x16 h[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) h[nb][r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          h[0] = MFMA(wop[q][0][s], xop[s], h[0]);
          h[1] = MFMA(wop[q][1][s], xop[s], h[1]);
        }
        relu2(h);
        fwd_last_acc<NO>(pacc2, h, sWL + wl_off + q * NO * 64, half);
        // (one block at a time: left alone, the scheduler interleaves all blocks' products and parks their
        // accumulators in AGPRs, a copy in and a copy out per use)
        __builtin_amdgcn_sched_barrier(0);
      }
      PH(7);
      fwd_last_finish<NO>(out, pacc2, sBL);
    }
    PH(0);
    // ---- loss and its gradient w.r.t. out (arithmetic of k_ppo_loss)
    float gout[NO];
    loss_grad<NO>(P, net, A, valid, half, LossConst{invB, gmin}, out, ret, ad, act, olp, den, var, logsig, gout, s_main,
                  s_ent, s_dls);
#pragma unroll
    for (int n = 0; n < NO; ++n)
      if (half == 0) dbL[n] += gout[n];
    if (half == 0) {
#pragma unroll
      for (int n = 0; n < NO; ++n) sGo[n * LDT + r32] = gout[n];
    }
    wave_lds_fence();
    // rows 8 j + 4 half + i of the strip, register 4 j + i: the output gradients (broadcast reads) and, for the
    // 4x4 products of column group g, column 4 g + (lane & 3) of x' at those rows
    // (HOLDX: held for the whole strip; with many blocks and groups the registers go to the accumulators instead and
    // the x' quads are read again per 32 features, under the recompute's MFMAs)
    constexpr bool HOLDX = HB * NG <= 4;
    float4 go[NO][4], xq[NG][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int n = 0; n < NO; ++n) go[n][j] = ld4(sGo + n * LDT + 8 * j + 4 * half);
      if (HOLDX) {
#pragma unroll
        for (int g = 0; g < NG; ++g) xq[g][j] = ld4(sX + (4 * g + (lane & 3)) * LDT + 8 * j + 4 * half);
      }
    }
    PH(1);

    // ---- backward (lane = feature)
#pragma unroll
    for (int q = 0; q < HB; ++q)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        f32x16 pre;
#pragma unroll
        for (int r = 0; r < 16; ++r) pre[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) pre = MFMA(xop[s], wop[q][nb][s], pre);
        if (!HOLDX) {
          const float *xb = sX + wl_off + (lane & 3) * LDT + 4 * half;      // (wl_off = 0: keeps these reads in the loop)
#pragma unroll
          for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) xq[g][j] = ld4(xb + 4 * g * LDT + 8 * j);
        }
        float wlb[NO];       // the feature's own last-layer weights (read under the MFMAs)
#pragma unroll
        for (int n = 0; n < NO; ++n) wlb[n] = NO == 1 ? 0.f : sWL[wl_off + (q * NO + n) * 64 + nb * 32 + r32];
        // two rows at a time: the pair (r, r + 1) is a register pair of the product and of the 16 B reads, so the
        // FMAs are v_pk_fma_f32 (two per instruction, no moves); sums of even and odd rows are kept apart
        f32x2 sdb = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 x = {pre[r], pre[r + 1]};
          f32x2 gn[NO], av;
#pragma unroll
          for (int n = 0; n < NO; ++n) {
            const float4 g4 = go[n][r >> 2];
            gn[n] = (r & 2) ? f32x2{g4.z, g4.w} : f32x2{g4.x, g4.y};
          }
          if constexpr (NO == 1) {
            av = f32x2{x.x > 0.f ? gn[0].x : 0.f, x.y > 0.f ? gn[0].y : 0.f};      // (WL[f] is applied at the end)
            dwl[q][nb][0] = __builtin_elementwise_fma(x, av, dwl[q][nb][0]);    // x > 0 wherever av != 0
          } else {
            const f32x2 hr = {__int_as_float(__float_as_int(x.x) > 0 ? __float_as_int(x.x) : 0),
                              __int_as_float(__float_as_int(x.y) > 0 ? __float_as_int(x.y) : 0)};
            f32x2 t = f32x2{wlb[0], wlb[0]} * gn[0];
#pragma unroll
            for (int n = 1; n < NO; ++n) t = __builtin_elementwise_fma(f32x2{wlb[n], wlb[n]}, gn[n], t);
#pragma unroll
            for (int n = 0; n < NO; ++n) dwl[q][nb][n] = __builtin_elementwise_fma(hr, gn[n], dwl[q][nb][n]);
            av = f32x2{x.x > 0.f ? t.x : 0.f, x.y > 0.f ? t.y : 0.f};
          }
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const float4 x4 = xq[g][r >> 2];
            dw0m[q][nb][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.x, (r & 2) ? x4.z : x4.x, dw0m[q][nb][g], 0, 0, 0);
            dw0m[q][nb][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av.y, (r & 2) ? x4.w : x4.y, dw0m[q][nb][g], 0, 0, 0);
          }
          if (BV) sdb += av;
        }
        if (BV) vdb0[q][nb] += sdb;
        __builtin_amdgcn_sched_barrier(0);
      }
    wave_lds_fence();     // the next strip overwrites sX and sGo
    PH(3);
  }
  PH_OUT;

  STAMP(2);
  // =========================================================================== partials out
  float *part = P.ws + ((int64_t)net * gridDim.x + blockIdx.x) * P.pstride;
  float sc[2 + 2 * NO];
  sc[0] = trl_wave_sum(s_main);
  sc[1] = trl_wave_sum(s_ent);
#pragma unroll
  for (int n = 0; n < NO; ++n) {
    sc[2 + n] = trl_wave_sum(s_dls[n]);
    sc[2 + NO + n] = trl_wave_sum(dbL[n]);
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 2 + 2 * NO; ++j) scal[wave * 16 + j] = sc[j];
  }
  // per-feature sums: the two lane halves (the same features over different rows) are added first, then one copy per
  // wave, in wave order
  float *mine = sR + wave * NS * F;
#pragma unroll
  for (int q = 0; q < HB; ++q)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int fb = q * 64 + nb * 32;
#pragma unroll
      for (int n = 0; n < NO; ++n) {
        const float v = dwl[q][nb][n].x + dwl[q][nb][n].y;
        const float t = v + __shfl_xor(v, 32, 64);
        if (half == 0) mine[n * F + fb + r32] = t;
      }
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float v = dw0m[q][nb][g][i];
          const float t = v + __shfl_xor(v, 32, 64);
          if (half == 0) mine[(NO + 4 * g + (lane & 3)) * F + fb + 4 * (r32 >> 2) + i] = t;
        }
      {
        const float v = vdb0[q][nb].x + vdb0[q][nb].y;
        const float t = v + __shfl_xor(v, 32, 64);
        if (half == 0) mine[(NO + 4 * NG) * F + fb + r32] = t;
      }
    }
  __syncthreads();
  for (int e = tid; e < NS * F; e += NT) {
    const int sl = e / F, f = e - sl * F;
    float v = sR[e];
#pragma unroll
    for (int c = 1; c < NW; ++c) v += sR[c * NS * F + e];
    if (f >= H1) continue;
    if (sl < NO) {
      if (sl < NOr) part[a.poff[1] + sl * H1 + f] = v;
    } else {
      const int k = sl - NO;
      if (NO == 1) v *= sWL[(f >> 6) * 64 + (f & 63)];
      if (k < K0) part[a.poff[0] + f * K0 + k] = v;
      else if (BV ? k == 4 * NG : k == K0) part[a.poff[0] + H1 * K0 + f] = v;      // db0: the VALU sum, or the ones column
    }
  }
  if (tid < 2 + 2 * NO) {
    float v = scal[tid];
#pragma unroll
    for (int w = 1; w < NW; ++w) v += scal[w * 16 + tid];
    if (tid < 2 + NO) {                       // s_main, s_ent, dls[n]
      if (tid < 2 + A) part[a.nparams + tid] = v;
    } else {                                  // dbL[n]
      const int n = tid - 2 - NO;
      if (n < NOr) part[a.poff[1] + NOr * H1 + n] = v;
    }
  }
  STAMP(3);
}

// ---- inference: the forward alone over B rows (PPOController.compute_return's value / mean /
// log-prob pass, ppo_controller.py:34,48): only the outputs are written, several workgroups per CU
// ---------------------------------------------------------------------------------------------------------------
// Role-split variant of k_ac_learn for two hidden layers (<= 64 wide) over a fan-in <= 4 (PPO / A2C on Pendulum- or
// CartPole-sized observations): 512 threads, and the two waves that share a SIMD (w and w + 4) share every strip.
//
// Why (tools/mfma_valu_probe.hip).  v_mfma_f32_32x32x2_f32 holds ITS OWN wave for its 64 cycles: a VALU or LDS
// instruction behind it adds its full issue time (16 v_fma behind each MFMA: 64 -> 129 cycles per MFMA).  One wave
// per SIMD therefore pays MFMA + VALU + every LDS round trip (k_ac_learn: 20.5 k cycles per strip for 12.5 k of MFMA).
// A second wave on the SIMD does get work done meanwhile -- LDS traffic at full rate, and about two VALU
// instructions per partner MFMA (26-28 cycles each beside an MFMA stream, whatever their independence) -- but two
// symmetric waves at 256 registers each do not fit this step without spills (64 dW1 accumulators per wave).  So the
// pair splits the STEP instead of the strips:
//   wave A (w < 4): forward, loss, last layer backward, the ReLU mask of layer 1, the transposing writes of the
//     g2 and h1^T tiles, db1; it holds no weight-gradient tile;
//   wave B (w + 4): everything that only needs those tiles -- dW1 (64 MFMAs), the input gradient of layer 1 (64
//     MFMAs: nothing A computes depends on it), its mask, the g1 tile, db0 and dW0 -- with the 64 dW1 accumulators.
// Each wave's LDS round trips, waits and part of its VALU work run under the other's MFMAs.  Hand-off through one
// set of LDS tiles per pair and counters (LDS operations of the CU execute in issue order, so a counter written
// after a tile is seen after it):
//   pub[0] strips whose g2 tile is written, pub[1] strips whose h1^T tile is written (A)
//   pub[2] strips whose g2 / h1^T tiles B has read (B reads them into registers at once, so A can write the next)
//   pub[3] strips whose x^T tile B is done with (four x tiles in rotation: A never waits on this one)
// Every wait is bounded (a pair that lost its partner finishes with wrong numbers rather than hanging the GPU).
// Strip assignment and partial-sum order are those of k_ac_learn<3, 1, NO, true> (pair p = its wave p); all results
// but the layer-0 gradients (summed row by row on the 4x4 MFMA here, in four interleaved partial sums there) are the
// same bits.  Measured with the in-kernel trace (tools/learn_pair_trace.py): the two waves' MFMA streams do not
// interleave -- whichever streams keeps the pipe (wave priorities change nothing) -- and the period is B's 128 MFMAs
// + A's 68 + about 3 k cycles of dependent LDS round trips at the strip boundary in which neither has one to issue.
__device__ __forceinline__ void pair_wait(const int *ctr, int want) {
  for (int spin = 0; spin < (1 << 20); ++spin) {
    const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (v >= want) break;
    __builtin_amdgcn_s_sleep(1);
  }
  wave_lds_fence();
}
__device__ __forceinline__ void pair_post(int *ctr, int value, int lane) {
  wave_lds_fence();      // (the tile accesses above are issued, and LDS keeps issue order)
  if (lane == 0) __hip_atomic_store(ctr, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int NO>
__global__ __launch_bounds__(512) void k_ac_learn_pair(LearnArgs P) {
  constexpr int D = 3, NT = 512, LDW0 = 5;
  constexpr int XT = 4 * LDT + 4;                     // x^T tile: 4 rows (+ a dummy slot for idle prefetch lanes)
  constexpr int OFF_B = 0;                            // b0, b1, bL
  constexpr int OFF_WL = OFF_B + 64 + 64 + 4;
  constexpr int OFF_W0 = OFF_WL + NO * 64;
  constexpr int OFF_W1 = OFF_W0 + 64 * LDW0;
  constexpr int OFF_W1T = OFF_W1 + 64 * LDW;
  constexpr int OFF_T = OFF_W1T + 64 * LDW;
  constexpr int NXB = 4;                              // x tiles in rotation (see wave A's loop)
  constexpr int PT = NXB * XT + 3 * 64 * LDT + 8;     // per pair: the x tiles, g2, h1^T, g1 tiles, the counters
  __shared__ __align__(16) float smem[OFF_T + 4 * PT];
  float *const sB = smem + OFF_B, *const sWL = smem + OFF_WL, *const sW0 = smem + OFF_W0, *const sW1 = smem + OFF_W1,
               *const sW1T = smem + OFF_W1T, *const sT = smem + OFF_T;
  static_assert(PT >= 32 * 64 && 4 * PT >= 4 * 12 * 64 + 64, "the end-of-kernel reduction re-uses the pair regions");

  const int net = blockIdx.y;
  const NetArgs &a = P.net[net];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5,
            r32 = lane & 31;
  const int pair = wave & 3;
#ifdef TRL_TUNING
  const bool roleA = (P.tune_prio & 16) ? wave >= 4 : wave < 4;      // A/B runs of tools/: bit 4 swaps the roles,
  const int dynp = roleA ? (P.tune_prio & 3) : ((P.tune_prio >> 2) & 3);   // bits 1:0 / 3:2: priority of A's / B's matrix phases
#define MATRIX_PHASE() do { if (dynp == 1) __builtin_amdgcn_s_setprio(1); else if (dynp == 2) __builtin_amdgcn_s_setprio(2); else if (dynp == 3) __builtin_amdgcn_s_setprio(3); } while (0)
#define VECTOR_PHASE() do { if (dynp) __builtin_amdgcn_s_setprio(0); } while (0)
  const bool hold = (P.tune_prio & 32) != 0;                          // bit 5: B's stream waits for A's next forward
#else
  const bool roleA = wave < 4;
  constexpr bool hold = false;
#define MATRIX_PHASE()
#define VECTOR_PHASE()
#endif
  const int K0 = a.dims[0], H1 = a.dims[1], HL = a.dims[2], NOr = a.dims[3];
  const int64_t B = P.B;
  const int A = P.A;

  stage_weights<D, LDW0, NO, true, NT>(a, sW0, sW1, sW1T, sWL, sB, tid);
  float *sXp = sT + pair * PT;          // [NXB][XT]   x^T of the strip, in rotation
  float *sG2 = sXp + NXB * XT;            // [64][LDT]   gradient w.r.t. layer 1's pre-activation
  float *sH1 = sG2 + 64 * LDT;          // [64][LDT]   h1^T
  float *sG1 = sH1 + 64 * LDT;          // [64][LDT]   gradient w.r.t. layer 0's pre-activation
  int *pub = (int *)(sG1 + 64 * LDT);   // [4]         hand-off counters
  if (roleA) {
    for (int e = lane; e < NXB * XT; e += 64) sXp[e] = 0.f;   // rows >= K0 stay zero for good ...
    if (K0 < 4 && lane < 32) {                                // ... but row 3, where there is room: ones (W0's column 3
#pragma unroll                                                 // is zero, so the forward does not see them), which make
      for (int t = 0; t < NXB; ++t) sXp[t * XT + 3 * LDT + lane] = 1.f;   // the layer-0 bias gradient a column of dW0
    }
    if (lane < 8) pub[lane] = 0;
  }
  __syncthreads();

  const int64_t nstrips = (B + 31) >> 5;
  const int64_t sstride = (int64_t)gridDim.x * 4;
  int64_t strip = (int64_t)blockIdx.x * 4 + pair;

  // What a role hands to the end-of-kernel sums.  The big per-role state (A: the last layer's per-lane gradient
  // partials; B: the dW1 accumulators) is declared inside its role's branch, so that it takes no registers in the
  // other role's loop; each branch parks it in LDS before the common code.  Both branches execute the same number
  // of workgroup barriers.
  float vdb0 = 0.f;                                                  // B
  f32x4 dw0m = {0.f, 0.f, 0.f, 0.f};                                 // B: dW0[4 (lane / 4) + i][lane % 4] in element i
  float vdb1 = 0.f, vec_wl[NO], sc[2 + 2 * NO];                      // A
#pragma unroll
  for (int n = 0; n < NO; ++n) vec_wl[n] = 0.f;
#pragma unroll
  for (int j = 0; j < 2 + 2 * NO; ++j) sc[j] = 0.f;
  float *mine = sT + pair * PT;         // after the loops: [64][64] dW1 tile of the pair (B), then A's transposes
  static_assert(PT >= 64 * 64 + 64 * LDT, "the pair region holds B's dW1 tile and A's transpose tile");

  if (roleA) {
    // =================================================================================== wave A: the chain
    f32x16 dWL[NO][2];
    float dbL[NO], s_main = 0.f, s_ent = 0.f, s_dls[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      dbL[n] = 0.f; s_dls[n] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dWL[n][kb][r] = 0.f;
    }
    float den[NO], var[NO], logsig[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      den[n] = var[n] = 1.f; logsig[n] = 0.f;
      if (net == 1 && P.kind == KIND_PPO && n < A) {
        const float sigma = expf(P.log_std[n]);   // self.log_std.exp()
        var[n] = uni(sigma * sigma);              // scale ** 2
        logsig[n] = uni(logf(sigma));             // scale.log()
        den[n] = uni(2.0f * (sigma * sigma));
      }
    }
    const LossConst lc{uni(1.0f / (float)B), uni(-(1.0f / (float)(B * A)))};
    const int q64 = 64 / K0, m64 = 64 % K0;
    const int ksteps0 = (K0 + 1) >> 1;
    constexpr int XPF = 2;               // 32 rows x fan-in <= 4: at most 128 elements
    const int nelem = 32 * K0;
    int xoff[XPF];
    {
      int rr = lane / K0, kk = lane % K0;
#pragma unroll
      for (int j = 0; j < XPF; ++j) {
        xoff[j] = lane + 64 * j < nelem ? kk * LDT + rr : XT - 4;
        rr += q64; kk += m64;
        if (kk >= K0) { kk -= K0; ++rr; }
      }
    }
    const float *in0 = net == 0 ? P.returns : P.adv;
    const float *in_act[NO], *in_olp[NO];
    const bool ppo_actor = net == 1 && P.kind == KIND_PPO;
    const int astr = ppo_actor ? A : 1;
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      in_act[n] = net == 0 ? in0 : (ppo_actor ? P.action + (n < A ? n : 0) : P.action);
      in_olp[n] = ppo_actor ? P.old_logp + (n < A ? n : 0) : in0;
    }
    float xr[XPF], p_in0 = 0.f, p_act[NO], p_olp[NO];
    // (st is wave-uniform: scalar bases, 32-bit lane offsets)
    auto fetch_x = [&](int64_t st) {     // observations of a strip
      st = st < nstrips ? st : nstrips - 1;
      const int64_t row0 = st << 5;
      const int64_t left = B - row0;
      const int rows = (int)(left < 32 ? left : 32);
      const float *base = P.obs + row0 * K0;
      const int last = rows * K0 - 1;
#pragma unroll
      for (int j = 0; j < XPF; ++j) {
        const int e = lane + 64 * j;
        xr[j] = base[e < last ? e : last];
      }
    };
    auto fetch_p = [&](int64_t st) {     // per-row loss inputs of a strip
      st = st < nstrips ? st : nstrips - 1;
      const int64_t row0 = st << 5;
      const int64_t left = B - row0;
      const int rows = (int)(left < 32 ? left : 32);
      const int rl = r32 < rows ? r32 : rows - 1;
      p_in0 = (in0 + row0)[rl];
#pragma unroll
      for (int n = 0; n < NO; ++n) {
        p_act[n] = (in_act[n] + row0 * astr)[rl * astr];
        p_olp[n] = (in_olp[n] + row0 * astr)[rl * astr];
      }
    };
    // The x^T tile of strip it + 1 is written in the middle of strip it (its loads were issued a strip earlier), so
    // that nothing stands between the end of B's MFMA stream and this wave's next forward but the loss inputs'
    // copy.  NXB tiles in rotation: tile (it + 1) % NXB was last read by B's layer-0 gradient of strip it - 3,
    // which the tile wait of strip it - 1 has long seen.
    auto put_x = [&](int slot) {
      float *sXw = sXp + (slot % NXB) * XT;
#pragma unroll
      for (int j = 0; j < XPF; ++j) sXw[xoff[j]] = xr[j];
    };
    fetch_x(strip);
    fetch_p(strip);
    put_x(0);
    fetch_x(strip + sstride);
    for (int it = 0; strip < nstrips; strip += sstride, ++it) {
      const int64_t row0 = strip << 5;
      const bool valid = row0 + r32 < B;
      TRACE(0);
      float *sX = sXp + (it % NXB) * XT;      // written in the middle of the previous strip (put_x)
      const float ret = p_in0, ad = p_in0;
      float act[NO], olp[NO];
#pragma unroll
      for (int n = 0; n < NO; ++n) { act[n] = p_act[n]; olp[n] = p_olp[n]; }
      fetch_p(strip + sstride);
      wave_lds_fence();

      f32x16 h1[2], h2[2];
      float out[NO];
      TRACE(1);
      MATRIX_PHASE();
      fwd_layer0<LDW0>(h1, sX, sW0, sB, ksteps0, half, r32);
      // h1^T goes out as soon as it exists (B has read the previous strip's tiles long since): B picks it up while
      // this wave is still in its forward, and only the g2 tile is left between the loss and B's MFMAs
      TRACE(3);
      pair_wait(pub + 2, it);
      TRACE(4);
      put_tile(sH1, h1, half, r32);
      pair_post(pub + 1, it + 1, lane);
      fwd_layer1(h2, h1, sW1, sB + 64, half, r32);
      VECTOR_PHASE();
      pair_wait(pub + 3, it - 2);             // (never blocks: see put_x)
      put_x(it + 1);
      fetch_x(strip + 2 * sstride);
      if (hold && lane == 0) __hip_atomic_store(pub + 4, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      TRACE(2);
      fwd_last<NO>(out, h2, sWL, sB + 128, half);

      float gout[NO];
      loss_grad<NO>(P, net, A, valid, half, lc, out, ret, ad, act, olp, den, var, logsig, gout, s_main, s_ent, s_dls);
#pragma unroll
      for (int n = 0; n < NO; ++n)
        if (half == 0) dbL[n] += gout[n];

      // last layer backward (VALU): dWL partials per lane; g = (WL^T gout) * (h2 > 0)
      f32x16 g[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x2 t0 = {0.f, 0.f}, t1 = {0.f, 0.f};      // (two features per v_pk_fma_f32)
#pragma unroll
          for (int n = 0; n < NO; ++n) {
            const float4 w4 = *(const float4 *)(sWL + n * 64 + kb * 32 + 8 * q + 4 * half);
            t0 = __builtin_elementwise_fma(f32x2{w4.x, w4.y}, f32x2{gout[n], gout[n]}, t0);
            t1 = __builtin_elementwise_fma(f32x2{w4.z, w4.w}, f32x2{gout[n], gout[n]}, t1);
          }
          const float t[4] = {t0.x, t0.y, t1.x, t1.y};
#pragma unroll
          for (int i = 0; i < 4; ++i) g[kb][4 * q + i] = h2[kb][4 * q + i] > 0.f ? t[i] : 0.f;
        }
      // publish g2: the last thing B's MFMAs wait for
      TRACE(7);
      put_tile(sG2, g, half, r32);
      pair_post(pub + 0, it + 1, lane);
      TRACE(5);

      // off the critical path (B's MFMA stream has started: VALU work here runs at 26-28 cycles per instruction,
      // tools/mfma_valu_probe.hip, but this wave's next forward waits for that stream to end anyway)
#pragma unroll
      for (int n = 0; n < NO; ++n)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) dWL[n][kb][r] = fmaf(gout[n], h2[kb][r], dWL[n][kb][r]);
      vdb1 += row_sum32(sG2 + lane * LDT);     // db1[lane]: the row of the tile this wave has just written
      TRACE(6);
    }
    if (lane == 0) __hip_atomic_store(pub + 4, 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // no forward left to wait for
    __syncthreads();      // (1) every pair is through its strips: the pair regions are free
    // per-lane partial vectors dWL[n] -> per-feature sums over the wave's 32 rows, through a tile behind B's
    float *tr = mine + 64 * 64;
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      put_tile(tr, dWL[n], half, r32);
      wave_lds_fence();
      float sum = 0.f;
      for (int r = 0; r < 32; ++r) sum += tr[lane * LDT + r];
      vec_wl[n] = sum;
      wave_lds_fence();
    }
    sc[0] = trl_wave_sum(s_main);
    sc[1] = trl_wave_sum(s_ent);
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      sc[2 + n] = trl_wave_sum(s_dls[n]);
      sc[2 + NO + n] = trl_wave_sum(dbL[n]);
    }
  } else {
    // =================================================================================== wave B: the tiles' readers
    f32x16 accW1[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW1[nb][kb][r] = 0.f;
    const float *ga = sG2 + r32 * LDT + 4 * half, *ha = sH1 + r32 * LDT + 4 * half;
    for (int it = 0; strip < nstrips; strip += sstride, ++it) {
      // Everything this strip needs from the two tiles is read up front, into registers, and the tiles handed back:
      // A writes the next strip's while this wave's 128 MFMAs run (held for the 128 MFMAs, the single set of tiles
      // put A's tile writes and this wave's vector phase in series with them).  h1^T comes first, while A is still
      // in its forward:
      //   h1 in the accumulator layout (lane = row, register r = feature feat(r, half)): the ReLU mask of layer 0;
      //   as rows in the order (j, half, i) -> row 8j + 4half + i (a lane's four steps are one 16 B read): the B
      //   operand of dW1 += g2 h1^T
      TRACE(0);
      pair_wait(pub + 1, it + 1);
      f32x16 hm[2], gB[2];
      float4 G0[4], G1[4], X0[4], X1[4];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) hm[kb][r] = sH1[(kb * 32 + feat(r, half)) * LDT + r32];
#pragma unroll
      for (int j = 0; j < 4; ++j) { X0[j] = ld4(ha + 8 * j); X1[j] = ld4(ha + 32 * LDT + 8 * j); }
      //   then g2, in the accumulator layout (the B operand of the input-gradient product) and as rows (the A
      //   operand of dW1)
      pair_wait(pub + 0, it + 1);
      TRACE(1);
#pragma unroll
      for (int j = 0; j < 4; ++j) { G0[j] = ld4(ga + 8 * j); G1[j] = ld4(ga + 32 * LDT + 8 * j); }
      // Order of the two MFMA streams.  A wave that streams MFMAs keeps the SIMD's matrix pipe, and its partner's
      // VALU work crawls meanwhile (26-28 cycles per instruction).  Left to the race, this wave's stream goes first
      // and A's forward of the NEXT strip after it; A's loss and last layer then run with the pipe idle.  Held back
      // until that forward has ended, this stream covers them instead, and A's forward covers this wave's mask.
      if (hold) pair_wait(pub + 4, it + 2);
      TRACE(2);
      MATRIX_PHASE();
      // The 8 row reads are all the first product needs: its MFMAs start behind them, and the 32 reads of the
      // accumulator-layout copy (the second product's operand) go out two behind each of the first 16 MFMAs -- LDS
      // traffic runs at full rate beside an MFMA stream -- instead of ahead of the first one.
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        accW1[0][0] = MFMA(f4(G0[0], i), f4(X0[0], i), accW1[0][0]);
        accW1[0][1] = MFMA(f4(G0[0], i), f4(X1[0], i), accW1[0][1]);
        accW1[1][0] = MFMA(f4(G1[0], i), f4(X0[0], i), accW1[1][0]);
        accW1[1][1] = MFMA(f4(G1[0], i), f4(X1[0], i), accW1[1][1]);
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) gB[kb][r] = sG2[(kb * 32 + feat(r, half)) * LDT + r32];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // two LDS reads
      }
      pair_post(pub + 2, it + 1, lane);         // (LDS keeps issue order: the store follows the reads above)
#pragma unroll
      for (int j = 1; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          accW1[0][0] = MFMA(f4(G0[j], i), f4(X0[j], i), accW1[0][0]);
          accW1[0][1] = MFMA(f4(G0[j], i), f4(X1[j], i), accW1[0][1]);
          accW1[1][0] = MFMA(f4(G1[j], i), f4(X0[j], i), accW1[1][0]);
          accW1[1][1] = MFMA(f4(G1[j], i), f4(X1[j], i), accW1[1][1]);
        }
      // layer 1 backward, input side: t^T = W1^T g2^T (A holds g2 in registers too and would run this product
      // itself; here it costs A nothing), then the mask: g1 = t * (h1 > 0)
      f32x16 t[2];
      TRACE(3);
      bwd_dx1(t, gB, sW1T, half, r32);
      VECTOR_PHASE();
      TRACE(4);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[kb][r] = hm[kb][r] > 0.f ? t[kb][r] : 0.f;
      put_tile(sG1, t, half, r32);
      wave_lds_fence();
      TRACE(5);
      // layer 0: dW0[n][k] += sum_row g1[n][row] x[k][row] as 32 one-step products of v_mfma_f32_4x4x1 (16 blocks of
      // 4 features x 4 inputs: lane = feature on the A side, lane % 4 = input on the B side; 8 cycles each, against
      // 128 FMAs on the VALU).  With a fan-in < 4 the x tile's row 3 is ones and column 3 is the bias gradient.
      {
        const float *grow = sG1 + lane * LDT, *xrow = sXp + (it % NXB) * XT + (lane & 3) * LDT;
        float4 gv[8], xv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { gv[i] = ld4(grow + 4 * i); xv[i] = ld4(xrow + 4 * i); }
        MATRIX_PHASE();
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) dw0m = __builtin_amdgcn_mfma_f32_4x4x1f32(f4(gv[i], c), f4(xv[i], c), dw0m, 0, 0, 0);
        VECTOR_PHASE();
        if (K0 == 4) {
          float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) { s0 += gv[i].x; s1 += gv[i].y; s2 += gv[i].z; s3 += gv[i].w; }
          vdb0 += (s0 + s1) + (s2 + s3);
        }
      }
      pair_post(pub + 3, it + 1, lane);
      TRACE(6);
    }
    __syncthreads();      // (1)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[(nb * 32 + feat(r, half)) * 64 + kb * 32 + r32] = accW1[nb][kb][r];
  }

  // =========================================================================== partials out
  __syncthreads();        // (2)
  float *part = P.ws + ((int64_t)net * gridDim.x + blockIdx.x) * P.pstride;
  {
    // dW1: the four pairs' tiles, added in pair order
    const int N = a.dims[2], K = H1, off = a.poff[1];
    for (int e = tid; e < N * K; e += NT) {
      const int i = (e / K) * 64 + (e % K);
      part[off + e] = ((sT[i] + sT[PT + i]) + sT[2 * PT + i]) + sT[3 * PT + i];
    }
  }
  __syncthreads();
  // per-feature vectors [pair][slot][64] and scalars [pair][16], summed in pair order
  constexpr int VS = 12;     // slots: db0, db1, NO last-layer rows, 4 dW0 columns
  float *vsum = sT, *scal = sT + 4 * VS * 64;
  if (roleA) {
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 2 + 2 * NO; ++j) scal[pair * 16 + j] = sc[j];
    }
#pragma unroll
    for (int n = 0; n < NO; ++n) vsum[(pair * VS + 2 + n) * 64 + lane] = vec_wl[n];
    vsum[(pair * VS + 1) * 64 + lane] = vdb1;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) vsum[(pair * VS + 2 + NO + (lane & 3)) * 64 + 4 * (lane >> 2) + i] = dw0m[i];
    if (K0 == 4) {
      vsum[(pair * VS + 0) * 64 + lane] = vdb0;
    } else if ((lane & 3) == 3) {          // column 3 of the products: the ones row
#pragma unroll
      for (int i = 0; i < 4; ++i) vsum[(pair * VS + 0) * 64 + 4 * (lane >> 2) + i] = dw0m[i];
    }
  }
  __syncthreads();
  for (int sl = tid >> 6; sl < 2 + NO + 4; sl += 8) {
    const int f = tid & 63;
    const float v = ((vsum[(0 * VS + sl) * 64 + f] + vsum[(1 * VS + sl) * 64 + f]) + vsum[(2 * VS + sl) * 64 + f]) +
                    vsum[(3 * VS + sl) * 64 + f];
    if (sl >= 2 + NO) {
      const int k = sl - 2 - NO;
      if (f < H1 && k < K0) part[a.poff[0] + f * K0 + k] = v;
    } else if (sl == 0) {
      if (f < H1) part[a.poff[0] + H1 * K0 + f] = v;
    } else if (sl == 1) {
      if (f < a.dims[2]) part[a.poff[1] + a.dims[2] * H1 + f] = v;
    } else {
      const int n = sl - 2;
      if (f < HL && n < NOr) part[a.poff[2] + n * HL + f] = v;
    }
  }
  if (tid < 2 + 2 * NO) {
    const float v = ((scal[0 * 16 + tid] + scal[1 * 16 + tid]) + scal[2 * 16 + tid]) + scal[3 * 16 + tid];
    if (tid < 2 + NO) {                       // s_main, s_ent, dls[n]
      if (tid < 2 + A) part[a.nparams + tid] = v;
    } else {                                  // dbL[n]
      const int n = tid - 2 - NO;
      if (n < NOr) part[a.poff[2] + NOr * HL + n] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_ac_learn_pair_bx: the role-split kernel with the three 64 x 64 products of a network -- layer 1 forward, dW1, the
// input gradient of layer 1 -- on v_mfma_f32_32x32x16_bf16 instead of v_mfma_f32_32x32x2_f32.  NOT the parity mode
// (k_ac_learn_pair stays that, and the default): selected by the *_bx6 entry points only.
//
// Arithmetic.  An f32 is exactly the sum of three bf16: hi = its top 16 bits, mid = the top 16 bits of the exact
// remainder, lo = what is then left (8 + 8 + 8 significant bits; two masks and two exact subtractions).  Both operands
// of a product are split this way and six of the nine partial products kept,
//   a b ~ a_lo b_hi + a_hi b_lo + a_mid b_mid + a_mid b_hi + a_hi b_mid + a_hi b_hi       (added in this order),
// each partial product exact (8 x 8 bits) and all sums taken in the MFMA's f32 accumulators.  What is dropped
// (a_mid b_lo + a_lo b_mid + a_lo b_lo) is at most 2^-23 |a| |b| per term, i.e. the result is within
// 1.2e-7 sum |a_k| |b_k| of the exact dot product plus the f32 rounding of the accumulation: the error class of an f32
// dot product in a different summation order ("f32-equivalent"), not the same bits.  One 32-cycle instruction covers
// 16 k for a 32 x 32 tile where the f32 instruction needs eight of 64 cycles: 6 x 32 against 8 x 64 cycles.
//
// Layouts.  The 32 x 32 accumulator layout is the same for both instructions (lane = column, register r = row
// feat(r, half)), and a bf16 operand is 8 consecutive k per lane (lane half h holds k = 8 h .. 8 h + 7 of a step).
// Which k that is may be any permutation as long as both operands agree, so an accumulator is the next product's
// operand as it stands: step s' of block kb takes registers 8 s' .. 8 s' + 7, i.e. features kb 32 + feat(8 s' + j, half),
// and the weights' parts are laid out in LDS in that order, once per workgroup (sW1p / sW1Tp: [part][block][step][lane]
// x 16 B, a wave's read of one operand is 1 KB contiguous).  dW1 sums over samples, the lane index of everything A
// computes, so its operands still cross LDS: the g2 tile as in k_ac_learn_pair; h1 is NOT handed over -- wave B
// recomputes it from the x tile, transposed (x as the A operand, W0 as the B operand of the same four f32 MFMAs:
// lane = feature, register r = sample feat(r, half), which is exactly the sample order of the g2 rows' 16 B reads) and
// once more in the accumulator layout for the ReLU mask.  That removes the h1^T tile (LDS: the parts are 48 KB against
// 34 KB of f32 weights) and A's 32 transposing writes per strip.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// the top halves of two floats as one dword of two bf16: low 16 bits from a, high 16 bits from b
__device__ __forceinline__ unsigned int bx_top2(float a, float b) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float bx_top16(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }
// two floats -> one dword of each part
__device__ __forceinline__ void bx_split2(float x0, float x1, unsigned int &hi, unsigned int &mid, unsigned int &lo) {
  const f32x2 w = {x0, x1};
  hi = bx_top2(w[0], w[1]);
  const f32x2 r = w - f32x2{bx_top16(w[0]), bx_top16(w[1])};
  mid = bx_top2(r[0], r[1]);
  const f32x2 l = r - f32x2{bx_top16(r[0]), bx_top16(r[1])};
  lo = bx_top2(l[0], l[1]);
}
// registers 8 s .. 8 s + 7 of an accumulator -> the three operands of one k step
__device__ __forceinline__ void bx_split8(const f32x16 &v, int s, u32x4 (&part)[3]) {
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    unsigned int hi, mid, lo;
    bx_split2(v[8 * s + 2 * d], v[8 * s + 2 * d + 1], hi, mid, lo);
    part[0][d] = hi; part[1][d] = mid; part[2][d] = lo;
  }
}

// W1's parts into LDS in operand order.  sW1p [3][nb][ks][64][4 dwords]: lane (n = nb 32 + r32, half), step ks = (kb, s'):
// W1[n][kb 32 + feat(8 s' + j, half)], j = 0 .. 7.  sW1Tp [3][kbo][ks][64][4]: lane (k = kbo 32 + r32, half), step
// ks = (nb, s'): W1[nb 32 + feat(8 s' + j, half)][k].  Zero beyond the layer's real size.
template <int NT = 512, bool WITH_T = true>
__device__ __forceinline__ void bx_stage_w1(const NetArgs &a, unsigned int *sW1p, unsigned int *sW1Tp, int tid) {
  const int H1 = a.dims[1], H2 = a.dims[2];
  constexpr int NPOS = 2 * 4 * 64 * 4;      // dword positions of one part
  const float *w1 = a.w[1];
  for (int pos = tid; pos < (WITH_T ? 2 : 1) * NPOS; pos += NT) {
    const bool tr = pos >= NPOS;
    const int q = tr ? pos - NPOS : pos;
    const int d = q & 3, ln = (q >> 2) & 63, ks = (q >> 8) & 3, blk = q >> 10;
    const int r32 = ln & 31, half = ln >> 5, kb = ks >> 1, sp = ks & 1;
    const int f0 = kb * 32 + feat(8 * sp + 2 * d, half), f1 = kb * 32 + feat(8 * sp + 2 * d + 1, half);
    const int o = blk * 32 + r32;
    float x0, x1;
    if (!tr) {      // row n = o, columns f0 / f1
      x0 = (o < H2 && f0 < H1) ? w1[o * H1 + f0] : 0.f;
      x1 = (o < H2 && f1 < H1) ? w1[o * H1 + f1] : 0.f;
    } else {        // column k = o, rows f0 / f1
      x0 = (f0 < H2 && o < H1) ? w1[f0 * H1 + o] : 0.f;
      x1 = (f1 < H2 && o < H1) ? w1[f1 * H1 + o] : 0.f;
    }
    unsigned int hi, mid, lo;
    bx_split2(x0, x1, hi, mid, lo);
    unsigned int *dst = tr ? sW1Tp : sW1p;
    dst[q] = hi; dst[NPOS + q] = mid; dst[2 * NPOS + q] = lo;
  }
}

// out[0] (rows blk 0) and out[1] (rows blk 1) += W-parts x b-parts over the four k steps; bsrc[ks >> 1] registers
// 8 (ks & 1) .. + 7 are the B operand of step ks.  The A operands are fetched one step ahead; the two output blocks
// alternate so that no MFMA waits for the one before it.
__device__ __forceinline__ void bx_product(f32x16 (&out)[2], const f32x16 (&bsrc)[2], const unsigned int *sWp, int lane) {
  constexpr int NPOS = 2 * 4 * 64 * 4;
  const u32x4 *base = (const u32x4 *)sWp + lane;
  u32x4 a0[3], a1[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) { a0[p] = base[(p * NPOS) / 4 + 0 * 256 + 0 * 64]; a1[p] = base[(p * NPOS) / 4 + 1 * 256 + 0 * 64]; }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    u32x4 b[3];
    bx_split8(bsrc[ks >> 1], ks & 1, b);
    u32x4 n0[3], n1[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      n0[p] = a0[p]; n1[p] = a1[p];
      if (ks + 1 < 4) { n0[p] = base[(p * NPOS) / 4 + 0 * 256 + (ks + 1) * 64]; n1[p] = base[(p * NPOS) / 4 + 1 * 256 + (ks + 1) * 64]; }
    }
    issue_fence();
    out[0] = MFMA_BF(a0[2], b[0], out[0]); out[1] = MFMA_BF(a1[2], b[0], out[1]);
    out[0] = MFMA_BF(a0[0], b[2], out[0]); out[1] = MFMA_BF(a1[0], b[2], out[1]);
    out[0] = MFMA_BF(a0[1], b[1], out[0]); out[1] = MFMA_BF(a1[1], b[1], out[1]);
    out[0] = MFMA_BF(a0[1], b[0], out[0]); out[1] = MFMA_BF(a1[1], b[0], out[1]);
    out[0] = MFMA_BF(a0[0], b[1], out[0]); out[1] = MFMA_BF(a1[0], b[1], out[1]);
    out[0] = MFMA_BF(a0[0], b[0], out[0]); out[1] = MFMA_BF(a1[0], b[0], out[1]);
#pragma unroll
    for (int p = 0; p < 3; ++p) { a0[p] = n0[p]; a1[p] = n1[p]; }
  }
}

// layer 0 over a fan-in <= 4 as exactly two k steps, no loop (x rows / W0 columns beyond the fan-in are zero, or a ones
// row against a zero column): pre-activations.  TR = false: the accumulator layout of fwd_layer0 (lane = sample, W0 the
// A operand).  TR = true: transposed -- x the A operand, W0 the B operand: lane = feature, register r = sample feat(r, half).
template <int LDW0, bool TR>
__device__ __forceinline__ void bx_layer0(f32x16 (&h)[2], const float *sX, const float *sW0, int half, int r32) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int k = 2 * s + half;
    const float xv = sX[k * LDT + r32];
    const float w0 = sW0[r32 * LDW0 + k];
    const float w1 = sW0[(32 + r32) * LDW0 + k];
    if (TR) { h[0] = MFMA(xv, w0, h[0]); h[1] = MFMA(xv, w1, h[1]); }
    else { h[0] = MFMA(w0, xv, h[0]); h[1] = MFMA(w1, xv, h[1]); }
  }
}

template <int NO>
__global__ __launch_bounds__(512) void k_ac_learn_pair_bx(LearnArgs P) {
  constexpr int NT = 512, LDW0 = 5;
  constexpr int XT = 4 * LDT + 4;                     // x^T tile: 4 rows (+ a dummy slot for idle prefetch lanes)
  constexpr int NPOS = 2 * 4 * 64 * 4;                // dwords of one part of W1 in operand order
  constexpr int OFF_B = 0;                            // b0, b1, bL
  constexpr int OFF_WL = OFF_B + 64 + 64 + 4;
  constexpr int OFF_W0 = OFF_WL + NO * 64;
  constexpr int OFF_P1 = OFF_W0 + 64 * LDW0;
  constexpr int OFF_P1T = OFF_P1 + 3 * NPOS;
  constexpr int OFF_T = OFF_P1T + 3 * NPOS;
  constexpr int NXB = 4;                              // x tiles in rotation
  constexpr int PT_USED = NXB * XT + 2 * 64 * LDT + 8;     // per pair: the x tiles, g2, g1 tiles, the counters
  constexpr int PT = PT_USED > 64 * 64 + 64 * LDT ? PT_USED : 64 * 64 + 64 * LDT;      // (the end-of-kernel sums re-use the pair regions)
  static_assert(OFF_P1 % 4 == 0 && (NXB * XT) % 4 == 0, "16 B alignment of the operand arrays and tiles");
  static_assert((OFF_T + 4 * PT) * 4 <= 160 * 1024, "LDS");
  __shared__ __align__(16) float smem[OFF_T + 4 * PT];
  float *const sB = smem + OFF_B, *const sWL = smem + OFF_WL, *const sW0 = smem + OFF_W0, *const sT = smem + OFF_T;
  unsigned int *const sW1p = (unsigned int *)(smem + OFF_P1), *const sW1Tp = (unsigned int *)(smem + OFF_P1T);
  static_assert(PT >= 32 * 64 && 4 * PT >= 4 * 12 * 64 + 64, "the end-of-kernel reduction re-uses the pair regions");

  const int net = blockIdx.y;
  const NetArgs &a = P.net[net];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5,
            r32 = lane & 31;
  const int pair = wave & 3;
  const bool roleA = wave < 4;
  const int K0 = a.dims[0], H1 = a.dims[1], HL = a.dims[2], NOr = a.dims[3];
  const int64_t B = P.B;
  const int A = P.A;

  for (int e = tid; e < 64 * LDW0; e += NT) {
    const int n = e / LDW0, k = e % LDW0;
    sW0[e] = (n < H1 && k < K0) ? a.w[0][n * K0 + k] : 0.f;
  }
  for (int e = tid; e < NO * 64; e += NT) {
    const int n = e >> 6, k = e & 63;
    sWL[e] = (n < NOr && k < HL) ? a.w[2][n * HL + k] : 0.f;
  }
  for (int e = tid; e < 64 + 64 + 4; e += NT) {      // b0 (64), b1 (64), bL (4)
    float v;
    if (e < 64) v = e < H1 ? a.b[0][e] : 0.f;
    else if (e < 128) v = e - 64 < HL ? a.b[1][e - 64] : 0.f;
    else v = e - 128 < NOr ? a.b[2][e - 128] : 0.f;
    sB[e] = v;
  }
  bx_stage_w1(a, sW1p, sW1Tp, tid);
  float *sXp = sT + pair * PT;          // [NXB][XT]   x^T of the strip, in rotation
  float *sG2 = sXp + NXB * XT;          // [64][LDT]   gradient w.r.t. layer 1's pre-activation
  float *sG1 = sG2 + 64 * LDT;          // [64][LDT]   gradient w.r.t. layer 0's pre-activation
  int *pub = (int *)(sG1 + 64 * LDT);   // [8]         hand-off counters:
  //   pub[0] strips whose g2 tile is written (A)          pub[1] x tiles written (A): strips 0 .. pub[1] - 1
  //   pub[2] strips whose g2 tile B has read               pub[3] strips whose x^T tile B is done with
  if (roleA) {
    for (int e = lane; e < NXB * XT; e += 64) sXp[e] = 0.f;   // rows >= K0 stay zero for good ...
    if (K0 < 4 && lane < 32) {                                // ... but row 3, where there is room: ones (W0's column 3
#pragma unroll                                                 // is zero, so the forward does not see them), which make
      for (int t = 0; t < NXB; ++t) sXp[t * XT + 3 * LDT + lane] = 1.f;   // the layer-0 bias gradient a column of dW0
    }
    if (lane < 8) pub[lane] = 0;
  }
  __syncthreads();

  const int64_t nstrips = (B + 31) >> 5;
  const int64_t sstride = (int64_t)gridDim.x * 4;
  int64_t strip = (int64_t)blockIdx.x * 4 + pair;

  float vdb0 = 0.f;                                                  // B
  f32x4 dw0m = {0.f, 0.f, 0.f, 0.f};                                 // B: dW0[4 (lane / 4) + i][lane % 4] in element i
  float vdb1 = 0.f, vec_wl[NO], sc[2 + 2 * NO];                      // A
#pragma unroll
  for (int n = 0; n < NO; ++n) vec_wl[n] = 0.f;
#pragma unroll
  for (int j = 0; j < 2 + 2 * NO; ++j) sc[j] = 0.f;
  float *mine = sT + pair * PT;         // after the loops: [64][64] dW1 tile of the pair (B), then A's transposes

  if (roleA) {
    // =================================================================================== wave A: the chain
    f32x16 dWL[NO][2];
    float dbL[NO], s_main = 0.f, s_ent = 0.f, s_dls[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      dbL[n] = 0.f; s_dls[n] = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) dWL[n][kb][r] = 0.f;
    }
    float den[NO], var[NO], logsig[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      den[n] = var[n] = 1.f; logsig[n] = 0.f;
      if (net == 1 && P.kind == KIND_PPO && n < A) {
        const float sigma = expf(P.log_std[n]);   // self.log_std.exp()
        var[n] = uni(sigma * sigma);              // scale ** 2
        logsig[n] = uni(logf(sigma));             // scale.log()
        den[n] = uni(2.0f * (sigma * sigma));
      }
    }
    const LossConst lc{uni(1.0f / (float)B), uni(-(1.0f / (float)(B * A)))};
    const int q64 = 64 / K0, m64 = 64 % K0;
    constexpr int XPF = 2;               // 32 rows x fan-in <= 4: at most 128 elements
    const int nelem = 32 * K0;
    int xoff[XPF];
    {
      int rr = lane / K0, kk = lane % K0;
#pragma unroll
      for (int j = 0; j < XPF; ++j) {
        xoff[j] = lane + 64 * j < nelem ? kk * LDT + rr : XT - 4;
        rr += q64; kk += m64;
        if (kk >= K0) { kk -= K0; ++rr; }
      }
    }
    const float *in0 = net == 0 ? P.returns : P.adv;
    const float *in_act[NO], *in_olp[NO];
    const bool ppo_actor = net == 1 && P.kind == KIND_PPO;
    const int astr = ppo_actor ? A : 1;
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      in_act[n] = net == 0 ? in0 : (ppo_actor ? P.action + (n < A ? n : 0) : P.action);
      in_olp[n] = ppo_actor ? P.old_logp + (n < A ? n : 0) : in0;
    }
    float xr[XPF], p_in0 = 0.f, p_act[NO], p_olp[NO];
    auto fetch_x = [&](int64_t st) {     // observations of a strip
      st = st < nstrips ? st : nstrips - 1;
      const int64_t row0 = st << 5;
      const int64_t left = B - row0;
      const int rows = (int)(left < 32 ? left : 32);
      const float *base = P.obs + row0 * K0;
      const int last = rows * K0 - 1;
#pragma unroll
      for (int j = 0; j < XPF; ++j) {
        const int e = lane + 64 * j;
        xr[j] = base[e < last ? e : last];
      }
    };
    auto fetch_p = [&](int64_t st) {     // per-row loss inputs of a strip
      st = st < nstrips ? st : nstrips - 1;
      const int64_t row0 = st << 5;
      const int64_t left = B - row0;
      const int rows = (int)(left < 32 ? left : 32);
      const int rl = r32 < rows ? r32 : rows - 1;
      p_in0 = (in0 + row0)[rl];
#pragma unroll
      for (int n = 0; n < NO; ++n) {
        p_act[n] = (in_act[n] + row0 * astr)[rl * astr];
        p_olp[n] = (in_olp[n] + row0 * astr)[rl * astr];
      }
    };
    // x tile of strip it + 1 is written in the middle of strip it; tile (it + 1) % NXB was last read by B in strip it - 3
    auto put_x = [&](int slot) {
      float *sXw = sXp + (slot % NXB) * XT;
#pragma unroll
      for (int j = 0; j < XPF; ++j) sXw[xoff[j]] = xr[j];
    };
    fetch_x(strip);
    fetch_p(strip);
    put_x(0);
    pair_post(pub + 1, 1, lane);
    fetch_x(strip + sstride);
    for (int it = 0; strip < nstrips; strip += sstride, ++it) {
      const int64_t row0 = strip << 5;
      const bool valid = row0 + r32 < B;
      float *sX = sXp + (it % NXB) * XT;      // written in the middle of the previous strip (put_x)
      const float ret = p_in0, ad = p_in0;
      float act[NO], olp[NO];
#pragma unroll
      for (int n = 0; n < NO; ++n) { act[n] = p_act[n]; olp[n] = p_olp[n]; }
      fetch_p(strip + sstride);
      wave_lds_fence();

      f32x16 h1[2], h2[2];
      float out[NO];
      load_feat_vec(h1, sB, half);
      bx_layer0<LDW0, false>(h1, sX, sW0, half, r32);
      relu2(h1);
      load_feat_vec(h2, sB + 64, half);
      bx_product(h2, h1, sW1p, lane);
      relu2(h2);
      pair_wait(pub + 3, it - 2);             // (never blocks: see put_x)
      put_x(it + 1);
      pair_post(pub + 1, it + 2, lane);
      fetch_x(strip + 2 * sstride);
      fwd_last<NO>(out, h2, sWL, sB + 128, half);

      float gout[NO];
      loss_grad<NO>(P, net, A, valid, half, lc, out, ret, ad, act, olp, den, var, logsig, gout, s_main, s_ent, s_dls);
#pragma unroll
      for (int n = 0; n < NO; ++n)
        if (half == 0) dbL[n] += gout[n];

      // last layer backward (VALU): dWL partials per lane; g = (WL^T gout) * (h2 > 0)
      f32x16 g[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x2 t0 = {0.f, 0.f}, t1 = {0.f, 0.f};      // (two features per v_pk_fma_f32)
#pragma unroll
          for (int n = 0; n < NO; ++n) {
            const float4 w4 = *(const float4 *)(sWL + n * 64 + kb * 32 + 8 * q + 4 * half);
            t0 = __builtin_elementwise_fma(f32x2{w4.x, w4.y}, f32x2{gout[n], gout[n]}, t0);
            t1 = __builtin_elementwise_fma(f32x2{w4.z, w4.w}, f32x2{gout[n], gout[n]}, t1);
          }
          const float t[4] = {t0.x, t0.y, t1.x, t1.y};
#pragma unroll
          for (int i = 0; i < 4; ++i) g[kb][4 * q + i] = h2[kb][4 * q + i] > 0.f ? t[i] : 0.f;
        }
      // publish g2 (B has read the previous strip's at the start of its own previous strip: never blocks)
      pair_wait(pub + 2, it);
      put_tile(sG2, g, half, r32);
      pair_post(pub + 0, it + 1, lane);

      // off the critical path
#pragma unroll
      for (int n = 0; n < NO; ++n)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) dWL[n][kb][r] = fmaf(gout[n], h2[kb][r], dWL[n][kb][r]);
      vdb1 += row_sum32(sG2 + lane * LDT);     // db1[lane]: the row of the tile this wave has just written
    }
    __syncthreads();      // (1) every pair is through its strips: the pair regions are free
    float *tr = mine + 64 * 64;
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      put_tile(tr, dWL[n], half, r32);
      wave_lds_fence();
      float sum = 0.f;
      for (int r = 0; r < 32; ++r) sum += tr[lane * LDT + r];
      vec_wl[n] = sum;
      wave_lds_fence();
    }
    sc[0] = trl_wave_sum(s_main);
    sc[1] = trl_wave_sum(s_ent);
#pragma unroll
    for (int n = 0; n < NO; ++n) {
      sc[2 + n] = trl_wave_sum(s_dls[n]);
      sc[2 + NO + n] = trl_wave_sum(dbL[n]);
    }
  } else {
    // =================================================================================== wave B: the tiles' readers
    f32x16 accW1[2][2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW1[nb][kb][r] = 0.f;
    const float *ga = sG2 + r32 * LDT + 4 * half;
    const float bias0 = sB[r32], bias1 = sB[32 + r32];
    for (int it = 0; strip < nstrips; strip += sstride, ++it) {
      const float *sX = sXp + (it % NXB) * XT;
      pair_wait(pub + 1, it + 1);
      // h1 of the strip, transposed: lane = feature kb 32 + r32, register r = sample feat(r, half) -- the B operand of
      // dW1 += g2 h1^T as three bf16 parts per k step (t = samples' registers 8 t .. 8 t + 7)
      u32x4 Xp[2][2][3];
      {
        f32x16 xt[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { xt[0][r] = bias0; xt[1][r] = bias1; }
        bx_layer0<LDW0, true>(xt, sX, sW0, half, r32);
        relu2(xt);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int t = 0; t < 2; ++t) bx_split8(xt[kb], t, Xp[kb][t]);
      }
      // g2: rows (the A operand of dW1; a lane's 16 B reads are samples 8 j + 4 half + i = feat(4 j + i, half)) and in the
      // accumulator layout (the B operand of the input-gradient product); then the tile goes back to A
      pair_wait(pub + 0, it + 1);
      u32x4 Gp[2][2][3];
      f32x16 gB[2];
      {
        f32x16 gr[2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v0 = ld4(ga + 8 * j), v1 = ld4(ga + 32 * LDT + 8 * j);
          gr[0][4 * j] = v0.x; gr[0][4 * j + 1] = v0.y; gr[0][4 * j + 2] = v0.z; gr[0][4 * j + 3] = v0.w;
          gr[1][4 * j] = v1.x; gr[1][4 * j + 1] = v1.y; gr[1][4 * j + 2] = v1.z; gr[1][4 * j + 3] = v1.w;
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) gB[kb][r] = sG2[(kb * 32 + feat(r, half)) * LDT + r32];
        pair_post(pub + 2, it + 1, lane);         // (LDS keeps issue order: the store follows the reads above)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int t = 0; t < 2; ++t) bx_split8(gr[nb], t, Gp[nb][t]);
      }
      // dW1[n][k] += sum over samples: 2 x 2 blocks x 2 steps x 6 partial products, the four accumulators in rotation
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#define BX_DW1(pa, pb)                                                          \
  accW1[0][0] = MFMA_BF(Gp[0][t][pa], Xp[0][t][pb], accW1[0][0]);               \
  accW1[0][1] = MFMA_BF(Gp[0][t][pa], Xp[1][t][pb], accW1[0][1]);               \
  accW1[1][0] = MFMA_BF(Gp[1][t][pa], Xp[0][t][pb], accW1[1][0]);               \
  accW1[1][1] = MFMA_BF(Gp[1][t][pa], Xp[1][t][pb], accW1[1][1]);
        BX_DW1(2, 0) BX_DW1(0, 2) BX_DW1(1, 1) BX_DW1(1, 0) BX_DW1(0, 1) BX_DW1(0, 0)
#undef BX_DW1
      }
      // layer 1 backward, input side: t^T = W1^T g2^T, then the mask: g1 = t * (h1 > 0), h1 recomputed in the
      // accumulator layout (the same four MFMAs wave A ran)
      f32x16 t[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[kb][r] = 0.f;
      bx_product(t, gB, sW1Tp, lane);
      {
        f32x16 hm[2];      // (pre-activations: relu(z) > 0 is z > 0)
        load_feat_vec(hm, sB, half);
        bx_layer0<LDW0, false>(hm, sX, sW0, half, r32);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) t[kb][r] = hm[kb][r] > 0.f ? t[kb][r] : 0.f;
      }
      put_tile(sG1, t, half, r32);
      wave_lds_fence();
      // layer 0: dW0[n][k] += sum_row g1[n][row] x[k][row] as 32 one-step products of v_mfma_f32_4x4x1
      {
        const float *grow = sG1 + lane * LDT, *xrow = sXp + (it % NXB) * XT + (lane & 3) * LDT;
        float4 gv[8], xv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { gv[i] = ld4(grow + 4 * i); xv[i] = ld4(xrow + 4 * i); }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) dw0m = __builtin_amdgcn_mfma_f32_4x4x1f32(f4(gv[i], c), f4(xv[i], c), dw0m, 0, 0, 0);
        if (K0 == 4) {
          float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) { s0 += gv[i].x; s1 += gv[i].y; s2 += gv[i].z; s3 += gv[i].w; }
          vdb0 += (s0 + s1) + (s2 + s3);
        }
      }
      pair_post(pub + 3, it + 1, lane);
    }
    __syncthreads();      // (1)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[(nb * 32 + feat(r, half)) * 64 + kb * 32 + r32] = accW1[nb][kb][r];
  }

  // =========================================================================== partials out (as k_ac_learn_pair)
  __syncthreads();        // (2)
  float *part = P.ws + ((int64_t)net * gridDim.x + blockIdx.x) * P.pstride;
  {
    const int N = a.dims[2], K = H1, off = a.poff[1];
    for (int e = tid; e < N * K; e += NT) {
      const int i = (e / K) * 64 + (e % K);
      part[off + e] = ((sT[i] + sT[PT + i]) + sT[2 * PT + i]) + sT[3 * PT + i];
    }
  }
  __syncthreads();
  constexpr int VS = 12;     // slots: db0, db1, NO last-layer rows, 4 dW0 columns
  float *vsum = sT, *scal = sT + 4 * VS * 64;
  if (roleA) {
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < 2 + 2 * NO; ++j) scal[pair * 16 + j] = sc[j];
    }
#pragma unroll
    for (int n = 0; n < NO; ++n) vsum[(pair * VS + 2 + n) * 64 + lane] = vec_wl[n];
    vsum[(pair * VS + 1) * 64 + lane] = vdb1;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) vsum[(pair * VS + 2 + NO + (lane & 3)) * 64 + 4 * (lane >> 2) + i] = dw0m[i];
    if (K0 == 4) {
      vsum[(pair * VS + 0) * 64 + lane] = vdb0;
    } else if ((lane & 3) == 3) {          // column 3 of the products: the ones row
#pragma unroll
      for (int i = 0; i < 4; ++i) vsum[(pair * VS + 0) * 64 + 4 * (lane >> 2) + i] = dw0m[i];
    }
  }
  __syncthreads();
  for (int sl = tid >> 6; sl < 2 + NO + 4; sl += 8) {
    const int f = tid & 63;
    const float v = ((vsum[(0 * VS + sl) * 64 + f] + vsum[(1 * VS + sl) * 64 + f]) + vsum[(2 * VS + sl) * 64 + f]) +
                    vsum[(3 * VS + sl) * 64 + f];
    if (sl >= 2 + NO) {
      const int k = sl - 2 - NO;
      if (f < H1 && k < K0) part[a.poff[0] + f * K0 + k] = v;
    } else if (sl == 0) {
      if (f < H1) part[a.poff[0] + H1 * K0 + f] = v;
    } else if (sl == 1) {
      if (f < a.dims[2]) part[a.poff[1] + a.dims[2] * H1 + f] = v;
    } else {
      const int n = sl - 2;
      if (f < HL && n < NOr) part[a.poff[2] + n * HL + f] = v;
    }
  }
  if (tid < 2 + 2 * NO) {
    const float v = ((scal[0 * 16 + tid] + scal[1 * 16 + tid]) + scal[2 * 16 + tid]) + scal[3 * 16 + tid];
    if (tid < 2 + NO) {                       // s_main, s_ent, dls[n]
      if (tid < 2 + A) part[a.nparams + tid] = v;
    } else {                                  // dbL[n]
      const int n = tid - 2 - NO;
      if (n < NOr) part[a.poff[2] + NOr * HL + n] = v;
    }
  }
}

// y[64][n] = w[64][64] x[64][n] through bx_stage_w1 + bx_product, i.e. exactly the operand split, partial products and
// accumulation order of the learner's bf16x6 products (a test surface: tests/test_gpu_learn_bx6.py bounds its error
// against float64 on cancelling inputs).  One wave per 32 columns.
__global__ __launch_bounds__(512) void k_bx6_probe(const float *w, const float *x, float *y, int64_t n) {
  constexpr int NPOS = 2 * 4 * 64 * 4;
  __shared__ __align__(16) unsigned int sP[6 * NPOS];
  NetArgs a{};
  a.dims[1] = 64; a.dims[2] = 64; a.w[1] = w;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, r32 = lane & 31;
  bx_stage_w1(a, sP, sP + 3 * NPOS, tid);
  __syncthreads();
  for (int64_t c0 = ((int64_t)blockIdx.x * 8 + wave) * 32; c0 < n; c0 += (int64_t)gridDim.x * 8 * 32) {
    const int64_t col = c0 + r32 < n ? c0 + r32 : n - 1;
    f32x16 b[2], out[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) { b[kb][r] = x[(int64_t)(kb * 32 + feat(r, half)) * n + col]; out[kb][r] = 0.f; }
    bx_product(out, b, sP, lane);
    if (c0 + r32 < n) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[(int64_t)(nb * 32 + feat(r, half)) * n + col] = out[nb][r];
    }
  }
}

struct InferArgs {
  NetArgs net[2];
  int32_t which[2];     // blockIdx.y -> network index
  const float *obs, *action, *log_std;
  int64_t B;
  int32_t A;
  float *value, *mean, *logp;
};

// (amdgpu_waves_per_eu(2): a register budget of 256, so that the MFMA results are taken in VGPRs -- see k_ac_learn_wide)
// BX (two hidden layers): layer 1 on the bf16 MFMA with exactly split operands -- the forward of k_ac_learn_pair_bx, the same
// staging, split and accumulation order (trl_ppo_forward_fused_bx6)
template <int D, int KB0, int NO, bool XP, bool BX = false>     // (hidden layers of at most 64 units; a wider one is k_ac_infer_wide's)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_ac_infer(InferArgs P) {
  constexpr int LDW0 = 32 * KB0 + 1;
  constexpr int XT = 32 * KB0 * LDT + 4;
  static_assert(!BX || D == 3, "the bf16x6 product is layer 1's");
  __shared__ __align__(16) float sW0[64 * LDW0];
  __shared__ __align__(16) float sW1[D == 3 ? (BX ? 3 * 2 * 4 * 64 * 4 : 64 * LDW) : 4];
  __shared__ __align__(16) float sWL[NO * 64];
  __shared__ __align__(16) float sB[64 + 64 + 4];
  __shared__ __align__(16) float sT[4 * XT];
  const int net = P.which[blockIdx.y];
  const NetArgs &a = P.net[net];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, r32 = lane & 31;
  const int K0 = a.dims[0];
  const int64_t B = P.B;
  const int A = P.A;
  if (BX) {      // (stage_weights<2> leaves out W1 and takes layer 1 for the last one: redo the last layer and the biases)
    stage_weights<2, LDW0, NO, false>(a, sW0, nullptr, nullptr, sWL, sB, tid);
    __syncthreads();
    const int HL = a.dims[2], NOr = a.dims[3];
    for (int e = tid; e < NO * 64; e += 256) {
      const int n = e >> 6, k = e & 63;
      sWL[e] = (n < NOr && k < HL) ? a.w[2][n * HL + k] : 0.f;
    }
    for (int e = tid; e < 64 + 4; e += 256) {
      if (e < 64) sB[64 + e] = e < HL ? a.b[1][e] : 0.f;
      else sB[128 + e - 64] = e - 64 < NOr ? a.b[2][e - 64] : 0.f;
    }
    bx_stage_w1<256, false>(a, (unsigned int *)sW1, nullptr, tid);
  } else {
    stage_weights<D, LDW0, NO, false>(a, sW0, sW1, nullptr, sWL, sB, tid);
  }
  float *sX = sT + wave * XT;
  for (int e = lane; e < XT; e += 64) sX[e] = 0.f;
  __syncthreads();

  float var2[NO], logsig[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) {
    var2[n] = 2.f; logsig[n] = 0.f;
    if (net == 1 && P.logp && n < A) {
      const float sigma = expf(P.log_std[n]);
      var2[n] = 2.0f * (sigma * sigma);
      logsig[n] = logf(sigma);
    }
  }
  const int q64 = 64 / K0, m64 = 64 % K0;
  const int ksteps0 = (K0 + 1) >> 1;
  constexpr int XPF = 6;          // branch-free one-strip-ahead fetch, as in k_ac_learn
  const int nelem = 32 * K0;
  int xoff[XPF];
  {
    int rr = lane / K0, kk = lane % K0;
#pragma unroll
    for (int j = 0; j < XPF; ++j) {
      xoff[j] = lane + 64 * j < nelem ? kk * LDT + rr : XT - 4;
      rr += q64; kk += m64;
      if (kk >= K0) { kk -= K0; ++rr; }
    }
  }
  const bool want_lp = net == 1 && P.logp != nullptr;
  const float *in_act[NO];
  const int astr = want_lp ? A : 0;
#pragma unroll
  for (int n = 0; n < NO; ++n) in_act[n] = want_lp ? P.action + (n < A ? n : 0) : P.obs;
  float xr[XPF], p_act[NO];
#pragma unroll
  for (int j = 0; j < XPF; ++j) xr[j] = 0.f;
  const int64_t nstrips = (B + 31) >> 5;
  auto fetch = [&](int64_t strip) {
    strip = strip < nstrips ? strip : nstrips - 1;
    const int64_t row0 = strip << 5;
    const int64_t row = row0 + r32;
    const int64_t rowc = row < B ? row : B - 1;
    if (XP) {
      const float *base = P.obs + row0 * K0;
      const int64_t avail = (B - row0) * K0;
      const int last = (int)(avail < nelem ? avail : nelem) - 1;
#pragma unroll
      for (int j = 0; j < XPF; ++j) {
        const int e = lane + 64 * j;
        xr[j] = base[e < last ? e : last];
      }
    }
#pragma unroll
    for (int n = 0; n < NO; ++n) p_act[n] = in_act[n][rowc * astr];
  };
  const int64_t sstride = (int64_t)gridDim.x * 4;
  int64_t strip = (int64_t)blockIdx.x * 4 + wave;
  fetch(strip);
  for (; strip < nstrips; strip += sstride) {
    const int64_t row0 = strip << 5;
    const int64_t row = row0 + r32;
    const bool valid = row < B;
    if (XP) {
#pragma unroll
      for (int j = 0; j < XPF; ++j) sX[xoff[j]] = xr[j];
    } else {
      const float *base = P.obs + row0 * K0;
      const int64_t avail = (B - row0) * K0;
      int rr = lane / K0, kk = lane % K0;
      for (int e = lane; e < nelem; e += 64) {
        sX[kk * LDT + rr] = e < avail ? base[e] : 0.f;
        rr += q64; kk += m64;
        if (kk >= K0) { kk -= K0; ++rr; }
      }
    }
    float act[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) act[n] = p_act[n];
    fetch(strip + sstride);
    wave_lds_fence();
    f32x16 h1[2], h2[2];
    float out[NO];
    fwd_layer0<LDW0>(h1, sX, sW0, sB, ksteps0, half, r32);
    if (D == 3 && BX) {
      load_feat_vec(h2, sB + 64, half);
      bx_product(h2, h1, (const unsigned int *)sW1, lane);
      relu2(h2);
    } else if (D == 3) {
      fwd_layer1(h2, h1, sW1, sB + 64, half, r32);
    }
    fwd_last<NO>(out, D == 3 ? h2 : h1, sWL, sB + 128, half);
    if (valid && half == 0) {
      if (net == 0) {
        P.value[row] = out[0];
      } else {
#pragma unroll
        for (int n = 0; n < NO; ++n)
          if (n < A) {
            if (P.mean) P.mean[row * A + n] = out[n];
            if (P.logp) {
              const float d = act[n] - out[n];
              P.logp[row * A + n] = ((-(d * d)) / var2[n] - logsig[n]) - LOG_SQRT_2PI;   // Normal.log_prob
            }
          }
      }
    }
    wave_lds_fence();     // the next strip overwrites sX
  }
}

// The forward of k_ac_learn_wide alone (one wide hidden layer over a fan-in <= 4): the same operand registers, the
// same k order ([W0 | b0] [x | 1]^T from C = 0) and the same packed dot product, so a value or log-prob computed here
// is the bit pattern the learner kernel computes for that row (PPO's first-epoch ratio is exactly 1; A2C's detached
// advantage uses the learner's own V).  Small LDS footprint and <= 128 registers: several workgroups per CU.
template <int NO, int HB, int KS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void k_ac_infer_wide(InferArgs P) {
  constexpr int NW = 4, NT = 64 * NW;
  constexpr int XR = 2 * KS, XT = XR * LDT + 4;
  __shared__ __align__(16) float sWL[HB * NO * 64];
  __shared__ __align__(16) float sBL[4];
  __shared__ __align__(16) float sT[NW * XT];
  const int net = P.which[blockIdx.y];
  const NetArgs &a = P.net[net];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, half = lane >> 5,
            r32 = lane & 31;
  const int K0 = a.dims[0], H1 = a.dims[1], NOr = a.dims[2];
  const int64_t B = P.B;
  const int A = P.A;
  for (int e = tid; e < HB * NO * 64; e += NT) {
    const int q = e / (NO * 64), r = e - q * NO * 64, n = r >> 6, k = q * 64 + (r & 63);
    sWL[e] = (n < NOr && k < H1) ? a.w[1][n * H1 + k] : 0.f;
  }
  if (tid < 4) sBL[tid] = tid < NOr ? a.b[1][tid] : 0.f;
  float *sX = sT + wave * XT;
  for (int e = lane; e < XT; e += 64) sX[e] = 0.f;
  if (lane < 32) sX[K0 * LDT + lane] = 1.f;
  float wop[HB][2][KS];
#pragma unroll
  for (int q = 0; q < HB; ++q)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int f = q * 64 + nb * 32 + r32;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k = 2 * s + half;
        float v = 0.f;
        if (f < H1 && k < K0) v = a.w[0][f * K0 + k];
        if (f < H1 && k == K0) v = a.b[0][f];
        wop[q][nb][s] = v;
      }
    }
  __syncthreads();

  float var2[NO], logsig[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n) {
    var2[n] = 2.f; logsig[n] = 0.f;
    if (net == 1 && P.logp && n < A) {
      const float sigma = expf(P.log_std[n]);
      var2[n] = 2.0f * (sigma * sigma);
      logsig[n] = logf(sigma);
    }
  }
  constexpr int XPF = KS;
  const int nelem = 32 * K0;
  int xoff[XPF];
#pragma unroll
  for (int j = 0; j < XPF; ++j) {
    const int e = lane + 64 * j;
    xoff[j] = e < nelem ? (e % K0) * LDT + e / K0 : XT - 4;
  }
  const bool want_lp = net == 1 && P.logp != nullptr;
  const float *in_act[NO];
  const int astr = want_lp ? A : 0;
#pragma unroll
  for (int n = 0; n < NO; ++n) in_act[n] = want_lp ? P.action + (n < A ? n : 0) : P.obs;
  float xr[XPF], p_act[NO];
  const int64_t nstrips = (B + 31) >> 5;
  auto fetch = [&](int64_t strip) {
    strip = strip < nstrips ? strip : nstrips - 1;
    const int64_t row0 = strip << 5;
    const int64_t row = row0 + r32;
    const int64_t rowc = row < B ? row : B - 1;
    const float *base = P.obs + row0 * K0;
    const int64_t avail = (B - row0) * K0;
    const int last = (int)(avail < nelem ? avail : nelem) - 1;
#pragma unroll
    for (int j = 0; j < XPF; ++j) {
      const int e = lane + 64 * j;
      xr[j] = base[e < last ? e : last];
    }
#pragma unroll
    for (int n = 0; n < NO; ++n) p_act[n] = in_act[n][rowc * astr];
  };
  const int64_t sstride = (int64_t)gridDim.x * NW;
  int64_t strip = (int64_t)blockIdx.x * NW + wave;
  fetch(strip);
  for (; strip < nstrips; strip += sstride) {
    const int64_t row = (strip << 5) + r32;
    const bool valid = row < B;
#pragma unroll
    for (int j = 0; j < XPF; ++j) sX[xoff[j]] = xr[j];
    float act[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) act[n] = p_act[n];
    fetch(strip + sstride);
    wave_lds_fence();
    float xop[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) xop[s] = sX[(2 * s + half) * LDT + r32];
    int wl_off = 0;      // (as in k_ac_learn_wide: keeps the last layer's rows out of registers)
    asm volatile("" : "+v"(wl_off));
    f32x2 pacc[NO];
#pragma unroll
    for (int n = 0; n < NO; ++n) pacc[n] = f32x2{0.f, 0.f};
#pragma unroll
    for (int q = 0; q < HB; ++q) {
      f32x16 h[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[nb][r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        h[0] = MFMA(wop[q][0][s], xop[s], h[0]);
        h[1] = MFMA(wop[q][1][s], xop[s], h[1]);
      }
      relu2(h);
      fwd_last_acc<NO>(pacc, h, sWL + wl_off + q * NO * 64, half);
      __builtin_amdgcn_sched_barrier(0);
    }
    float out[NO];
    fwd_last_finish<NO>(out, pacc, sBL);
    if (valid && half == 0) {
      if (net == 0) {
        P.value[row] = out[0];
      } else {
#pragma unroll
        for (int n = 0; n < NO; ++n)
          if (n < A) {
            if (P.mean) P.mean[row * A + n] = out[n];
            if (P.logp) {
              const float d = act[n] - out[n];
              P.logp[row * A + n] = ((-(d * d)) / var2[n] - logsig[n]) - LOG_SQRT_2PI;   // Normal.log_prob
            }
          }
      }
    }
    wave_lds_fence();     // the next strip overwrites sX
  }
}

// ---- second launch: sums the partial sets of every workgroup in a fixed order and writes
// each parameter's gradient to its destination; one extra block finishes the scalars.
struct GradDst {
  int32_t nseg;
  int32_t start[7];
  float *dst[6];
};

struct ReduceArgs {
  const float *ws;
  int32_t nwg, pstride;
  int32_t nparams[2];
  GradDst gd[2];
  float *losses;      // [4] actor, critic, entropy, total
  float *dlog_std;    // [A]
  int64_t B;
  int32_t A;
  int32_t kind;
  float alpha, beta, dls_ent;
  // optional update stage: param / moments of gradient element g are at (g - grad) in these
  int32_t do_adam;
  float *p, *g, *m, *v;
  TrlAdamScalars adam;
};

__global__ __launch_bounds__(256) void k_learn_reduce(ReduceArgs R, int param_blocks) {
  __shared__ float partS[16][17];
  const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
  if ((int)blockIdx.x < param_blocks) {
    const int net = blockIdx.y;
    const int np = R.nparams[net];
    const float *ws = R.ws + (int64_t)net * R.nwg * R.pstride;
    const int i = blockIdx.x * 16 + e;
    float s = 0.f;
    if (i < np)
      for (int w = grp; w < R.nwg; w += 16) s += ws[(int64_t)w * R.pstride + i];
    partS[grp][e] = s;
    __syncthreads();
    if (grp == 0 && i < np) {
      float t = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < 16; ++g2) t += partS[g2][e];
      const GradDst &gd = R.gd[net];
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (j < gd.nseg && i >= gd.start[j] && i < gd.start[j + 1]) {
          float *gp = gd.dst[j] + (i - gd.start[j]);
          *gp = t;
          if (R.do_adam) {
            const int64_t o = gp - R.g;
            trl_adam_elem(R.p + o, gp, R.m + o, R.v + o, R.adam);
          }
        }
    }
    return;
  }
  if (blockIdx.y != 0) return;
  // scalars: slot e < 8: [0] critic sum; [1] actor min-sum; [2] entropy sum; [3 + a] dlog_std[a]
  float s = 0.f;
  if (e < 3 + R.A && e < 8) {
    const int net = e == 0 ? 0 : 1;
    const int j = e == 0 ? 0 : e - 1;
    const float *ws = R.ws + (int64_t)net * R.nwg * R.pstride + R.nparams[net] + j;
    for (int w = grp; w < R.nwg; w += 16) s += ws[(int64_t)w * R.pstride];
  }
  partS[grp][e] = s;
  __syncthreads();
  if (grp == 0) {
    float t = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 16; ++g2) t += partS[g2][e];
    partS[0][e] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // PPO: the surrogate and the entropy are means over [B, A]; A2C: means over the batch
    const float invBA = 1.0f / (float)(R.kind == KIND_A2C ? R.B : R.B * R.A);
    const float actor = -(partS[0][1] * invBA);
    const float critic = partS[0][0] * (1.0f / (float)R.B);
    const float ent = partS[0][2] * invBA;
    R.losses[0] = actor;
    R.losses[1] = critic;
    R.losses[2] = ent;
    R.losses[3] = (actor + R.alpha * critic) - R.beta * ent;
  }
  if (R.dlog_std && (int)threadIdx.x < R.A && threadIdx.x < 5) {
    float *gp = R.dlog_std + threadIdx.x;
    *gp = partS[0][3 + threadIdx.x] + R.dls_ent;
    if (R.do_adam) {
      const int64_t o = gp - R.g;
      trl_adam_elem(R.p + o, gp, R.m + o, R.v + o, R.adam);
    }
  }
}

constexpr int WIDE_FANIN_MAX = 9;      // k_ac_learn_wide / k_ac_infer_wide: K0 + 1 <= 2 KS, KS <= 5
inline int wide_ks(int K0) { return K0 <= 3 ? 2 : (K0 <= 5 ? 3 : (K0 <= 7 ? 4 : 5)); }
inline int wide_class(int K0) { return K0 <= 3 ? 0 : (K0 == 4 ? 1 : (K0 == 5 ? 2 : (K0 <= 7 ? 3 : (K0 == 8 ? 4 : 5)))); }

int fill_net(const char *fn, const trl_mlp_desc *d, int out_max, NetArgs *a, int hidden_max = 64) {
  TRL_REQUIRE(d != nullptr, "%s: null network description", fn);
  TRL_REQUIRE(d->depth == 2 || d->depth == 3, "%s: depth %d not in {2, 3}", fn, d->depth);
  a->depth = d->depth;
  int off = 0;
  for (int l = 0; l <= d->depth; ++l) {
    // one hidden layer over a fan-in of at most WIDE_FANIN_MAX: up to 256 units, walked as 64-wide blocks (HB)
    const int wide = (d->depth == 2 && l == 1 && hidden_max > 64 && d->dims[0] <= WIDE_FANIN_MAX) ? hidden_max : 64;
    const int lim = l == d->depth ? out_max : wide;
    TRL_REQUIRE(d->dims[l] >= 1 && d->dims[l] <= lim, "%s: dims[%d]=%d must be in [1, %d]", fn, l, d->dims[l], lim);
    a->dims[l] = d->dims[l];
  }
  for (int l = 0; l < d->depth; ++l) {
    TRL_REQUIRE(d->w[l] && d->b[l], "%s: null weight/bias of layer %d", fn, l);
    a->w[l] = d->w[l];
    a->b[l] = d->b[l];
    a->poff[l] = off;
    off += d->dims[l + 1] * d->dims[l] + d->dims[l + 1];
  }
  a->nparams = off;
  return TRL_OK;
}

int fill_dst(const char *fn, const NetArgs &a, float *const *pg, GradDst *gd) {
  gd->nseg = 2 * a.depth;
  int off = 0;
  for (int l = 0; l < a.depth; ++l) {
    TRL_REQUIRE(pg[2 * l] && pg[2 * l + 1], "%s: null gradient destination of layer %d", fn, l);
    gd->start[2 * l] = off;
    gd->dst[2 * l] = pg[2 * l];
    off += a.dims[l + 1] * a.dims[l];
    gd->start[2 * l + 1] = off;
    gd->dst[2 * l + 1] = pg[2 * l + 1];
    off += a.dims[l + 1];
  }
  gd->start[gd->nseg] = off;
  return TRL_OK;
}

inline int learn_wgs(int64_t B) {
  // one workgroup per CU over both networks: 128 per network, fewer when the batch is small
  const int64_t need = trl_cdiv(trl_cdiv(B, 32), 4);
  return (int)(need < 128 ? need : 128);
}

inline int pstride_of(int np0, int np1) {
  const int m = (np0 > np1 ? np0 : np1) + NEXTRA;
  return (m + 3) & ~3;
}

}  // namespace

extern "C" int64_t trl_ppo_learn_workspace_bytes(const trl_mlp_desc *critic, const trl_mlp_desc *actor, int64_t B) {
  if (!critic || !actor || B <= 0) return 0;
  int np[2] = {0, 0};
  const trl_mlp_desc *d[2] = {critic, actor};
  for (int g = 0; g < 2; ++g) {
    if (d[g]->depth < 1 || d[g]->depth > 3) return 0;
    for (int l = 0; l < d[g]->depth; ++l) np[g] += d[g]->dims[l + 1] * d[g]->dims[l] + d[g]->dims[l + 1];
  }
  return 2ll * MAX_WG * pstride_of(np[0], np[1]) * 4;
}

namespace {
int learn_fused(const char *fn, int kind, bool bx6, const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std,
                const float *obs, const float *action, const float *returns, const float *old_logp, const float *adv,
                int64_t B, int32_t A, double clip, double alpha, double beta, float *losses_out,
                float *const *critic_grads, float *const *actor_grads, float *dlog_std, const trl_adam_args *adam,
                void *ws, void *stream) {
  TRL_REQUIRE(obs && action && returns && adv && losses_out && critic_grads && actor_grads && ws, "%s: null pointer",
              fn);
  TRL_REQUIRE(kind == KIND_A2C || (log_std && old_logp && dlog_std), "%s: null pointer", fn);
  TRL_REQUIRE(B > 0 && B < (1ll << 31), "%s: B=%lld out of range", fn, (long long)B);
  TRL_REQUIRE(A >= 1 && A <= 4, "%s: action size %d not in [1, 4]", fn, A);
  TRL_REQUIRE(critic && actor, "%s: null network description", fn);
  TRL_REQUIRE(A <= 2 || actor->depth == 2, "%s: two hidden layers take an action size of at most 2 (got %d)", fn, A);
  LearnArgs P{};
  int rc = fill_net(fn, critic, 1, &P.net[0], 256);
  if (rc) return rc;
  rc = fill_net(fn, actor, A, &P.net[1], 256);
  if (rc) return rc;
  TRL_REQUIRE(P.net[1].dims[actor->depth] == A, "%s: the actor's fan-out %d is not the action size %d", fn,
              P.net[1].dims[actor->depth], A);
  TRL_REQUIRE(critic->depth == actor->depth && critic->dims[0] == actor->dims[0],
              "%s: critic and actor must have the same depth and observation size", fn);
  P.obs = obs; P.action = action; P.returns = returns; P.old_logp = old_logp; P.adv = adv; P.log_std = log_std;
  P.B = B; P.A = A; P.kind = kind;
  P.lo = (float)(1.0 - clip); P.hi = (float)(1.0 + clip); P.alpha = (float)alpha; P.beta = (float)beta;
  P.ws = (float *)ws;
  P.pstride = pstride_of(P.net[0].nparams, P.net[1].nparams);
  // two hidden layers over a fan-in <= 4 and enough strips for 2 x 1024 waves: the eight-wave variant
  const int hmax = critic->dims[1] > actor->dims[1] ? critic->dims[1] : actor->dims[1];
  // two hidden layers over a fan-in <= 4 and enough strips to keep 1024 wave pairs busy: the role-split kernel
  bool pairk = critic->depth == 3 && critic->dims[0] <= 4 && A <= 2 && (bx6 || B >= 8 * 32 * 128);
  bool onewide = true;     // one hidden layer <= 64 wide over a fan-in <= 4: k_ac_learn_wide with one block
#ifdef TRL_TUNING
  if (const char *e = getenv("TRL_LEARN_ONEWIDE")) onewide = atoi(e) != 0;       // A/B runs of tools/ only
  if (const char *e = getenv("TRL_LEARN_NW")) pairk = pairk && atoi(e) != 4;     // A/B runs of tools/ only
  if (const char *e = getenv("TRL_LEARN_PRIO")) P.tune_prio = atoi(e);
#endif
  const int nwg = learn_wgs(B);
  ReduceArgs R{};
  R.ws = (const float *)ws; R.nwg = nwg; R.pstride = P.pstride;
  R.nparams[0] = P.net[0].nparams; R.nparams[1] = P.net[1].nparams;
  rc = fill_dst(fn, P.net[0], critic_grads, &R.gd[0]);
  if (rc) return rc;
  rc = fill_dst(fn, P.net[1], actor_grads, &R.gd[1]);
  if (rc) return rc;
  R.losses = losses_out; R.dlog_std = dlog_std; R.B = B; R.A = A; R.kind = kind;
  R.alpha = P.alpha; R.beta = P.beta; R.dls_ent = (float)(-beta / (double)A);
  if (adam) {
    TRL_REQUIRE(adam->dev_state == nullptr, "%s: the in-launch update keeps its step count on the host (dev_state must be NULL)", fn);
    TRL_REQUIRE(adam->param && adam->grad && adam->exp_avg && adam->exp_avg_sq && adam->n > 0 && adam->step >= 1,
                "%s: incomplete adam arguments", fn);
    const float *lo = adam->grad, *hi = adam->grad + adam->n;
    bool inside = dlog_std == nullptr || (dlog_std >= lo && dlog_std + A <= hi);
    for (int g = 0; g < 2; ++g)
      for (int j = 0; j < R.gd[g].nseg; ++j)
        inside = inside && R.gd[g].dst[j] >= lo && R.gd[g].dst[j] + (R.gd[g].start[j + 1] - R.gd[g].start[j]) <= hi;
    TRL_REQUIRE(inside, "%s: with adam, every gradient destination must lie inside adam->grad", fn);
    R.do_adam = 1;
    R.p = adam->param; R.g = adam->grad; R.m = adam->exp_avg; R.v = adam->exp_avg_sq;
    R.adam = trl_adam_scalars(adam->lr, adam->beta1, adam->beta2, adam->eps, adam->step, adam->grad_scale,
                              adam->clip_value);
  }

  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)nwg, 2), block(pairk ? 512 : 256);
  const int D = critic->depth, KB0 = critic->dims[0] <= 32 ? 1 : 2, NOv = A <= 1 ? 1 : (A <= 2 ? 2 : 4);
  const bool xp = critic->dims[0] <= 12;
#define TRL_LEARN_CASE(d, kb, no)                                                                  \
  if (D == d && KB0 == kb && NOv == no) {                                                          \
    if (kb == 1 && xp) hipLaunchKernelGGL((k_ac_learn<d, kb, no, kb == 1>), grid, block, 0, st, P); \
    else hipLaunchKernelGGL((k_ac_learn<d, kb, no, false>), grid, block, 0, st, P);               \
  }
  TRL_REQUIRE(!bx6 || pairk,
              "%s: the bf16x6 products exist for the role-split kernel only: two hidden layers <= 64 wide, at most 4 "
              "observations, at most 2 actions", fn);
  if (pairk && bx6) {
    if (NOv == 1) hipLaunchKernelGGL((k_ac_learn_pair_bx<1>), grid, block, 0, st, P);
    else hipLaunchKernelGGL((k_ac_learn_pair_bx<2>), grid, block, 0, st, P);
  } else if (pairk) {
    if (NOv == 1) hipLaunchKernelGGL((k_ac_learn_pair<1>), grid, block, 0, st, P);
    else hipLaunchKernelGGL((k_ac_learn_pair<2>), grid, block, 0, st, P);
  } else if (D == 2 && (hmax > 64 || (onewide && critic->dims[0] <= 4 && NOv <= 2))) {
    // one hidden layer: 65 .. 256 wide over a fan-in <= WIDE_FANIN_MAX (fill_net has checked it), or <= 64 wide over a
    // fan-in <= 4 -- 1, 2 or 4 blocks of 64
    TRL_REQUIRE(NOv <= 2, "%s: a hidden layer of %d units takes an action size of at most 2 (got %d)", fn, hmax, A);
    const int HBv = hmax <= 64 ? 1 : (hmax <= 128 ? 2 : 4), cls = wide_class(critic->dims[0]);
    // (four blocks, two outputs and 8 - 9 inputs: the accumulators alone leave no registers -- 184 / 328 B of scratch)
    TRL_REQUIRE(!(HBv == 4 && NOv == 2 && cls >= 4),
                "%s: a hidden layer of %d units with two actions takes at most 7 observations (got %d)", fn, hmax,
                critic->dims[0]);
#define TRL_WIDE_CASE(no, hb, c, ks, ng, bv) \
  if (NOv == no && HBv == hb && cls == c) hipLaunchKernelGGL((k_ac_learn_wide<no, hb, ks, ng, bv>), grid, block, 0, st, P);
#define TRL_WIDE_CASES(no, hb)                                                                  \
  TRL_WIDE_CASE(no, hb, 0, 2, 1, false) TRL_WIDE_CASE(no, hb, 1, 3, 1, true) TRL_WIDE_CASE(no, hb, 2, 3, 2, false) \
  TRL_WIDE_CASE(no, hb, 3, 4, 2, false) TRL_WIDE_CASE(no, hb, 4, 5, 2, true) TRL_WIDE_CASE(no, hb, 5, 5, 3, false)
    TRL_WIDE_CASE(1, 1, 0, 2, 1, false) TRL_WIDE_CASE(1, 1, 1, 3, 1, true)
    TRL_WIDE_CASE(2, 1, 0, 2, 1, false) TRL_WIDE_CASE(2, 1, 1, 3, 1, true)
    TRL_WIDE_CASES(1, 2) TRL_WIDE_CASES(1, 4) TRL_WIDE_CASES(2, 2)
    TRL_WIDE_CASE(2, 4, 0, 2, 1, false) TRL_WIDE_CASE(2, 4, 1, 3, 1, true) TRL_WIDE_CASE(2, 4, 2, 3, 2, false)
    TRL_WIDE_CASE(2, 4, 3, 4, 2, false)
#undef TRL_WIDE_CASES
#undef TRL_WIDE_CASE
  } else {
  TRL_LEARN_CASE(2, 1, 1) TRL_LEARN_CASE(2, 1, 2) TRL_LEARN_CASE(2, 1, 4)
  TRL_LEARN_CASE(2, 2, 1) TRL_LEARN_CASE(2, 2, 2) TRL_LEARN_CASE(2, 2, 4)
  TRL_LEARN_CASE(3, 1, 1) TRL_LEARN_CASE(3, 1, 2)
  TRL_LEARN_CASE(3, 2, 1) TRL_LEARN_CASE(3, 2, 2)
  }
#undef TRL_LEARN_CASE
  TRL_CHECK_LAUNCH(fn);
  const int np_max = R.nparams[0] > R.nparams[1] ? R.nparams[0] : R.nparams[1];
  const int pblocks = (int)trl_cdiv(np_max, 16);
  hipLaunchKernelGGL(k_learn_reduce, dim3((unsigned)pblocks + 1, 2), dim3(256), 0, st, R, pblocks);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}
}  // namespace

extern "C" int trl_ppo_learn_fused(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std,
                                   const float *obs, const float *action, const float *returns,
                                   const float *old_logp, const float *adv, int64_t B, int32_t A, double clip,
                                   double alpha, double beta, float *losses_out, float *const *critic_grads,
                                   float *const *actor_grads, float *dlog_std, const trl_adam_args *adam, void *ws,
                                   void *stream) {
  return learn_fused("trl_ppo_learn_fused", KIND_PPO, false, critic, actor, log_std, obs, action, returns, old_logp, adv, B, A,
                     clip, alpha, beta, losses_out, critic_grads, actor_grads, dlog_std, adam, ws, stream);
}

extern "C" int trl_a2c_learn_fused(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *obs,
                                   const float *action, const float *returns, const float *adv, int64_t B, int32_t A,
                                   double alpha, double beta, float *losses_out, float *const *critic_grads,
                                   float *const *actor_grads, const trl_adam_args *adam, void *ws, void *stream) {
  return learn_fused("trl_a2c_learn_fused", KIND_A2C, false, critic, actor, nullptr, obs, action, returns, nullptr, adv, B, A,
                     0.0, alpha, beta, losses_out, critic_grads, actor_grads, nullptr, adam, ws, stream);
}

extern "C" int trl_ppo_learn_fused_bx6(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std,
                                       const float *obs, const float *action, const float *returns,
                                       const float *old_logp, const float *adv, int64_t B, int32_t A, double clip,
                                       double alpha, double beta, float *losses_out, float *const *critic_grads,
                                       float *const *actor_grads, float *dlog_std, const trl_adam_args *adam, void *ws,
                                       void *stream) {
  return learn_fused("trl_ppo_learn_fused_bx6", KIND_PPO, true, critic, actor, log_std, obs, action, returns, old_logp, adv,
                     B, A, clip, alpha, beta, losses_out, critic_grads, actor_grads, dlog_std, adam, ws, stream);
}

extern "C" int trl_a2c_learn_fused_bx6(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *obs,
                                       const float *action, const float *returns, const float *adv, int64_t B, int32_t A,
                                       double alpha, double beta, float *losses_out, float *const *critic_grads,
                                       float *const *actor_grads, const trl_adam_args *adam, void *ws, void *stream) {
  return learn_fused("trl_a2c_learn_fused_bx6", KIND_A2C, true, critic, actor, nullptr, obs, action, returns, nullptr, adv,
                     B, A, 0.0, alpha, beta, losses_out, critic_grads, actor_grads, nullptr, adam, ws, stream);
}

extern "C" int trl_bx6_product_probe(const float *w, const float *x, float *y, int64_t n, void *stream) {
  const char *fn = "trl_bx6_product_probe";
  TRL_REQUIRE(w && x && y, "%s: null pointer", fn);
  TRL_REQUIRE(n > 0 && n < (1ll << 31), "%s: n=%lld out of range", fn, (long long)n);
  const int64_t need = trl_cdiv(n, 256);
  hipLaunchKernelGGL(k_bx6_probe, dim3((unsigned)(need < 256 ? need : 256)), dim3(512), 0, (hipStream_t)stream, w, x, y, n);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

namespace {
int forward_fused(const char *fn, bool bx6, const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std,
                  const float *obs, const float *action, int64_t B, int32_t A, float *value, float *mean, float *logp,
                  void *stream) {
  TRL_REQUIRE(obs, "%s: null obs", fn);
  TRL_REQUIRE(B > 0 && B < (1ll << 36), "%s: B=%lld out of range", fn, (long long)B);
  TRL_REQUIRE(value || mean || logp, "%s: no output requested", fn);
  const bool want_actor = mean || logp;
  TRL_REQUIRE(!value || critic, "%s: value requested without a critic", fn);
  TRL_REQUIRE(!want_actor || actor, "%s: mean / logp requested without an actor", fn);
  TRL_REQUIRE(!logp || (action && log_std), "%s: logp needs action and log_std", fn);
  InferArgs P{};
  int ny = 0, D = 0, K0 = 0;
  if (value) {
    int rc = fill_net(fn, critic, 1, &P.net[0], 256);
    if (rc) return rc;
    P.which[ny++] = 0;
    D = critic->depth; K0 = critic->dims[0];
  }
  if (want_actor) {
    TRL_REQUIRE(A >= 1 && A <= 4, "%s: action size %d not in [1, 4]", fn, A);
    int rc = fill_net(fn, actor, A, &P.net[1], 256);
    if (rc) return rc;
    TRL_REQUIRE(P.net[1].dims[actor->depth] == A, "%s: the actor's fan-out is not the action size %d", fn, A);
    TRL_REQUIRE(!value || (actor->depth == D && actor->dims[0] == K0),
                "%s: critic and actor must have the same depth and observation size", fn);
    P.which[ny++] = 1;
    D = actor->depth; K0 = actor->dims[0];
  }
  P.obs = obs; P.action = action; P.log_std = log_std; P.B = B; P.A = want_actor ? A : 1;
  P.value = value; P.mean = mean; P.logp = logp;
  int hmax = 0;
  for (int i = 0; i < ny; ++i) hmax = P.net[P.which[i]].dims[1] > hmax ? P.net[P.which[i]].dims[1] : hmax;
  // one hidden layer the learner takes with k_ac_learn_wide: its own forward (65 .. 256 wide; at most 64 wide
  // only when the learner takes that kernel too, i.e. at most two actions -- A is read for this even when only the value
  // is asked for)
  const bool wide = D == 2 && (hmax > 64 || (K0 <= 4 && (A < 1 || A <= 2)));
  const int64_t need = trl_cdiv(trl_cdiv(B, 32), 4);
  const int64_t cap = (wide ? 1024 : 768) / ny;      // ~3 workgroups per CU over the launch (4 of the wide kernel's)
  const dim3 grid((unsigned)(need < cap ? need : cap), (unsigned)ny), block(256);
  const int KB0 = K0 <= 32 ? 1 : 2, NOv = (!want_actor || A <= 1) ? 1 : (A <= 2 ? 2 : 4);
  hipStream_t st = (hipStream_t)stream;
  const bool xp = K0 <= 12;
#define TRL_INFER_CASE(d, kb, no)                                                                  \
  if (D == d && KB0 == kb && NOv == no) {                                                          \
    if (kb == 1 && xp) hipLaunchKernelGGL((k_ac_infer<d, kb, no, kb == 1>), grid, block, 0, st, P); \
    else hipLaunchKernelGGL((k_ac_infer<d, kb, no, false>), grid, block, 0, st, P);               \
  }
  if (bx6) {
    TRL_REQUIRE(D == 3 && K0 <= 4 && NOv <= 2,
                "%s: the bf16x6 product exists for two hidden layers <= 64 wide over at most 4 observations and at most 2 actions", fn);
    if (NOv == 1) hipLaunchKernelGGL((k_ac_infer<3, 1, 1, true, true>), grid, block, 0, st, P);
    else hipLaunchKernelGGL((k_ac_infer<3, 1, 2, true, true>), grid, block, 0, st, P);
  } else if (wide) {     // wide single hidden layer (fill_net has checked fan-in <= 4): blocks of 64
    const int HBv = hmax <= 64 ? 1 : (hmax <= 128 ? 2 : 4), KSv = wide_ks(K0);
#define TRL_WIDE_CASE(no, hb, ks) \
  if (NOv == no && HBv == hb && KSv == ks) hipLaunchKernelGGL((k_ac_infer_wide<no, hb, ks>), grid, block, 0, st, P);
#define TRL_WIDE_CASES(no)                                                                              \
  TRL_WIDE_CASE(no, 2, 2) TRL_WIDE_CASE(no, 2, 3) TRL_WIDE_CASE(no, 2, 4) TRL_WIDE_CASE(no, 2, 5)         \
  TRL_WIDE_CASE(no, 4, 2) TRL_WIDE_CASE(no, 4, 3) TRL_WIDE_CASE(no, 4, 4) TRL_WIDE_CASE(no, 4, 5)
    TRL_WIDE_CASE(1, 1, 2) TRL_WIDE_CASE(1, 1, 3) TRL_WIDE_CASE(2, 1, 2) TRL_WIDE_CASE(2, 1, 3)
    TRL_WIDE_CASES(1) TRL_WIDE_CASES(2) TRL_WIDE_CASES(4)
#undef TRL_WIDE_CASES
#undef TRL_WIDE_CASE
  } else {
  TRL_INFER_CASE(2, 1, 1) TRL_INFER_CASE(2, 1, 2) TRL_INFER_CASE(2, 1, 4)
  TRL_INFER_CASE(2, 2, 1) TRL_INFER_CASE(2, 2, 2) TRL_INFER_CASE(2, 2, 4)
  TRL_INFER_CASE(3, 1, 1) TRL_INFER_CASE(3, 1, 2) TRL_INFER_CASE(3, 1, 4)
  TRL_INFER_CASE(3, 2, 1) TRL_INFER_CASE(3, 2, 2) TRL_INFER_CASE(3, 2, 4)
  }
#undef TRL_INFER_CASE
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}
}  // namespace

extern "C" int trl_ppo_forward_fused(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std,
                                     const float *obs, const float *action, int64_t B, int32_t A, float *value,
                                     float *mean, float *logp, void *stream) {
  return forward_fused("trl_ppo_forward_fused", false, critic, actor, log_std, obs, action, B, A, value, mean, logp, stream);
}

extern "C" int trl_ppo_forward_fused_bx6(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std,
                                         const float *obs, const float *action, int64_t B, int32_t A, float *value,
                                         float *mean, float *logp, void *stream) {
  return forward_fused("trl_ppo_forward_fused_bx6", true, critic, actor, log_std, obs, action, B, A, value, mean, logp, stream);
}
