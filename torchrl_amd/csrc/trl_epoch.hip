// trl_epoch.hip -- the reference's PPO minibatch loop in ONE launch (gfx950).
//
//   for epoch in range(n_epochs):                      examples/ppo.py:70-73
//     for minibatch in loader:                         (sequential, un-shuffled, last partial batch kept: ppo.py:66-68)
//       controller.learn(*minibatch)                   ppo_controller.py:53-80
//
// At the reference's minibatch sizes (32 default, 64 in its spec) one learn() is a dependency chain of a few
// microseconds of arithmetic: as two launches per call it is bound by launch boundaries, weight staging and the
// partial-sum flush (20.8 us of device time, 26.9 us with the host).  Every update depends on the
// one before it, so the loop cannot be spread over the chip; what it can do is never leave a CU.  And the critic and
// the actor never meet: the loss is a sum of a critic term and actor terms, `advantages` and `returns` are inputs,
// clip_grad_value_ and Adam are elementwise -- two independent chains.  So the launch is TWO workgroups, one per
// network, each walking all minibatches of all epochs on its own CU (8 wavefronts, two per SIMD):
//
//   * the hidden x hidden weight matrix and both its Adam moments live in REGISTERS of the lane that owns them -- the
//     lane that holds that element of the weight-gradient product's accumulator tile -- for the whole launch; layer 0's
//     and the small vectors' (last layer, biases, log_std) state is spread one element per thread; LDS holds the
//     copies of the weights the products read;
//   * a minibatch is walked in tiles of 64 samples; every product is a 16x16 output tile of v_mfma_f32_16x16x4_f32
//     (exact f32) with both operands read from LDS with the reduction index contiguous (one ds_read_b128 feeds four
//     MFMAs), which is why activations / gradients / W1 that are summed over along either index are kept in both
//     orientations (the second copy is 4 extra 4 B stores per tile element);
//   * the last layer (fan-out <= 4) is folded into the epilogue of the tile that produces its input: a 16-lane DPP
//     row sum per output; the loss (k_ppo_loss's arithmetic, op for op) runs on one wave, lane = sample; the masked
//     outer product and the last layer's gradient run with lane = feature;
//   * clip_grad_value_ + Adam (trl_adam_elem: torch's op order) are applied by the owning lane straight from the
//     accumulator; the step's bias corrections come from a table a small launch derives ahead (double precision).
//
// Results: the same arithmetic as trl_ppo_learn_fused up to the order of the f32 sums (tests/test_gpu_epoch.py: K
// consecutive learn() calls, the oracle's torch-CPU loop, the reference's golden two-step fixture).
#include <string.h>

#include <type_traits>

#include "trl_common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int NW = 8, NT = 64 * NW;  // wavefronts (two per SIMD: 256 registers each), threads per workgroup
constexpr int T1 = 16 / NW;         // accumulator tiles a wave owns of a product with up to 16 output tiles
constexpr int U0 = 4096 / NT;       // elements of the padded layer-0 image per thread at most (256 x 16)
constexpr int G0 = U0 / 4;          // groups of four layer-0 parameters a thread owns at most
constexpr int S = 64;               // samples per tile
constexpr int LS = S + 4;           // row stride (floats) of every 64-column LDS image: 16 lanes x 16 B on 16 distinct bank quads
constexpr int KP = 16;              // fan-in + 1 (the bias column), padded to one k round
constexpr int MAXO = 4;
// the tile pipeline's wide form (k_ppo_tiles<..., KPW>, one hidden layer, one batch spread over the chip): two k rounds at
// layer 0 (fan-in + bias up to 32) and up to six outputs
constexpr int KPW = 32;
constexpr int MAXO_T = 6;
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;   // math.log(math.sqrt(2 * math.pi))
constexpr float ENT_CONST = 1.41893853320467274178f;      // 0.5 + 0.5 * math.log(2 * math.pi)

struct Net {
  int depth;                 // 2 or 3 Linear layers
  int d0, H1, H2, O;         // fan-in, hidden widths (H2 = 0 when depth == 2), fan-out
  int64_t w_off[3], b_off[3];      // offsets (floats) of the layers' parameters in the flat buffers
  int64_t ls_off;            // log_std (actor), -1 for the critic
};

struct Args {
  Net net[2];                // 0 = critic, 1 = actor
  float *param, *grad, *m, *v;
  const float *obs, *action, *returns, *old_logp, *adv;
  int64_t n_rows;
  int B, n_epochs, A;
  float lo, hi, alpha, beta, clip;
  float w1, beta2, c2, eps;  // Adam's step-independent scalars
  const float *steps;        // [K][4]: bc2_sqrt, neg_step, RN(1 / bc2_sqrt), - of update k (k_epoch_prep)
  float *losses;             // [K][4]
  // one big batch spread over many workgroups (k_ppo_tiles): workgroup (role, g) takes rows [g * slab, (g + 1) * slab) and
  // writes its partial gradient sums to gpart + g * n_flat (indexed like the flat buffers) and loss terms to lpart
  int64_t slab;
  float *gpart, *lpart;      // lpart: [2][G][4]
  int64_t n_flat;
  // k_ppo_tiles launched one role at a time (A2C without a caller-supplied advantage): the critic's workgroups write
  // adv_out[row] = returns[row] - V(row) on their way, the actor's launch behind them reads it as `adv`
  int role_base;
  float *adv_out;
  // forward only (trl_ac_forward_tiles): the tiles stop behind the loss phase, whose waves write the critic's value /
  // the actor's mean and Normal log-prob of `action` instead; nothing else leaves the kernel
  int fwd_only;
  float *fwd_value, *fwd_mean, *fwd_logp;
};

// LDS map of one workgroup (offsets in floats).  Aliases: g2 (the [sample][feature] copy of layer 1's output gradient)
// lives where h1's [sample][feature] copy was (dead once layer 1 has run); at update time -- every product of the
// minibatch done -- the partial sums of the lane = feature pass lie over h1 / h1t and layer 0's flushed gradient over h2t.
struct Lay {
  int xp[2], xpt[2], w0p, gw0, h1, h1t, h2t, g1t, w1, w1t, red, gout, part, vp_p, vp_m, vp_v, tail_g, total;
  int HL, HLP, NB, wpb, nred, NV, o_bl, o_b1, o_ls, PO;
};

__host__ __device__ inline Lay make_lay(const Net &n, int kp = KP) {
  Lay L;
  const int ldk = kp + 4;
  // the wide form runs in tiles mode only (no update in the kernel: no moments, layer 0's gradient leaves the accumulators
  // for the partial buffer directly) and keeps the partial sums of the lane = feature pass over the x' images, which are
  // dead by then -- what makes 256 units with 32 columns and six outputs fit the 160 KB
  const bool wide = kp > KP;
  int o = 0;
  auto take = [&](int f) { int at = o; o += (f + 3) / 4 * 4; return at; };
  const bool d3 = n.depth == 3;
  L.HL = d3 ? n.H2 : n.H1;
  L.NB = (L.HL + 63) / 64;
  L.HLP = L.NB * 64;
  L.wpb = L.NB == 1 ? NW : (L.NB == 2 ? NW / 2 : NW / 4);      // wavefronts per block of 64 features in the lane = feature pass
  // partial dot products of the last layer, one plane per slot: two hidden layers -> a slot per 16 features of h2; one
  // hidden layer -> a wave sums over its own tiles first, slot = w >> 2
  L.nred = d3 ? L.HL / 16 : (L.HL / 16 < NW / 4 ? L.HL / 16 : NW / 4);
  L.PO = n.O + (d3 ? 1 : 0);                         // partial-sum planes: dwL[o] and (two hidden layers) db1
  L.xp[0] = take(S * ldk);
  L.xp[1] = take(S * ldk);
  L.xpt[0] = take(kp * LS);
  L.xpt[1] = take(kp * LS);
  L.w0p = take(n.H1 * ldk);
  const int part_f = L.wpb * L.PO * L.HLP;
  if (d3) {
    L.h1 = take(S * LS);
    L.h1t = take(n.H1 * LS);
    if (part_f > o - L.h1) o = L.h1 + (part_f + 3) / 4 * 4;
    L.part = L.h1;
    L.h2t = take(n.H2 * LS > n.H1 * ldk ? n.H2 * LS : n.H1 * ldk);
    L.gw0 = L.h2t;
    L.g1t = take(n.H1 * LS);
    L.w1 = take(n.H2 * LS);
    L.w1t = take(n.H1 * LS);
  } else {
    L.h1 = 0;
    L.h1t = take(n.H1 * LS);
    L.h2t = L.h1t;
    L.g1t = L.h1t;
    L.w1 = L.w1t = 0;
    if (wide) {
      L.part = L.xp[0];      // part_f <= 512 PO <= 2 S ldk for every width (PO <= 6)
      L.gw0 = L.w0p;         // (not used)
    } else {
      L.part = take(part_f);
      L.gw0 = take(n.H1 * ldk);
    }
  }
  L.red = take(L.nred * n.O * LS);
  L.gout = take(n.O * LS);
  // the small vectors, in this order: last layer's weights [O][HL], hidden bias b1 [H2] (two hidden layers) -- both whole
  // groups of four, updated four to a thread -- then the last layer's bias [O] and log_std [O] (actor), one to a thread
  L.o_b1 = n.O * L.HL;
  L.o_bl = L.o_b1 + (d3 ? n.H2 : 0);
  L.o_ls = L.o_bl + n.O;
  L.NV = L.o_ls + (n.ls_off >= 0 ? n.O : 0);
  L.vp_p = take(L.NV);
  L.vp_m = wide ? L.vp_p : take(L.NV);
  L.vp_v = wide ? L.vp_p : take(L.NV);
  L.tail_g = take(L.NV - L.o_bl);
  L.total = o;
  return L;
}

// One 16x16 output tile: sum over k of A[row i][k] * B[col i][k], both operands k-contiguous in LDS (row strides LDA,
// LDB floats), K a multiple of 16.  Lane (i = lane & 15, q = lane >> 4) supplies k = k0 + 4 q + c to the c-th MFMA of a
// round for both operands, so any k order is the same sum; the next round's 16 B reads are issued ahead of the MFMAs.
template <int LDA, int LDB>
__device__ __forceinline__ v4f mma(const float *A, const float *B, int K, v4f acc, int i, int q) {
  const float *pa = A + i * LDA + 4 * q, *pb = B + i * LDB + 4 * q;
  v4f a = *(const v4f *)pa, b = *(const v4f *)pb;
  for (int k0 = 16; k0 < K; k0 += 16) {
    const v4f na = *(const v4f *)(pa + k0), nb = *(const v4f *)(pb + k0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
    a = na;
    b = nb;
  }
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  return acc;
}

// Two tiles in one loop (independent accumulators: a wave's own MFMAs then fill the dependent-issue gap, 40 cycles
// against 32); the tiles usually share one operand (same pointer passed twice).
template <int LDA, int LDB>
__device__ __forceinline__ void mma2(const float *A0, const float *A1, const float *B0, const float *B1, int K, v4f &acc0,
                                     v4f &acc1, int i, int q) {
  const int oa = i * LDA + 4 * q, ob = i * LDB + 4 * q;
  v4f a0 = *(const v4f *)(A0 + oa), a1 = *(const v4f *)(A1 + oa), b0 = *(const v4f *)(B0 + ob), b1 = *(const v4f *)(B1 + ob);
  for (int k0 = 16; k0 < K; k0 += 16) {
    const v4f na0 = *(const v4f *)(A0 + oa + k0), na1 = *(const v4f *)(A1 + oa + k0), nb0 = *(const v4f *)(B0 + ob + k0),
              nb1 = *(const v4f *)(B1 + ob + k0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
    a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
  }
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc1, 0, 0, 0);
  acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
}

__device__ __forceinline__ v4f relu4(v4f x) {
  x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);
  return x;
}

__device__ __forceinline__ v4f mask4(v4f g, v4f h) {
  g.x = h.x > 0.f ? g.x : 0.f; g.y = h.y > 0.f ? g.y : 0.f; g.z = h.z > 0.f ? g.z : 0.f; g.w = h.w > 0.f ? g.w : 0.f;
  return g;
}

// sum over the 16 lanes of a DPP row (the lanes that share q), the same bits in every lane of the row
template <int CTRL>
__device__ __forceinline__ float dpp_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_sum16(float x) {
  x += dpp_f<0xB1>(x);       // quad_perm [1, 0, 3, 2]
  x += dpp_f<0x4E>(x);       // quad_perm [2, 3, 0, 1]
  x += dpp_f<0x141>(x);      // row_half_mirror
  x += dpp_f<0x140>(x);      // row_mirror
  return x;
}

// ---- clip_grad_value_ + Adam on four elements at once, trl_adam_elem's arithmetic (torch's op order, one f32 rounding
// per op) with the two IEEE divisions spelled out so that they cost what they must and no more:
//   * sqrt(v) / bias_correction2_sqrt divides by a per-step CONSTANT c: with rc = RN(1 / c) from the step table the
//     quotient is q0 = x rc, two residual corrections r = fma(-c, q, x), q += r rc -- the refinement the compiler's
//     own expansion ends with, minus its range scaling (x = sqrt(v) >= 0 and c in (0.03, 1]: nothing to scale; a
//     subnormal x may come out an ulp of a subnormal off, and is then added to eps);
//   * (step_size m) / denom: v_rcp_f32 seed, one Newton step, the same two residual corrections; denom >= eps, the
//     quotient is far from overflow, and a quotient in the subnormal range is added to a parameter.
// Correctly rounded quotients are unique, so in the ranges above these are the bits of __fdiv_rn
// (tests/test_gpu_epoch.py::test_fast_adam_is_trl_adam_elem_bit_for_bit, on the diagnostic build).  Written on 4-vectors:
// the multiplies, adds and fmas pair up as v_pk_*_f32.  clip = +inf means none (med3 then returns the gradient).
struct AdamV {
  float w1, beta2, c2, eps, clip, c, rc, neg_step;
};

__device__ __forceinline__ v4f vfma(v4f a, v4f b, v4f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v4f splat(float x) { return (v4f){x, x, x, x}; }

__device__ __forceinline__ void adam4(v4f &p, v4f &g, v4f &m, v4f &v, const AdamV s) {
  g.x = __builtin_amdgcn_fmed3f(g.x, -s.clip, s.clip); g.y = __builtin_amdgcn_fmed3f(g.y, -s.clip, s.clip);
  g.z = __builtin_amdgcn_fmed3f(g.z, -s.clip, s.clip); g.w = __builtin_amdgcn_fmed3f(g.w, -s.clip, s.clip);
  m = m + splat(s.w1) * (g - m);                                  // exp_avg.lerp_(grad, 1 - beta1)
  v = v * splat(s.beta2) + (splat(s.c2) * g) * g;                 // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
  const v4f x = {__fsqrt_rn(v.x), __fsqrt_rn(v.y), __fsqrt_rn(v.z), __fsqrt_rn(v.w)};
  const v4f c = splat(s.c), rc = splat(s.rc);
  v4f q = x * rc;
  q = vfma(vfma(-c, q, x), rc, q);
  q = vfma(vfma(-c, q, x), rc, q);
  const v4f den = q + splat(s.eps);                               // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
  const v4f n = splat(s.neg_step) * m;
  v4f y = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y), __builtin_amdgcn_rcpf(den.z), __builtin_amdgcn_rcpf(den.w)};
  y = vfma(vfma(-den, y, splat(1.0f)), y, y);
  v4f d = n * y;
  d = vfma(vfma(-den, d, n), y, d);
  d = vfma(vfma(-den, d, n), y, d);
  p = p + d;                                                      // param.addcdiv_(exp_avg, denom, value=-step_size)
}

__device__ __forceinline__ void adam1(float &p, float &g, float &m, float &v, const AdamV s) {
  v4f pp = splat(p), gg = splat(g), mm = splat(m), vv = splat(v);
  adam4(pp, gg, mm, vv, s);
  p = pp.x; g = gg.x; m = mm.x; v = vv.x;
}

#ifdef TRL_TUNING      // diagnostic build: clock stamps of every wave at the phase boundaries of an update (tools/epoch_stamps.py); taken at
                       // every update of the launch's second half so that the stamp code itself is warm, update K - 2's survive
__device__ unsigned long long g_epoch_stamp[2][8][16];
// (tiles mode: slab 0's middle tile, stamps 0 .. 8, and the next tile's start as stamp 9: tools/tiles_stamps.py)
#define STAMP(j)                                                                                              \
  do {                                                                                                        \
    if (TILES) {                                                                                              \
      if (lane == 0 && blockIdx.y == 0 && (j) <= 8 && t == ntile / 2) g_epoch_stamp[role][w][j] = __builtin_amdgcn_s_memtime(); \
      if (lane == 0 && blockIdx.y == 0 && (j) == 0 && t == ntile / 2 + 1) g_epoch_stamp[role][w][9] = __builtin_amdgcn_s_memtime(); \
    } else if (lane == 0 && k >= K / 2 && k + 1 < K && t == 0) {                                              \
      g_epoch_stamp[role][w][j] = __builtin_amdgcn_s_memtime();                                               \
    }                                                                                                         \
  } while (0)
// (tiles mode, inside a phase: any stamp index)
#define STAMPX(j)                                                                                             \
  do {                                                                                                        \
    if (TILES && lane == 0 && blockIdx.y == 0 && t == ntile / 2) g_epoch_stamp[role][w][j] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define STAMP(j) \
  do {           \
  } while (0)
#define STAMPX(j) \
  do {            \
  } while (0)
#endif

// the update table: bias corrections of steps step0 .. step0 + K - 1 (torch: bias_correction2 ** 0.5 and lr / bias_correction1,
// in double, narrowed at the point of use)
__global__ void k_epoch_prep(float *steps, int K, double lr, double beta1, double beta2, double eps, int64_t step0) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) {
    const TrlAdamScalars s = trl_adam_scalars(lr, beta1, beta2, eps, step0 + k, 1.0, 0.0);
    steps[4 * k] = s.bc2_sqrt;
    steps[4 * k + 1] = s.neg_step;
    steps[4 * k + 2] = (float)(1.0 / (double)s.bc2_sqrt);      // RN(1 / c) of the f32 constant c the division uses
    steps[4 * k + 3] = 0.f;
  }
}

__global__ void k_epoch_totals(float *losses, int K, float alpha, float beta) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) losses[4 * k + 3] = (losses[4 * k] + alpha * losses[4 * k + 1]) - beta * losses[4 * k + 2];
}

// role: 0 critic, 1 actor; D3: two hidden layers; O: the network's fan-out; KIND: 0 PPO (Normal policy, clipped
// surrogate), 1 A2C (softmax policy, adv * log-prob: a2c_controller.py:48-59 in k_a2c_loss's arithmetic; tiles mode only)
constexpr float F32_EPS = 1.1920928955078125e-07f;        // torch.finfo(torch.float32).eps
constexpr float F32_LOWEST = -3.4028234663852886e+38f;
template <int role, bool D3, int O, bool TILES, int KIND = 0, int KPT = 16>
__device__ __forceinline__ void epoch_body(const Args &P, float *lds) {
  constexpr int KP = KPT, LDK = KP + 4, CT = KP / 16;      // (these shadow the file's: the wide form's x' images have 32 columns)
  constexpr bool WIDE = KP > 16;
  static_assert(!WIDE || (TILES && !D3), "the wide form is tiles mode, one hidden layer");
  const Net &N = P.net[role];
  const Lay L = make_lay(N, KP);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, q = lane >> 4;
  const int H1 = N.H1, H2 = N.H2, HL = L.HL, d0 = N.d0;
  float *w0p = lds + L.w0p, *gw0 = lds + L.gw0, *h1 = lds + L.h1, *h1t = lds + L.h1t, *h2t = lds + L.h2t, *g1t = lds + L.g1t,
        *w1 = lds + L.w1, *w1t = lds + L.w1t, *red = lds + L.red, *gout = lds + L.gout, *part = lds + L.part,
        *vpp = lds + L.vp_p, *vpm = lds + L.vp_m, *vpv = lds + L.vp_v, *tailg = lds + L.tail_g;
  const int xp0 = L.xp[0], xp1 = L.xp[1], xpt0 = L.xpt[0], xpt1 = L.xpt[1];      // (no run-time index into the map: that would put it on the stack)
  float *hlt = D3 ? h2t : h1t;               // the last hidden layer's [feature][sample] image
  float *g2 = h1;                            // (alias, see Lay)

  // ---- ownership.  W1 (two hidden layers): tiles w + NW u of dW1 [H2][H1], element r of an accumulator = (row r1 + 4 q + r,
  // column c1 + i): both moments in this lane's registers, the parameter itself in LDS (w1 / w1t).  Layer 0's
  // [W0 | b0]: element e = tid + u * NT, moments in registers, parameter in LDS (w0p).
  const int KT1 = D3 ? H1 / 16 : 1;
  bool own1[T1];
  int r1[T1], c1[T1];
  v4f m1[T1], v1[T1], acc1[T1];
#pragma unroll
  for (int u = 0; u < T1; ++u) {
    const int tt = w + NW * u;
    own1[u] = D3 && tt < (H2 / 16) * KT1;
    r1[u] = (tt / KT1) * 16;
    c1[u] = (tt % KT1) * 16;
    m1[u] = v1[u] = acc1[u] = (v4f){0, 0, 0, 0};
    if (own1[u]) {
      float p[4], m[4], v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t g = N.w_off[1] + (int64_t)(r1[u] + 4 * q + r) * H1 + c1[u] + i;
        p[r] = P.param[g]; m[r] = TILES ? 0.f : P.m[g]; v[r] = TILES ? 0.f : P.v[g];      // (tiles mode: no update here, no moments)
        w1[(r1[u] + 4 * q + r) * LS + c1[u] + i] = p[r];
      }
      m1[u] = (v4f){m[0], m[1], m[2], m[3]}; v1[u] = (v4f){v[0], v[1], v[2], v[3]};
      *(v4f *)(w1t + (c1[u] + i) * LS + r1[u] + 4 * q) = (v4f){p[0], p[1], p[2], p[3]};
    }
  }
  // layer 0: the padded [H1][KP] image, element e = (row n = e >> 4, column j = e & 15); j < d0 a weight, j == d0 the
  // bias, j > d0 padding (zero parameter, zero gradient, never updated, never written back).  The H1 (d0 + 1) real
  // elements, numbered row by row, are owned four consecutive ones to a thread (group tid + u NT): one adam4 per group
  // with all four lanes of the vector at work, and the groups of a small layer fall on the first waves only.
  const int n0e = H1 * KP;
  const int C0 = d0 + 1, n0v = H1 * C0;
  // Tiles mode keeps the columns of the layer-0 images PERMUTED within each block of 16: column j sits in slot
  // 4 (j & 3) + ((j & 15) >> 2) (a 4 x 4 transpose: its own inverse).  The c-th MFMA of a k round multiplies slots
  // {c, 4 + c, 8 + c, 12 + c}, so the real columns -- fan-in + bias, e.g. 9 of 16 or 18 of 32 -- fill the first
  // ceil(columns / 4) MFMAs of a round and the forward pass skips the rest (all-zero products).  The persistent form
  // keeps the natural order (its per-thread ownership walks consecutive columns).
  constexpr bool PERM = TILES;
  auto slot_of = [&](int j) { return PERM ? (j & ~15) + 4 * (j & 3) + ((j & 15) >> 2) : j; };
  const int nck0 = C0 >= 16 ? 4 : (C0 + 3) / 4;                            // MFMAs of the first k round
  const int nck1 = C0 <= 16 ? 0 : (C0 >= 32 ? 4 : (C0 - 16 + 3) / 4);  // ... of the second
  auto w0_global = [&](int e) -> int64_t {
    const int n = e / KP, j = slot_of(e % KP);      // (slot -> column)
    return j < d0 ? N.w_off[0] + (int64_t)n * d0 + j : (j == d0 ? N.b_off[0] + n : -1);
  };
  v4f m0[G0], v0[G0];
  int first0[G0];      // the group's first element: column j << 16 | offset n LDK + j in the padded images (w0p, gw0)
#pragma unroll
  for (int u = 0; u < G0; ++u) {
    const int ce = 4 * (tid + u * NT), n = ce / C0, j = ce - n * C0;
    first0[u] = ce < n0v ? (j << 16 | (n * LDK + j)) : 0;
  }
  // the offsets of group u's four elements; -1 beyond the layer
  auto group_at = [&](int u, int at[4]) {
    int a = first0[u] & 0xffff, j = first0[u] >> 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      at[r] = 4 * (tid + u * NT) + r < n0v ? a : -1;
      ++j; ++a;
      if (j == C0) { j = 0; a += LDK - C0; }
    }
  };
  for (int e = tid; e < H1 * LDK; e += NT) w0p[e] = 0.f;      // (the padding columns stay zero for good)
  __syncthreads();
  if (PERM) {
    for (int e = tid; e < n0v; e += NT) {
      const int n = e / C0, j = e - n * C0;
      w0p[n * LDK + slot_of(j)] = P.param[w0_global(n * KP + slot_of(j))];
    }
  }
#pragma unroll
  for (int u = 0; u < (PERM ? 0 : G0); ++u) {
    float m[4] = {0.f, 0.f, 0.f, 0.f}, v[4] = {0.f, 0.f, 0.f, 0.f};
    int at[4];
    group_at(u, at);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (at[r] >= 0) {
        const int64_t g = w0_global((at[r] / LDK) * KP + at[r] % LDK);
        if (!TILES) { m[r] = P.m[g]; v[r] = P.v[g]; }
        w0p[at[r]] = P.param[g];
      }
    }
    m0[u] = (v4f){m[0], m[1], m[2], m[3]}; v0[u] = (v4f){v[0], v[1], v[2], v[3]};
  }
  bool own0[T1];      // accumulator tiles of dW0' [H1][KP]: rows 16 (w + NW u), columns 16 c
  v4f acc0[T1][CT];
#pragma unroll
  for (int u = 0; u < T1; ++u) {
    own0[u] = w + NW * u < H1 / 16;
#pragma unroll
    for (int c = 0; c < CT; ++c) acc0[u][c] = (v4f){0, 0, 0, 0};
  }
  // the small vectors: last layer [O][HL], its bias [O], hidden bias b1 [H2] (two hidden layers), log_std [O] (actor)
  auto vp_global = [&](int e) -> int64_t {
    if (e < L.o_b1) return N.w_off[D3 ? 2 : 1] + e;
    if (e < L.o_bl) return N.b_off[1] + (e - L.o_b1);
    if (e < L.o_ls) return N.b_off[D3 ? 2 : 1] + (e - L.o_bl);
    return N.ls_off + (e - L.o_ls);
  };
  for (int e = tid; e < L.NV; e += NT) {
    const int64_t g = vp_global(e);
    vpp[e] = P.param[g];
    if (!TILES) { vpm[e] = P.m[g]; vpv[e] = P.v[g]; }
  }

  // tiles mode: this workgroup's slab of the one batch is walked like one minibatch, normalised by the whole batch
  const int64_t base = TILES ? (int64_t)blockIdx.y * P.slab : 0;
  const int64_t n_rows = TILES ? (P.n_rows - base < P.slab ? P.n_rows - base : P.slab) : P.n_rows;
  const int B = TILES ? (int)n_rows : P.B;
  const int mb_per_epoch = TILES ? 1 : (int)((n_rows + B - 1) / B);
  const int K = TILES ? 1 : mb_per_epoch * P.n_epochs;

  // ---- input prefetch: a tile's observations are 64 * d0 consecutive floats (consecutive rows), one per thread.
  // Wave 0's lane s also carries sample s's loss inputs.
  constexpr int XU = (S * (KP - 1) + NT - 1) / NT;
  float xin[XU];
#pragma unroll
  for (int u = 0; u < XU; ++u) xin[u] = 0.f;
  // PPO's actor loss is independent per (sample, action dimension): wave a < O takes dimension a of the tile's 64 samples
  // (the critic's and A2C's -- a softmax over the actions -- stay on wave 0), so the loss phase is one dimension long
  constexpr bool LW = role == 1 && KIND == 0;
  float in_act[1], in_old[1], in_adv = 0.f, in_ret = 0.f;
  in_act[0] = in_old[0] = 0.f;
  auto prefetch = [&](int64_t row0, int rows) {
    row0 += base;
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int e = tid + u * NT;
      xin[u] = e < rows * d0 ? P.obs[row0 * d0 + e] : 0.f;
    }
    if (LW ? w < O : w == 0) {
      const bool ok = lane < rows;
      const int64_t r = row0 + (ok ? lane : 0);
      if (role == 0) {
        in_ret = ok ? P.returns[r] : 0.f;
      } else {
        in_adv = ok ? P.adv[r] : 0.f;
        if (KIND == 1) {
          in_act[0] = ok ? P.action[r] : 0.f;      // the action's index
        } else {      // (LW: wave a carries action dimension a)
          in_act[0] = ok ? P.action[r * O + w] : 0.f; in_old[0] = ok ? P.old_logp[r * O + w] : 0.f;
        }
      }
    }
  };
  // registers -> the x' images [S][KP] and [KP][S]: columns j < d0 the observation, column d0 ones (the bias), the rest
  // -- and every entry of a sample beyond the minibatch -- zeros (such a sample contributes exact zeros downstream)
  auto store_x = [&](float *X, float *XT, int rows) {
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int e = tid + u * NT;
      if (e < S * d0) {
        const int s = e / d0, j = e - s * d0;
        X[s * LDK + slot_of(j)] = xin[u];
        XT[slot_of(j) * LS + s] = xin[u];
      }
    }
    for (int e = tid; e < (KP - d0) * S; e += NT) {
      const int s = e & 63, j = d0 + (e >> 6);
      const float v = (j == d0 && s < rows) ? 1.f : 0.f;
      X[s * LDK + slot_of(j)] = v;
      XT[slot_of(j) * LS + s] = v;
    }
  };
  auto tile_of = [&](int k, int t, int64_t &row0, int &rows, int &bm) {      // update k, tile t -> rows of the data
    const int mb = k % mb_per_epoch;
    const int64_t mb0 = (int64_t)mb * B;
    bm = (int)(n_rows - mb0 < B ? n_rows - mb0 : B);
    row0 = mb0 + (int64_t)t * S;
    rows = bm - t * S < S ? bm - t * S : S;
  };

  // Input pipeline, two tiles deep: at the top of tile T the registers hold tile T + 1 (loaded during tile T - 1: long
  // arrived), they go to the other x' buffers, and the loads of tile T + 2 are issued into them -- a load is never
  // waited for less than a whole tile after its issue (an observation row comes from HBM the first time: ~2 us).
  auto advance = [&](int &k, int &t) {      // the tile after (k, t); k == K when there is none
    int64_t r0; int rr, bm;
    tile_of(k, 0, r0, rr, bm);
    if ((t + 1) * S < bm) { ++t; } else { ++k; t = 0; }
  };
  int cur = 0;
  int pk = 0, pt = 0;              // the tile the registers hold / are loading
  int prow = 0;                    // its row count
  {
    int64_t row0; int bm;
    tile_of(0, 0, row0, prow, bm);
    prefetch(row0, prow);
    store_x(lds + L.xp[0], lds + L.xpt[0], prow);
  }
  float cur_act[1], cur_old[1], cur_adv = in_adv, cur_ret = in_ret;
  float nxt_act[1], nxt_old[1], nxt_adv = 0.f, nxt_ret = 0.f;
  cur_act[0] = in_act[0]; cur_old[0] = in_old[0]; nxt_act[0] = nxt_old[0] = 0.f;
  advance(pk, pt);
  if (pk < K) {
    int64_t row0; int bm;
    tile_of(pk, pt, row0, prow, bm);
    prefetch(row0, prow);
  }
  __syncthreads();

  // lane = feature pass: this wave's block of 64 features and its share of the tile's samples
  const int gp_nb = w / L.wpb, gp_sr = w % L.wpb;      // (the wave takes S / wpb samples of its block)
  const int gp_n = gp_nb * 64 + lane;
  const bool gp_on = gp_nb < L.NB && gp_n < HL;
  // forward tiles of the last hidden layer: this wave's sample tile and slot of partial dot products
  const int f_s0 = (w & 3) * 16;

  for (int k = 0; k < K; ++k) {
    int bm;
    { int64_t r0; int rr; tile_of(k, 0, r0, rr, bm); }
    const int ntile = (bm + S - 1) / S;
    const int64_t nb = TILES ? P.n_rows : (int64_t)bm;      // rows the means are taken over
    const float invB = 1.0f / (float)nb;
    const v4f step_k = TILES ? (v4f){1.f, 0.f, 1.f, 0.f} : *(const v4f *)(P.steps + 4 * k);
    float a_crit = 0.f, a_min = 0.f, a_ent = 0.f, a_dls[O], a_gb[O], dwl[O], db1 = 0.f;
#pragma unroll
    for (int o = 0; o < O; ++o) a_dls[o] = a_gb[o] = dwl[o] = 0.f;

    for (int t = 0; t < ntile; ++t) {
      const float *X = lds + (cur ? xp1 : xp0), *XT = lds + (cur ? xpt1 : xpt0);
      int rows_t;
      { int64_t r0; int b2; tile_of(k, t, r0, rows_t, b2); }
      // registers (tile T + 1) -> the other x' buffers and wave 0's "next" loss inputs; then the loads of tile T + 2
      const bool more = pk < K;
      if (more) {
        store_x(lds + (cur ? xp0 : xp1), lds + (cur ? xpt0 : xpt1), prow);
        nxt_adv = in_adv; nxt_ret = in_ret;
        nxt_act[0] = in_act[0]; nxt_old[0] = in_old[0];
        advance(pk, pt);
        if (pk < K) {
          int64_t row0; int bm;
          tile_of(pk, pt, row0, prow, bm);
          prefetch(row0, prow);
        }
      }

      STAMP(0);
      // ---- layer 0: h1[s][n] = relu(x'[s] . W0'[n]) as C[s][n] tiles (lane: 4 consecutive samples of feature n0 + i).
      // One hidden layer: the last layer's dot products ride in the epilogue (16-lane row sums), partials per slot.
      // A wave's tiles w + NW u share the sample tile; they are computed NTW at a time as straight-line code (every
      // operand read issued ahead of the first MFMA; a tile is one to four dependent instructions long, so a loop over
      // tiles waits out each tile's LDS round trip in turn).  A slot beyond the layer repeats tile w & 3 -- the same
      // values stored to the same place -- with zero last-layer weights.
      {
        const int ntl = 4 * (H1 / 16);
        float pr[O][4];
#pragma unroll
        for (int o = 0; o < O; ++o) pr[o][0] = pr[o][1] = pr[o][2] = pr[o][3] = 0.f;
        auto tiles = [&](auto ntw_c, int u0) {      // the wave's tiles w + NW (u0 + u), u < NTW
          constexpr int NTW = decltype(ntw_c)::value;
          int n0[NTW];
          float wl[NTW][O];
          v4f c[NTW];
#pragma unroll
          for (int u = 0; u < NTW; ++u) {
            const int t0 = w + NW * (u0 + u);
            const bool ok = t0 < ntl;
            n0[u] = ((ok ? t0 : (w & 3)) >> 2) * 16;
#pragma unroll
            for (int o = 0; o < O; ++o) wl[u][o] = (!D3 && ok) ? vpp[o * HL + n0[u] + i] : 0.f;
            c[u] = (v4f){0, 0, 0, 0};
          }
          if (d0 < 4) {      // fan-in + bias fit one MFMA: k = q
            const float xq = X[(f_s0 + i) * LDK + slot_of(q)];
            float wq[NTW];
#pragma unroll
            for (int u = 0; u < NTW; ++u) wq[u] = w0p[(n0[u] + i) * LDK + slot_of(q)];
#pragma unroll
            for (int u = 0; u < NTW; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq, wq[u], c[u], 0, 0, 0);
          } else {
            v4f xa[CT], wb[NTW][CT];
#pragma unroll
            for (int kc = 0; kc < CT; ++kc) {
              xa[kc] = *(const v4f *)(X + (f_s0 + i) * LDK + 16 * kc + 4 * q);
#pragma unroll
              for (int u = 0; u < NTW; ++u) wb[u][kc] = *(const v4f *)(w0p + (n0[u] + i) * LDK + 16 * kc + 4 * q);
            }
#pragma unroll
            for (int kc = 0; kc < CT; ++kc) {
              const int nc = PERM ? (kc == 0 ? nck0 : nck1) : 4;      // (wave-uniform)
              if (nc > 0) {
#pragma unroll
                for (int u = 0; u < NTW; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kc].x, wb[u][kc].x, c[u], 0, 0, 0);
              }
              if (nc > 1) {
#pragma unroll
                for (int u = 0; u < NTW; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kc].y, wb[u][kc].y, c[u], 0, 0, 0);
              }
              if (nc > 2) {
#pragma unroll
                for (int u = 0; u < NTW; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kc].z, wb[u][kc].z, c[u], 0, 0, 0);
              }
              if (nc > 3) {
#pragma unroll
                for (int u = 0; u < NTW; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kc].w, wb[u][kc].w, c[u], 0, 0, 0);
              }
            }
          }
#pragma unroll
          for (int u = 0; u < NTW; ++u) {
            const v4f cu = relu4(c[u]);
            *(v4f *)(h1t + (n0[u] + i) * LS + f_s0 + 4 * q) = cu;
            if (D3) {
              float *hp = h1 + (f_s0 + 4 * q) * LS + n0[u] + i;
              hp[0] = cu.x; hp[LS] = cu.y; hp[2 * LS] = cu.z; hp[3 * LS] = cu.w;
            } else {
#pragma unroll
              for (int o = 0; o < O; ++o) {
                pr[o][0] = __builtin_fmaf(wl[u][o], cu.x, pr[o][0]); pr[o][1] = __builtin_fmaf(wl[u][o], cu.y, pr[o][1]);
                pr[o][2] = __builtin_fmaf(wl[u][o], cu.z, pr[o][2]); pr[o][3] = __builtin_fmaf(wl[u][o], cu.w, pr[o][3]);
              }
            }
          }
        };
        // (two hidden layers are at most 64 wide: two tiles a wave; one wide layer: four at a time -- eight would hold the
        // registers of a second kernel)
        if (D3 || ntl <= 2 * NW) {
          tiles(std::integral_constant<int, 2>(), 0);
        } else {
          constexpr int CH = (O <= 2 && !WIDE) ? 4 : 2;      // (three and more outputs: their partial dots take the registers)
          for (int u0 = 0; w + NW * u0 < ntl; u0 += CH) tiles(std::integral_constant<int, CH>(), u0);
        }
        STAMPX(10);
        if (!D3 && w < ntl) {
#pragma unroll
          for (int o = 0; o < O; ++o) {
            v4f sres;
            sres.x = row_sum16(pr[o][0]); sres.y = row_sum16(pr[o][1]); sres.z = row_sum16(pr[o][2]); sres.w = row_sum16(pr[o][3]);
            if (i == 0) *(v4f *)(red + ((w >> 2) * O + o) * LS + f_s0 + 4 * q) = sres;
          }
        }
        STAMPX(11);
      }
      __syncthreads();
      STAMP(1);
      // ---- layer 1 (two hidden layers): h2[s][n] = relu(h1[s] . W1[n] + b1[n]), the last layer's partial dots behind it
      if (D3) {
        const int ntl = 4 * (H2 / 16);
        for (int t0 = w; t0 < ntl; t0 += 2 * NW) {      // tiles t0 and t0 + NW: the same sample tile, two feature tiles
          const bool two = t0 + NW < ntl;
          const int n0 = (t0 >> 2) * 16, n1 = two ? ((t0 + NW) >> 2) * 16 : n0;
          float wl0[O], wl1[O];
#pragma unroll
          for (int o = 0; o < O; ++o) { wl0[o] = vpp[o * HL + n0 + i]; wl1[o] = vpp[o * HL + n1 + i]; }
          const float bb0 = vpp[L.o_b1 + n0 + i], bb1 = vpp[L.o_b1 + n1 + i];
          v4f c0 = {bb0, bb0, bb0, bb0}, c1 = {bb1, bb1, bb1, bb1};
          mma2<LS, LS>(h1 + f_s0 * LS, h1 + f_s0 * LS, w1 + n0 * LS, w1 + n1 * LS, H1, c0, c1, i, q);
          c0 = relu4(c0);
          c1 = relu4(c1);
          *(v4f *)(h2t + (n0 + i) * LS + f_s0 + 4 * q) = c0;
          if (two) *(v4f *)(h2t + (n1 + i) * LS + f_s0 + 4 * q) = c1;
#pragma unroll
          for (int o = 0; o < O; ++o) {
            v4f s0r, s1r;
            s0r.x = row_sum16(wl0[o] * c0.x); s0r.y = row_sum16(wl0[o] * c0.y); s0r.z = row_sum16(wl0[o] * c0.z); s0r.w = row_sum16(wl0[o] * c0.w);
            s1r.x = row_sum16(wl1[o] * c1.x); s1r.y = row_sum16(wl1[o] * c1.y); s1r.z = row_sum16(wl1[o] * c1.z); s1r.w = row_sum16(wl1[o] * c1.w);
            if (i == 0) {
              *(v4f *)(red + ((n0 >> 4) * O + o) * LS + f_s0 + 4 * q) = s0r;
              if (two) *(v4f *)(red + ((n1 >> 4) * O + o) * LS + f_s0 + 4 * q) = s1r;
            }
          }
        }
        __syncthreads();
      }
      STAMP(2);
      // ---- loss and its derivative at the outputs (wave 0, lane = sample): k_ppo_loss's arithmetic
      if (LW) {
        // PPO's actor: wave a < O, action dimension a of the tile's samples (k_ppo_loss's arithmetic)
        if (w < O) {
          const bool ok = lane < rows_t;
          const int a = w;
          float outa = vpp[L.o_bl + a];
          for (int sl = 0; sl < L.nred; ++sl) outa += red[(sl * O + a) * LS + lane];
          const float gmin = -(1.0f / (float)(nb * O));
          const float sigma = expf(vpp[L.o_ls + a]);
          const float var = sigma * sigma;
          const float logsig = logf(sigma);
          const float den = 2.0f * var;
          const float d = cur_act[0] - outa;
          const float sq = d * d;
          const float lp = ((-sq) / den - logsig) - LOG_SQRT_2PI;
          if (TILES && P.fwd_only && ok) {
            const int64_t at = (base + (int64_t)t * S + lane) * O + a;
            if (P.fwd_mean) P.fwd_mean[at] = outa;
            if (P.fwd_logp) P.fwd_logp[at] = lp;
          }
          const float ratio = expf(lp - cur_old[0]);
          const float s1 = ratio * cur_adv;
          const float rc = fminf(fmaxf(ratio, P.lo), P.hi);
          const float s2 = rc * cur_adv;
          const float g1 = s1 < s2 ? gmin : (s1 == s2 ? 0.5f * gmin : 0.f);
          const float g2v = s2 < s1 ? gmin : (s1 == s2 ? 0.5f * gmin : 0.f);
          const bool inr = ratio >= P.lo && ratio <= P.hi;
          const float g_ratio = g1 * cur_adv + (inr ? g2v * cur_adv : 0.f);
          const float g_lp = g_ratio * ratio;
          const float g = ok ? (g_lp / den) * (2.0f * d) : 0.f;
          a_min += ok ? fminf(s1, s2) : 0.f;
          a_ent += ok ? ENT_CONST + logsig : 0.f;
          a_dls[0] += ok ? g_lp * (sq / var - 1.0f) : 0.f;
          a_gb[0] += g;
          gout[a * LS + lane] = g;
        }
      } else if (w == 0) {
        const bool ok = lane < rows_t;
        float out[O];
#pragma unroll
        for (int o = 0; o < O; ++o) {
          float sacc = vpp[L.o_bl + o];
          for (int sl = 0; sl < L.nred; ++sl) sacc += red[(sl * O + o) * LS + lane];
          out[o] = sacc;
        }
        if (role == 0) {      // critic: mse_loss(values, returns), weight alpha
          if (TILES && P.adv_out && ok) P.adv_out[base + (int64_t)t * S + lane] = cur_ret - out[0];
          if (TILES && P.fwd_only && ok) P.fwd_value[base + (int64_t)t * S + lane] = out[0];
          const float dv = out[0] - cur_ret;
          const float g = ok ? P.alpha * ((2.0f * invB) * dv) : 0.f;
          a_crit += ok ? dv * dv : 0.f;
          a_gb[0] += g;
          gout[lane] = g;
        } else if (KIND == 1) {
          // out = the actor's logits; nn.Softmax(dim=1) (models.py:99-103), Categorical(probs), the loss of
          // a2c_controller.py:48-59; cur_adv is the detached advantage returns - V(obs) (the caller's forward pass)
          float zmax = out[0];
#pragma unroll
          for (int n = 1; n < O; ++n) zmax = fmaxf(zmax, out[n]);
          float pr[O], esum = 0.f;
#pragma unroll
          for (int n = 0; n < O; ++n) { pr[n] = expf(out[n] - zmax); esum += pr[n]; }
          float sp = 0.f;
#pragma unroll
          for (int n = 0; n < O; ++n) { pr[n] = pr[n] / esum; sp += pr[n]; }      // Categorical: probs / probs.sum(-1, keepdim=True)
          int ai = (int)cur_act[0];      // action.long()
          ai = ai < 0 ? 0 : (ai >= O ? O - 1 : ai);
          float ent = 0.f, gdot = 0.f, logp = 0.f, gg[O];
#pragma unroll
          for (int n = 0; n < O; ++n) {
            const float pn = pr[n] / sp;
            const float cl = fminf(fmaxf(pn, F32_EPS), 1.0f - F32_EPS);      // clamp_probs
            const float logit = logf(cl);
            ent += fmaxf(logit, F32_LOWEST) * pn;
            if (n == ai) logp = logit;
            const float dlogit = (pn >= F32_EPS && pn <= 1.0f - F32_EPS) ? 1.0f / cl : 0.f;
            float g1 = (P.beta * invB) * (logit + pn * dlogit);      // from -beta * entropy
            if (n == ai) g1 += (-(cur_adv * invB)) * dlogit;          // from the actor term
            gg[n] = g1;
            gdot += g1 * pn;
          }
          float dp[O], dsum = 0.f;
#pragma unroll
          for (int n = 0; n < O; ++n) { dp[n] = (gg[n] - gdot) / sp; dsum += dp[n] * pr[n]; }      // back through the normalisation
#pragma unroll
          for (int n = 0; n < O; ++n) {
            const float g = ok ? (dp[n] - dsum) * pr[n] : 0.f;      // softmax backward
            a_gb[n] += g;
            gout[n * LS + lane] = g;
          }
          a_min += ok ? cur_adv * logp : 0.f;
          a_ent += ok ? -ent : 0.f;
        }
      }
      __syncthreads();
      STAMP(3);
      const bool bwd = !(TILES && P.fwd_only);      // (forward only: the tile ends here)
      // ---- lane = feature: dwL[o][n] += sum_s gout[s][o] h[n][s];  g[n][s] = (sum_o gout[s][o] wL[o][n]) (h > 0), written
      // over h's image and (two hidden layers) as the [sample][feature] copy the next product sums over
      if (bwd && gp_on) {
        float wl_n[O];
#pragma unroll
        for (int o = 0; o < O; ++o) wl_n[o] = vpp[o * HL + gp_n];
        // the wave's gp_ns = 64 / wpb samples, four at a time, as straight-line code: every read ahead of the first store
        // (the gradient overwrites h in place, which keeps a loop's reads behind the previous iteration's store)
        auto pass = [&](auto nit_c, int s0) {
          constexpr int NIT = decltype(nit_c)::value;
          v4f h[NIT], go[NIT][O];
#pragma unroll
          for (int j = 0; j < NIT; ++j) {
            h[j] = *(const v4f *)(hlt + gp_n * LS + s0 + 4 * j);
#pragma unroll
            for (int o = 0; o < O; ++o) go[j][o] = *(const v4f *)(gout + o * LS + s0 + 4 * j);
          }
#pragma unroll
          for (int j = 0; j < NIT; ++j) {
            v4f g = {0, 0, 0, 0};
#pragma unroll
            for (int o = 0; o < O; ++o) {
              dwl[o] = __builtin_fmaf(go[j][o].x, h[j].x, dwl[o]); dwl[o] = __builtin_fmaf(go[j][o].y, h[j].y, dwl[o]);
              dwl[o] = __builtin_fmaf(go[j][o].z, h[j].z, dwl[o]); dwl[o] = __builtin_fmaf(go[j][o].w, h[j].w, dwl[o]);
              g.x = __builtin_fmaf(go[j][o].x, wl_n[o], g.x); g.y = __builtin_fmaf(go[j][o].y, wl_n[o], g.y);
              g.z = __builtin_fmaf(go[j][o].z, wl_n[o], g.z); g.w = __builtin_fmaf(go[j][o].w, wl_n[o], g.w);
            }
            g = mask4(g, h[j]);
            const int s = s0 + 4 * j;
            *(v4f *)(hlt + gp_n * LS + s) = g;
            if (D3) {
              db1 += (g.x + g.y) + (g.z + g.w);
              float *gp = g2 + s * LS + gp_n;
              gp[0] = g.x; gp[LS] = g.y; gp[2 * LS] = g.z; gp[3 * LS] = g.w;
            }
          }
        };
        const int ns = S / L.wpb;      // 8 (one block of 64 features), 16 or 32 samples
        if (D3 || ns == 8) {
          pass(std::integral_constant<int, 2>(), gp_sr * ns);
        } else {
          constexpr int CH = O <= 2 ? 4 : 2;
          for (int c0 = 0; c0 < ns; c0 += 4 * CH) pass(std::integral_constant<int, CH>(), gp_sr * ns + c0);
        }
      }
      __syncthreads();
      STAMP(4);
      if (D3 && bwd) {
        // ---- dW1[n][k] += sum_s g2[n][s] h1[k][s]
        if (T1 == 2 && own1[T1 - 1]) {
          mma2<LS, LS>(h2t + r1[0] * LS, h2t + r1[T1 - 1] * LS, h1t + c1[0] * LS, h1t + c1[T1 - 1] * LS, S, acc1[0], acc1[T1 - 1], i, q);
        } else {
#pragma unroll
          for (int u = 0; u < T1; ++u)
            if (own1[u]) acc1[u] = mma<LS, LS>(h2t + r1[u] * LS, h1t + c1[u] * LS, S, acc1[u], i, q);
        }
        STAMP(5);
        // ---- g1[s][k] = (sum_n g2[s][n] W1[n][k]) (h1[s][k] > 0) as C[s][k] tiles -> [feature][sample] image
        {
          const int ntl = 4 * (H1 / 16);
          for (int t0 = w; t0 < ntl; t0 += 2 * NW) {
            const bool two = t0 + NW < ntl;
            const int k0 = (t0 >> 2) * 16, k1 = two ? ((t0 + NW) >> 2) * 16 : k0;
            v4f c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
            mma2<LS, LS>(g2 + f_s0 * LS, g2 + f_s0 * LS, w1t + k0 * LS, w1t + k1 * LS, H2, c0, c1, i, q);
            const int at0 = (k0 + i) * LS + f_s0 + 4 * q, at1 = (k1 + i) * LS + f_s0 + 4 * q;
            *(v4f *)(g1t + at0) = mask4(c0, *(const v4f *)(h1t + at0));
            if (two) *(v4f *)(g1t + at1) = mask4(c1, *(const v4f *)(h1t + at1));
          }
        }
        __syncthreads();
      }
      STAMP(6);
      // ---- dW0'[n][j] += sum_s g1[n][s] x'[j][s]
      if (!bwd) {
      } else if (WIDE) {      // a row tile's two column tiles share its operand
#pragma unroll
        for (int u = 0; u < T1; ++u)
          if (own0[u]) {
            if (nck1 > 0) mma2<LS, LS>(g1t + (w + NW * u) * 16 * LS, g1t + (w + NW * u) * 16 * LS, XT, XT + 16 * LS, S, acc0[u][0], acc0[u][CT - 1], i, q);
            else acc0[u][0] = mma<LS, LS>(g1t + (w + NW * u) * 16 * LS, XT, S, acc0[u][0], i, q);      // (five or six actions over <= 15 observations)
          }
      } else if (T1 == 2 && own0[T1 - 1]) {
        mma2<LS, LS>(g1t + w * 16 * LS, g1t + (w + NW) * 16 * LS, XT, XT, S, acc0[0][0], acc0[T1 - 1][0], i, q);
      } else {
#pragma unroll
        for (int u = 0; u < T1; ++u)
          if (own0[u]) acc0[u][0] = mma<LS, LS>(g1t + (w + NW * u) * 16 * LS, XT, S, acc0[u][0], i, q);
      }
      STAMP(7);
      // the next tile's x' images are in the other buffers (stored at the top of this tile); its loss inputs become current
      if (more) {
        cur ^= 1;
        cur_adv = nxt_adv; cur_ret = nxt_ret;
        cur_act[0] = nxt_act[0]; cur_old[0] = nxt_old[0];
      }
      STAMP(8);
      if (t + 1 < ntile) __syncthreads();
    }
    const int t = 0;      // (for the stamps below)
    (void)t;
    if (TILES && P.fwd_only) return;

    // ---- the update.  Layer 0's gradient tile -> LDS (its state is spread one element per thread); partial sums of
    // the lane = feature pass; the loss values and the gradients that come out of them.  (Two hidden layers: the
    // partial sums lie over h1 / h1t and layer 0's gradient over h2t -- every wave is past the barrier behind the last
    // product that reads those.)
#pragma unroll
    for (int u = 0; u < T1; ++u) {
      if (own0[u] && !WIDE) {      // (the wide form writes the accumulators to the partial buffer itself, below)
        float *gp = gw0 + ((w + NW * u) * 16 + 4 * q) * LDK + i;
        gp[0] = acc0[u][0].x; gp[LDK] = acc0[u][0].y; gp[2 * LDK] = acc0[u][0].z; gp[3 * LDK] = acc0[u][0].w;
        acc0[u][0] = (v4f){0, 0, 0, 0};
      }
    }
    if (gp_on) {
#pragma unroll
      for (int o = 0; o < O; ++o) part[(gp_sr * L.PO + o) * L.HLP + gp_n] = dwl[o];
      if (D3) part[(gp_sr * L.PO + O) * L.HLP + gp_n] = db1;
    }
    float *lout = TILES ? P.lpart + ((size_t)role * gridDim.y + blockIdx.y) * 4 : P.losses + 4 * k;
    if (LW) {
      // every loss wave's sums; wave 0 adds the dimensions' loss sums in order behind the barrier (`red` is free: the last
      // tile's loss phase, which read it, is two barriers back)
      if (w < O) {
        const float s_min = trl_wave_sum(a_min), s_ent = trl_wave_sum(a_ent);
        const float gb = trl_wave_sum(a_gb[0]), gl = trl_wave_sum(a_dls[0]);
        // the entropy term's share of d loss / d log_std[a], d(-beta * mean(ent)) = -beta / A: once per batch
        const float ent_share = (!TILES || blockIdx.y == 0) ? -P.beta / (float)O : 0.f;
        if (lane == 0) {
          red[w] = s_min; red[8 + w] = s_ent;
          tailg[w] = gb;
          tailg[L.o_ls - L.o_bl + w] = gl + ent_share;
        }
      }
    } else if (w == 0) {
      if (role == 0) {
        const float s_crit = trl_wave_sum(a_crit), gb = trl_wave_sum(a_gb[0]);
        if (lane == 0) {
          lout[1] = TILES ? s_crit : s_crit * (1.0f / (float)nb);      // (tiles mode: the slab's raw sum; the reduce launch normalises)
          tailg[0] = gb;
        }
      } else {
        const float invBA = 1.0f / (float)(nb * O);
        const float s_min = trl_wave_sum(a_min), s_ent = trl_wave_sum(a_ent);
        if (lane == 0) {
          lout[0] = TILES ? s_min : -(s_min * invBA);
          lout[2] = TILES ? s_ent : s_ent * invBA;
        }
        // the entropy term's share of d loss / d log_std[a], d(-beta * mean(ent)) = -beta / A: once per batch
        const float ent_share = (!TILES || blockIdx.y == 0) ? -P.beta / (float)O : 0.f;
#pragma unroll
        for (int a = 0; a < O; ++a) {
          const float gb = trl_wave_sum(a_gb[a]), gl = trl_wave_sum(a_dls[a]);
          if (lane == 0) {
            tailg[a] = gb;
            if (KIND == 0) tailg[L.o_ls - L.o_bl + a] = gl + ent_share;      // (A2C: no log_std)
          }
        }
      }
    }
    __syncthreads();
    if (LW && tid == 0) {
      float s_min = 0.f, s_ent = 0.f;
      for (int a = 0; a < O; ++a) { s_min += red[a]; s_ent += red[8 + a]; }
      const float invBA = 1.0f / (float)(nb * O);
      lout[0] = TILES ? s_min : -(s_min * invBA);
      lout[2] = TILES ? s_ent : s_ent * invBA;
    }
    STAMP(9);
    if (TILES) {
      // ---- partial sums of this slab -> the workgroup's plane of the partial buffer (the reduce launch adds the planes
      // in order and applies clip + Adam)
      float *gp = P.gpart + (size_t)blockIdx.y * P.n_flat;
      for (int e = tid; e < L.NV; e += NT) {
        float g;
        if (e < L.o_b1) {
          const int o = e / HL, n = e - o * HL;
          g = 0.f;
          for (int sr = 0; sr < L.wpb; ++sr) g += part[(sr * L.PO + o) * L.HLP + n];
        } else if (e < L.o_bl) {
          const int n = e - L.o_b1;
          g = 0.f;
          for (int sr = 0; sr < L.wpb; ++sr) g += part[(sr * L.PO + O) * L.HLP + n];
        } else {
          g = tailg[e - L.o_bl];
        }
        gp[vp_global(e)] = g;
      }
      if (WIDE) {
#pragma unroll
        for (int u = 0; u < T1; ++u)
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            if (!own0[u]) continue;
            const float g[4] = {acc0[u][c].x, acc0[u][c].y, acc0[u][c].z, acc0[u][c].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int64_t go = w0_global(((w + NW * u) * 16 + 4 * q + r) * KP + 16 * c + i);
              if (go >= 0) gp[go] = g[r];
            }
          }
      }
#pragma unroll
      for (int u = 0; u < (WIDE ? 0 : U0); ++u) {
        const int e = tid + u * NT;
        if (e < n0e) {
          const int64_t go = w0_global(e);
          if (go >= 0) gp[go] = gw0[(e / KP) * LDK + (e % KP)];
        }
      }
#pragma unroll
      for (int u = 0; u < T1; ++u) {
        if (own1[u]) {
          const float g[4] = {acc1[u].x, acc1[u].y, acc1[u].z, acc1[u].w};
#pragma unroll
          for (int r = 0; r < 4; ++r) gp[N.w_off[1] + (int64_t)(r1[u] + 4 * q + r) * H1 + c1[u] + i] = g[r];
        }
      }
      return;
    }
    // (by value, the per-step scalars made wave-uniform: the block must live in registers, not on the stack)
    auto uni = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); };
    const AdamV sc = {P.w1, P.beta2, P.c2, P.eps, P.clip, uni(step_k.x), uni(step_k.z), uni(step_k.y)};
    const bool last = k + 1 == K;
    STAMP(10);
    STAMP(11);
    // The update of everything this wave owns as ONE straight-line stream per combination of what it owns, so that the
    // adam4s -- each a dependent chain of ~50 instructions (square root, two divisions) -- overlap one another instead
    // of running one after the other behind wave-uniform branches:
    //   * the small vectors, dealt from the far end of the workgroup (layer 0's groups start at thread 0): thread
    //     NT - 1 - T takes group T of four consecutive elements of [wL | b1] -- its gradient the partial planes of the
    //     lane = feature pass added in plane order -- or, behind the groups, ONE element of [bL | log_std];
    //   * layer 0: the thread's groups of four (the clipped gradient goes back to gw0: the last update's is written out
    //     after the loop);
    //   * W1's accumulator tiles (two hidden layers), merged into the stream when the wave owns both of its slots.
    const int T = NT - 1 - tid, nvec = L.o_bl >> 2, nsmall = nvec + (L.NV - L.o_bl);
    auto tile_update = [&](int u) {
      float *wp = w1 + (r1[u] + 4 * q) * LS + c1[u] + i;
      v4f p4 = {wp[0], wp[LS], wp[2 * LS], wp[3 * LS]};
      adam4(p4, acc1[u], m1[u], v1[u], sc);
      wp[0] = p4.x; wp[LS] = p4.y; wp[2 * LS] = p4.z; wp[3 * LS] = p4.w;
      *(v4f *)(w1t + (c1[u] + i) * LS + r1[u] + 4 * q) = p4;
      if (!last) acc1[u] = (v4f){0, 0, 0, 0};      // (the last update's clipped gradient is written out after the loop)
    };
    auto update = [&](auto sv_c, auto nl0_c, auto t2_c) {
      constexpr bool SV = decltype(sv_c)::value, T2 = decltype(t2_c)::value;
      constexpr int NL0 = decltype(nl0_c)::value;
      // ---- reads
      const bool sv_on = SV && T < nsmall, vec = T < nvec;
      const int e0 = vec ? 4 * T : L.o_bl + (T - nvec);
      v4f sg = {0, 0, 0, 0}, sp = {0, 0, 0, 0}, sm = {0, 0, 0, 0}, svv = {0, 0, 0, 0};
      if (SV) {
        if (sv_on && vec) {
          int plane = O, n = e0 - L.o_b1;      // (b1's plane, two hidden layers)
          if (e0 < L.o_b1) {
            plane = 0; n = e0;
#pragma unroll
            for (int o = 1; o < O; ++o)
              if (e0 >= o * HL) { plane = o; n = e0 - o * HL; }
          }
          v4f pl[NW];
#pragma unroll
          for (int sr = 0; sr < NW; ++sr) {      // (always a valid address: planes beyond the layout's count read plane 0 and count as zeros)
            const v4f x = *(const v4f *)(part + ((sr < L.wpb ? sr : 0) * L.PO + plane) * L.HLP + n);
            pl[sr] = sr < L.wpb ? x : (v4f){0, 0, 0, 0};
          }
#pragma unroll
          for (int sr = 0; sr < NW; ++sr) sg = sg + pl[sr];
          sp = *(const v4f *)(vpp + e0); sm = *(const v4f *)(vpm + e0); svv = *(const v4f *)(vpv + e0);
        } else if (sv_on) {
          sg.x = tailg[T - nvec]; sp.x = vpp[e0]; sm.x = vpm[e0]; svv.x = vpv[e0];
        }
      }
      int at0[NL0 > 0 ? NL0 : 1][4];
      v4f lp[NL0 > 0 ? NL0 : 1], lg[NL0 > 0 ? NL0 : 1];
      bool l_on[NL0 > 0 ? NL0 : 1];
#pragma unroll
      for (int u = 0; u < NL0; ++u) {
        l_on[u] = 4 * (tid + u * NT) < n0v;
        group_at(u, at0[u]);
        float pp[4], gg[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int at = at0[u][r] >= 0 ? at0[u][r] : (at0[u][0] >= 0 ? at0[u][0] : 0);      // (always a valid address)
          pp[r] = w0p[at]; gg[r] = gw0[at];
        }
        lp[u] = (v4f){pp[0], pp[1], pp[2], pp[3]}; lg[u] = (v4f){gg[0], gg[1], gg[2], gg[3]};
      }
      v4f tp[T1];
      if (T2) {
#pragma unroll
        for (int u = 0; u < T1; ++u) {
          const float *wp = w1 + (r1[u] + 4 * q) * LS + c1[u] + i;
          tp[u] = (v4f){wp[0], wp[LS], wp[2 * LS], wp[3 * LS]};
        }
      }
      // ---- the arithmetic, one block
      if (SV) adam4(sp, sg, sm, svv, sc);
#pragma unroll
      for (int u = 0; u < NL0; ++u) adam4(lp[u], lg[u], m0[u], v0[u], sc);
      if (T2) {
#pragma unroll
        for (int u = 0; u < T1; ++u) adam4(tp[u], acc1[u], m1[u], v1[u], sc);
      }
      // ---- writes
      if (SV && sv_on) {
        if (vec) {
          *(v4f *)(vpp + e0) = sp; *(v4f *)(vpm + e0) = sm; *(v4f *)(vpv + e0) = svv;
          if (last) {      // the gradients the last learn() leaves behind (after clip_grad_value_)
            float *gd = P.grad + vp_global(e0);
            gd[0] = sg.x; gd[1] = sg.y; gd[2] = sg.z; gd[3] = sg.w;
          }
        } else {
          vpp[e0] = sp.x; vpm[e0] = sm.x; vpv[e0] = svv.x;
          if (last) P.grad[vp_global(e0)] = sg.x;
        }
      }
#pragma unroll
      for (int u = 0; u < NL0; ++u) {
        if (l_on[u]) {
          const float po[4] = {lp[u].x, lp[u].y, lp[u].z, lp[u].w}, go[4] = {lg[u].x, lg[u].y, lg[u].z, lg[u].w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (at0[u][r] >= 0) { w0p[at0[u][r]] = po[r]; gw0[at0[u][r]] = go[r]; }
          }
        }
      }
      if (T2) {
#pragma unroll
        for (int u = 0; u < T1; ++u) {
          float *wp = w1 + (r1[u] + 4 * q) * LS + c1[u] + i;
          wp[0] = tp[u].x; wp[LS] = tp[u].y; wp[2 * LS] = tp[u].z; wp[3 * LS] = tp[u].w;
          *(v4f *)(w1t + (c1[u] + i) * LS + r1[u] + 4 * q) = tp[u];
          if (!last) acc1[u] = (v4f){0, 0, 0, 0};
        }
      } else if (D3) {
#pragma unroll
        for (int u = 0; u < T1; ++u)
          if (own1[u]) tile_update(u);
      }
    };
    {
      // what this WAVE owns (wave-uniform: its last lane has the smallest T, its first lane the smallest group number)
      const bool sv_wave = NT - 64 * (w + 1) < nsmall;
      const int nl0_wave = 4 * (64 * w) < n0v ? ((G0 > 1 && 4 * (64 * w + NT) < n0v) ? 2 : 1) : 0;
      const bool t2 = D3 && T1 == 2 && own1[0] && own1[T1 - 1];
      using std::integral_constant;
#define TRL_UPD(sv, nl0, tt) update(integral_constant<bool, sv>(), integral_constant<int, nl0>(), integral_constant<bool, tt>())
      if (D3) {      // (at most 64 wide: layer 0 is at most 256 groups, the small vectors at most 88: never in one wave)
        if (t2) {
          if (sv_wave) TRL_UPD(true, 0, D3); else if (nl0_wave) TRL_UPD(false, 1, D3); else TRL_UPD(false, 0, D3);
        } else {
          if (sv_wave) TRL_UPD(true, 0, false); else if (nl0_wave) TRL_UPD(false, 1, false); else TRL_UPD(false, 0, false);
        }
      } else {
        if (sv_wave) {
          if (nl0_wave == 2) TRL_UPD(true, 2, false); else if (nl0_wave == 1) TRL_UPD(true, 1, false); else TRL_UPD(true, 0, false);
        } else {
          if (nl0_wave == 2) TRL_UPD(false, 2, false); else if (nl0_wave == 1) TRL_UPD(false, 1, false);
        }
      }
#undef TRL_UPD
    }
    STAMP(12);
    __syncthreads();
    STAMP(13);
  }

  // ---- parameters, moments and the last update's clipped gradients (what the last learn() leaves in .grad) back to the
  // flat buffers
#pragma unroll
  for (int u = 0; u < T1; ++u) {
    if (own1[u]) {
      const float m[4] = {m1[u].x, m1[u].y, m1[u].z, m1[u].w}, v[4] = {v1[u].x, v1[u].y, v1[u].z, v1[u].w},
                  gl[4] = {acc1[u].x, acc1[u].y, acc1[u].z, acc1[u].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t g = N.w_off[1] + (int64_t)(r1[u] + 4 * q + r) * H1 + c1[u] + i;
        P.param[g] = w1[(r1[u] + 4 * q + r) * LS + c1[u] + i]; P.m[g] = m[r]; P.v[g] = v[r]; P.grad[g] = gl[r];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < G0; ++u) {
    const float m[4] = {m0[u].x, m0[u].y, m0[u].z, m0[u].w}, v[4] = {v0[u].x, v0[u].y, v0[u].z, v0[u].w};
    int at0[4];
    group_at(u, at0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int at = at0[r];
      if (at >= 0) {
        const int64_t g = w0_global((at / LDK) * KP + at % LDK);
        P.param[g] = w0p[at]; P.m[g] = m[r]; P.v[g] = v[r]; P.grad[g] = gw0[at];
      }
    }
  }
  for (int e = tid; e < L.NV; e += NT) {
    const int64_t g = vp_global(e);
    P.param[g] = vpp[e]; P.m[g] = vpm[e]; P.v[g] = vpv[e];
  }
}

template <bool D3, int AO>
__global__ __launch_bounds__(NT) void k_ppo_epoch(const Args P) {
  extern __shared__ __align__(16) float lds[];
  if (blockIdx.x == 0) epoch_body<0, D3, 1, false>(P, lds);
  else epoch_body<1, D3, AO, false>(P, lds);
}

// One big batch: workgroup (role = blockIdx.x, g = blockIdx.y) carries slab g of the rows through the same tile pipeline
// and leaves its partial gradient sums; k_tiles_reduce adds the slabs in order and applies clip_grad_value_ + Adam.
template <bool D3, int AO, int KIND = 0, int KPT = 16>
__global__ __launch_bounds__(NT) void k_ppo_tiles(const Args P) {
  extern __shared__ __align__(16) float lds[];
  if (blockIdx.x + P.role_base == 0) epoch_body<0, D3, 1, true, KIND, KPT>(P, lds);
  else epoch_body<1, D3, AO, true, KIND, KPT>(P, lds);
}

struct Segs {
  int n;
  int64_t off[16];
  int len[16];
  int start[17];      // prefix sums of len
};

__global__ __launch_bounds__(256) void k_tiles_reduce(const float *__restrict__ gpart, const float *__restrict__ lpart, int G,
                                                      int64_t n_flat, Segs sg, float *param, float *grad, float *m, float *v,
                                                      TrlAdamScalars sc, int update, int64_t n_rows, int A,
                                                      float *__restrict__ losses) {
  const int total = sg.start[sg.n];
  for (int t = blockIdx.x * 256 + threadIdx.x; t < total; t += gridDim.x * 256) {
    int si = 0;
#pragma unroll
    for (int j = 1; j < 16; ++j)
      if (j < sg.n && t >= sg.start[j]) si = j;
    const int64_t e = sg.off[si] + (t - sg.start[si]);
    float g = 0.f;
    for (int gi = 0; gi < G; ++gi) g += gpart[(size_t)gi * n_flat + e];      // slab order: reproducible
    grad[e] = g;
    if (update) trl_adam_elem(param + e, grad + e, m + e, v + e, sc);
  }
  if (blockIdx.x == 0 && threadIdx.x < 3) {
    // the loss values: the slabs' raw sums added in double (128 partial sums of up to millions of terms each), then the
    // means.  [1]: critic, from the critic's workgroups; [0] = -mean(min(surr1, surr2)) and [2] = mean(entropy), the actor's
    const int role = threadIdx.x == 1 ? 0 : 1;
    double a = 0.0;
    for (int gi = 0; gi < G; ++gi) a += (double)lpart[((size_t)role * G + gi) * 4 + threadIdx.x];
    const double cnt = threadIdx.x == 1 ? (double)n_rows : (double)n_rows * (double)A;
    losses[threadIdx.x] = (float)((threadIdx.x == 0 ? -a : a) / cnt);
  }
}

__global__ void k_tiles_total(float *losses, float alpha, float beta) {
  if (threadIdx.x == 0 && blockIdx.x == 0) losses[3] = (losses[0] + alpha * losses[1]) - beta * losses[2];
}

constexpr int64_t MAX_LDS_BYTES = 160 * 1024;

int fill_net(const char *fn, const char *name, const trl_mlp_desc *d, const trl_adam_args *adam, int O, Net &n, int kp = KP) {
  TRL_REQUIRE(d != nullptr, "%s: %s is NULL", fn, name);
  TRL_REQUIRE(d->depth == 2 || d->depth == 3, "%s: %s: depth 2 or 3 (one or two hidden layers), got %d", fn, name, d->depth);
  n.depth = d->depth;
  n.d0 = d->dims[0];
  n.H1 = d->dims[1];
  n.H2 = d->depth == 3 ? d->dims[2] : 0;
  n.O = d->dims[d->depth];
  TRL_REQUIRE(n.O == O, "%s: %s: fan-out %d, expected %d", fn, name, n.O, O);
  TRL_REQUIRE(n.d0 >= 1 && n.d0 <= kp - 1, "%s: %s: fan-in %d must be in [1, %d]", fn, name, n.d0, kp - 1);
  TRL_REQUIRE(n.H1 >= 16 && n.H1 % 16 == 0 && n.H1 <= 256, "%s: %s: hidden width %d must be a multiple of 16 in [16, 256]", fn,
              name, n.H1);
  if (d->depth == 3) {
    TRL_REQUIRE(n.H2 >= 16 && n.H2 % 16 == 0 && n.H1 <= 64 && n.H2 <= 64,
                "%s: %s: two hidden layers must be multiples of 16 of at most 64 (got %d, %d)", fn, name, n.H1, n.H2);
  }
  for (int l = 0; l < d->depth; ++l) {
    TRL_REQUIRE(d->w[l] && d->b[l], "%s: %s has a NULL parameter", fn, name);
    const int64_t wo = d->w[l] - adam->param, bo = d->b[l] - adam->param;
    TRL_REQUIRE(wo >= 0 && wo + (int64_t)d->dims[l + 1] * d->dims[l] <= adam->n && bo >= 0 && bo + d->dims[l + 1] <= adam->n,
                "%s: %s: parameters must lie inside the optimizer's flat buffer", fn, name);
    n.w_off[l] = wo;
    n.b_off[l] = bo;
  }
  for (int l = d->depth; l < 3; ++l) n.w_off[l] = n.b_off[l] = 0;
  n.ls_off = -1;
  return TRL_OK;
}

typedef void (*EpochKernel)(const Args);

EpochKernel pick_kernel(bool d3, int A) {
  switch (A) {
    case 1: return d3 ? k_ppo_epoch<true, 1> : k_ppo_epoch<false, 1>;
    case 2: return d3 ? k_ppo_epoch<true, 2> : k_ppo_epoch<false, 2>;
    case 3: return d3 ? k_ppo_epoch<true, 3> : k_ppo_epoch<false, 3>;
    default: return d3 ? k_ppo_epoch<true, 4> : k_ppo_epoch<false, 4>;
  }
}

EpochKernel pick_tiles_kernel_wide(int kind, int A) {
  if (kind == 1) {
    switch (A) {
      case 2: return k_ppo_tiles<false, 2, 1, KPW>;
      case 3: return k_ppo_tiles<false, 3, 1, KPW>;
      case 4: return k_ppo_tiles<false, 4, 1, KPW>;
      case 5: return k_ppo_tiles<false, 5, 1, KPW>;
      default: return k_ppo_tiles<false, 6, 1, KPW>;
    }
  }
  switch (A) {
    case 1: return k_ppo_tiles<false, 1, 0, KPW>;
    case 2: return k_ppo_tiles<false, 2, 0, KPW>;
    case 3: return k_ppo_tiles<false, 3, 0, KPW>;
    case 4: return k_ppo_tiles<false, 4, 0, KPW>;
    case 5: return k_ppo_tiles<false, 5, 0, KPW>;
    default: return k_ppo_tiles<false, 6, 0, KPW>;
  }
}

EpochKernel pick_tiles_kernel_a2c(bool d3, int A) {
  switch (A) {
    case 2: return d3 ? k_ppo_tiles<true, 2, 1> : k_ppo_tiles<false, 2, 1>;
    case 3: return d3 ? k_ppo_tiles<true, 3, 1> : k_ppo_tiles<false, 3, 1>;
    default: return d3 ? k_ppo_tiles<true, 4, 1> : k_ppo_tiles<false, 4, 1>;
  }
}

EpochKernel pick_tiles_kernel(bool d3, int A) {
  switch (A) {
    case 1: return d3 ? k_ppo_tiles<true, 1> : k_ppo_tiles<false, 1>;
    case 2: return d3 ? k_ppo_tiles<true, 2> : k_ppo_tiles<false, 2>;
    case 3: return d3 ? k_ppo_tiles<true, 3> : k_ppo_tiles<false, 3>;
    default: return d3 ? k_ppo_tiles<true, 4> : k_ppo_tiles<false, 4>;
  }
}

// slabs of a batch of n rows: whole 64-row tiles, at most MAX_SLABS of them (two workgroups a slab: one per CU), or twice
// that when the roles are launched one after the other
constexpr int MAX_SLABS = 128;
void slab_plan(int64_t n, int64_t *slab, int *G, int max_slabs = MAX_SLABS) {
  const int64_t tiles = (n + S - 1) / S;
  const int64_t per = (tiles + max_slabs - 1) / max_slabs;
  *slab = per * S;
  *G = (int)((n + *slab - 1) / *slab);
}

}  // namespace

#ifdef TRL_TUNING
namespace {
// diagnostic build: the fast Adam of this file against trl_adam_elem on caller-supplied elements; counts elements whose
// parameter, clipped gradient or moments differ in any bit
__global__ void k_adam_compare(const float *p, const float *g, const float *m, const float *v, int64_t n, TrlAdamScalars ref,
                               AdamV fast, unsigned long long *mismatch) {
  const int64_t e0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (e0 + 3 >= n) return;
  v4f p4 = {p[e0], p[e0 + 1], p[e0 + 2], p[e0 + 3]}, g4 = {g[e0], g[e0 + 1], g[e0 + 2], g[e0 + 3]};
  v4f m4 = {m[e0], m[e0 + 1], m[e0 + 2], m[e0 + 3]}, v4 = {v[e0], v[e0 + 1], v[e0 + 2], v[e0 + 3]};
  adam4(p4, g4, m4, v4, fast);
  const float pf[4] = {p4.x, p4.y, p4.z, p4.w}, gf[4] = {g4.x, g4.y, g4.z, g4.w}, mf[4] = {m4.x, m4.y, m4.z, m4.w},
              vf[4] = {v4.x, v4.y, v4.z, v4.w};
  unsigned bad = 0;
  for (int r = 0; r < 4; ++r) {
    float pr = p[e0 + r], gr = g[e0 + r], mr = m[e0 + r], vr = v[e0 + r];
    trl_adam_elem(&pr, &gr, &mr, &vr, ref);
    bad += __builtin_bit_cast(unsigned, pr) != __builtin_bit_cast(unsigned, pf[r]) ||
           __builtin_bit_cast(unsigned, gr) != __builtin_bit_cast(unsigned, gf[r]) ||
           __builtin_bit_cast(unsigned, mr) != __builtin_bit_cast(unsigned, mf[r]) ||
           __builtin_bit_cast(unsigned, vr) != __builtin_bit_cast(unsigned, vf[r]);
  }
  if (bad) atomicAdd(mismatch, (unsigned long long)bad);
}
}  // namespace

extern "C" int trl_tune_adam_compare(const float *p, const float *g, const float *m, const float *v, int64_t n, double lr,
                                     double beta1, double beta2, double eps, int64_t step, double clip_value,
                                     unsigned long long *mismatch_d, void *stream) {
  const TrlAdamScalars ref = trl_adam_scalars(lr, beta1, beta2, eps, step, 1.0, clip_value);
  AdamV f;
  f.w1 = ref.w1; f.beta2 = ref.beta2; f.c2 = ref.c2; f.eps = ref.eps;
  f.clip = clip_value > 0 ? (float)clip_value : __builtin_inff();
  f.c = ref.bc2_sqrt; f.rc = (float)(1.0 / (double)ref.bc2_sqrt); f.neg_step = ref.neg_step;
  hipLaunchKernelGGL(k_adam_compare, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, ref,
                     f, mismatch_d);
  return hipGetLastError() == hipSuccess ? TRL_OK : TRL_EHIP;
}

extern "C" int trl_tune_epoch_stamps(unsigned long long *out_h /*[2][8][16]*/) {
  return hipMemcpyFromSymbol(out_h, HIP_SYMBOL(g_epoch_stamp), sizeof(unsigned long long) * 256) == hipSuccess ? TRL_OK : TRL_EHIP;
}
#endif

extern "C" int64_t trl_ppo_learn_epoch_workspace_bytes(int64_t n_rows, int64_t minibatch, int32_t n_epochs) {
  if (n_rows <= 0 || minibatch <= 0 || n_epochs <= 0) return 0;
  return ((n_rows + minibatch - 1) / minibatch) * n_epochs * 4 * (int64_t)sizeof(float);
}

// 1 when trl_ppo_learn_epoch takes these networks (shapes and the LDS they need), 0 otherwise; no error text is set
extern "C" int trl_ppo_learn_epoch_supported(const trl_mlp_desc *critic, const trl_mlp_desc *actor, int32_t A) {
  if (!critic || !actor || A < 1 || A > MAXO || critic->depth != actor->depth) return 0;
  Net n[2];
  const trl_mlp_desc *d[2] = {critic, actor};
  for (int r = 0; r < 2; ++r) {
    const int dep = d[r]->depth;
    if (dep != 2 && dep != 3) return 0;
    n[r].depth = dep; n[r].d0 = d[r]->dims[0]; n[r].H1 = d[r]->dims[1]; n[r].H2 = dep == 3 ? d[r]->dims[2] : 0;
    n[r].O = d[r]->dims[dep]; n[r].ls_off = r == 1 ? 0 : -1;
    if (n[r].O != (r == 0 ? 1 : A) || n[r].d0 < 1 || n[r].d0 > KP - 1 || n[r].H1 < 16 || n[r].H1 % 16 || n[r].H1 > 256) return 0;
    if (dep == 3 && (n[r].H2 < 16 || n[r].H2 % 16 || n[r].H1 > 64 || n[r].H2 > 64)) return 0;
    if ((int64_t)make_lay(n[r]).total * 4 > MAX_LDS_BYTES) return 0;
  }
  return n[0].d0 == n[1].d0 ? 1 : 0;
}

extern "C" int trl_ppo_learn_epoch(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std,
                                   const float *obs, const float *action, const float *returns, const float *old_logp,
                                   const float *adv, int64_t n_rows, int64_t minibatch, int32_t n_epochs, int32_t A,
                                   double clip, double alpha, double beta, float *losses_out, const trl_adam_args *adam,
                                   void *ws, void *stream) {
  const char *fn = "trl_ppo_learn_epoch";
  TRL_REQUIRE(adam && adam->param && adam->grad && adam->exp_avg && adam->exp_avg_sq && adam->n > 0,
              "%s: the optimizer block (flat parameter, gradient and moment buffers) is required", fn);
  TRL_REQUIRE(!adam->dev_state, "%s: the step count must be on the host (not graph mode)", fn);
  TRL_REQUIRE(adam->step >= 1, "%s: adam->step is the 1-based count of the FIRST update of this call", fn);
  TRL_REQUIRE(adam->grad_scale == 1.0, "%s: single-rank update (grad_scale must be 1)", fn);
  TRL_REQUIRE(A >= 1 && A <= MAXO, "%s: action size %d must be in [1, %d]", fn, A, MAXO);
  TRL_REQUIRE(n_rows >= 1 && minibatch >= 1 && n_epochs >= 1, "%s: n_rows, minibatch and n_epochs must be positive", fn);
  TRL_REQUIRE(minibatch <= 4096, "%s: minibatch %lld: this entry is for small minibatches (<= 4096); larger ones are "
              "trl_ppo_learn_fused's", fn, (long long)minibatch);
  TRL_REQUIRE(obs && action && returns && old_logp && adv && losses_out && log_std && ws, "%s: NULL argument", fn);
  TRL_REQUIRE(((uintptr_t)ws & 15) == 0, "%s: the workspace must be 16 B aligned", fn);
  Args P = {};
  int rc;
  if ((rc = fill_net(fn, "critic", critic, adam, 1, P.net[0])) || (rc = fill_net(fn, "actor", actor, adam, A, P.net[1])))
    return rc;
  TRL_REQUIRE(P.net[0].d0 == P.net[1].d0, "%s: critic and actor must see the same observation", fn);
  TRL_REQUIRE(P.net[0].depth == P.net[1].depth, "%s: critic and actor must have the same depth", fn);
  const int64_t lo = log_std - adam->param;
  TRL_REQUIRE(lo >= 0 && lo + A <= adam->n, "%s: log_std must lie inside the optimizer's flat buffer", fn);
  P.net[1].ls_off = lo;
  const int64_t K64 = ((n_rows + minibatch - 1) / minibatch) * n_epochs;
  TRL_REQUIRE(K64 <= (1 << 24), "%s: %lld updates in one launch: too many", fn, (long long)K64);
  int64_t lds_bytes = 0;
  for (int r = 0; r < 2; ++r) {
    const int64_t b = (int64_t)make_lay(P.net[r]).total * 4;
    lds_bytes = b > lds_bytes ? b : lds_bytes;
  }
  TRL_REQUIRE(lds_bytes <= MAX_LDS_BYTES, "%s: networks of this size need %lld B of LDS (limit %lld)", fn,
              (long long)lds_bytes, (long long)MAX_LDS_BYTES);
  P.param = adam->param; P.grad = adam->grad; P.m = adam->exp_avg; P.v = adam->exp_avg_sq;
  P.obs = obs; P.action = action; P.returns = returns; P.old_logp = old_logp; P.adv = adv;
  P.n_rows = n_rows; P.B = (int)minibatch; P.n_epochs = n_epochs; P.A = A;
  P.lo = (float)(1.0 - clip); P.hi = (float)(1.0 + clip); P.alpha = (float)alpha; P.beta = (float)beta;
  P.clip = adam->clip_value > 0 ? (float)adam->clip_value : __builtin_inff();
  P.w1 = (float)(1.0 - adam->beta1); P.beta2 = (float)adam->beta2; P.c2 = (float)(1.0 - adam->beta2); P.eps = (float)adam->eps;
  P.steps = (const float *)ws;
  P.losses = losses_out;
  hipStream_t st = (hipStream_t)stream;
  const int K = (int)K64;
  EpochKernel kern = pick_kernel(P.net[0].depth == 3, A);
  if (lds_bytes > 48 * 1024 &&      // (per device and cheap: set whenever the launch needs more than the default limit)
      hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MAX_LDS_BYTES) != hipSuccess) {
    (void)hipGetLastError();
    trl_set_error("%s: could not raise the kernel's LDS limit", fn);
    return TRL_EHIP;
  }
  hipLaunchKernelGGL(k_epoch_prep, dim3((K + 255) / 256), dim3(256), 0, st, (float *)ws, K, adam->lr, adam->beta1, adam->beta2,
                     adam->eps, adam->step);
  TRL_CHECK_LAUNCH(fn);
  hipLaunchKernelGGL(kern, dim3(2), dim3(NT), (size_t)lds_bytes, st, P);
  TRL_CHECK_LAUNCH(fn);
  hipLaunchKernelGGL(k_epoch_totals, dim3((K + 255) / 256), dim3(256), 0, st, losses_out, K, P.alpha, P.beta);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

// ---- one learn() step over ONE batch of any size through the same tile pipeline, spread over the chip: the general
// whole-step path for the shapes trl_ppo_learn_epoch takes (fan-in <= 15, up to four actions, one hidden layer up to 256
// wide or two up to 64) -- some of which trl_ppo_learn_fused's register-chained kernels do not -- and, with ONE hidden
// layer, for fan-in up to 31 and up to six actions (the wide form: ActorCriticNet(17, 6, 256))
extern "C" int64_t trl_ppo_learn_tiles_workspace_bytes(int64_t n_flat, int64_t B) {
  if (n_flat <= 0 || B <= 0) return 0;
  int64_t slab; int G;
  slab_plan(B, &slab, &G, 2 * MAX_SLABS);      // (the larger of the two plans) + the advantage column of trl_a2c_learn_tiles
  return ((int64_t)G * n_flat + 2 * (int64_t)G * 4 + 64 + (B + 63) / 64 * 64) * (int64_t)sizeof(float);
}

namespace {
// kind 0: PPO (log_std, old_logp, action [B, A]); kind 1: A2C (softmax actor, action [B] = the index, no log_std / old_logp)
int learn_tiles_impl(const char *fn, int kind, const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std,
                     const float *obs, const float *action, const float *returns, const float *old_logp,
                     const float *adv, int64_t B, int32_t A, double clip, double alpha, double beta,
                     float *losses_out, const trl_adam_args *adam, int32_t update, void *ws, void *stream) {
  TRL_REQUIRE(adam && adam->param && adam->grad && adam->n > 0, "%s: the optimizer block (flat buffers) is required", fn);
  TRL_REQUIRE(!update || (adam->exp_avg && adam->exp_avg_sq && adam->step >= 1 && !adam->dev_state),
              "%s: update needs the moment buffers and a host step count >= 1", fn);
  TRL_REQUIRE(A >= (kind == 1 ? 2 : 1) && A <= MAXO_T, "%s: action size %d must be in [%d, %d]", fn, A, kind == 1 ? 2 : 1, MAXO_T);
  TRL_REQUIRE(critic && actor, "%s: NULL network", fn);
  // the wide form (32 columns at layer 0, up to six outputs) for what the 16-column kernels do not take; one hidden layer only
  const bool wide = A > MAXO || critic->dims[0] > KP - 1;
  TRL_REQUIRE(!wide || (critic->depth == 2 && actor->depth == 2),
              "%s: more than %d observations or %d actions need networks of ONE hidden layer (got depth %d)", fn, KP - 1, MAXO,
              critic->depth);
  const int kp = wide ? KPW : KP;
  TRL_REQUIRE(B >= 1 && B <= (1ll << 31) - 64, "%s: bad batch size", fn);
  TRL_REQUIRE(obs && action && returns && losses_out && ws, "%s: NULL argument", fn);
  TRL_REQUIRE(kind == 1 ? true : (adv && old_logp && log_std), "%s: NULL argument", fn);
  // One role per launch, each over twice the slabs: needed when the critic's launch is to compute the advantage for the
  // actor's (A2C, adv == NULL), and faster for any big batch -- the actor's workgroups (A outputs) take longer than the
  // critic's (one), so in one launch of both half the chip waits for the other half
  const bool own_adv = kind == 1 && adv == nullptr;
  const bool split = own_adv || B >= 4 * MAX_SLABS * S;
  TRL_REQUIRE(((uintptr_t)ws & 15) == 0, "%s: the workspace must be 16 B aligned", fn);
  Args P = {};
  int rc;
  if ((rc = fill_net(fn, "critic", critic, adam, 1, P.net[0], kp)) || (rc = fill_net(fn, "actor", actor, adam, A, P.net[1], kp)))
    return rc;
  TRL_REQUIRE(P.net[0].d0 == P.net[1].d0 && P.net[0].depth == P.net[1].depth,
              "%s: critic and actor must have the same fan-in and depth", fn);
  const int64_t lo = kind == 1 ? -1 : log_std - adam->param;
  TRL_REQUIRE(kind == 1 || (lo >= 0 && lo + A <= adam->n), "%s: log_std must lie inside the optimizer's flat buffer", fn);
  P.net[1].ls_off = lo;
  int64_t lds_bytes = 0;
  for (int r = 0; r < 2; ++r) {
    const int64_t b = (int64_t)make_lay(P.net[r], kp).total * 4;
    lds_bytes = b > lds_bytes ? b : lds_bytes;
  }
  TRL_REQUIRE(lds_bytes <= MAX_LDS_BYTES, "%s: networks of this size need %lld B of LDS (limit %lld)", fn,
              (long long)lds_bytes, (long long)MAX_LDS_BYTES);
  int G;
  slab_plan(B, &P.slab, &G, split ? 2 * MAX_SLABS : MAX_SLABS);
  P.param = adam->param; P.grad = adam->grad; P.m = adam->exp_avg; P.v = adam->exp_avg_sq;
  P.obs = obs; P.action = action; P.returns = returns; P.old_logp = old_logp; P.adv = adv;
  P.n_rows = B; P.B = 0; P.n_epochs = 1; P.A = A;
  P.lo = (float)(1.0 - clip); P.hi = (float)(1.0 + clip); P.alpha = (float)alpha; P.beta = (float)beta;
  P.clip = __builtin_inff();
  P.n_flat = adam->n;
  P.gpart = (float *)ws;
  P.lpart = P.gpart + (size_t)G * adam->n;
  P.losses = losses_out;
  float *adv_ws = P.lpart + ((size_t)2 * G * 4 + 63) / 64 * 64;
  hipStream_t st = (hipStream_t)stream;
  EpochKernel kern = wide ? pick_tiles_kernel_wide(kind, A)
                          : (kind == 1 ? pick_tiles_kernel_a2c(P.net[0].depth == 3, A) : pick_tiles_kernel(P.net[0].depth == 3, A));
  if (lds_bytes > 48 * 1024 &&
      hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MAX_LDS_BYTES) != hipSuccess) {
    (void)hipGetLastError();
    trl_set_error("%s: could not raise the kernel's LDS limit", fn);
    return TRL_EHIP;
  }
  if (split) {
    P.adv_out = own_adv ? adv_ws : nullptr;
    hipLaunchKernelGGL(kern, dim3(1, G), dim3(NT), (size_t)lds_bytes, st, P);      // the critic's workgroups
    TRL_CHECK_LAUNCH(fn);
    P.adv_out = nullptr; P.role_base = 1;
    if (own_adv) P.adv = adv_ws;
    hipLaunchKernelGGL(kern, dim3(1, G), dim3(NT), (size_t)lds_bytes, st, P);      // the actor's
  } else {
    hipLaunchKernelGGL(kern, dim3(2, G), dim3(NT), (size_t)lds_bytes, st, P);
  }
  TRL_CHECK_LAUNCH(fn);
  // the parameters' places in the flat buffers, as segments: both networks' layers and log_std
  Segs sg = {};
  auto add = [&](int64_t off, int64_t len) {
    sg.off[sg.n] = off; sg.len[sg.n] = (int)len; sg.start[sg.n + 1] = sg.start[sg.n] + (int)len; ++sg.n;
  };
  const trl_mlp_desc *nets[2] = {critic, actor};
  for (int r = 0; r < 2; ++r)
    for (int l = 0; l < nets[r]->depth; ++l) {
      add(P.net[r].w_off[l], (int64_t)nets[r]->dims[l + 1] * nets[r]->dims[l]);
      add(P.net[r].b_off[l], nets[r]->dims[l + 1]);
    }
  if (kind == 0) add(lo, A);
  TrlAdamScalars sc = {};
  if (update) sc = trl_adam_scalars(adam->lr, adam->beta1, adam->beta2, adam->eps, adam->step, adam->grad_scale, adam->clip_value);
  const int total = sg.start[sg.n];
  hipLaunchKernelGGL(k_tiles_reduce, dim3((total + 255) / 256), dim3(256), 0, st, (const float *)P.gpart, (const float *)P.lpart, G,
                     P.n_flat, sg, adam->param, adam->grad, adam->exp_avg, adam->exp_avg_sq, sc, update ? 1 : 0, B,
                     kind == 1 ? 1 : (int)A,      // (A2C's actor and entropy terms are means over the batch, PPO's over [B, A])
                     losses_out);
  TRL_CHECK_LAUNCH(fn);
  hipLaunchKernelGGL(k_tiles_total, dim3(1), dim3(64), 0, st, losses_out, P.alpha, P.beta);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}
}  // namespace

extern "C" int trl_ppo_learn_tiles(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std,
                                   const float *obs, const float *action, const float *returns, const float *old_logp,
                                   const float *adv, int64_t B, int32_t A, double clip, double alpha, double beta,
                                   float *losses_out, const trl_adam_args *adam, int32_t update, void *ws, void *stream) {
  return learn_tiles_impl("trl_ppo_learn_tiles", 0, critic, actor, log_std, obs, action, returns, old_logp, adv, B, A, clip,
                          alpha, beta, losses_out, adam, update, ws, stream);
}

extern "C" int trl_a2c_learn_tiles(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *obs,
                                   const float *action, const float *returns, const float *adv, int64_t B, int32_t A,
                                   double alpha, double beta, float *losses_out, const trl_adam_args *adam, int32_t update,
                                   void *ws, void *stream) {
  return learn_tiles_impl("trl_a2c_learn_tiles", 1, critic, actor, nullptr, obs, action, returns, nullptr, adv, B, A, 0.0,
                          alpha, beta, losses_out, adam, update, ws, stream);
}

// ---- forward only through the tile pipeline: V(obs), the actor's mean and Normal(mean, exp(log_std)).log_prob(action)
// (ppo_controller.py:32-51's forward passes; a2c_controller.py:30's values) for the shapes trl_ppo_forward_fused does not
// take -- the tiles stop behind the loss phase, whose waves write the outputs.  actor == NULL: the critic alone.
extern "C" int trl_ac_forward_tiles(const trl_mlp_desc *critic, const trl_mlp_desc *actor, const float *log_std,
                                    const float *obs, const float *action, int64_t B, int32_t A, float *value_out,
                                    float *mean_out, float *logp_out, void *stream) {
  const char *fn = "trl_ac_forward_tiles";
  TRL_REQUIRE(critic && obs && value_out, "%s: NULL argument", fn);
  TRL_REQUIRE(!actor || (log_std && (mean_out || logp_out) && (!logp_out || action)),
              "%s: the actor's pass needs log_std, an output and (for log-probs) the actions", fn);
  TRL_REQUIRE(B >= 1 && B <= (1ll << 31) - 64, "%s: bad batch size", fn);
  if (!actor) A = 1;
  TRL_REQUIRE(A >= 1 && A <= MAXO_T, "%s: action size %d must be in [1, %d]", fn, A, MAXO_T);
  const bool wide = A > MAXO || critic->dims[0] > KP - 1;
  TRL_REQUIRE(!wide || (critic->depth == 2 && (!actor || actor->depth == 2)),
              "%s: more than %d observations or %d actions need networks of ONE hidden layer", fn, KP - 1, MAXO);
  const int kp = wide ? KPW : KP;
  // the kernels address parameters as offsets into one buffer: the span of these networks' parameters
  const trl_mlp_desc *nets[2] = {critic, actor};
  const float *lo = nullptr, *hi = nullptr;
  auto span = [&](const float *p, int64_t n) {
    if (!lo || p < lo) lo = p;
    if (!hi || p + n > hi) hi = p + n;
  };
  for (int r = 0; r < (actor ? 2 : 1); ++r) {
    TRL_REQUIRE(nets[r]->depth == 2 || nets[r]->depth == 3, "%s: depth 2 or 3 (one or two hidden layers)", fn);
    for (int l = 0; l < nets[r]->depth; ++l) {
      TRL_REQUIRE(nets[r]->w[l] && nets[r]->b[l], "%s: a network has a NULL parameter", fn);
      span(nets[r]->w[l], (int64_t)nets[r]->dims[l + 1] * nets[r]->dims[l]);
      span(nets[r]->b[l], nets[r]->dims[l + 1]);
    }
  }
  if (actor) span(log_std, A);
  trl_adam_args flat = {};
  flat.param = const_cast<float *>(lo);
  flat.n = hi - lo;
  Args P = {};
  int rc;
  if ((rc = fill_net(fn, "critic", critic, &flat, 1, P.net[0], kp))) return rc;
  if (actor) {
    if ((rc = fill_net(fn, "actor", actor, &flat, A, P.net[1], kp))) return rc;
    TRL_REQUIRE(P.net[0].d0 == P.net[1].d0 && P.net[0].depth == P.net[1].depth,
                "%s: critic and actor must have the same fan-in and depth", fn);
    P.net[1].ls_off = log_std - lo;
  } else {
    P.net[1] = P.net[0];
  }
  int64_t lds_bytes = 0;
  for (int r = 0; r < (actor ? 2 : 1); ++r) {
    const int64_t b = (int64_t)make_lay(P.net[r], kp).total * 4;
    lds_bytes = b > lds_bytes ? b : lds_bytes;
  }
  TRL_REQUIRE(lds_bytes <= MAX_LDS_BYTES, "%s: networks of this size need %lld B of LDS (limit %lld)", fn,
              (long long)lds_bytes, (long long)MAX_LDS_BYTES);
  const bool split = B >= 4 * MAX_SLABS * S;
  int G;
  slab_plan(B, &P.slab, &G, split || !actor ? 2 * MAX_SLABS : MAX_SLABS);
  P.param = flat.param;
  P.obs = obs;
  // the loss inputs the forward does not use are read all the same: any readable column of the right length
  P.action = action ? action : (logp_out ? logp_out : mean_out);
  P.returns = value_out; P.adv = value_out; P.old_logp = logp_out ? logp_out : mean_out;
  P.n_rows = B; P.n_epochs = 1; P.A = A;
  P.lo = 0.f; P.hi = 2.f; P.clip = __builtin_inff();
  P.lpart = value_out;      // (never written: the kernel returns ahead of its sums)
  P.fwd_only = 1; P.fwd_value = value_out; P.fwd_mean = mean_out; P.fwd_logp = logp_out;
  hipStream_t st = (hipStream_t)stream;
  EpochKernel kern = wide ? pick_tiles_kernel_wide(0, A) : pick_tiles_kernel(P.net[0].depth == 3, A);
  if (lds_bytes > 48 * 1024 &&
      hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MAX_LDS_BYTES) != hipSuccess) {
    (void)hipGetLastError();
    trl_set_error("%s: could not raise the kernel's LDS limit", fn);
    return TRL_EHIP;
  }
  if (!actor) {
    hipLaunchKernelGGL(kern, dim3(1, G), dim3(NT), (size_t)lds_bytes, st, P);
  } else if (split) {
    hipLaunchKernelGGL(kern, dim3(1, G), dim3(NT), (size_t)lds_bytes, st, P);
    TRL_CHECK_LAUNCH(fn);
    P.role_base = 1;
    hipLaunchKernelGGL(kern, dim3(1, G), dim3(NT), (size_t)lds_bytes, st, P);
  } else {
    hipLaunchKernelGGL(kern, dim3(2, G), dim3(NT), (size_t)lds_bytes, st, P);
  }
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}
