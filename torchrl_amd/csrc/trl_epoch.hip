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
// network, each walking all minibatches of all epochs on its own CU:
//
//   * the network's parameters and both Adam moments live in REGISTERS of the lane that owns them (the lane that holds
//     the element of the weight-gradient product's accumulator tile), for the whole launch; LDS holds the copy of the
//     weights the products read; the small vectors (last layer, its bias, hidden bias, log_std) live in LDS;
//   * a minibatch is walked in tiles of 64 samples; all products are 16x16 output tiles of v_mfma_f32_16x16x4_f32
//     (exact f32) with both operands read from LDS, k four at a time per lane: one ds_read_b128 where the reduction
//     index is the contiguous one of the operand's LDS image, four ds_read_b32 (same bytes, same banks) where it is
//     the strided one -- so every activation / gradient is stored ONCE, as [feature][sample];
//   * the last layer (fan-out <= 4), the loss (k_ppo_loss's arithmetic, op for op), its backward seed and the masked
//     outer product run on the VALU with lane = feature;
//   * clip_grad_value_ + Adam (trl_adam_elem: torch's op order) are applied by the owning lane straight from the
//     accumulator; the step's bias corrections come from a table a small launch derives ahead (double precision).
//
// Results: the same arithmetic as trl_ppo_learn_fused up to the order of the f32 sums (tests/test_gpu_epoch.py: K
// consecutive learn() calls and the oracle's torch-CPU loop).
#include <string.h>

#include "trl_common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int NT = 512, NW = 8;     // threads, wavefronts per workgroup (two per SIMD)
constexpr int S = 64;               // samples per tile
constexpr int LDS_S = S + 4;        // row stride of a [feature][sample] image: 16 lanes x 16 B on 16 distinct bank quads
constexpr int KP = 16;              // fan-in + 1 (the bias column), padded to one k round
constexpr int LDK = KP + 4;
constexpr int MAXO = 4;
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
  const float *steps;        // [K][2]: bc2_sqrt, neg_step of update k (k_epoch_prep)
  float *losses;             // [K][4]
};

// LDS map of one workgroup (offsets in floats)
struct Lay {
  int xpt[2], w0p, h1t, h2t, w1, red, gout, part, vp_p, vp_m, vp_v, vp_g, total;
  int ldw1, HL, HLP, wpb, NV, o_wl, o_bl, o_b1, o_ls;
};

__host__ __device__ inline Lay make_lay(const Net &n) {
  Lay L;
  int o = 0;
  auto take = [&](int f) { int at = o; o += (f + 3) / 4 * 4; return at; };
  L.xpt[0] = take(KP * LDS_S);
  L.xpt[1] = take(KP * LDS_S);
  L.w0p = take(n.H1 * LDK);
  L.h1t = take(n.H1 * LDS_S);
  L.ldw1 = n.H1 + 4;
  L.h2t = n.depth == 3 ? take(n.H2 * LDS_S) : 0;
  L.w1 = n.depth == 3 ? take(n.H2 * L.ldw1) : 0;
  L.HL = n.depth == 3 ? n.H2 : n.H1;
  const int NB = (L.HL + 63) / 64;
  L.HLP = NB * 64;
  L.wpb = NW / NB;           // wavefronts per block of 64 features in the lane = feature pass
  L.red = take(NW * MAXO * S);
  L.gout = take(MAXO * LDS_S);
  L.part = take(L.wpb * (MAXO + 1) * L.HLP);
  L.o_wl = 0;
  L.o_bl = n.O * L.HL;
  L.o_b1 = L.o_bl + n.O;
  L.o_ls = L.o_b1 + (n.depth == 3 ? n.H2 : 0);
  L.NV = L.o_ls + (n.ls_off >= 0 ? n.O : 0);
  L.vp_p = take(L.NV);
  L.vp_m = take(L.NV);
  L.vp_v = take(L.NV);
  L.vp_g = take(L.NV);
  L.total = o;
  return L;
}

// Two 16x16 output tiles at once (independent accumulators: the f32 MFMA's dependent latency is 40 cycles against 32
// of issue).  Operand X of tile t: element (row-or-column i, reduction index k) at X[i * ld + k] when XKC (k contiguous:
// one 16 B read per four k) or X[k * ld + i] (k strided: four 4 B reads).  Lane (i = lane & 15, q = lane >> 4) supplies
// k = k0 + 4 q + c to the c-th MFMA of a round for both operands, so any k order is the same sum.
template <bool AKC, bool BKC>
__device__ __forceinline__ void mma2(const float *A0, const float *A1, int lda, const float *B0, const float *B1, int ldb,
                                     int K, v4f &acc0, v4f &acc1, int i, int q) {
  auto ld = [&](const float *X, int ldx, bool kc, int k0, float (&r)[4]) {
    if (kc) {
      const v4f t = *(const v4f *)(X + i * ldx + k0 + 4 * q);
      r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    } else {
      const float *p = X + (k0 + 4 * q) * ldx + i;
#pragma unroll
      for (int c = 0; c < 4; ++c) r[c] = p[c * ldx];
    }
  };
  float a0[4], a1[4], b0[4], b1[4];
  ld(A0, lda, AKC, 0, a0); ld(A1, lda, AKC, 0, a1); ld(B0, ldb, BKC, 0, b0); ld(B1, ldb, BKC, 0, b1);
  for (int k0 = 0; k0 < K; k0 += 16) {
    float na0[4], na1[4], nb0[4], nb1[4];
    const int kn = k0 + 16 < K ? k0 + 16 : k0;      // (the last round re-reads itself: no branch around the loads)
    ld(A0, lda, AKC, kn, na0); ld(A1, lda, AKC, kn, na1); ld(B0, ldb, BKC, kn, nb0); ld(B1, ldb, BKC, kn, nb1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c], b0[c], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c], b1[c], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { a0[c] = na0[c]; a1[c] = na1[c]; b0[c] = nb0[c]; b1[c] = nb1[c]; }
  }
}

__device__ __forceinline__ v4f relu4(v4f x) {
  x.x = fmaxf(x.x, 0.f); x.y = fmaxf(x.y, 0.f); x.z = fmaxf(x.z, 0.f); x.w = fmaxf(x.w, 0.f);
  return x;
}

// a parameter tile owned by this lane for the whole launch: accumulator element r is (row m0 + 4 q + r, column c0 + i)
struct Owned {
  v4f p, m, v, acc;
};

__device__ __forceinline__ void adam4(Owned &t, const TrlAdamScalars &s, v4f &g_out) {
  float p[4] = {t.p.x, t.p.y, t.p.z, t.p.w}, m[4] = {t.m.x, t.m.y, t.m.z, t.m.w}, v[4] = {t.v.x, t.v.y, t.v.z, t.v.w},
        g[4] = {t.acc.x, t.acc.y, t.acc.z, t.acc.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) trl_adam_elem(&p[r], &g[r], &m[r], &v[r], s);
  t.p = (v4f){p[0], p[1], p[2], p[3]};
  t.m = (v4f){m[0], m[1], m[2], m[3]};
  t.v = (v4f){v[0], v[1], v[2], v[3]};
  g_out = (v4f){g[0], g[1], g[2], g[3]};
  t.acc = (v4f){0.f, 0.f, 0.f, 0.f};
}

// the update table: bias corrections of steps step0 .. step0 + K - 1 (torch: bias_correction2 ** 0.5 and lr / bias_correction1,
// in double, narrowed at the point of use)
__global__ void k_epoch_prep(float *steps, int K, double lr, double beta1, double beta2, double eps, int64_t step0) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) {
    const TrlAdamScalars s = trl_adam_scalars(lr, beta1, beta2, eps, step0 + k, 1.0, 0.0);
    steps[2 * k] = s.bc2_sqrt;
    steps[2 * k + 1] = s.neg_step;
  }
}

__global__ void k_epoch_totals(float *losses, int K, float alpha, float beta) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < K) losses[4 * k + 3] = (losses[4 * k] + alpha * losses[4 * k + 1]) - beta * losses[4 * k + 2];
}

template <int role>      // 0 critic, 1 actor
__device__ __forceinline__ void epoch_body(const Args &P, float *lds) {
  const Net &N = P.net[role];
  const Lay L = make_lay(N);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 15, q = lane >> 4;
  const bool d3 = N.depth == 3;
  const int H1 = N.H1, H2 = N.H2, O = N.O, HL = L.HL, d0 = N.d0, A = P.A;
  float *xpt[2] = {lds + L.xpt[0], lds + L.xpt[1]};
  float *w0p = lds + L.w0p, *h1t = lds + L.h1t, *h2t = lds + L.h2t, *w1 = lds + L.w1, *red = lds + L.red,
        *gout = lds + L.gout, *part = lds + L.part, *vpp = lds + L.vp_p, *vpm = lds + L.vp_m, *vpv = lds + L.vp_v,
        *vpg = lds + L.vp_g;
  float *hlt = d3 ? h2t : h1t;               // the last hidden layer's image
  const int ldw1 = L.ldw1;

  // ---- ownership of the two weight matrices.  Main product (depth 3: dW1 [H2][H1]; depth 2: dW0' [H1][KP]): tiles
  // w and w + NW; depth 3 also owns dW0' tile w (w < H1 / 16) in the third slot.
  const int KT1 = d3 ? H1 / 16 : 1;                        // column tiles of the main product
  const int nmain = d3 ? (H2 / 16) * KT1 : H1 / 16;
  Owned own[3];
  int t_row[3], t_col[3];
  bool t_ok[3], t_is_w1[3];
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) {
    const int t = sl < 2 ? w + NW * sl : w;
    t_is_w1[sl] = d3 && sl < 2;
    t_ok[sl] = sl < 2 ? t < nmain : (d3 && t < H1 / 16);
    t_row[sl] = (t_is_w1[sl] ? t / KT1 : t) * 16;
    t_col[sl] = (t_is_w1[sl] ? t % KT1 : 0) * 16;
    own[sl].acc = (v4f){0.f, 0.f, 0.f, 0.f};
    float p[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
    if (t_ok[sl]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = t_row[sl] + 4 * q + r, col = t_col[sl] + i;
        int64_t g = -1;
        if (t_is_w1[sl]) g = N.w_off[1] + (int64_t)row * H1 + col;
        else if (col < d0) g = N.w_off[0] + (int64_t)row * d0 + col;
        else if (col == d0) g = N.b_off[0] + row;
        if (g >= 0) { p[r] = P.param[g]; m[r] = P.m[g]; v[r] = P.v[g]; }
        if (t_is_w1[sl]) w1[row * ldw1 + col] = p[r];
        else w0p[row * LDK + col] = p[r];
      }
    }
    own[sl].p = (v4f){p[0], p[1], p[2], p[3]};
    own[sl].m = (v4f){m[0], m[1], m[2], m[3]};
    own[sl].v = (v4f){v[0], v[1], v[2], v[3]};
  }
  // the small vectors: last layer [O][HL], its bias [O], hidden bias b1 [H2] (depth 3), log_std [A] (actor)
  auto vp_global = [&](int e) -> int64_t {
    if (e < L.o_bl) return N.w_off[N.depth - 1] + e;
    if (e < L.o_b1) return N.b_off[N.depth - 1] + (e - L.o_bl);
    if (e < L.o_ls) return N.b_off[1] + (e - L.o_b1);
    return N.ls_off + (e - L.o_ls);
  };
  for (int e = tid; e < L.NV; e += NT) {
    const int64_t g = vp_global(e);
    vpp[e] = P.param[g]; vpm[e] = P.m[g]; vpv[e] = P.v[g]; vpg[e] = 0.f;
  }

  const int B = P.B;
  const int64_t n_rows = P.n_rows;
  const int mb_per_epoch = (int)((n_rows + B - 1) / B);
  const int K = mb_per_epoch * P.n_epochs;

  // ---- input prefetch: a tile's observations are 64 * d0 consecutive floats (consecutive rows); thread t takes
  // elements t and t + NT.  Wave 0's lane s also carries sample s's loss inputs.
  float xin[2] = {0.f, 0.f};
  float in_act[MAXO] = {0, 0, 0, 0}, in_old[MAXO] = {0, 0, 0, 0}, in_adv = 0.f, in_ret = 0.f;
  auto prefetch = [&](int64_t row0, int rows) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + u * NT;
      xin[u] = e < rows * d0 ? P.obs[row0 * d0 + e] : 0.f;
    }
    if (w == 0) {
      const bool ok = lane < rows;
      const int64_t r = row0 + (ok ? lane : 0);
      if (role == 0) {
        in_ret = ok ? P.returns[r] : 0.f;
      } else {
        in_adv = ok ? P.adv[r] : 0.f;
#pragma unroll
        for (int a = 0; a < MAXO; ++a)
          if (a < A) { in_act[a] = ok ? P.action[r * A + a] : 0.f; in_old[a] = ok ? P.old_logp[r * A + a] : 0.f; }
      }
    }
  };
  // registers -> the x' image [KP][S]: rows j < d0 the observation, row d0 ones (the bias column), the rest and every
  // column of a sample beyond the minibatch zeros (such a sample then contributes exact zeros everywhere downstream)
  auto store_x = [&](float *X, int rows) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = tid + u * NT;
      if (e < S * d0) { const int s = e / d0, j = e - s * d0; X[j * LDS_S + s] = xin[u]; }
    }
    for (int e = tid; e < (KP - d0) * S; e += NT) {
      const int j = d0 + e / S, s = e % S;
      X[j * LDS_S + s] = (j == d0 && s < rows) ? 1.f : 0.f;
    }
  };
  auto tile_of = [&](int k, int t, int64_t &row0, int &rows, int &bm) {      // update k, tile t -> rows of the data
    const int mb = k % mb_per_epoch;
    const int64_t mb0 = (int64_t)mb * B;
    bm = (int)(n_rows - mb0 < B ? n_rows - mb0 : B);
    row0 = mb0 + (int64_t)t * S;
    rows = bm - t * S < S ? bm - t * S : S;
  };

  int cur = 0;
  {
    int64_t row0; int rows, bm;
    tile_of(0, 0, row0, rows, bm);
    prefetch(row0, rows);
    store_x(xpt[0], rows);
  }
  float cur_act[MAXO], cur_old[MAXO], cur_adv = in_adv, cur_ret = in_ret;
#pragma unroll
  for (int a = 0; a < MAXO; ++a) { cur_act[a] = in_act[a]; cur_old[a] = in_old[a]; }
  __syncthreads();

  // lane = feature pass: this wave's block of 64 features and its share of the tile's samples
  const int gp_nb = w / L.wpb, gp_sr = w % L.wpb, gp_ns = S / L.wpb;
  const int gp_n = gp_nb * 64 + lane;
  const bool gp_on = gp_nb * 64 < HL && gp_n < HL;

  for (int k = 0; k < K; ++k) {
    int bm;
    { int64_t r0; int rr; tile_of(k, 0, r0, rr, bm); }
    const int ntile = (bm + S - 1) / S;
    const float invB = 1.0f / (float)bm;
    const float bc2_sqrt = P.steps[2 * k], neg_step = P.steps[2 * k + 1];
    float a_crit = 0.f, a_min = 0.f, a_ent = 0.f, a_dls[MAXO] = {0, 0, 0, 0}, a_gb[MAXO] = {0, 0, 0, 0};
    float dwl[MAXO] = {0, 0, 0, 0}, db1 = 0.f;

    for (int t = 0; t < ntile; ++t) {
      const float *X = xpt[cur];
      int rows_t;
      { int64_t r0; int b2; tile_of(k, t, r0, rows_t, b2); }
      // the tile after this one (next tile of the minibatch, or the first of the next update): loads in flight from here
      const bool more = t + 1 < ntile || k + 1 < K;
      int64_t nrow0 = 0; int nrows = 0;
      if (more) { int b2; tile_of(t + 1 < ntile ? k : k + 1, t + 1 < ntile ? t + 1 : 0, nrow0, nrows, b2); prefetch(nrow0, nrows); }

      // ---- layer 0: h1[s][n] = relu(x'[s] . W0'[n]) as C[s][n] tiles -> [feature][sample] image, 16 B stores
      {
        const int ntl = 4 * (H1 / 16);
        for (int t0 = w; t0 < ntl; t0 += 2 * NW) {
          const bool two = t0 + NW < ntl;
          const int t1 = two ? t0 + NW : t0;
          const int s0 = (t0 & 3) * 16, n0 = (t0 >> 2) * 16, n1 = (t1 >> 2) * 16;      // (t1 & 3 == t0 & 3)
          v4f c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
          mma2<false, true>(X + s0, X + s0, LDS_S, w0p + n0 * LDK, w0p + n1 * LDK, LDK, KP, c0, c1, i, q);
          *(v4f *)(h1t + (n0 + i) * LDS_S + s0 + 4 * q) = relu4(c0);
          if (two) *(v4f *)(h1t + (n1 + i) * LDS_S + s0 + 4 * q) = relu4(c1);
        }
      }
      __syncthreads();
      // ---- layer 1 (depth 3): h2[s][n] = relu(h1[s] . W1[n] + b1[n])
      if (d3) {
        const int ntl = 4 * (H2 / 16);
        for (int t0 = w; t0 < ntl; t0 += 2 * NW) {
          const bool two = t0 + NW < ntl;
          const int t1 = two ? t0 + NW : t0;
          const int s0 = (t0 & 3) * 16, n0 = (t0 >> 2) * 16, n1 = (t1 >> 2) * 16;
          const float bb0 = vpp[L.o_b1 + n0 + i], bb1 = vpp[L.o_b1 + n1 + i];
          v4f c0 = {bb0, bb0, bb0, bb0}, c1 = {bb1, bb1, bb1, bb1};
          mma2<false, true>(h1t + s0, h1t + s0, LDS_S, w1 + n0 * ldw1, w1 + n1 * ldw1, ldw1, H1, c0, c1, i, q);
          *(v4f *)(h2t + (n0 + i) * LDS_S + s0 + 4 * q) = relu4(c0);
          if (two) *(v4f *)(h2t + (n1 + i) * LDS_S + s0 + 4 * q) = relu4(c1);
        }
        __syncthreads();
      }
      // ---- last layer: out[s][o] = bL[o] + wL[o] . h[s]; lane = sample, the waves split the features
      {
        const int nper = HL / NW, n0 = w * nper;
        float pr[MAXO] = {0, 0, 0, 0};
        for (int n = n0; n < n0 + nper; ++n) {
          const float h = hlt[n * LDS_S + lane];
#pragma unroll
          for (int o = 0; o < MAXO; ++o)
            if (o < O) pr[o] = __builtin_fmaf(vpp[L.o_wl + o * HL + n], h, pr[o]);
        }
#pragma unroll
        for (int o = 0; o < MAXO; ++o)
          if (o < O) red[(w * MAXO + o) * S + lane] = pr[o];
      }
      __syncthreads();
      // ---- loss and its derivative at the outputs (wave 0, lane = sample): k_ppo_loss's arithmetic
      if (w == 0) {
        const bool ok = lane < rows_t;
        float out[MAXO];
#pragma unroll
        for (int o = 0; o < MAXO; ++o) {
          out[o] = 0.f;
          if (o < O) {
            float sacc = vpp[L.o_bl + o];
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) sacc += red[(ww * MAXO + o) * S + lane];
            out[o] = sacc;
          }
        }
        if (role == 0) {      // critic: mse_loss(values, returns), weight alpha
          const float dv = out[0] - cur_ret;
          const float g = ok ? P.alpha * ((2.0f * invB) * dv) : 0.f;
          a_crit += ok ? dv * dv : 0.f;
          a_gb[0] += g;
          gout[lane] = g;
        } else {
          const float gmin = -(1.0f / (float)((int64_t)bm * A));
#pragma unroll
          for (int a = 0; a < MAXO; ++a) {
            if (a < A) {
              const float sigma = expf(vpp[L.o_ls + a]);
              const float var = sigma * sigma;
              const float logsig = logf(sigma);
              const float den = 2.0f * var;
              const float d = cur_act[a] - out[a];
              const float sq = d * d;
              const float lp = ((-sq) / den - logsig) - LOG_SQRT_2PI;
              const float ratio = expf(lp - cur_old[a]);
              const float s1 = ratio * cur_adv;
              const float rc = fminf(fmaxf(ratio, P.lo), P.hi);
              const float s2 = rc * cur_adv;
              const float g1 = s1 < s2 ? gmin : (s1 == s2 ? 0.5f * gmin : 0.f);
              const float g2 = s2 < s1 ? gmin : (s1 == s2 ? 0.5f * gmin : 0.f);
              const bool inr = ratio >= P.lo && ratio <= P.hi;
              const float g_ratio = g1 * cur_adv + (inr ? g2 * cur_adv : 0.f);
              const float g_lp = g_ratio * ratio;
              const float g = ok ? (g_lp / den) * (2.0f * d) : 0.f;
              a_min += ok ? fminf(s1, s2) : 0.f;
              a_ent += ok ? ENT_CONST + logsig : 0.f;
              a_dls[a] += ok ? g_lp * (sq / var - 1.0f) : 0.f;
              a_gb[a] += g;
              gout[a * LDS_S + lane] = g;
            }
          }
        }
      }
      __syncthreads();
      // ---- lane = feature: dwL[o][n] += sum_s gout[s][o] h[n][s];  g[n][s] = (sum_o gout[s][o] wL[o][n]) (h > 0), in place
      if (gp_on) {
        float wl_n[MAXO];
#pragma unroll
        for (int o = 0; o < MAXO; ++o) wl_n[o] = o < O ? vpp[L.o_wl + o * HL + gp_n] : 0.f;
        for (int s = gp_sr * gp_ns; s < (gp_sr + 1) * gp_ns; s += 4) {
          v4f h = *(const v4f *)(hlt + gp_n * LDS_S + s);
          v4f g = {0, 0, 0, 0};
#pragma unroll
          for (int o = 0; o < MAXO; ++o) {
            if (o < O) {
              const v4f go = *(const v4f *)(gout + o * LDS_S + s);
              dwl[o] = __builtin_fmaf(go.x, h.x, dwl[o]); dwl[o] = __builtin_fmaf(go.y, h.y, dwl[o]);
              dwl[o] = __builtin_fmaf(go.z, h.z, dwl[o]); dwl[o] = __builtin_fmaf(go.w, h.w, dwl[o]);
              g.x = __builtin_fmaf(go.x, wl_n[o], g.x); g.y = __builtin_fmaf(go.y, wl_n[o], g.y);
              g.z = __builtin_fmaf(go.z, wl_n[o], g.z); g.w = __builtin_fmaf(go.w, wl_n[o], g.w);
            }
          }
          g.x = h.x > 0.f ? g.x : 0.f; g.y = h.y > 0.f ? g.y : 0.f; g.z = h.z > 0.f ? g.z : 0.f; g.w = h.w > 0.f ? g.w : 0.f;
          db1 += (g.x + g.y) + (g.z + g.w);
          *(v4f *)(hlt + gp_n * LDS_S + s) = g;
        }
      }
      __syncthreads();
      if (d3) {
        // ---- dW1[n][k] += sum_s g2[n][s] h1[k][s]  (both operands sample-contiguous)
        {
          const bool two = t_ok[1];
          const int r0 = t_row[0], c0 = t_col[0], r1 = two ? t_row[1] : r0, c1 = two ? t_col[1] : c0;
          if (t_ok[0]) {
            v4f z = {0, 0, 0, 0};
            mma2<true, true>(h2t + r0 * LDS_S, h2t + r1 * LDS_S, LDS_S, h1t + c0 * LDS_S, h1t + c1 * LDS_S, LDS_S, S,
                             own[0].acc, two ? own[1].acc : z, i, q);
          }
        }
        __syncthreads();
        // ---- g1[s][k] = (sum_n g2[s][n] W1[n][k]) (h1 > 0) as C[s][k] tiles, written over h1's image
        {
          const int ntl = 4 * (H1 / 16);
          for (int t0 = w; t0 < ntl; t0 += 2 * NW) {
            const bool two = t0 + NW < ntl;
            const int t1 = two ? t0 + NW : t0;
            const int s0 = (t0 & 3) * 16, k0 = (t0 >> 2) * 16, k1 = (t1 >> 2) * 16;
            v4f c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
            mma2<false, false>(h2t + s0, h2t + s0, LDS_S, w1 + k0, w1 + k1, ldw1, H2, c0, c1, i, q);
            float *m0 = h1t + (k0 + i) * LDS_S + s0 + 4 * q, *m1 = h1t + (k1 + i) * LDS_S + s0 + 4 * q;
            const v4f h0 = *(const v4f *)m0;
            c0.x = h0.x > 0.f ? c0.x : 0.f; c0.y = h0.y > 0.f ? c0.y : 0.f; c0.z = h0.z > 0.f ? c0.z : 0.f; c0.w = h0.w > 0.f ? c0.w : 0.f;
            *(v4f *)m0 = c0;
            if (two) {
              const v4f hh = *(const v4f *)m1;
              c1.x = hh.x > 0.f ? c1.x : 0.f; c1.y = hh.y > 0.f ? c1.y : 0.f; c1.z = hh.z > 0.f ? c1.z : 0.f; c1.w = hh.w > 0.f ? c1.w : 0.f;
              *(v4f *)m1 = c1;
            }
          }
        }
        __syncthreads();
        // ---- dW0'[n][j] += sum_s g1[n][s] x'[j][s]
        if (t_ok[2]) {
          v4f z = {0, 0, 0, 0};
          mma2<true, true>(h1t + t_row[2] * LDS_S, h1t + t_row[2] * LDS_S, LDS_S, X, X, LDS_S, S, own[2].acc, z, i, q);
        }
      } else if (t_ok[0]) {
        const bool two = t_ok[1];
        const int r1 = two ? t_row[1] : t_row[0];
        v4f z = {0, 0, 0, 0};
        mma2<true, true>(h1t + t_row[0] * LDS_S, h1t + r1 * LDS_S, LDS_S, X, X, LDS_S, S, own[0].acc, two ? own[1].acc : z, i, q);
      }
      // the next tile's x' image goes into the other buffer (nobody reads that one now); its loss inputs become current
      if (more) {
        store_x(xpt[cur ^ 1], nrows);
        cur ^= 1;
        cur_adv = in_adv; cur_ret = in_ret;
#pragma unroll
        for (int a = 0; a < MAXO; ++a) { cur_act[a] = in_act[a]; cur_old[a] = in_old[a]; }
      }
      if (t + 1 < ntile) __syncthreads();
    }

    // ---- the update.  Partial sums of the lane = feature pass, the loss values, the small vectors' gradients
    if (gp_on) {
#pragma unroll
      for (int o = 0; o < MAXO; ++o)
        if (o < O) part[(gp_sr * (MAXO + 1) + o) * L.HLP + gp_n] = dwl[o];
      part[(gp_sr * (MAXO + 1) + MAXO) * L.HLP + gp_n] = db1;
    }
    if (w == 0) {
      if (role == 0) {
        const float s_crit = trl_wave_sum(a_crit), gb = trl_wave_sum(a_gb[0]);
        if (lane == 0) {
          P.losses[4 * k + 1] = s_crit * (1.0f / (float)bm);
          vpg[L.o_bl] = gb;
        }
      } else {
        const float invBA = 1.0f / (float)((int64_t)bm * A);
        const float s_min = trl_wave_sum(a_min), s_ent = trl_wave_sum(a_ent);
        if (lane == 0) {
          P.losses[4 * k + 0] = -(s_min * invBA);
          P.losses[4 * k + 2] = s_ent * invBA;
        }
#pragma unroll
        for (int a = 0; a < MAXO; ++a) {
          if (a < A) {
            const float gb = trl_wave_sum(a_gb[a]), gl = trl_wave_sum(a_dls[a]);
            if (lane == 0) {
              vpg[L.o_bl + a] = gb;
              vpg[L.o_ls + a] = gl + (-P.beta / (float)A);      // the entropy term's share: d(-beta * mean(ent)) / d log_std[a]
            }
          }
        }
      }
    }
    __syncthreads();
    TrlAdamScalars sc;
    sc.w1 = P.w1; sc.beta2 = P.beta2; sc.c2 = P.c2; sc.bc2_sqrt = bc2_sqrt; sc.eps = P.eps; sc.neg_step = neg_step;
    sc.scale = 1.0f; sc.clip = P.clip;
    const bool last = k + 1 == K;
    for (int e = tid; e < L.NV; e += NT) {
      float g;
      if (e < L.o_bl) {
        const int o = e / HL, n = e - o * HL;
        g = 0.f;
        for (int sr = 0; sr < L.wpb; ++sr) g += part[(sr * (MAXO + 1) + o) * L.HLP + n];
      } else if (e >= L.o_b1 && e < L.o_ls) {
        const int n = e - L.o_b1;
        g = 0.f;
        for (int sr = 0; sr < L.wpb; ++sr) g += part[(sr * (MAXO + 1) + MAXO) * L.HLP + n];
      } else {
        g = vpg[e];
      }
      float p = vpp[e], m = vpm[e], v = vpv[e];
      trl_adam_elem(&p, &g, &m, &v, sc);
      vpp[e] = p; vpm[e] = m; vpv[e] = v; vpg[e] = g;
    }
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
      if (t_ok[sl]) {
        v4f g;
        adam4(own[sl], sc, g);
        const float p[4] = {own[sl].p.x, own[sl].p.y, own[sl].p.z, own[sl].p.w};
        const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = t_row[sl] + 4 * q + r, col = t_col[sl] + i;
          if (t_is_w1[sl]) w1[row * ldw1 + col] = p[r];
          else if (col <= d0) w0p[row * LDK + col] = p[r];
          if (last) {      // the gradients the last learn() leaves behind (after clip_grad_value_)
            int64_t go = -1;
            if (t_is_w1[sl]) go = N.w_off[1] + (int64_t)row * H1 + col;
            else if (col < d0) go = N.w_off[0] + (int64_t)row * d0 + col;
            else if (col == d0) go = N.b_off[0] + row;
            if (go >= 0) P.grad[go] = gg[r];
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- parameters and moments back to the flat buffers
#pragma unroll
  for (int sl = 0; sl < 3; ++sl) {
    if (t_ok[sl]) {
      const float p[4] = {own[sl].p.x, own[sl].p.y, own[sl].p.z, own[sl].p.w};
      const float m[4] = {own[sl].m.x, own[sl].m.y, own[sl].m.z, own[sl].m.w};
      const float v[4] = {own[sl].v.x, own[sl].v.y, own[sl].v.z, own[sl].v.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = t_row[sl] + 4 * q + r, col = t_col[sl] + i;
        int64_t g = -1;
        if (t_is_w1[sl]) g = N.w_off[1] + (int64_t)row * H1 + col;
        else if (col < d0) g = N.w_off[0] + (int64_t)row * d0 + col;
        else if (col == d0) g = N.b_off[0] + row;
        if (g >= 0) { P.param[g] = p[r]; P.m[g] = m[r]; P.v[g] = v[r]; }
      }
    }
  }
  for (int e = tid; e < L.NV; e += NT) {
    const int64_t g = vp_global(e);
    P.param[g] = vpp[e]; P.m[g] = vpm[e]; P.v[g] = vpv[e]; P.grad[g] = vpg[e];
  }
}

__global__ __launch_bounds__(NT) void k_ppo_epoch(const Args P) {
  extern __shared__ __align__(16) float lds[];
  if (blockIdx.x == 0) epoch_body<0>(P, lds);
  else epoch_body<1>(P, lds);
}

constexpr int64_t MAX_LDS_BYTES = 160 * 1024;

int fill_net(const char *fn, const char *name, const trl_mlp_desc *d, const trl_adam_args *adam, int O, Net &n) {
  TRL_REQUIRE(d != nullptr, "%s: %s is NULL", fn, name);
  TRL_REQUIRE(d->depth == 2 || d->depth == 3, "%s: %s: depth 2 or 3 (one or two hidden layers), got %d", fn, name, d->depth);
  n.depth = d->depth;
  n.d0 = d->dims[0];
  n.H1 = d->dims[1];
  n.H2 = d->depth == 3 ? d->dims[2] : 0;
  n.O = d->dims[d->depth];
  TRL_REQUIRE(n.O == O, "%s: %s: fan-out %d, expected %d", fn, name, n.O, O);
  TRL_REQUIRE(n.d0 >= 1 && n.d0 <= KP - 1, "%s: %s: fan-in %d must be in [1, %d]", fn, name, n.d0, KP - 1);
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

}  // namespace

extern "C" int64_t trl_ppo_learn_epoch_workspace_bytes(int64_t n_rows, int64_t minibatch, int32_t n_epochs) {
  if (n_rows <= 0 || minibatch <= 0 || n_epochs <= 0) return 0;
  return ((n_rows + minibatch - 1) / minibatch) * n_epochs * 2 * (int64_t)sizeof(float);
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
  Args P = {};
  int rc;
  if ((rc = fill_net(fn, "critic", critic, adam, 1, P.net[0])) || (rc = fill_net(fn, "actor", actor, adam, A, P.net[1])))
    return rc;
  TRL_REQUIRE(P.net[0].d0 == P.net[1].d0, "%s: critic and actor must see the same observation", fn);
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
  P.clip = (float)adam->clip_value;
  P.w1 = (float)(1.0 - adam->beta1); P.beta2 = (float)adam->beta2; P.c2 = (float)(1.0 - adam->beta2); P.eps = (float)adam->eps;
  P.steps = (const float *)ws;
  P.losses = losses_out;
  hipStream_t st = (hipStream_t)stream;
  const int K = (int)K64;
  if (lds_bytes > 48 * 1024 &&      // (per device and cheap: set whenever the launch needs more than the default limit)
      hipFuncSetAttribute((const void *)k_ppo_epoch, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MAX_LDS_BYTES) != hipSuccess) {
    (void)hipGetLastError();
    trl_set_error("%s: could not raise the kernel's LDS limit", fn);
    return TRL_EHIP;
  }
  hipLaunchKernelGGL(k_epoch_prep, dim3((K + 255) / 256), dim3(256), 0, st, (float *)ws, K, adam->lr, adam->beta1, adam->beta2,
                     adam->eps, adam->step);
  TRL_CHECK_LAUNCH(fn);
  hipLaunchKernelGGL(k_ppo_epoch, dim3(2), dim3(NT), (size_t)lds_bytes, st, P);
  TRL_CHECK_LAUNCH(fn);
  hipLaunchKernelGGL(k_epoch_totals, dim3((K + 255) / 256), dim3(256), 0, st, losses_out, K, P.alpha, P.beta);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}
