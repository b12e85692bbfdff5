// Whole learner step of the small-batch off-policy controllers as a short chain of launches.
//   DDPGController.learn   ddpg_controller.py:101-128 (+ polyak_average_ :42-57)
//   DQNController.learn    dqn_controller.py:61-83
// on the reference networks (models.py: DDPGActorNet :27-53, DDPGCriticNet :56-81, QNet :6-24:
// two hidden ReLU layers, 128-256 wide) at the reference batch sizes (32-256 rows).
//
// At these sizes a step is ~170 MFLOP: nothing is bound by arithmetic or by HBM (every operand is
// L2-resident), the step is bound by HOW MANY dependent launches it takes.  Built from BLAS calls
// and elementwise kernels it is ~70 launches; here it is 10 phases (DDPG: in one launch) / 6 launches (DQN):
//   * one launch = one *phase*: up to 4 independent matrix products of the step that have the same
//     dependency depth (e.g. layer 1 of actor, target actor and critic), each cut into 16x16 output
//     tiles, one workgroup per tile, so a phase fills 100-400 CUs;
//   * a workgroup's 4 waves split the reduction dimension (split-K), issue all their operand loads
//     up front, run v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulate) and add their four
//     partial tiles in a fixed order through LDS: the critical path of a phase is one L2 round trip
//     + <= 16 MFMAs + one LDS exchange;
//   * everything elementwise rides in a product's operand fetch or epilogue: bias / ReLU / tanh,
//     the input concatenation [obs | action] of the critic, the ReLU / tanh masks of the backward,
//     the last layer's backward (a masked outer product, formed on the fly as the next product's
//     operand instead of being materialised), bias gradients (column sums of the same operand), and
//     the optimizer: a weight-gradient tile is consumed where it is produced -- Adam on the tile's
//     parameters, then the Polyak average into the target network's same tile.
// Products whose weights are still read later in the chain are ordered before the launch that
// updates them; nothing inside one launch reads a parameter another workgroup of it writes.
// Reductions have a fixed order, so a step is bitwise reproducible run to run.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

#include "trl_common.h"

namespace {

typedef float f4v __attribute__((ext_vector_type(4)));

enum { T_FWD = 0, T_DX = 1, T_DW = 2 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };
enum { UPD_GRAD = 0, UPD_ADAM = 1, UPD_ADAM_POLYAK = 2 };
constexpr int MAX_TILES = 4;

// A gradient operand G [B, Kg]: materialised (g != NULL), or the backward of the layer above taken
// on the fly:  G[i, k] = (h[i, k] > 0) * sum_a seed[i, a] * w2[a, k]   (h = this layer's ReLU output,
// w2 [As, Kg] = the next layer's weight, seed [B, As] = the gradient at that layer's output, or the
// constant cseed when seed == NULL).
struct Grad {
  const float *g;
  const float *seed;
  const float *w2;
  const float *h;
  int ld;
  int As;
  float cseed;
};

struct Tile {
  int type;
  int M, N, K;        // output M x N, reduction length K
  int tiles_n;
  int wg0;            // first workgroup of this problem in the launch
  int fast;           // FWD: 16 B operand loads (single source, K % 4 == 0, aligned bases)
  int act;            // FWD: activation; DX: the mask applied to the product
  // FWD: y[B, N] = act([x1 | x2][B, K1 + K2] @ w[N, K]^T + b);   DW: X = [x1 | x2] [B, N]
  const float *x1, *x2;
  int K1, K2;
  const float *w;     // FWD: [N, K];  DX: [K, ldw], columns woff .. woff + N - 1 are used
  int ldw, woff;
  const float *b;
  float *y;           // FWD / DX output [B, ldy]
  int ldy;
  // DX: y[B, N] = (G[B, K] @ w[K, woff : woff + N]) * mask(hin);  gout: optional copy of G [B, K]
  // DW: dW[M, N] = G[B, M]^T @ X[B, N]  (K = B);  db[M] = column sums of G
  Grad G;
  const float *hin;
  int ldhin;
  float *gout;
  int64_t w_off, b_off;  // offsets of the weight / bias in the optimizer's flat buffers
  int upd;               // which Update of the launch
  // FWD, N == 1, optional: y is the target critic's next-state value; the row's TD target, loss term and the loss's
  // derivative at `cur` follow in the same epilogue (ddpg_controller.py:105-112) instead of in a phase of their own
  const float *td_cur, *td_reward, *td_done;
  float *td_dcur, *td_lterm;
  float td_gamma, td_hdelta;
  int td_kind;
};

struct Update {
  float *param, *grad, *m, *v, *target;
  const TrlAdamScalars *dev;   // graph replay: the scalars live in the device Adam state, `s` is unused
  TrlAdamScalars s;
  float tau, omt;
  int mode;
};

// (one-launch chain only) a phase that is not a set of tiles: the DDPG losses, run by workgroup 0
// the two loss values: sums over the batch of the per-row terms, by one workgroup that rides in a later phase
struct LossArgs {
  const float *lterm, *q_pi;
  int B;
  float *losses_out;
  // kind == 1: not sums but the whole TD loss of a DQN step (dqn_controller.py:64-78; the arithmetic of k_td_loss,
  // trl_loss.hip) over the B rows by this one workgroup: loss -> losses_out[0], d loss / d q -> dq [B, A]
  int kind;
  const float *q, *qnt, *qno, *action, *reward, *done;
  float *dq;
  int A, td_kind;
  float gamma, hdelta;
};

struct PhaseArgs {
  Tile t[MAX_TILES];
  Update u[2];
  int nt;
  int wgs;            // workgroups' worth of tiles in this phase
  int sum_extra;      // 1: one more workgroup (index wgs) forms the losses L
  LossArgs L;
};

// Every pointer of a Tile is global memory.  Said explicitly: in k_chain the tile descriptions come out of LDS and the
// compiler would otherwise take the pointers for generic ones (flat_ instructions: both memory counters, more waits).
typedef __attribute__((address_space(1))) float gfloat;
typedef __attribute__((address_space(1))) f4v gf4v;
__device__ __forceinline__ const gfloat *gm(const float *p) { return (const gfloat *)p; }
__device__ __forceinline__ gfloat *gm(float *p) { return (gfloat *)p; }
__device__ __forceinline__ f4v ld4(const float *p) { return *(const gf4v *)p; }
__device__ __forceinline__ float ld1(const float *p) { return *gm(p); }
// x where `on`, else a zero -- as a product with 1.0 or 0.0.  Written `on ? x : 0.f` (or as a bit mask, which the
// compiler turns back into that select) the LOAD of x is sunk under a branch on `on` and waited for there, one load at
// a time (seen in the ISA: saveexec / load / vmcnt(0) per element).  Every address here is clamped to a valid, finite
// element, so the loads can all be in flight together; a product's sign of zero does not reach a sum that starts at +0.
__device__ __forceinline__ float keep(float x, bool on) { return x * (on ? 1.0f : 0.0f); }
__device__ __forceinline__ f4v keep(f4v x, bool on) {
  f4v r;
#pragma unroll
  for (int c = 0; c < 4; ++c) r[c] = keep(x[c], on);
  return r;
}

// COH (the one-launch chain, k_chain): what one phase writes and a later phase of the SAME launch reads is stored
// write-through at agent scope (sc1), so that it is in memory, past the writer's XCD's L2, once the store is
// acknowledged; the barrier between two phases then needs no L2 write-back / invalidate, which is what makes it
// cheaper than a kernel boundary (tools/grid_barrier_probe.hip: 1.7-2.5 us against 11-19 us with release / acquire,
// ~7 us for a launch).  The READS stay plain, L2-cached 16 B loads: a reader's L2 or L1 could only hold a stale copy of
// a line it had fetched earlier in this launch, and no line of these buffers is read before the phase that writes it
// (every buffer is written whole by one phase, buffers are 256 B apart, and a launch starts with clean caches).
// Agent-scope LOADS were measured and cost 3-4 us per tile (every operand fetched from memory, four dwords at a time).
template <bool COH>
__device__ __forceinline__ float ldf(const float *p) {
  return *gm(p);
}
template <bool COH>
__device__ __forceinline__ f4v ldv(const float *p) {
  return ld4(p);
}
template <bool COH>
__device__ __forceinline__ void stf(float *p, float v) {
  if (COH) __hip_atomic_store(gm(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *gm(p) = v;
}

// AS1: one output (a critic's value, a one-dimensional action), the common case -- straight-line and branch-free (the
// seed pointer's null test becomes arithmetic on a load from a valid address), so that a caller's loads for several
// quads / rows are all issued before the first is waited for; a branch or a run-time loop between two loads keeps them
// in separate blocks and the second is issued only after the first has come back.
template <bool COH, bool AS1>
__device__ __forceinline__ f4v grad_row4(const Grad &G, int i, int k, int Kg, bool virt) {
  if (!virt) return ldv<COH>(G.g + (size_t)i * G.ld + k);
  const f4v h = ldv<COH>(G.h + (size_t)i * Kg + k);
  f4v acc = {0.f, 0.f, 0.f, 0.f};
  if (AS1) {
    const float fs = G.seed ? 1.0f : 0.0f, cs = G.seed ? 0.0f : G.cseed;
    const float s = ldf<COH>(G.seed ? G.seed + i : G.w2) * fs + cs;      // seed[i], or cseed (x * 1 + 0 and x * 0 + c are exact)
    acc += s * ld4(G.w2 + k);
  } else {
    for (int a = 0; a < G.As; ++a) {
      const float s = G.seed ? ldf<COH>(G.seed + (size_t)i * G.As + a) : G.cseed;
      const f4v w = ld4(G.w2 + (size_t)a * Kg + k);
      acc += s * w;
    }
  }
  f4v r;
#pragma unroll
  for (int c = 0; c < 4; ++c) r[c] = h[c] > 0.f ? acc[c] : 0.f;
  return r;
}

template <bool COH, bool AS1>
__device__ __forceinline__ float grad_at(const Grad &G, int i, int m, int Kg, bool virt) {
  if (!virt) return ldf<COH>(G.g + (size_t)i * G.ld + m);
  const float h = ldf<COH>(G.h + (size_t)i * Kg + m);
  float acc = 0.f;
  if (AS1) {
    const float fs = G.seed ? 1.0f : 0.0f, cs = G.seed ? 0.0f : G.cseed;
    const float s = ldf<COH>(G.seed ? G.seed + i : G.w2) * fs + cs;
    acc += s * ld1(G.w2 + m);
  } else {
    for (int a = 0; a < G.As; ++a) {
      const float s = G.seed ? ldf<COH>(G.seed + (size_t)i * G.As + a) : G.cseed;
      acc += s * ld1(G.w2 + (size_t)a * Kg + m);
    }
  }
  return h > 0.f ? acc : 0.f;
}

// X = [x1 | x2] at (row i, column k), k < K1 + K2
template <bool COH>
__device__ __forceinline__ float cat_at(const Tile &T, int i, int k) {
  const float *p = k < T.K1 ? T.x1 + (size_t)i * T.K1 + k : T.x2 + (size_t)i * T.K2 + (k - T.K1);
  return ldf<COH>(p);
}

// One wave's share [kb, ke) of the reduction: operand quads for 64 k's are fetched before the 16
// MFMAs that consume them.  Lane (r = lane & 15, q = lane >> 4) feeds row r of A and column r of B at
// the four k's kk .. kk + 3, kk = kc + 16 u + 4 q -- the same k's on both sides, which is all the
// instruction needs (it sums over its k slots; which k sits in which slot is free).
template <class LA, class LB, class SA>
__device__ __forceinline__ f4v mma_range(int kb, int ke, int q, LA load_a, LB load_b, SA after_a) {
  f4v acc = {0.f, 0.f, 0.f, 0.f};
  for (int kc = kb; kc < ke; kc += 64) {
    f4v a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int kk = kc + 16 * u + 4 * q;
      a[u] = load_a(kk);
      b[u] = load_b(kk);
    }
    // (what is done with an A quad besides the product -- the DX tiles' copy of it -- comes after ALL the loads have
    // been issued: a store between two loads waits for the first and holds the second back)
#pragma unroll
    for (int u = 0; u < 4; ++u) after_a(kc + 16 * u + 4 * q, a[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][c], b[u][c], acc, 0, 0, 0);
  }
  return acc;
}

// A gradient element's update (Adam, then the Polyak average into the target network) with the element's optimizer
// state read AHEAD of the product that yields the gradient: a tile is a chain of dependent memory round trips
// (operands, product, state, stores), and this takes one out of the chain.
struct AdamPre {
  float p, m, v, t;
};
__device__ __forceinline__ AdamPre update_prefetch(const Update &U, int64_t off, bool on) {
  AdamPre a = {0.f, 0.f, 0.f, 0.f};
  if (on && U.mode >= UPD_ADAM) {
    a.p = U.param[off]; a.m = U.m[off]; a.v = U.v[off];
    if (U.mode == UPD_ADAM_POLYAK) a.t = U.target[off];
  }
  return a;
}
__device__ __forceinline__ void apply_update(const Update &U, int64_t off, float g, const AdamPre &a) {
  if (U.mode < UPD_ADAM) {
    U.grad[off] = g;
    return;
  }
  const TrlAdamScalars s = U.dev ? *U.dev : U.s;      // (trl_adam_elem's arithmetic, op for op)
  float gi = g;
  if (s.scale != 1.0f) gi = __fmul_rn(gi, s.scale);
  if (s.clip > 0.f) gi = fminf(fmaxf(gi, -s.clip), s.clip);
  U.grad[off] = gi;
  const float mi = __fadd_rn(a.m, __fmul_rn(s.w1, __fsub_rn(gi, a.m)));
  const float vi = __fadd_rn(__fmul_rn(a.v, s.beta2), __fmul_rn(__fmul_rn(s.c2, gi), gi));
  const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vi), s.bc2_sqrt), s.eps);
  const float pn = __fadd_rn(a.p, __fdiv_rn(__fmul_rn(s.neg_step, mi), denom));
  U.param[off] = pn;
  U.m[off] = mi;
  U.v[off] = vi;
  if (U.mode == UPD_ADAM_POLYAK) U.target[off] = __fadd_rn(__fmul_rn(U.tau, pn), __fmul_rn(U.omt, a.t));
}

template <int TYPE, bool FLAG, bool COH, bool AS1 = false>   // FLAG: FWD -> fast loads;  DX / DW -> G is taken on the fly
__device__ __forceinline__ f4v tile_product(const Tile &T, int m0, int n0, int tn, int kb, int ke, int r, int q) {
  if (TYPE == T_FWD) {
    const int i = m0 + r;                    // batch row (B % 16 == 0: always valid)
    const int n = n0 + r, nc = n < T.N ? n : 0;
    const int K = T.K;
    if (FLAG) {
      const float *xa = T.x1 + (size_t)i * K;
      const float *wb = T.w + (size_t)nc * K;
      const bool nok = n < T.N;
      return mma_range(kb, ke, q,
          [&](int kk) { const bool ok = kk < ke; f4v v = ldv<COH>(xa + (ok ? kk : 0)); return keep(v, ok); },
          [&](int kk) { const bool ok = kk < ke && nok; f4v v = ld4(wb + (kk < ke ? kk : 0)); return keep(v, ok); },
          [](int, const f4v &) {});
    }
    return mma_range(kb, ke, q,
        [&](int kk) {
          f4v v;
#pragma unroll
          for (int c = 0; c < 4; ++c) { const int k = kk + c; const bool ok = k < ke; const float x = cat_at<COH>(T, i, ok ? k : 0); v[c] = keep(x, ok); }
          return v;
        },
        [&](int kk) {
          f4v v;
#pragma unroll
          for (int c = 0; c < 4; ++c) { const int k = kk + c; const bool ok = k < ke; const float x = ld1(T.w + (size_t)nc * K + (ok ? k : 0)); v[c] = keep(x, ok && n < T.N); }
          return v;
        },
        [](int, const f4v &) {});
  }
  if (TYPE == T_DX) {
    const int i = m0 + r;
    const int n = n0 + r, nc = n < T.N ? n : 0;
    const bool wr = T.gout != nullptr && tn == 0;
    return mma_range(kb, ke, q,
        [&](int kk) {
          const bool ok = kk < ke;
          return keep(grad_row4<COH, AS1>(T.G, i, ok ? kk : 0, T.K, FLAG), ok);
        },
        [&](int kk) {
          f4v v;
#pragma unroll
          for (int c = 0; c < 4; ++c) { const int k = kk + c; const bool ok = k < ke; const float x = ld1(T.w + (size_t)(ok ? k : 0) * T.ldw + T.woff + nc); v[c] = keep(x, ok && n < T.N); }
          return v;
        },
        [&](int kk, const f4v &v) {      // the optional copy of G (rows of the first column of tiles write it)
          if (wr && kk < ke) {
            if (COH) {
#pragma unroll
              for (int c = 0; c < 4; ++c) stf<true>(T.gout + (size_t)i * T.K + kk + c, v[c]);
            } else {
              *(gf4v *)(T.gout + (size_t)i * T.K + kk) = v;
            }
          }
        });
  }
  // T_DW: rows of the product = output features m, columns = input features n, reduction over the batch
  const int m = m0 + r, mc = m < T.M ? m : 0;
  const int n = n0 + r, nc = n < T.N ? n : 0;
  return mma_range(kb, ke, q,
      [&](int kk) {
        f4v v;
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int i = kk + c; const bool ok = i < ke; const float x = grad_at<COH, AS1>(T.G, ok ? i : 0, mc, T.M, FLAG); v[c] = keep(x, ok && m < T.M); }
        return v;
      },
      [&](int kk) {
        f4v v;
#pragma unroll
        for (int c = 0; c < 4; ++c) { const int i = kk + c; const bool ok = i < ke; const float x = cat_at<COH>(T, ok ? i : 0, nc); v[c] = keep(x, ok && n < T.N); }
        return v;
      },
      [](int, const f4v &) {});
}

#ifdef TRL_TUNING      // diagnostic build: 100 MHz stamps inside workgroup 0's tile of each phase of k_chain
__device__ unsigned long long g_dbg[16 * 8];
__device__ int g_dbg_phase;
#define DBG(k)                                                                                     \
  do {                                                                                             \
    if (COH && blockIdx.x == 0 && threadIdx.x == 0) g_dbg[(g_dbg_phase & 15) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define DBG(k)
#endif

// one 16 x 16 output tile of problem T (tile index tl within the problem) by the whole workgroup
template <bool COH>
__device__ __forceinline__ void phase_tile(const Tile &T, const Update &U, int tl, float (*red)[256]) {
  const int tm = tl / T.tiles_n, tn = tl - tm * T.tiles_n;
  const int m0 = tm * 16, n0 = tn * 16;
  const int t = threadIdx.x, w = t >> 6, lane = t & 63, r = lane & 15, q = lane >> 4;
  const int Kq = (T.K + 63) / 64 * 16;      // per-wave share of the reduction, a multiple of 16
  const int kb = w * Kq;
  const int ke = min(T.K, kb + Kq);
  const bool virt = T.G.g == nullptr;
  const bool as1 = T.G.As == 1;
  const int row = t >> 4, col = t & 15;
  const int m = m0 + row, n = n0 + col;

  // a weight-gradient tile: the optimizer state of the tile's elements and (first tile of a row of tiles) the bias
  // gradient's partial column sums are fetched / formed up front, beside the product's own operand loads
  AdamPre pre_w = {0.f, 0.f, 0.f, 0.f}, pre_b = {0.f, 0.f, 0.f, 0.f};
  float sbias = 0.f;
  DBG(0);
  if (T.type == T_DW) {
    pre_w = update_prefetch(U, T.w_off + (int64_t)m * T.N + n, m < T.M && n < T.N);
    if (tn == 0) {
      pre_b = update_prefetch(U, T.b_off + m0 + t, t < 16 && m0 + t < T.M);
      // bias gradient of rows m0 .. m0 + 15: column sums of G over the batch, 16 row segments per column; eight rows'
      // loads in flight at a time, added in row order
      const int mm = m0 + col, mc = mm < T.M ? mm : 0;
      const int rps = T.K / 16;
      const int iend = (row + 1) * rps;
      for (int i0 = row * rps; i0 < iend; i0 += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool on = i0 + j < iend;
          const float x = (virt && as1) ? grad_at<COH, true>(T.G, on ? i0 + j : i0, mc, T.M, true)
                                        : grad_at<COH, false>(T.G, on ? i0 + j : i0, mc, T.M, virt);
          v[j] = keep(x, on);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) sbias += v[j];
      }
    }
  }

  DBG(1);
  f4v acc;
  if (T.type == T_FWD)
    acc = T.fast ? tile_product<T_FWD, true, COH>(T, m0, n0, tn, kb, ke, r, q) : tile_product<T_FWD, false, COH>(T, m0, n0, tn, kb, ke, r, q);
  else if (T.type == T_DX)
    acc = !virt ? tile_product<T_DX, false, COH>(T, m0, n0, tn, kb, ke, r, q)
                : (as1 ? tile_product<T_DX, true, COH, true>(T, m0, n0, tn, kb, ke, r, q)
                       : tile_product<T_DX, true, COH, false>(T, m0, n0, tn, kb, ke, r, q));
  else
    acc = !virt ? tile_product<T_DW, false, COH>(T, m0, n0, tn, kb, ke, r, q)
                : (as1 ? tile_product<T_DW, true, COH, true>(T, m0, n0, tn, kb, ke, r, q)
                       : tile_product<T_DW, true, COH, false>(T, m0, n0, tn, kb, ke, r, q));

  // lane (r, q), register c holds element (row 4 q + c, column r) of the wave's partial tile
#pragma unroll
  for (int c = 0; c < 4; ++c) red[w][(4 * q + c) * 16 + r] = acc[c];
  DBG(2);
  __syncthreads();
  float v = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
  DBG(3);

  if (T.type == T_FWD) {
    if (n < T.N) {
      v += ld1(T.b + n);
      if (T.act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (T.act == ACT_TANH) v = tanhf(v);
      stf<COH>(T.y + (size_t)m * T.ldy + n, v);
      if (T.td_dcur != nullptr && n == 0) {
        const float y = ld1(T.td_reward + m) + ((1.0f - ld1(T.td_done + m)) * T.td_gamma) * v;
        const float td = ldf<COH>(T.td_cur + m) - y;
        float g;
        const float l = trl_loss_elem(td, T.td_kind, T.td_hdelta, &g);
        stf<COH>(T.td_dcur + m, (1.0f / (float)T.M) * g);
        stf<COH>(T.td_lterm + m, l);
      }
    }
    return;
  }
  if (T.type == T_DX) {
    if (n < T.N) {
      if (T.act != ACT_NONE) {
        const float hv = ldf<COH>(T.hin + (size_t)m * T.ldhin + n);
        v = T.act == ACT_RELU ? (hv > 0.f ? v : 0.f) : v * (1.0f - hv * hv);
      }
      stf<COH>(T.y + (size_t)m * T.ldy + n, v);
    }
    return;
  }
  if (m < T.M && n < T.N) apply_update(U, T.w_off + (int64_t)m * T.N + n, v, pre_w);
  DBG(4);
  if (tn != 0) return;
  __syncthreads();
  red[0][t] = sbias;
  __syncthreads();
  if (t < 16 && m0 + t < T.M) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += red[0][g * 16 + t];
    apply_update(U, T.b_off + m0 + t, s, pre_b);
  }
  DBG(5);
}

// critic loss = mean of the rows' loss terms, actor loss = -mean(q_pi) (ddpg_controller.py:105-112).  One workgroup;
// thread-strided partial sums, then the block sum: a fixed order.
template <bool COH>
__device__ __forceinline__ void ddpg_sums_body(const LossArgs &L, float *red) {
  const float invB = 1.0f / (float)L.B;
  float s_c = 0.f, s_a = 0.f;
  for (int b = threadIdx.x; b < L.B; b += 256) {
    s_c += ldf<COH>(L.lterm + b);
    s_a += ldf<COH>(L.q_pi + b);
  }
  s_c = trl_block_sum<4>(s_c, red);
  s_a = trl_block_sum<4>(s_a, red);
  if (threadIdx.x == 0) {
    L.losses_out[0] = -(s_a * invB);
    L.losses_out[1] = s_c * invB;
  }
}

// The TD loss of a DQN step as a member of a chain (B <= 4096 rows: one workgroup).  Row by row the arithmetic of
// k_td_loss (trl_loss.hip); a thread's rows are added in order, then the block sum: for B <= 256 (a row a thread) the
// same summation as k_td_loss's single block, i.e. the same bits; beyond that another fixed order.
template <bool COH>
__device__ __forceinline__ void dqn_td_body(const LossArgs &L, float *red) {
  const float invB = 1.0f / (float)L.B;
  const int A = L.A;
  float acc = 0.f;
  for (int b = threadIdx.x; b < L.B; b += 256) {
    int a = (int)ld1(L.action + b);  // action.long()
    a = a < 0 ? 0 : (a >= A ? A - 1 : a);
    const float *tn = L.qnt + (size_t)b * A;
    float nq;
    if (L.qno) {  // Double-DQN: first argmax of the online net, value from the target net
      const float *on = L.qno + (size_t)b * A;
      int am = 0;
      float best = ldf<COH>(on);
      for (int i = 1; i < A; ++i) {
        const float x = ldf<COH>(on + i);
        if (x > best) { best = x; am = i; }
      }
      nq = ldf<COH>(tn + am);
    } else {
      nq = ldf<COH>(tn);
      for (int i = 1; i < A; ++i) nq = fmaxf(nq, ldf<COH>(tn + i));
    }
    const float y = ld1(L.reward + b) + (L.gamma * nq) * (1.0f - ld1(L.done + b));      // reward + gamma * next_q * (1 - done)
    const float td = y - ldf<COH>(L.q + (size_t)b * A + a);
    float g;
    acc += trl_loss_elem(td, L.td_kind, L.hdelta, &g);
    const float gq = -(invB * g);  // d loss / d q[b, a]
    for (int i = 0; i < A; ++i) stf<COH>(L.dq + (size_t)b * A + i, (i == a) ? gq : 0.0f);
  }
  const float s = trl_block_sum<4>(acc, red);
  if (threadIdx.x == 0) L.losses_out[0] = s * invB;
}

__global__ __launch_bounds__(256) void k_phase(PhaseArgs P) {
  __shared__ float red[4][256];
  const int wg = blockIdx.x;
  if (P.sum_extra && wg == P.wgs) {
    if (P.L.kind == 1) dqn_td_body<false>(P.L, &red[0][0]);
    else ddpg_sums_body<false>(P.L, &red[0][0]);
    return;
  }
  int pi = 0;
#pragma unroll
  for (int i = 1; i < MAX_TILES; ++i)
    if (i < P.nt && wg >= P.t[i].wg0) pi = i;
  const Tile &T = P.t[pi];
  phase_tile<false>(T, P.u[T.upd], wg - T.wg0, red);
}

// y = reward + (1 - done) * gamma * next_q;  critic loss = mean(l(cur - y));  actor loss = -mean(q_pi)
// (ddpg_controller.py:105-112), and the gradient of the critic loss at cur.  One workgroup.
// ---- the whole chain in ONE launch: every workgroup walks the phases, taking the tiles wg, wg + gridDim.x, ... of each,
// with a device-wide barrier between two phases.  The barrier needs every workgroup resident at once: the host sizes the
// grid from the occupancy the runtime reports for this kernel, but another process or stream on the card can still
// take the CUs away, so every wait is bounded AND a wait that runs out is reported, never passed over: the workgroup
// raises the launch's abort flag (which the polling workgroups, and workgroups that only start now, see and leave on),
// writes 1 + the phase index to a pinned host word and NaN to the step's losses; the host finds the word at its next
// call on the stream, clears the barrier block, returns TRL_EHIP and runs that stream's later steps as launches.
// bar (32-dword lines): line 0 the arrival count, line 1 the generation word; the generation word holding CHAIN_POISON
// IS the abort flag, so the polls see it for free and the workgroup that arrives last cannot complete a barrier the
// others have given up on (it publishes the new generation by compare-and-swap from the old one).
constexpr unsigned CHAIN_SPIN_LIMIT = 1u << 20;      // polls of ~1.5 us: about 1.5 s (a phase is under 10 us)
constexpr unsigned CHAIN_POISON = 0xFFFFFFFFu;

// returns false when the launch is aborted (this workgroup's wait ran out, or another's did)
__device__ __forceinline__ bool chain_barrier(unsigned *bar, unsigned &gen, int nsleep, unsigned spin_limit,
                                              unsigned *fault_h, int phase, int *ok_lds) {
  __builtin_amdgcn_s_waitcnt(0);     // this thread's write-through stores have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned want = gen + 1;
    if (want == CHAIN_POISON) want = 0;     // (every workgroup passes every barrier, so equality is the test and wrapping is harmless)
    int ok = 0;
    if (__hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
      __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_waitcnt(0);
      unsigned expect = gen;
      ok = __hip_atomic_compare_exchange_strong(bar + 32, &expect, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;      // fails only on the poison
    } else {
      bool poisoned = false;
      for (unsigned spin = 0; spin < spin_limit; ++spin) {
        const unsigned v = __hip_atomic_load(bar + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v == want) { ok = 1; break; }
        if (v == CHAIN_POISON) { poisoned = true; break; }
        for (int j = 0; j < nsleep; ++j) __builtin_amdgcn_s_sleep(4);     // (polls of idle workgroups slow the busy ones)
      }
      if (!ok && !poisoned) {      // this workgroup's wait ran out: abort the launch, tell the host (the first one's phase stays)
        if (__hip_atomic_exchange(bar + 32, CHAIN_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != CHAIN_POISON)
          __hip_atomic_store(fault_h, (unsigned)(1 + phase), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    gen = want;
    *ok_lds = ok;
  }
  __syncthreads();
  return *ok_lds != 0;
}

constexpr int MAX_PHASES = 12;

__global__ __launch_bounds__(256) void k_chain(const PhaseArgs *__restrict__ table, int nphase, unsigned *bar, int nsleep,
                                               unsigned spin_limit, unsigned *fault_h, int flags) {
  __shared__ float red[4][256];
  __shared__ __align__(16) int tab[MAX_PHASES * (sizeof(PhaseArgs) / 4)];
  __shared__ int ok_lds;
  static_assert(sizeof(PhaseArgs) % 4 == 0, "dword copy");
  constexpr int PW = sizeof(PhaseArgs) / 4;
#ifdef TRL_TUNING      // diagnostic build: the last workgroup never arrives (the test of the barrier's time-out path)
  if ((flags & 1) && blockIdx.x == gridDim.x - 1 && gridDim.x > 1) return;
#endif
  // the phase table (pinned host memory, written by the host for this launch) -> LDS, one round trip
  for (int e = threadIdx.x; e < nphase * PW; e += 256) tab[e] = ((const int *)table)[e];
  unsigned gen = 0;
  if (threadIdx.x == 0) {
    gen = __hip_atomic_load(bar + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok_lds = gen == CHAIN_POISON ? 0 : 1;    // an earlier launch on this stream was aborted and the host has not yet
  }                                          // cleared the block: do nothing (the host reports it at its next call)
  __syncthreads();
  if (!ok_lds) return;
  for (int p = 0; p < nphase; ++p) {
#ifdef TRL_TUNING      // diagnostic build: 100 MHz stamps of workgroup 0 at each phase's start and at its barrier
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      ((unsigned long long *)(bar + 64))[2 * p] = __builtin_amdgcn_s_memrealtime();
      g_dbg_phase = p;
    }
#endif
    const PhaseArgs &P = *(const PhaseArgs *)(tab + p * PW);
    {
      for (int wg = blockIdx.x; wg < P.wgs + P.sum_extra; wg += gridDim.x) {
        if (wg == P.wgs) {      // (sum_extra) the losses of the phase before
          union { LossArgs l; int w[sizeof(LossArgs) / 4]; } L;
          const int *lsrc = (const int *)&P.L;
#pragma unroll
          for (int i = 0; i < (int)(sizeof(LossArgs) / 4); ++i) L.w[i] = __builtin_amdgcn_readfirstlane(lsrc[i]);
          if (L.l.kind == 1) dqn_td_body<true>(L.l, &red[0][0]);
          else ddpg_sums_body<true>(L.l, &red[0][0]);
          __syncthreads();
          continue;
        }
        int pi = 0;
#pragma unroll
        for (int i = 1; i < MAX_TILES; ++i)
          if (i < P.nt && wg >= P.t[i].wg0) pi = i;
        // the tile's description as scalars (it is the same for the whole workgroup)
        union { Tile t; int w[sizeof(Tile) / 4]; } T;
        const int *src = (const int *)&P.t[pi];
#pragma unroll
        for (int i = 0; i < (int)(sizeof(Tile) / 4); ++i) T.w[i] = __builtin_amdgcn_readfirstlane(src[i]);
        union { Update u; int w[sizeof(Update) / 4]; } U;
        const int *usrc = (const int *)&P.u[T.t.upd];
#pragma unroll
        for (int i = 0; i < (int)(sizeof(Update) / 4); ++i) U.w[i] = __builtin_amdgcn_readfirstlane(usrc[i]);
        phase_tile<true>(T.t, U.u, wg - T.t.wg0, red);
        __syncthreads();     // `red` is free for the next tile
      }
    }
#ifdef TRL_TUNING
    if (blockIdx.x == 0 && threadIdx.x == 0) ((unsigned long long *)(bar + 64))[2 * p + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    if (p + 1 < nphase && !chain_barrier(bar, gen, nsleep, spin_limit, fault_h, p, &ok_lds)) {
      // aborted: no further phase runs (no further update is applied); the step's loss values say so
      if (threadIdx.x == 0)
        for (int q = 0; q < nphase; ++q) {
          const PhaseArgs &Q = *(const PhaseArgs *)(tab + q * PW);
          if (Q.sum_extra && Q.L.losses_out) {
            Q.L.losses_out[0] = __builtin_nanf("");
            if (Q.L.kind == 0) Q.L.losses_out[1] = __builtin_nanf("");
          }
        }
      return;
    }
  }
#ifdef TRL_TUNING
  if (blockIdx.x == 0 && threadIdx.x == 0) ((unsigned long long *)(bar + 64))[2 * nphase] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---------------------------------------------------------------------------------------------- host side
struct Phase {
  PhaseArgs a;
  int wgs;
  Phase() : wgs(0) { memset(&a, 0, sizeof(a)); }
  Tile &add(int type, int M, int N, int K, int tiles_m) {
    Tile &T = a.t[a.nt++];
    T.type = type;
    T.M = M;
    T.N = N;
    T.K = K;
    T.tiles_n = (N + 15) / 16;
    T.wg0 = wgs;
    wgs += tiles_m * T.tiles_n;
    return T;
  }
};

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

// y[B, N] = act([x1 | x2] @ w^T + b)
void add_fwd(Phase &ph, int B, const float *x1, int K1, const float *x2, int K2, const float *w, const float *b, int N,
             int act, float *y) {
  Tile &T = ph.add(T_FWD, B, N, K1 + K2, B / 16);
  T.x1 = x1; T.K1 = K1; T.x2 = x2; T.K2 = K2;
  T.w = w; T.ldw = K1 + K2; T.b = b; T.act = act; T.y = y; T.ldy = N;
  T.fast = K2 == 0 && K1 % 4 == 0 && aligned16(x1) && aligned16(w);
}

Grad g_virtual(const float *seed, float cseed, int As, const float *w2, const float *h) {
  Grad G;
  G.g = nullptr; G.ld = 0; G.seed = seed; G.cseed = cseed; G.As = As; G.w2 = w2; G.h = h;
  return G;
}

Grad g_material(const float *g, int ld) {
  Grad G;
  G.g = g; G.ld = ld; G.seed = nullptr; G.cseed = 0.f; G.As = 0; G.w2 = nullptr; G.h = nullptr;
  return G;
}

// y[B, N] = (G[B, K] @ w[K, woff : woff + N]) * mask(hin)
void add_dx(Phase &ph, int B, Grad G, int K, const float *w, int ldw, int woff, int N, int act, const float *hin,
            int ldhin, float *y, float *gout) {
  Tile &T = ph.add(T_DX, B, N, K, B / 16);
  T.G = G; T.w = w; T.ldw = ldw; T.woff = woff; T.act = act; T.hin = hin; T.ldhin = ldhin; T.y = y; T.ldy = N;
  T.gout = gout;
}

// dW[M, N] = G[B, M]^T @ [x1 | x2][B, N], db[M] = column sums of G; consumed by Update `upd`
void add_dw(Phase &ph, int B, Grad G, int M, const float *x1, int K1, const float *x2, int K2, int64_t w_off,
            int64_t b_off, int upd) {
  Tile &T = ph.add(T_DW, M, K1 + K2, B, (M + 15) / 16);
  T.G = G; T.x1 = x1; T.K1 = K1; T.x2 = x2; T.K2 = K2; T.w_off = w_off; T.b_off = b_off; T.upd = upd;
}

// ---- the one-launch chain's host state, one per (device, stream): a barrier block in device memory and a ring of
// phase tables in pinned host memory (a slot is reused only after the launch that read it has finished).  Launches
// on one stream run one after the other, so they can share the barrier block; the number of states is bounded by the
// number of streams the caller uses, whatever the number of learners and workspaces.
constexpr int CHAIN_SLOTS = 16;
struct ChainState {
  unsigned *bar = nullptr;          // device, 256 dwords, zeroed once; the barrier leaves it clean
  PhaseArgs *tables = nullptr;      // pinned host: CHAIN_SLOTS x MAX_PHASES
  hipEvent_t ev[CHAIN_SLOTS];
  bool used[CHAIN_SLOTS];
  uint64_t n = 0;
  int max_wgs = 0;                  // resident-workgroup budget: what the runtime says fits of k_chain, at most two per CU
  unsigned *fault_h = nullptr;      // pinned host word: 0, or 1 + the phase whose barrier timed out (written by the kernel)
  bool launches_only = false;       // after a reported time-out this stream's steps run as one launch per phase
};
std::mutex g_chain_mu;
std::unordered_map<uint64_t, ChainState *> g_chains;     // key: stream handle ^ (device << 56)
int g_chain_on = 1;

ChainState *chain_state(const char *fn, hipStream_t st) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    trl_set_error("%s: hipGetDevice failed", fn);
    return nullptr;
  }
  const uint64_t key = (uint64_t)(uintptr_t)st ^ ((uint64_t)dev << 56);
  std::lock_guard<std::mutex> lk(g_chain_mu);
  auto it = g_chains.find(key);
  if (it != g_chains.end()) return it->second;
  ChainState *c = new ChainState();
  int per_cu = 0;      // workgroups of k_chain (256 threads, its static LDS, its registers) the runtime says fit on one CU
  bool ok = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chain, 256, 0) == hipSuccess &&
            hipMalloc((void **)&c->bar, 256 * sizeof(unsigned)) == hipSuccess &&
            hipMemset(c->bar, 0, 256 * sizeof(unsigned)) == hipSuccess &&
            hipHostMalloc((void **)&c->tables, sizeof(PhaseArgs) * CHAIN_SLOTS * MAX_PHASES, hipHostMallocDefault) == hipSuccess &&
            hipHostMalloc((void **)&c->fault_h, 64, hipHostMallocDefault) == hipSuccess;
  if (ok) *(volatile unsigned *)c->fault_h = 0u;
  for (int i = 0; ok && i < CHAIN_SLOTS; ++i) {
    ok = hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming) == hipSuccess;
    c->used[i] = false;
  }
  if (!ok) {
    (void)hipGetLastError();
    trl_set_error("%s: could not set up the one-launch chain (device / pinned allocation)", fn);
    delete c;
    return nullptr;
  }
  c->max_wgs = (per_cu < 2 ? per_cu : 2) * cus;      // 0: the kernel does not fit at all -> launches (run_chain)
  g_chains[key] = c;
  return c;
}

// a step's phases, in order; run as one launch each, or all in one launch (k_chain)
struct Chain {
  PhaseArgs ph[MAX_PHASES];
  int n = 0;
  bool full = false;      // (more phases than the table holds: refused by the runners)
  void add(const Phase &p) {
    if (n >= MAX_PHASES) { full = true; return; }
    ph[n] = p.a; ph[n].wgs = p.wgs; ++n;
  }
  // one more workgroup in the phase added last: the two loss values from the rows' terms
  void add_sums(const LossArgs &L) { ph[n - 1].L = L; ph[n - 1].sum_extra = 1; }
  // a phase of its own that is one workgroup's loss body and no tiles
  void add_loss_phase(const LossArgs &L) {
    if (n >= MAX_PHASES) { full = true; return; }
    memset(&ph[n], 0, sizeof(PhaseArgs));
    ph[n].L = L; ph[n].sum_extra = 1; ++n;
  }
};

int run_launches(const Chain &c, hipStream_t st, const char *what) {
  TRL_REQUIRE(!c.full, "%s: internal: more than %d phases", what, MAX_PHASES);
  for (int i = 0; i < c.n; ++i) {
    hipLaunchKernelGGL(k_phase, dim3(c.ph[i].wgs + c.ph[i].sum_extra), dim3(256), 0, st, c.ph[i]);
    TRL_CHECK_LAUNCH(what);
  }
  return TRL_OK;
}

int run_chain(const Chain &c, hipStream_t st, const char *what) {
  TRL_REQUIRE(!c.full, "%s: internal: more than %d phases", what, MAX_PHASES);
  ChainState *cs = chain_state(what, st);
  if (!cs) return TRL_EHIP;
  if (const unsigned f = *(volatile unsigned *)cs->fault_h) {
    // a launch of the chain on this stream gave up at a barrier (a workgroup was not resident: the card is shared, or
    // the occupancy estimate was wrong).  That step -- and any step queued behind it, which did nothing -- is lost; say
    // so, clean the barrier block behind them on the stream, and keep to one launch per phase on this stream from now on.
    *(volatile unsigned *)cs->fault_h = 0u;
    cs->launches_only = true;
    if (hipMemsetAsync(cs->bar, 0, 256 * sizeof(unsigned), st) != hipSuccess) (void)hipGetLastError();
    trl_set_error("%s: an earlier one-launch step on this stream timed out at the device-wide barrier after phase %u (not "
                  "every workgroup was resident); its losses are NaN, its update is incomplete and steps queued behind it "
                  "did not run.  This stream now runs the step as one launch per phase", what, f - 1u);
    return TRL_EHIP;
  }
  if (cs->launches_only || cs->max_wgs < 1) return run_launches(c, st, what);
  const int slot = (int)(cs->n++ % CHAIN_SLOTS);
  if (cs->used[slot] && hipEventSynchronize(cs->ev[slot]) != hipSuccess) {    // (16 launches back: long finished)
    trl_set_error("%s: waiting for an earlier launch of the chain failed", what);
    return TRL_EHIP;
  }
  PhaseArgs *tab = cs->tables + (size_t)slot * MAX_PHASES;
  memcpy(tab, c.ph, sizeof(PhaseArgs) * c.n);
  int wgs = 1;
  for (int i = 0; i < c.n; ++i) wgs = c.ph[i].wgs + c.ph[i].sum_extra > wgs ? c.ph[i].wgs + c.ph[i].sum_extra : wgs;
  wgs = wgs < cs->max_wgs ? wgs : cs->max_wgs;
#ifdef TRL_TUNING
  if (const char *e = getenv("TRL_CHAIN_WGS")) wgs = atoi(e) < wgs ? atoi(e) : wgs;     // A/B runs of tools/ only
#endif
  int nsleep = 2, flags = 0;
  unsigned spin_limit = CHAIN_SPIN_LIMIT;
#ifdef TRL_TUNING
  if (const char *e = getenv("TRL_CHAIN_SLEEP")) nsleep = atoi(e);
  if (const char *e = getenv("TRL_CHAIN_SPIN")) spin_limit = (unsigned)atoi(e);
  if (const char *e = getenv("TRL_CHAIN_WITHHOLD")) {      // the n-th chain launch of the stream loses a workgroup (n = value)
    if ((uint64_t)atoll(e) == cs->n) flags |= 1;
  }
#endif
  hipLaunchKernelGGL(k_chain, dim3(wgs), dim3(256), 0, st, (const PhaseArgs *)tab, c.n, cs->bar, nsleep, spin_limit,
                     cs->fault_h, flags);
  TRL_CHECK_LAUNCH(what);
  if (hipEventRecord(cs->ev[slot], st) != hipSuccess) {
    trl_set_error("%s: hipEventRecord failed", what);
    return TRL_EHIP;
  }
  cs->used[slot] = true;
#ifdef TRL_TUNING
  if (getenv("TRL_CHAIN_STAMPS") && cs->n % 200 == 0) {      // (synchronises: diagnostics only)
    unsigned long long h[2 * MAX_PHASES + 1];
    if (hipMemcpy(h, cs->bar + 64, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
      fprintf(stderr, "k_chain, %d workgroups; phase: tiles' workgroups, us of workgroup 0 in the phase, us to the next phase's start\n", wgs);
      for (int i = 0; i < c.n; ++i)
        fprintf(stderr, "  phase %2d: %4d  %6.2f  %6.2f\n", i, c.ph[i].wgs, (double)(h[2 * i + 1] - h[2 * i]) / 100.0,
                (double)(h[2 * i + 2] - h[2 * i + 1]) / 100.0);
      fprintf(stderr, "  total %6.2f us\n", (double)(h[2 * c.n] - h[0]) / 100.0);
      unsigned long long d[16 * 8];
      if (hipMemcpyFromSymbol(d, HIP_SYMBOL(g_dbg), sizeof(d)) == hipSuccess)
        for (int i = 0; i < c.n; ++i)
          fprintf(stderr, "  tile of phase %2d: prefetch %5.2f  product %5.2f  lds %5.2f  update %5.2f  bias %5.2f\n", i,
                  (double)(d[i * 8 + 1] - d[i * 8]) / 100.0, (double)(d[i * 8 + 2] - d[i * 8 + 1]) / 100.0,
                  (double)(d[i * 8 + 3] - d[i * 8 + 2]) / 100.0, (double)(d[i * 8 + 4] - d[i * 8 + 3]) / 100.0,
                  (double)(d[i * 8 + 5] - d[i * 8 + 4]) / 100.0);
    }
  }
#endif
  return TRL_OK;
}

int launch(Phase &ph, hipStream_t st, const char *what) {
  ph.a.wgs = ph.wgs;
  hipLaunchKernelGGL(k_phase, dim3(ph.wgs), dim3(256), 0, st, ph.a);
  TRL_CHECK_LAUNCH(what);
  return TRL_OK;
}

Update make_update(const trl_adam_args *o, float *target, double tau) {
  Update U;
  memset(&U, 0, sizeof(U));
  U.param = o->param; U.grad = o->grad; U.m = o->exp_avg; U.v = o->exp_avg_sq; U.target = target;
  U.mode = UPD_GRAD;
  if (o->dev_state) {
    U.dev = (const TrlAdamScalars *)o->dev_state->derived;
    U.mode = target ? UPD_ADAM_POLYAK : UPD_ADAM;
    U.tau = (float)tau;
    U.omt = (float)(1.0 - tau);
  } else if (o->step > 0) {
    U.s = trl_adam_scalars(o->lr, o->beta1, o->beta2, o->eps, o->step, o->grad_scale, o->clip_value);
    U.mode = target ? UPD_ADAM_POLYAK : UPD_ADAM;
    U.tau = (float)tau;
    U.omt = (float)(1.0 - tau);
  }
  return U;
}

constexpr int MAX_WIDTH = 1024, MAX_BATCH = 4096;

int check_net(const char *fn, const char *name, const trl_mlp_desc *n, const trl_adam_args *opt, bool u8 = false) {
  TRL_REQUIRE(n != nullptr, "%s: %s is NULL", fn, name);
  TRL_REQUIRE(n->depth == 3, "%s: %s must have two hidden layers (depth 3), got depth %d", fn, name, n->depth);
  TRL_REQUIRE(n->dims[0] >= 1 && (u8 ? n->dims[0] <= (1 << 20) && n->dims[0] % 32 == 0 : n->dims[0] <= MAX_WIDTH) &&
              n->dims[3] >= 1 && n->dims[3] <= MAX_WIDTH, "%s: %s fan-in / fan-out out of range", fn, name);
  TRL_REQUIRE(!u8 || n->dims[1] % 128 == 0, "%s: %s: the uint8 first layer needs a fan-out that is a multiple of 128", fn,
              name);
  for (int l = 1; l <= 2; ++l)
    TRL_REQUIRE(n->dims[l] >= 16 && n->dims[l] % 16 == 0 && n->dims[l] <= MAX_WIDTH,
                "%s: %s hidden width %d must be a multiple of 16 in [16, %d]", fn, name, n->dims[l], MAX_WIDTH);
  for (int l = 0; l < 3; ++l) {
    TRL_REQUIRE(n->w[l] != nullptr && n->b[l] != nullptr, "%s: %s has a NULL parameter", fn, name);
    TRL_REQUIRE(aligned16(n->w[l]), "%s: %s weights must be 16 B aligned", fn, name);
    if (opt) {
      const int64_t wo = n->w[l] - opt->param, bo = n->b[l] - opt->param;
      TRL_REQUIRE(wo >= 0 && wo + (int64_t)n->dims[l + 1] * n->dims[l] <= opt->n && bo >= 0 && bo + n->dims[l + 1] <= opt->n,
                  "%s: %s parameters must lie inside the optimizer's flat buffer", fn, name);
    }
  }
  return TRL_OK;
}

bool same_shape(const trl_mlp_desc *a, const trl_mlp_desc *b) {
  return a->depth == b->depth && a->dims[0] == b->dims[0] && a->dims[1] == b->dims[1] && a->dims[2] == b->dims[2] &&
         a->dims[3] == b->dims[3];
}

struct Carve {
  float *base;
  size_t off;
  float *take(size_t n) {
    float *p = base ? base + off : nullptr;
    off += (n + 63) / 64 * 64;
    return p;
  }
};

struct DdpgWs {
  float *a_h0, *a_h1, *ta_h0, *ta_h1, *a_pi, *a_next;
  float *c_h0, *c_h1, *p_h0, *p_h1, *tc_h0, *tc_h1, *cur, *q_pi, *nq, *dcur, *lterm;
  float *g_p0, *g_c0, *dz2, *g_a0, *g_a1;
  size_t floats;
  DdpgWs(float *base, const trl_mlp_desc *actor, const trl_mlp_desc *critic, size_t B) {
    Carve c{base, 0};
    const size_t A = actor->dims[3];
    a_h0 = c.take(B * actor->dims[1]); a_h1 = c.take(B * actor->dims[2]);
    ta_h0 = c.take(B * actor->dims[1]); ta_h1 = c.take(B * actor->dims[2]);
    a_pi = c.take(B * A); a_next = c.take(B * A);
    c_h0 = c.take(B * critic->dims[1]); c_h1 = c.take(B * critic->dims[2]);
    p_h0 = c.take(B * critic->dims[1]); p_h1 = c.take(B * critic->dims[2]);
    tc_h0 = c.take(B * critic->dims[1]); tc_h1 = c.take(B * critic->dims[2]);
    cur = c.take(B); q_pi = c.take(B); nq = c.take(B); dcur = c.take(B); lterm = c.take(B);
    g_p0 = c.take(B * critic->dims[1]); g_c0 = c.take(B * critic->dims[1]);
    dz2 = c.take(B * A);
    g_a0 = c.take(B * actor->dims[1]); g_a1 = c.take(B * actor->dims[2]);
    floats = c.off;
  }
};

struct DqnWs {
  float *h0[3], *h1[3], *q[3], *dq, *g0, *g1;
  size_t floats;
  DqnWs(float *base, const trl_mlp_desc *net, size_t B) {
    Carve c{base, 0};
    for (int j = 0; j < 3; ++j) {
      h0[j] = c.take(B * net->dims[1]);
      h1[j] = c.take(B * net->dims[2]);
      q[j] = c.take(B * net->dims[3]);
    }
    dq = c.take(B * net->dims[3]);
    g0 = c.take(B * net->dims[1]);
    g1 = c.take(B * net->dims[2]);
    floats = c.off;
  }
};

}  // namespace

extern "C" void trl_offpolicy_set_chain(int32_t on) { g_chain_on = on < 0 ? 0 : (on > 2 ? 2 : on); }

extern "C" int64_t trl_ddpg_learn_workspace_bytes(const trl_mlp_desc *actor, const trl_mlp_desc *critic, int64_t B) {
  if (!actor || !critic || B <= 0) return 0;
  return (int64_t)(DdpgWs(nullptr, actor, critic, (size_t)B).floats * sizeof(float));
}

extern "C" int trl_ddpg_learn_fused(const trl_mlp_desc *actor, const trl_mlp_desc *critic,
                                    const trl_mlp_desc *target_actor, const trl_mlp_desc *target_critic,
                                    const float *obs, const float *action, const float *reward, const float *next_obs,
                                    const float *done, int64_t B64, double gamma, int32_t kind, double huber_delta,
                                    float *losses_out, const trl_adam_args *actor_opt, const trl_adam_args *critic_opt,
                                    float *target_actor_flat, float *target_critic_flat, double tau, void *ws,
                                    void *stream) {
  const char *fn = "trl_ddpg_learn_fused";
  TRL_REQUIRE(actor_opt && critic_opt, "%s: both optimizer blocks are required (step == 0: gradients only)", fn);
  int rc;
  if ((rc = check_net(fn, "actor", actor, actor_opt)) || (rc = check_net(fn, "critic", critic, critic_opt)) ||
      (rc = check_net(fn, "target_actor", target_actor, nullptr)) ||
      (rc = check_net(fn, "target_critic", target_critic, nullptr)))
    return rc;
  TRL_REQUIRE(same_shape(actor, target_actor) && same_shape(critic, target_critic),
              "%s: target networks must have the shapes of the online ones", fn);
  const int S = actor->dims[0], A = actor->dims[3];
  TRL_REQUIRE(critic->dims[0] == S + A && critic->dims[3] == 1,
              "%s: critic must map [obs | action] (%d + %d) to one value", fn, S, A);
  TRL_REQUIRE(B64 >= 16 && B64 % 16 == 0 && B64 <= MAX_BATCH, "%s: B = %lld must be a multiple of 16 in [16, %d]", fn,
              (long long)B64, MAX_BATCH);
  TRL_REQUIRE(obs && action && reward && next_obs && done && losses_out && ws, "%s: NULL argument", fn);
  TRL_REQUIRE(kind == TRL_LOSS_MSE || (kind == TRL_LOSS_HUBER && huber_delta > 0), "%s: bad loss kind", fn);
  TRL_REQUIRE(aligned16(ws), "%s: workspace must be 16 B aligned", fn);
  const bool update = actor_opt->step > 0 || actor_opt->dev_state != nullptr;
  TRL_REQUIRE(update == (critic_opt->step > 0 || critic_opt->dev_state != nullptr), "%s: update both networks or neither", fn);
  TRL_REQUIRE((actor_opt->dev_state != nullptr) == (critic_opt->dev_state != nullptr),
              "%s: both optimizers keep their state on the device, or neither", fn);
  TRL_REQUIRE(!update || (target_actor_flat && target_critic_flat), "%s: target flat buffers are required", fn);
  TRL_REQUIRE(!update || actor_opt->dev_state || (actor_opt->grad_scale == 1.0 && critic_opt->grad_scale == 1.0),
              "%s: the in-launch update takes unscaled gradients", fn);

  const int B = (int)B64;
  hipStream_t st = (hipStream_t)stream;
  const DdpgWs W((float *)ws, actor, critic, (size_t)B);
  const int Ha0 = actor->dims[1], Ha1 = actor->dims[2], Hc0 = critic->dims[1], Hc1 = critic->dims[2];
  const trl_mlp_desc *ac = actor, *ta = target_actor, *cr = critic, *tc = target_critic;
  const Update Ua = make_update(actor_opt, update ? target_actor_flat : nullptr, tau);
  const Update Uc = make_update(critic_opt, update ? target_critic_flat : nullptr, tau);
  auto a_off = [&](const float *p) { return (int64_t)(p - actor_opt->param); };
  auto c_off = [&](const float *p) { return (int64_t)(p - critic_opt->param); };

  if (actor_opt->dev_state) {   // graph replay: advance the device step counts, derive this step's scalars
    if ((rc = trl_adam_prep_launch(fn, actor_opt->dev_state, st)) || (rc = trl_adam_prep_launch(fn, critic_opt->dev_state, st)))
      return rc;
  }
  Chain ch;
  {  // layer 0 of actor(obs), target_actor(next_obs), critic(obs, action)
    Phase ph;
    add_fwd(ph, B, obs, S, nullptr, 0, ac->w[0], ac->b[0], Ha0, ACT_RELU, W.a_h0);
    add_fwd(ph, B, next_obs, S, nullptr, 0, ta->w[0], ta->b[0], Ha0, ACT_RELU, W.ta_h0);
    add_fwd(ph, B, obs, S, action, A, cr->w[0], cr->b[0], Hc0, ACT_RELU, W.c_h0);
    ch.add(ph);
  }
  {  // layer 1 of the same three
    Phase ph;
    add_fwd(ph, B, W.a_h0, Ha0, nullptr, 0, ac->w[1], ac->b[1], Ha1, ACT_RELU, W.a_h1);
    add_fwd(ph, B, W.ta_h0, Ha0, nullptr, 0, ta->w[1], ta->b[1], Ha1, ACT_RELU, W.ta_h1);
    add_fwd(ph, B, W.c_h0, Hc0, nullptr, 0, cr->w[1], cr->b[1], Hc1, ACT_RELU, W.c_h1);
    ch.add(ph);
  }
  {  // pi(obs), pi'(next_obs) (tanh), Q(obs, action)
    Phase ph;
    add_fwd(ph, B, W.a_h1, Ha1, nullptr, 0, ac->w[2], ac->b[2], A, ACT_TANH, W.a_pi);
    add_fwd(ph, B, W.ta_h1, Ha1, nullptr, 0, ta->w[2], ta->b[2], A, ACT_TANH, W.a_next);
    add_fwd(ph, B, W.c_h1, Hc1, nullptr, 0, cr->w[2], cr->b[2], 1, ACT_NONE, W.cur);
    ch.add(ph);
  }
  {  // critic(obs, pi(obs)) and target_critic(next_obs, pi'(next_obs)), three layers
    Phase ph;
    add_fwd(ph, B, obs, S, W.a_pi, A, cr->w[0], cr->b[0], Hc0, ACT_RELU, W.p_h0);
    add_fwd(ph, B, next_obs, S, W.a_next, A, tc->w[0], tc->b[0], Hc0, ACT_RELU, W.tc_h0);
    ch.add(ph);
  }
  {
    Phase ph;
    add_fwd(ph, B, W.p_h0, Hc0, nullptr, 0, cr->w[1], cr->b[1], Hc1, ACT_RELU, W.p_h1);
    add_fwd(ph, B, W.tc_h0, Hc0, nullptr, 0, tc->w[1], tc->b[1], Hc1, ACT_RELU, W.tc_h1);
    ch.add(ph);
  }
  {
    Phase ph;
    add_fwd(ph, B, W.p_h1, Hc1, nullptr, 0, cr->w[2], cr->b[2], 1, ACT_NONE, W.q_pi);
    add_fwd(ph, B, W.tc_h1, Hc1, nullptr, 0, tc->w[2], tc->b[2], 1, ACT_NONE, W.nq);
    // y = reward + (1 - done) * gamma * nq, the critic loss's row terms and its derivative at `cur` in that tile's epilogue
    Tile &T = ph.a.t[ph.a.nt - 1];
    T.td_cur = W.cur; T.td_reward = reward; T.td_done = done; T.td_dcur = W.dcur; T.td_lterm = W.lterm;
    T.td_gamma = (float)gamma; T.td_hdelta = (float)huber_delta; T.td_kind = (int)kind;
    ch.add(ph);
  }

  const Grad G_p1 = g_virtual(nullptr, -1.0f / (float)B, 1, cr->w[2], W.p_h1);   // d(-mean Q)/d p_h1
  const Grad G_c1 = g_virtual(W.dcur, 0.f, 1, cr->w[2], W.c_h1);                 // d critic_loss / d c_h1
  {  // through critic layer 1, both paths
    Phase ph;
    add_dx(ph, B, G_p1, Hc1, cr->w[1], Hc0, 0, Hc0, ACT_RELU, W.p_h0, Hc0, W.g_p0, nullptr);
    add_dx(ph, B, G_c1, Hc1, cr->w[1], Hc0, 0, Hc0, ACT_RELU, W.c_h0, Hc0, W.g_c0, nullptr);
    ch.add(ph);
    LossArgs L;      // (the loss values themselves: nothing waits for them, one extra workgroup of this phase sums them)
    L.lterm = W.lterm; L.q_pi = W.q_pi; L.B = B; L.losses_out = losses_out;
    ch.add_sums(L);
  }
  {  // critic layer 1's gradient (+ update); actor path: through critic layer 0's action columns and the tanh
    Phase ph;
    ph.a.u[0] = Uc;
    add_dw(ph, B, G_c1, Hc1, W.c_h0, Hc0, nullptr, 0, c_off(cr->w[1]), c_off(cr->b[1]), 0);
    add_dx(ph, B, g_material(W.g_p0, Hc0), Hc0, cr->w[0], S + A, S, A, ACT_TANH, W.a_pi, A, W.dz2, nullptr);
    ch.add(ph);
  }
  {  // through actor layer 1 (keeping the masked outer product for the next launch); critic layers 0 and 2
    Phase ph;
    ph.a.u[0] = Uc;
    add_dx(ph, B, g_virtual(W.dz2, 0.f, A, ac->w[2], W.a_h1), Ha1, ac->w[1], Ha0, 0, Ha0, ACT_RELU, W.a_h0, Ha0,
           W.g_a0, W.g_a1);
    add_dw(ph, B, g_material(W.g_c0, Hc0), Hc0, obs, S, action, A, c_off(cr->w[0]), c_off(cr->b[0]), 0);
    add_dw(ph, B, g_material(W.dcur, 1), 1, W.c_h1, Hc1, nullptr, 0, c_off(cr->w[2]), c_off(cr->b[2]), 0);
    ch.add(ph);
  }
  {  // the actor's three layers
    Phase ph;
    ph.a.u[0] = Ua;
    add_dw(ph, B, g_material(W.g_a1, Ha1), Ha1, W.a_h0, Ha0, nullptr, 0, a_off(ac->w[1]), a_off(ac->b[1]), 0);
    add_dw(ph, B, g_material(W.g_a0, Ha0), Ha0, obs, S, nullptr, 0, a_off(ac->w[0]), a_off(ac->b[0]), 0);
    add_dw(ph, B, g_material(W.dz2, A), A, W.a_h1, Ha1, nullptr, 0, a_off(ac->w[2]), a_off(ac->b[2]), 0);
    ch.add(ph);
  }
  // One launch for the whole chain, or one per phase.  A stream capture must not meet the chain's host-side event
  // wait (and a replayed graph would re-read a table slot the host has since rewritten), so a capturing stream and
  // graph mode (step counts on the device) keep the launches.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) {
    (void)hipGetLastError();
    cap = hipStreamCaptureStatusNone;
  }
  // (hipStreamPerThread is one handle for a different stream in every thread: no shared barrier block for it)
  if (g_chain_on && !actor_opt->dev_state && cap == hipStreamCaptureStatusNone && st != hipStreamPerThread) {
    if ((rc = run_chain(ch, st, fn))) return rc;
  } else if ((rc = run_launches(ch, st, fn))) {
    return rc;
  }
  return TRL_OK;
}

namespace {

int64_t dqn_ws_floats(const trl_mlp_desc *q_net, int64_t B, bool u8, int npass) {
  int64_t f = (int64_t)DqnWs(nullptr, q_net, (size_t)B).floats + (trl_loss_workspace_bytes() + 3) / 4;
  f = (f + 63) / 64 * 64;
  if (u8) f += trl_u8_fwd_ws_floats(B, q_net->dims[1], q_net->dims[0], npass);
  return f;
}

// obs / next_obs are float32 [B, S], or (u8) uint8 [B, S]: then layer 0 runs on csrc/trl_linear_u8.hip --
// all passes in one split-K launch + one reduce launch, its weight gradient in one launch, its Adam step
// as a streaming launch -- and the rest of the chain is unchanged (9 launches in all).
// rec != NULL (u8 only): obs / next_obs are the ring's base plus the frame columns' offsets and batch row b is record
// rec[b], `stride` bytes apart -- the first layer reads the frames where the replay ring holds them
int dqn_learn_impl(const char *fn, bool u8, const trl_mlp_desc *q_net, const trl_mlp_desc *target_net, const void *obs,
                   const float *action, const float *reward, const void *next_obs, const float *done, int64_t B64,
                   double gamma, int32_t double_dqn, int32_t kind, double huber_delta, float *loss_out,
                   const trl_adam_args *opt, void *ws, void *stream, const int32_t *rec = nullptr, int64_t rstride = 0) {
  TRL_REQUIRE(opt != nullptr, "%s: the optimizer block is required (step == 0: gradients only)", fn);
  int rc;
  if ((rc = check_net(fn, "q_net", q_net, opt, u8)) || (rc = check_net(fn, "target_net", target_net, nullptr, u8)))
    return rc;
  TRL_REQUIRE(same_shape(q_net, target_net), "%s: the target network must have the online one's shape", fn);
  const int bmul = u8 ? 32 : 16;
  TRL_REQUIRE(B64 >= bmul && B64 % bmul == 0 && B64 <= MAX_BATCH, "%s: B = %lld must be a multiple of %d in [%d, %d]",
              fn, (long long)B64, bmul, bmul, MAX_BATCH);
  TRL_REQUIRE(obs && action && reward && next_obs && done && loss_out && ws, "%s: NULL argument", fn);
  TRL_REQUIRE(aligned16(ws), "%s: workspace must be 16 B aligned", fn);
  TRL_REQUIRE(opt->step == 0 || opt->dev_state || opt->grad_scale == 1.0, "%s: the in-launch update takes unscaled gradients", fn);

  const int B = (int)B64;
  hipStream_t st = (hipStream_t)stream;
  if (opt->dev_state && (rc = trl_adam_prep_launch(fn, opt->dev_state, st))) return rc;
  const DqnWs W((float *)ws, q_net, (size_t)B);
  void *loss_ws = (float *)ws + W.floats;
  const int S = q_net->dims[0], H0 = q_net->dims[1], H1 = q_net->dims[2], A = q_net->dims[3];
  const Update U = make_update(opt, nullptr, 1.0);
  auto off = [&](const float *p) { return (int64_t)(p - opt->param); };
  // 0: q_net(obs), 1: target_net(next_obs), 2: q_net(next_obs) (Double-DQN's argmax, dqn_controller.py:68-69)
  const trl_mlp_desc *nets[3] = {q_net, target_net, q_net};
  const void *xs[3] = {obs, next_obs, next_obs};
  const int npass = double_dqn ? 3 : 2;
  float *u8_ws = (float *)ws + (W.floats + (trl_loss_workspace_bytes() + 3) / 4 + 63) / 64 * 64;
  // The four small dependent steps between the first layer's forward and its weight gradient -- layer 1, layer 2, the TD
  // loss, the input gradients -- as ONE launch (k_chain) when the first layer is the uint8 kernels' (those stay launches
  // of their own on either side); same conditions as the DDPG chain, same bits as the launches for B <= 256.  OPT-IN
  // (trl_offpolicy_set_chain(2)): measured slower than the four launches on the bench's step (0.127 - 0.129 against 0.122 - 0.125 ms at
  // B = 256 on one box -- this step is bound by the device, not by the host's launches: a barrier plus a one-workgroup loss phase
  // under 500 polling workgroups costs more than three kernel boundaries).
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  const bool mid_chain = u8 && g_chain_on >= 2 && !opt->dev_state && cap == hipStreamCaptureStatusNone && st != hipStreamPerThread;
  Chain mid;
  for (int l = 0; l < 3; ++l) {
    if (l == 0 && u8) {
      const uint8_t *x8[3];
      const float *w8[3], *b8[3];
      float *y8[3];
      for (int j = 0; j < npass; ++j) {
        x8[j] = (const uint8_t *)xs[j]; w8[j] = nets[j]->w[0]; b8[j] = nets[j]->b[0]; y8[j] = W.h0[j];
      }
      if ((rc = trl_u8_fwd_launch(fn, npass, x8, w8, b8, y8, B, S, H0, ACT_RELU, u8_ws, st, rec, rstride))) return rc;
      continue;
    }
    Phase ph;
    for (int j = 0; j < npass; ++j) {
      const float *x = l == 0 ? (const float *)xs[j] : (l == 1 ? W.h0[j] : W.h1[j]);
      float *y = l == 0 ? W.h0[j] : (l == 1 ? W.h1[j] : W.q[j]);
      add_fwd(ph, B, x, nets[j]->dims[l], nullptr, 0, nets[j]->w[l], nets[j]->b[l], nets[j]->dims[l + 1],
              l < 2 ? ACT_RELU : ACT_NONE, y);
    }
    if (mid_chain && l >= 1) mid.add(ph);
    else if ((rc = launch(ph, st, fn))) return rc;
  }
  Phase dx;
  add_dx(dx, B, g_virtual(W.dq, 0.f, A, q_net->w[2], W.h1[0]), H1, q_net->w[1], H0, 0, H0, ACT_RELU, W.h0[0], H0,
         W.g0, W.g1);
  if (mid_chain) {
    LossArgs L;
    memset(&L, 0, sizeof(L));
    L.kind = 1; L.B = B; L.losses_out = loss_out;
    L.q = W.q[0]; L.qnt = W.q[1]; L.qno = double_dqn ? W.q[2] : nullptr; L.action = action; L.reward = reward; L.done = done;
    L.dq = W.dq; L.A = A; L.td_kind = kind; L.gamma = (float)gamma; L.hdelta = (float)huber_delta;
    mid.add_loss_phase(L);
    mid.add(dx);
    if ((rc = run_chain(mid, st, fn))) return rc;
  } else {
    if ((rc = trl_td_loss(W.q[0], W.q[1], double_dqn ? W.q[2] : nullptr, action, reward, done, B, A, gamma, kind,
                          huber_delta, loss_out, W.dq, loss_ws, stream)))
      return rc;
    if ((rc = launch(dx, st, fn))) return rc;
  }
  if (u8) {   // layer 0 first: it reads nothing the next launch writes
    // The gradient launch fills one wave of workgroups (K / 128 of them), so an Adam epilogue would run
    // after the MFMA work of ALL of them instead of beside it (measured 61 us against 25 + 15 us apart):
    // the 3.6 M-parameter update is its own streaming launch over the layer's slice of the flat buffers.
    const int64_t wo = off(q_net->w[0]), bo = off(q_net->b[0]);
    if ((rc = trl_u8_wgrad_launch(fn, W.g0, (const uint8_t *)obs, B, S, H0, opt->grad + wo, opt->grad + bo, nullptr, st, rec,
                                  rstride)))
      return rc;
    if (U.mode >= UPD_ADAM) {
      const int64_t nw = (int64_t)H0 * S;
      const bool joined = bo >= wo + nw && bo - (wo + nw) < 64;   // bias right behind the weight (alignment gap only)
      const int64_t spans[2][2] = {{wo, joined ? bo + H0 - wo : nw}, {bo, joined ? 0 : H0}};
      for (int i = 0; i < 2; ++i) {
        if (spans[i][1] <= 0) continue;
        float *p = opt->param + spans[i][0], *g = opt->grad + spans[i][0], *m = opt->exp_avg + spans[i][0],
              *v = opt->exp_avg_sq + spans[i][0];
        rc = opt->dev_state ? trl_adam_dev_launch(fn, p, g, m, v, spans[i][1], opt->dev_state, st)
                            : trl_adam_step(p, g, m, v, spans[i][1], opt->lr, opt->beta1, opt->beta2, opt->eps, opt->step,
                                            opt->grad_scale, opt->clip_value, stream);
        if (rc) return rc;
      }
    }
  }
  {
    Phase ph;
    ph.a.u[0] = U;
    add_dw(ph, B, g_material(W.g1, H1), H1, W.h0[0], H0, nullptr, 0, off(q_net->w[1]), off(q_net->b[1]), 0);
    if (!u8) add_dw(ph, B, g_material(W.g0, H0), H0, (const float *)obs, S, nullptr, 0, off(q_net->w[0]), off(q_net->b[0]), 0);
    add_dw(ph, B, g_material(W.dq, A), A, W.h1[0], H1, nullptr, 0, off(q_net->w[2]), off(q_net->b[2]), 0);
    if ((rc = launch(ph, st, fn))) return rc;
  }
  return TRL_OK;
}

}  // namespace

extern "C" int64_t trl_dqn_learn_workspace_bytes(const trl_mlp_desc *q_net, int64_t B) {
  if (!q_net || B <= 0) return 0;
  return dqn_ws_floats(q_net, B, false, 3) * (int64_t)sizeof(float);
}

extern "C" int trl_dqn_learn_fused(const trl_mlp_desc *q_net, const trl_mlp_desc *target_net, const float *obs,
                                   const float *action, const float *reward, const float *next_obs, const float *done,
                                   int64_t B64, double gamma, int32_t double_dqn, int32_t kind, double huber_delta,
                                   float *loss_out, const trl_adam_args *opt, void *ws, void *stream) {
  return dqn_learn_impl("trl_dqn_learn_fused", false, q_net, target_net, obs, action, reward, next_obs, done, B64, gamma,
                        double_dqn, kind, huber_delta, loss_out, opt, ws, stream);
}

extern "C" int64_t trl_dqn_learn_u8_workspace_bytes(const trl_mlp_desc *q_net, int64_t B) {
  if (!q_net || B <= 0 || q_net->depth != 3) return 0;
  return dqn_ws_floats(q_net, B, true, 3) * (int64_t)sizeof(float);
}

extern "C" int trl_dqn_learn_fused_u8(const trl_mlp_desc *q_net, const trl_mlp_desc *target_net, const uint8_t *obs,
                                      const float *action, const float *reward, const uint8_t *next_obs,
                                      const float *done, int64_t B64, double gamma, int32_t double_dqn, int32_t kind,
                                      double huber_delta, float *loss_out, const trl_adam_args *opt, void *ws,
                                      void *stream) {
  return dqn_learn_impl("trl_dqn_learn_fused_u8", true, q_net, target_net, obs, action, reward, next_obs, done, B64,
                        gamma, double_dqn, kind, huber_delta, loss_out, opt, ws, stream);
}

// ---- the same step with the frames read in place: the gather of examples/dqn.py:40-41 becomes the first layer's operand
// fetch (no [B, 28224] uint8 copies of obs and next_obs), a small prologue launch resolves the sampled rows' records and
// gathers the 12 B action / reward / done tail.  Workspace = the u8 step's + B x (4 + 12) bytes.
namespace {
int64_t ring_tail_floats(int64_t B) { return (4 * B + 63) / 64 * 64; }      // rec (as int32), action, reward, done
}  // namespace

extern "C" int64_t trl_dqn_learn_u8_ring_workspace_bytes(const trl_mlp_desc *q_net, int64_t B) {
  if (!q_net || B <= 0 || q_net->depth != 3) return 0;
  return (dqn_ws_floats(q_net, B, true, 3) + ring_tail_floats(B)) * (int64_t)sizeof(float);
}

extern "C" int trl_dqn_learn_fused_u8_ring(const trl_mlp_desc *q_net, const trl_mlp_desc *target_net, const void *ring,
                                           int64_t capacity, int64_t stride, const trl_ring_layout *layout, int64_t head,
                                           int64_t count, const int64_t *idx, int64_t B64, double gamma, int32_t double_dqn,
                                           int32_t kind, double huber_delta, float *loss_out, const trl_adam_args *opt,
                                           void *ws, void *stream) {
  const char *fn = "trl_dqn_learn_fused_u8_ring";
  TRL_REQUIRE(q_net && target_net && ring && layout && idx && loss_out && opt && ws, "%s: NULL argument", fn);
  TRL_REQUIRE(q_net->depth == 3, "%s: q_net must have two hidden layers", fn);
  // the record must be a Transition (torchrl/utils/storage.py:51-57): obs, action, reward, next_obs, done, with uint8
  // frames of the first layer's fan-in and one float32 each for the other three
  TRL_REQUIRE(layout->ncols == 5, "%s: the ring must hold the five columns of a Transition, got %d", fn, layout->ncols);
  for (int c : {0, 3})
    TRL_REQUIRE(layout->col_dtype[c] == TRL_U8 && layout->col_bytes[c] == q_net->dims[0] && layout->col_off[c] % 16 == 0,
                "%s: column %d must be %d uint8 values at a 16 B-aligned offset", fn, c, q_net->dims[0]);
  for (int c : {1, 2, 4})
    TRL_REQUIRE(layout->col_dtype[c] == TRL_F32 && layout->col_bytes[c] == 4, "%s: column %d must be one float32", fn, c);
  TRL_REQUIRE(stride % 16 == 0 && ((uintptr_t)ring & 15) == 0, "%s: ring base and record stride must be multiples of 16 B", fn);
  TRL_REQUIRE(B64 >= 32 && B64 % 32 == 0 && B64 <= MAX_BATCH, "%s: B = %lld must be a multiple of 32 in [32, %d]", fn,
              (long long)B64, MAX_BATCH);
  TRL_REQUIRE(aligned16(ws), "%s: workspace must be 16 B aligned", fn);
  float *tail = (float *)ws + dqn_ws_floats(q_net, B64, true, 3);
  int32_t *rec = (int32_t *)tail;
  float *action = tail + B64, *reward = tail + 2 * B64, *done = tail + 3 * B64;
  const int32_t offs[3] = {layout->col_off[1], layout->col_off[2], layout->col_off[4]};
  int rc;
  if ((rc = trl_ring_rows_launch(fn, ring, capacity, stride, head, count, idx, B64, offs, rec, action, reward, done,
                                 (hipStream_t)stream)))
    return rc;
  return dqn_learn_impl(fn, true, q_net, target_net, (const char *)ring + layout->col_off[0], action, reward,
                        (const char *)ring + layout->col_off[3], done, B64, gamma, double_dqn, kind, huber_delta, loss_out,
                        opt, ws, stream, rec, stride);
}
