// First layer of a Q-network on uint8 frames, read as uint8.
//
// The reference stores frames as float32 (`torch.Tensor(b)`, torchrl/utils/storage.py:65-67) and
// feeds them to `QNet.net[0]` (torchrl/contrib/models.py:12-18, a 28,224 -> 128 Linear for the
// 84x84x4 frames of BASELINE config 4); DQNController.learn runs that layer three times per step
// (dqn_controller.py:64-72: q_net(obs), q_net(next_obs), target_q_net(next_obs)) and its weight
// gradient once.  Here the replay ring keeps the frames uint8 and this layer takes them as they
// are: uint8 -> float32 is exact, so the values are the reference's, and the batch is never
// materialised as float32 (4x the bytes through the gather, the graph's input copy and the GEMM's
// operand fetch).
//
//   forward   y[B, N] = act(x_u8[B, K] @ W[N, K]^T + b)      split over K, partial tiles summed in a
//                                                            fixed order by a second launch
//   wgrad     dW[N, K] = g[B, N]^T @ x_u8[B, K]              one pass, each output written once
//
// Both run on v_mfma_f32_32x32x16_bf16 with operands that lose nothing: a uint8 IS a bf16, and the f32 operand
// (the weight; in wgrad the gradient) goes in as its three exact bf16 parts, so every partial product is exact and the
// sums are taken in the MFMA's f32 accumulators (see x_operand / w_operands below; until the end of round 2 this was
// v_mfma_f32_32x32x2_f32, eight 64-cycle instructions per 16 k where three 32-cycle ones do now).  The MFMA sums
// over its 16 k slots, and which k sits in which slot is free as long as A and B agree, so a lane walks 16
// CONSECUTIVE k of its row (one 16 B load = 16 uint8 of x, 4 x 16 B of W) as two groups of 8.
#include <stdlib.h>
#include <string.h>

#include "trl_common.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

typedef __bf16 bf8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float ub(unsigned int w, int j) { return (float)((w >> (8 * j)) & 255u); }

// ---- operands of v_mfma_f32_32x32x16_bf16 that lose nothing (the forward kernel).
// A uint8 is exactly a bf16 (8 significant bits), and an f32 is exactly the sum of three bf16 (hi = its top 16 bits,
// mid = the top 16 bits of the exact remainder, lo = what is then left: 8 + 8 + 8 significant bits).  So
// x_u8 * w = x * hi + x * mid + x * lo with every partial product exact (8 x 8 bits) and the sums taken in the MFMA's
// f32 accumulators: three 32-cycle instructions over 16 k instead of eight 64-cycle f32 ones, the same values up to the
// order of the f32 additions.
// the top halves of two floats as one dword of two bf16: low 16 bits from a, high 16 bits from b
__device__ __forceinline__ unsigned int top2(float a, float b) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float top16(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }
// 8 consecutive uint8 (two dwords) as one operand
__device__ __forceinline__ u4v x_operand(unsigned int w0, unsigned int w1) {
  u4v r;
  r[0] = top2(ub(w0, 0), ub(w0, 1));
  r[1] = top2(ub(w0, 2), ub(w0, 3));
  r[2] = top2(ub(w1, 0), ub(w1, 1));
  r[3] = top2(ub(w1, 2), ub(w1, 3));
  return r;
}
// 8 consecutive f32 as three operands: part[0] + part[1] + part[2] = w, exactly
__device__ __forceinline__ void w_operands(const f4v &a, const f4v &b, u4v (&part)[3]) {
  typedef float f2v __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const f2v w = d < 2 ? f2v{a[2 * d], a[2 * d + 1]} : f2v{b[2 * d - 4], b[2 * d - 3]};
    part[0][d] = top2(w[0], w[1]);
    const f2v r = w - f2v{top16(w[0]), top16(w[1])};
    part[1][d] = top2(r[0], r[1]);
    const f2v l = r - f2v{top16(r[0]), top16(r[1])};
    part[2][d] = top2(l[0], l[1]);
  }
}

constexpr int FWD_MT = 128, FWD_NT = 128;   // workgroup tile: 4 waves of 128 rows x 32 columns each (W is 4 B per element and x 1 B:
                                            // the tall wave tile loads the fewest bytes per MFMA, 8 KB per block against 10 KB for 64 x 64)
constexpr int KB = 32;                      // k per block: 16 per lane half

constexpr int MAX_PROBLEMS = 3;   // the three first-layer passes of a Double-DQN step share one launch

struct FwdArgs {
  const uint8_t *x[MAX_PROBLEMS];
  // rows read in place out of a replay ring: row b of problem p starts at x[p] + rec[b] * stride bytes (x[p] = the
  // ring's base plus the frame column's offset in a record, rec[b] = the physical record); NULL = dense [B, K] operands
  const int32_t *rec;
  int64_t stride;
  const float *w[MAX_PROBLEMS];
  float *ws;       // [P][S][Bpad][N] partial tiles
  int B, N, K, Bpad, bps, kblocks, S;
};

constexpr int WI = 4, WT = 1;   // 32 x 32 tiles per wave: rows x columns
struct Operands {
  u4v xa[WI];      // 16 uint8 of each row
  f4v wb[WT][4];   // 16 floats of each weight row
};

__device__ __forceinline__ void fwd_load(Operands &o, const uint8_t *const (&xr)[WI], const float *const (&wr)[WT], int k) {
#pragma unroll
  for (int i = 0; i < WI; ++i) o.xa[i] = *(const u4v *)(xr[i] + k);
#pragma unroll
  for (int t = 0; t < WT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) o.wb[t][q] = *(const f4v *)(wr[t] + k + 4 * q);
}

__global__ __launch_bounds__(256) void k_lin_u8_fwd(FwdArgs a) {
  const int s = blockIdx.x;
  const int pz = blockIdx.z;
  const uint8_t *xbase = pz == 0 ? a.x[0] : (pz == 1 ? a.x[1] : a.x[2]);
  const float *wbase = pz == 0 ? a.w[0] : (pz == 1 ? a.w[1] : a.w[2]);
  const int ntn = a.N / FWD_NT;
  const int mt = blockIdx.y / ntn, nt = blockIdx.y - mt * ntn;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, kh = lane >> 5;
  const int r0 = mt * FWD_MT, n0 = nt * FWD_NT + wave * 32;
  const uint8_t *xr[WI];
  const float *wr[WT];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    int row = r0 + 32 * i + lr;
    row = row < a.B ? row : a.B - 1;            // padding rows read a valid row; their results are never used
    xr[i] = xbase + (a.rec ? (size_t)a.rec[row] * (size_t)a.stride : (size_t)row * a.K) + 16 * kh;
  }
#pragma unroll
  for (int t = 0; t < WT; ++t) wr[t] = wbase + (size_t)(n0 + 32 * t + lr) * a.K + 16 * kh;   // N % 128 == 0: always valid

  f16v acc[WI][WT];
#pragma unroll
  for (int i = 0; i < WI; ++i)
#pragma unroll
    for (int t = 0; t < WT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  const int kb0 = s * a.bps;
  const int kb1 = min(a.kblocks, kb0 + a.bps);
  // Three operand sets in rotation: the loads of k block kb + 2 are issued before the 24 MFMAs of block kb, so a
  // block's operands have two blocks' worth of work to arrive (a fourth set, three blocks ahead, measured no gain: the
  // L1 lookup rate, 64 per load instruction of this layout, is what the loop waits for).
  // A lane's 16 consecutive k are two MFMA k groups of 8 (slot (lane half, j) <-> k = 16 * half + 8 m + j on both
  // operands); per group and tile: x * lo, x * mid, x * hi, smallest first.  24 MFMAs of 32 cycles per block of 32 k
  // where the exact-f32 instruction took 64 of 64 cycles.
  auto mma = [&](const Operands &o) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      u4v xo[WI], wp[WT][3];
#pragma unroll
      for (int i = 0; i < WI; ++i) xo[i] = x_operand(o.xa[i][2 * m], o.xa[i][2 * m + 1]);
#pragma unroll
      for (int t = 0; t < WT; ++t) w_operands(o.wb[t][2 * m], o.wb[t][2 * m + 1], wp[t]);
#pragma unroll
      for (int p = 2; p >= 0; --p)
#pragma unroll
        for (int i = 0; i < WI; ++i)
#pragma unroll
          for (int t = 0; t < WT; ++t)
            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, xo[i]),
                                                                __builtin_bit_cast(bf8v, wp[t][p]), acc[i][t], 0, 0, 0);
    }
  };
  // The MFMAs are unconditional (a branch around them makes the compiler copy all 64 accumulators at the join): a set
  // beyond the split's last block holds zeros, which add nothing.  The host makes a split a multiple of three blocks,
  // so only the launch's last split runs such a block.
  auto load = [&](Operands &o, int kb) {
    if (kb < kb1) {
      fwd_load(o, xr, wr, kb * KB);
    } else {
#pragma unroll
      for (int i = 0; i < WI; ++i) o.xa[i] = u4v{0u, 0u, 0u, 0u};
#pragma unroll
      for (int t = 0; t < WT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) o.wb[t][q] = f4v{0.f, 0.f, 0.f, 0.f};
    }
  };
  Operands o0, o1, o2;
  load(o0, kb0);
  load(o1, kb0 + 1);
  for (int kb = kb0; kb < kb1; kb += 3) {
    load(o2, kb + 2);
    mma(o0);
    load(o0, kb + 3);
    mma(o1);
    load(o1, kb + 4);
    mma(o2);
  }
  // accumulator layout: lane l, register r holds (row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31)
  float *ws = a.ws + ((size_t)pz * a.S + s) * a.Bpad * a.N;
#pragma unroll
  for (int i = 0; i < WI; ++i)
#pragma unroll
    for (int t = 0; t < WT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh;   // < Bpad by construction
        ws[(size_t)row * a.N + n0 + 32 * t + lr] = acc[i][t][r];
      }
}

// y[b, n] = act(sum_s ws[s][b][n] + bias[n]).  A thread sums one of 8 contiguous slices of the S partial
// tiles for four consecutive n (16 B loads, all of a slice in flight at once); the 8 slice sums are then
// added in slice order through LDS, so the result does not depend on timing.
struct ReduceArgs {
  const float *ws;
  const float *bias[MAX_PROBLEMS];
  float *y[MAX_PROBLEMS];
  int B, N, Bpad, S, act;
};

constexpr int RED_SLICES = 8, RED_MAX_PER_SLICE = 32;

__global__ __launch_bounds__(256) void k_lin_u8_reduce(ReduceArgs a) {
  __shared__ f4v part[RED_SLICES][32];
  const int pz = blockIdx.y;
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t quad = (int64_t)blockIdx.x * 32 + o;          // quad of outputs: (b, n .. n + 3)
  const int64_t nquads = (int64_t)a.B * a.N / 4;
  const bool live = quad < nquads;
  const int64_t e = live ? quad * 4 : 0;
  const int b = (int)(e / a.N), n = (int)(e - (int64_t)b * a.N);
  const size_t plane = (size_t)a.Bpad * a.N;
  const int per = (a.S + RED_SLICES - 1) / RED_SLICES;
  const int s0 = sl * per, s1 = min(a.S, s0 + per);
  const float *p = a.ws + (size_t)pz * a.S * plane + (size_t)b * a.N + n;
  f4v acc = {0.f, 0.f, 0.f, 0.f};
  f4v v[RED_MAX_PER_SLICE];
#pragma unroll
  for (int u = 0; u < RED_MAX_PER_SLICE; ++u)
    if (s0 + u < s1) v[u] = *(const f4v *)(p + (size_t)(s0 + u) * plane);
#pragma unroll
  for (int u = 0; u < RED_MAX_PER_SLICE; ++u)
    if (s0 + u < s1) acc += v[u];
  part[sl][o] = acc;
  __syncthreads();
  if (sl == 0 && live) {
    f4v t = part[0][o];
#pragma unroll
    for (int g = 1; g < RED_SLICES; ++g) t += part[g][o];
    const float *bias = pz == 0 ? a.bias[0] : (pz == 1 ? a.bias[1] : a.bias[2]);
    float *y = pz == 0 ? a.y[0] : (pz == 1 ? a.y[1] : a.y[2]);
    t += *(const f4v *)(bias + n);
    if (a.act == 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) t[c] = fmaxf(t[c], 0.f);
    }
    *(f4v *)(y + e) = t;
  }
}

// dW[n, k] = sum_b g[b, n] * x[b, k].  A workgroup owns 128 consecutive k; wave w owns output rows
// [n0 + 32 w, +32).  Lane (i = lane & 31, h = lane >> 5) reads ONE dword of x per batch row = the four
// consecutive k  kbase + 4 i .. + 3, which feed four MFMA tiles (tile t = the columns kbase + 4 i + t), so
// the four accumulators of a lane are four consecutive k of one output row: a 16 B store.
struct WgradArgs {
  const float *g;
  const uint8_t *x;
  const int32_t *rec;        // as FwdArgs::rec (x = ring base + column offset, row b at x + rec[b] * stride), or NULL
  int64_t stride;
  float *dw;       // [N, K] gradient (always written)
  float *db;       // [N] column sums of g, or NULL
  // update != 0: torch.optim.Adam's step on the weight (and bias) right where its gradient is produced
  float *pw, *mw, *vw, *pb, *mb, *vb;
  TrlAdamScalars s;
  int update;
  int B, N, K;
};

__global__ __launch_bounds__(256) void k_lin_u8_wgrad(WgradArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * 128 + wave * 32;
  const int kcol = blockIdx.x * 128 + 4 * i;
  const int kc = kcol < a.K ? kcol : a.K - 4;          // K % 4 == 0: a valid dword either way
  const float *gp = a.g + (size_t)(16 * h) * a.N + n0 + i;
  const int nblk = a.B / 32;
  if (blockIdx.x >= (unsigned)((a.K + 127) / 128)) {
    // one extra workgroup per 128 output rows: the bias gradient (column sums of g), beside the MFMA
    // workgroups instead of behind one of them.  The two lane halves take alternate 16-row groups of
    // the batch, added in a fixed order.
    float sum = 0.f;
    for (int bb = 0; bb < nblk; ++bb) {
      const float *g2 = gp + (size_t)bb * 32 * a.N;
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = g2[(size_t)j * a.N];
#pragma unroll
      for (int j = 0; j < 16; ++j) sum += v[j];
    }
    const float other = __shfl_xor(sum, 32, 64);
    if (h == 0) {
      const int n = n0 + i;
      a.db[n] = sum + other;
      if (a.update) trl_adam_elem(a.pb + n, a.db + n, a.mb + n, a.vb + n, a.s);
    }
    return;
  }
  // row r of x: dense at r * K, or the ring record rec[r].  The record numbers of a block's 16 rows are fetched a block
  // before the x loads that need them (the loads of block bb + 1 are issued under the MFMAs of block bb).
  const bool ring = a.rec != nullptr;
  auto xrow = [&](int r, int recno) -> const uint8_t * {
    return a.x + (ring ? (size_t)recno * (size_t)a.stride : (size_t)r * a.K) + kc;
  };
  f16v acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float ga[16], gn[16];
  unsigned int xa[16], xn[16];
  int rn[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int r = 16 * h + j;
    ga[j] = gp[(size_t)j * a.N];
    xa[j] = *(const unsigned int *)xrow(r, ring ? a.rec[r] : 0);
    rn[j] = (ring && nblk > 1) ? a.rec[32 + r] : 0;
  }
  for (int bb = 0; bb < nblk; ++bb) {
    const bool more = bb + 1 < nblk;
    if (more) {
      const float *g2 = gp + (size_t)(bb + 1) * 32 * a.N;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int r = (bb + 1) * 32 + 16 * h + j;
        gn[j] = g2[(size_t)j * a.N];
        xn[j] = *(const unsigned int *)xrow(r, rn[j]);
      }
      if (ring && bb + 2 < nblk) {
#pragma unroll
        for (int j = 0; j < 16; ++j) rn[j] = a.rec[(bb + 2) * 32 + 16 * h + j];
      }
    }
    // A lane's 16 batch rows are two MFMA k groups of 8 (slot (lane half, jj) <-> row 16 h + 8 m + jj on both operands);
    // g as its three exact bf16 parts, byte t of the 8 x dwords as the operand of tile t: 24 MFMAs of 32 cycles per 32
    // rows where the exact-f32 instruction took 64 of 64 cycles, the same values up to the order of the f32 additions.
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      u4v gq[3], xo[4];
      w_operands(f4v{ga[8 * m], ga[8 * m + 1], ga[8 * m + 2], ga[8 * m + 3]},
                 f4v{ga[8 * m + 4], ga[8 * m + 5], ga[8 * m + 6], ga[8 * m + 7]}, gq);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int d = 0; d < 4; ++d) xo[t][d] = top2(ub(xa[8 * m + 2 * d], t), ub(xa[8 * m + 2 * d + 1], t));
#pragma unroll
      for (int p = 2; p >= 0; --p)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8v, gq[p]), __builtin_bit_cast(bf8v, xo[t]),
                                                           acc[t], 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        ga[j] = gn[j];
        xa[j] = xn[j];
      }
    }
  }
  if (kcol < a.K) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const f4v v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
      const size_t off = (size_t)n * a.K + kcol;
      *(f4v *)(a.dw + off) = v;
      if (a.update) {
#pragma unroll
        for (int c = 0; c < 4; ++c) trl_adam_elem(a.pw + off + c, a.dw + off + c, a.mw + off + c, a.vw + off + c, a.s);
      }
    }
  }
}

struct FwdPlan {
  int mtiles, Bpad, kblocks, bps, S;
};

bool fwd_plan(int64_t B, int32_t N, int32_t K, int P, FwdPlan *p) {
  if (B <= 0 || N <= 0 || K <= 0 || N % FWD_NT || K % KB || B > (1 << 20) || P < 1 || P > MAX_PROBLEMS) return false;
  p->mtiles = (int)((B + FWD_MT - 1) / FWD_MT);
  p->Bpad = p->mtiles * FWD_MT;
  p->kblocks = K / KB;
  const int tiles = p->mtiles * (N / FWD_NT) * P;
  int total_wgs = 256;                                 // one workgroup per CU over the whole launch (tools/u8_fwd_sweep.py: 85 us per step's chain at 256, 92-100 at 192 / 384-1536)
#ifdef TRL_TUNING                                      // diagnostic build: sweep of the split count (tools/u8_fwd_sweep.py)
  if (const char *t = getenv("TRL_U8_FWD_WGS"))
    if (atoi(t) > 0) total_wgs = atoi(t);
#endif
  int target = (total_wgs + tiles - 1) / tiles;
  if (target < 1) target = 1;
  int bps = (p->kblocks + target - 1) / target;
  const int min_bps = (p->kblocks + RED_SLICES * RED_MAX_PER_SLICE - 1) / (RED_SLICES * RED_MAX_PER_SLICE);
  p->bps = bps > min_bps ? bps : min_bps;              // the reduce launch holds at most 256 partial tiles per output
  p->bps = (p->bps + 2) / 3 * 3;                       // whole rounds of the kernel's three operand sets
  p->S = (p->kblocks + p->bps - 1) / p->bps;
  return true;
}

}  // namespace

// Internal launchers (declared in trl_common.h): P first-layer problems of one shape in two launches, and
// the weight gradient with an optional in-place Adam step.  Arguments are validated by the callers.
int64_t trl_u8_fwd_ws_floats(int64_t B, int32_t N, int32_t K, int P) {
  FwdPlan p;
  if (!fwd_plan(B, N, K, P, &p)) return 0;
  return (int64_t)P * p.S * p.Bpad * N;
}

int trl_u8_fwd_launch(const char *fn, int P, const uint8_t *const *x, const float *const *w, const float *const *b,
                      float *const *y, int64_t B, int32_t K, int32_t N, int32_t act, float *ws, hipStream_t st,
                      const int32_t *rec, int64_t stride) {
  FwdPlan p;
  TRL_REQUIRE(fwd_plan(B, N, K, P, &p), "%s: uint8 first layer needs B in [1, 2^20], N %% 128 == 0, K %% 32 == 0 "
              "(B=%lld N=%d K=%d)", fn, (long long)B, N, K);
  FwdArgs a;
  ReduceArgs r;
  for (int i = 0; i < MAX_PROBLEMS; ++i) {
    const int j = i < P ? i : 0;
    TRL_REQUIRE(x[j] && w[j] && b[j] && y[j], "%s: null pointer", fn);
    TRL_REQUIRE(((uintptr_t)x[j] % 16 == 0) && ((uintptr_t)w[j] % 16 == 0) && ((uintptr_t)b[j] % 16 == 0) &&
                ((uintptr_t)y[j] % 16 == 0), "%s: x, w, b and y must be 16 B aligned", fn);
    a.x[i] = x[j]; a.w[i] = w[j];
    r.bias[i] = b[j]; r.y[i] = y[j];
  }
  TRL_REQUIRE(ws && (uintptr_t)ws % 16 == 0, "%s: workspace must be 16 B aligned", fn);
  TRL_REQUIRE(!rec || stride % 16 == 0, "%s: ring records must be a multiple of 16 B apart", fn);
  a.ws = ws;
  a.rec = rec; a.stride = stride;
  a.B = (int)B; a.N = N; a.K = K; a.Bpad = p.Bpad; a.bps = p.bps; a.kblocks = p.kblocks; a.S = p.S;
  hipLaunchKernelGGL(k_lin_u8_fwd, dim3(p.S, p.mtiles * (N / FWD_NT), P), dim3(256), 0, st, a);
  TRL_CHECK_LAUNCH(fn);
  r.ws = ws; r.B = (int)B; r.N = N; r.Bpad = p.Bpad; r.S = p.S; r.act = act;
  hipLaunchKernelGGL(k_lin_u8_reduce, dim3((unsigned)trl_cdiv(B * N / 4, 32), P), dim3(256), 0, st, r);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

int trl_u8_wgrad_launch(const char *fn, const float *g, const uint8_t *x, int64_t B, int32_t K, int32_t N, float *dw,
                        float *db, const TrlU8Update *u, hipStream_t st, const int32_t *rec, int64_t stride) {
  TRL_REQUIRE(g && x && dw, "%s: null pointer", fn);
  TRL_REQUIRE(B >= 32 && B % 32 == 0 && B <= (1 << 20) && N > 0 && N % 128 == 0 && K >= 4 && K % 4 == 0,
              "%s: uint8 weight gradient needs B %% 32 == 0, N %% 128 == 0, K %% 4 == 0 (B=%lld N=%d K=%d)", fn,
              (long long)B, N, K);
  TRL_REQUIRE(((uintptr_t)x % 4 == 0) && ((uintptr_t)dw % 16 == 0) && ((uintptr_t)g % 4 == 0),
              "%s: x must be 4 B aligned, dw 16 B aligned", fn);
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.g = g; a.x = x; a.dw = dw; a.db = db; a.B = (int)B; a.N = N; a.K = K;
  TRL_REQUIRE(!rec || stride % 4 == 0, "%s: ring records must be a multiple of 4 B apart", fn);
  a.rec = rec; a.stride = stride;
  if (u) {
    TRL_REQUIRE(db && u->pw && u->mw && u->vw && u->pb && u->mb && u->vb, "%s: the in-launch update needs every buffer", fn);
    a.update = 1;
    a.pw = u->pw; a.mw = u->mw; a.vw = u->vw; a.pb = u->pb; a.mb = u->mb; a.vb = u->vb; a.s = u->s;
  }
  hipLaunchKernelGGL(k_lin_u8_wgrad, dim3((unsigned)trl_cdiv(K, 128) + (db ? 1 : 0), N / 128), dim3(256), 0, st, a);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

extern "C" int64_t trl_linear_u8_fwd_workspace_bytes(int64_t B, int32_t N, int32_t K) {
  return trl_u8_fwd_ws_floats(B, N, K, 1) * (int64_t)sizeof(float);
}

extern "C" int trl_linear_u8_fwd(const uint8_t *x, const float *w, const float *b, int64_t B, int32_t K, int32_t N,
                                 int32_t act, float *y, void *ws, void *stream) {
  const char *fn = "trl_linear_u8_fwd";
  TRL_REQUIRE(x && w && b && y && ws, "%s: null pointer", fn);
  TRL_REQUIRE(act == 0 || act == 1, "%s: act must be 0 (none) or 1 (relu)", fn);
  return trl_u8_fwd_launch(fn, 1, &x, &w, &b, &y, B, K, N, act, (float *)ws, (hipStream_t)stream, nullptr, 0);
}

extern "C" int trl_linear_u8_wgrad(const float *g, const uint8_t *x, int64_t B, int32_t K, int32_t N, float *dw,
                                   void *stream) {
  return trl_u8_wgrad_launch("trl_linear_u8_wgrad", g, x, B, K, N, dw, nullptr, nullptr, (hipStream_t)stream, nullptr, 0);
}
