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
// Both run on v_mfma_f32_32x32x2_f32 (exact f32 products and accumulation); the MFMA sums over its
// two k slots, and which k sits in which slot is free as long as A and B agree, so a lane walks 16
// CONSECUTIVE k of its row (one 16 B load = 16 uint8 of x, 4 x 16 B of W) instead of stride-2 ones.
#include "trl_common.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ub(unsigned int w, int j) { return (float)((w >> (8 * j)) & 255u); }

constexpr int FWD_MT = 128, FWD_NT = 128;   // workgroup tile (4 waves as 2 x 2 of 64 x 64)
constexpr int KB = 32;                      // k per block: 16 per lane half

struct FwdArgs {
  const uint8_t *x;
  const float *w;
  float *ws;       // [S][Bpad][N] partial tiles
  int B, N, K, Bpad, bps, kblocks;
};

struct Operands {
  u4v xa[2];       // 16 uint8 of two rows
  f4v wb[2][4];    // 16 floats of two weight rows
};

__device__ __forceinline__ void fwd_load(Operands &o, const uint8_t *const (&xr)[2], const float *const (&wr)[2], int k) {
#pragma unroll
  for (int i = 0; i < 2; ++i) o.xa[i] = *(const u4v *)(xr[i] + k);
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) o.wb[t][q] = *(const f4v *)(wr[t] + k + 4 * q);
}

__global__ __launch_bounds__(256) void k_lin_u8_fwd(FwdArgs a) {
  const int s = blockIdx.x;
  const int ntn = a.N / FWD_NT;
  const int mt = blockIdx.y / ntn, nt = blockIdx.y - mt * ntn;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, kh = lane >> 5;
  const int r0 = mt * FWD_MT + (wave >> 1) * 64, n0 = nt * FWD_NT + (wave & 1) * 64;
  const uint8_t *xr[2];
  const float *wr[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int row = r0 + 32 * i + lr;
    row = row < a.B ? row : a.B - 1;            // padding rows read a valid row; their results are never used
    xr[i] = a.x + (size_t)row * a.K + 16 * kh;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) wr[t] = a.w + (size_t)(n0 + 32 * t + lr) * a.K + 16 * kh;   // N % 128 == 0: always valid

  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  const int kb0 = s * a.bps;
  const int kb1 = min(a.kblocks, kb0 + a.bps);
  Operands cur, nxt;
  if (kb0 < kb1) fwd_load(cur, xr, wr, kb0 * KB);
  for (int kb = kb0; kb < kb1; ++kb) {
    const bool more = kb + 1 < kb1;
    if (more) fwd_load(nxt, xr, wr, (kb + 1) * KB);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a0 = ub(cur.xa[0][q], j), a1 = ub(cur.xa[1][q], j);
        const float b0 = cur.wb[0][q][j], b1 = cur.wb[1][q][j];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    if (more) cur = nxt;
  }
  // accumulator layout: lane l, register r holds (row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31)
  float *ws = a.ws + (size_t)s * a.Bpad * a.N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = r0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh;   // < Bpad by construction
        ws[(size_t)row * a.N + n0 + 32 * t + lr] = acc[i][t][r];
      }
}

// y[b, n] = act(sum_s ws[s][b][n] + bias[n]), s in ascending order
__global__ __launch_bounds__(256) void k_lin_u8_reduce(const float *__restrict__ ws, const float *__restrict__ bias,
                                                       float *__restrict__ y, int B, int N, int Bpad, int S, int act) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)B * N) return;
  const int b = (int)(e / N), n = (int)(e - (int64_t)b * N);
  const size_t plane = (size_t)Bpad * N;
  const float *p = ws + (size_t)b * N + n;
  float acc = 0.f;
  int s = 0;
  for (; s + 8 <= S; s += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(s + u) * plane];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; s < S; ++s) acc += p[(size_t)s * plane];
  acc += bias[n];
  if (act == 1) acc = fmaxf(acc, 0.f);
  y[e] = acc;
}

// dW[n, k] = sum_b g[b, n] * x[b, k].  A workgroup owns 128 consecutive k; wave w owns output rows
// [n0 + 32 w, +32).  Lane (i = lane & 31, h = lane >> 5) reads ONE dword of x per batch row = the four
// consecutive k  kbase + 4 i .. + 3, which feed four MFMA tiles (tile t = the columns kbase + 4 i + t), so
// the four accumulators of a lane are four consecutive k of one output row: a 16 B store.
struct WgradArgs {
  const float *g;
  const uint8_t *x;
  float *dw;
  int B, N, K;
};

__global__ __launch_bounds__(256) void k_lin_u8_wgrad(WgradArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * 128 + wave * 32;
  const int kcol = blockIdx.x * 128 + 4 * i;
  const int kc = kcol < a.K ? kcol : a.K - 4;          // K % 4 == 0: a valid dword either way
  const float *gp = a.g + (size_t)(16 * h) * a.N + n0 + i;
  const uint8_t *xp = a.x + (size_t)(16 * h) * a.K + kc;
  f16v acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float ga[16], gn[16];
  unsigned int xa[16], xn[16];
  const int nblk = a.B / 32;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    ga[j] = gp[(size_t)j * a.N];
    xa[j] = *(const unsigned int *)(xp + (size_t)j * a.K);
  }
  for (int bb = 0; bb < nblk; ++bb) {
    const bool more = bb + 1 < nblk;
    if (more) {
      const float *g2 = gp + (size_t)(bb + 1) * 32 * a.N;
      const uint8_t *x2 = xp + (size_t)(bb + 1) * 32 * a.K;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        gn[j] = g2[(size_t)j * a.N];
        xn[j] = *(const unsigned int *)(x2 + (size_t)j * a.K);
      }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[j], ub(xa[j], t), acc[t], 0, 0, 0);
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
      *(f4v *)(a.dw + (size_t)n * a.K + kcol) = v;
    }
  }
}

struct FwdPlan {
  int mtiles, Bpad, kblocks, bps, S;
};

bool fwd_plan(int64_t B, int32_t N, int32_t K, FwdPlan *p) {
  if (B <= 0 || N <= 0 || K <= 0 || N % FWD_NT || K % KB || B > (1 << 20)) return false;
  p->mtiles = (int)((B + FWD_MT - 1) / FWD_MT);
  p->Bpad = p->mtiles * FWD_MT;
  p->kblocks = K / KB;
  const int tiles = p->mtiles * (N / FWD_NT);
  int target = (768 + tiles - 1) / tiles;              // ~3 workgroups per CU over the whole launch
  if (target < 1) target = 1;
  p->bps = (p->kblocks + target - 1) / target;
  p->S = (p->kblocks + p->bps - 1) / p->bps;
  return true;
}

}  // namespace

extern "C" int64_t trl_linear_u8_fwd_workspace_bytes(int64_t B, int32_t N, int32_t K) {
  FwdPlan p;
  if (!fwd_plan(B, N, K, &p)) return 0;
  return (int64_t)p.S * p.Bpad * N * (int64_t)sizeof(float);
}

extern "C" int trl_linear_u8_fwd(const uint8_t *x, const float *w, const float *b, int64_t B, int32_t K, int32_t N,
                                 int32_t act, float *y, void *ws, void *stream) {
  const char *fn = "trl_linear_u8_fwd";
  TRL_REQUIRE(x && w && b && y && ws, "%s: null pointer", fn);
  TRL_REQUIRE(act == 0 || act == 1, "%s: act must be 0 (none) or 1 (relu)", fn);
  FwdPlan p;
  TRL_REQUIRE(fwd_plan(B, N, K, &p), "%s: need B in [1, 2^20], N %% 128 == 0, K %% 32 == 0 (B=%lld N=%d K=%d)", fn,
              (long long)B, N, K);
  TRL_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)w % 16 == 0) && ((uintptr_t)ws % 16 == 0),
              "%s: x, w and ws must be 16 B aligned", fn);
  hipStream_t st = (hipStream_t)stream;
  FwdArgs a;
  a.x = x; a.w = w; a.ws = (float *)ws;
  a.B = (int)B; a.N = N; a.K = K; a.Bpad = p.Bpad; a.bps = p.bps; a.kblocks = p.kblocks;
  hipLaunchKernelGGL(k_lin_u8_fwd, dim3(p.S, p.mtiles * (N / FWD_NT)), dim3(256), 0, st, a);
  TRL_CHECK_LAUNCH(fn);
  hipLaunchKernelGGL(k_lin_u8_reduce, dim3((unsigned)trl_cdiv(B * N, 256)), dim3(256), 0, st, (const float *)ws, b, y,
                     (int)B, N, p.Bpad, p.S, act);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

extern "C" int trl_linear_u8_wgrad(const float *g, const uint8_t *x, int64_t B, int32_t K, int32_t N, float *dw,
                                   void *stream) {
  const char *fn = "trl_linear_u8_wgrad";
  TRL_REQUIRE(g && x && dw, "%s: null pointer", fn);
  TRL_REQUIRE(B >= 32 && B % 32 == 0 && B <= (1 << 20) && N > 0 && N % 128 == 0 && K >= 4 && K % 4 == 0,
              "%s: need B %% 32 == 0, N %% 128 == 0, K %% 4 == 0 (B=%lld N=%d K=%d)", fn, (long long)B, N, K);
  TRL_REQUIRE(((uintptr_t)x % 4 == 0) && ((uintptr_t)dw % 16 == 0) && ((uintptr_t)g % 4 == 0),
              "%s: x must be 4 B aligned, dw 16 B aligned", fn);
  WgradArgs a;
  a.g = g; a.x = x; a.dw = dw; a.B = (int)B; a.N = N; a.K = K;
  hipLaunchKernelGGL(k_lin_u8_wgrad, dim3((unsigned)trl_cdiv(K, 128), N / 128), dim3(256), 0, (hipStream_t)stream, a);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}
