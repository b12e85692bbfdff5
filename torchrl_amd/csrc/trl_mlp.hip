// Linear(+ReLU) forward and backward for the small policy / value MLPs, on f32 MFMA.
//
// Replaces, layer by layer, what torch runs for `nn.Linear` + `nn.ReLU` inside the networks of
// torchrl/contrib/models.py (QNet :14-20, DDPG nets :37-44/:66-72, A2CNet :92-103,
// ActorCriticNet :120-130) when the controllers learn on a large device batch.  The shapes are
// "skinny": B rows in the tens of thousands, fan-in / fan-out of 1..256.  BLAS libraries pick
// poor kernels there (the weight gradient reduces over B with no split; measured 187 us for
// 64x64 at B = 65536), and torch needs ~6 launches per layer for the backward (ReLU mask, dX,
// dW, bias sum, accumulation).  Here one launch does a layer's forward (bias + ReLU fused) and
// one its whole backward (ReLU mask, dX, dW and db partials), each a single pass over the
// activations, i.e. HBM-bound: fwd reads X writes Y; bwd reads dY, Y, X, writes dX.
//
// Arithmetic: `v_mfma_f32_32x32x2_f32`, exact f32 (a k-ordered fmaf chain per output, no
// reduced precision), so results agree with torch's fp32 GEMMs to summation-order effects.
// Tiles: a workgroup (4 waves) owns 128 rows; wave w owns rows [32w, 32w+32).  Operands are
// staged in LDS with an odd row stride so that the MFMA's one-f32-per-lane operand reads
// (lane l: A[l&31][l>>5], B[l>>5][l&31]) are bank-conflict free.
// The weight-gradient partials of each row slab go to a workspace and are summed by a second
// launch in slab order: bitwise reproducible run to run (no float atomics).
#include <stdlib.h>

#include <type_traits>

#include "trl_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;  // rows per tile

// C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int cd_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }


// ---- global -> register -> padded-LDS staging of a [BM x W] tile of a row-major [rows, ld] matrix
// (columns [c0, c0+W) of rows [row0, row0+BM)); zero outside the matrix.  W is 32 or 64, so a
// thread holds ROWS * W / 256 floats (ROWS = BM for activations, the padded fan-out for weights).  VEC: ld % 4 == 0 and 16 B-aligned base, so a
// thread's 4 consecutive columns are one 16 B load.
template <int W, bool VEC, int ROWS = BM>
struct TileRegs {
  static constexpr int NV = ROWS * W / 256 / 4;   // float4 groups per thread
  float4 v[NV];
  __device__ __forceinline__ void load(const float *__restrict__ m, int64_t rows, int ld, int64_t row0, int c0) {
    constexpr int QPR = W / 4;                  // float4 groups per tile row
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int q = (int)threadIdx.x + i * 256;
      const int r = q / QPR, c = c0 + 4 * (q % QPR);
      const int64_t gr = row0 + r;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < rows) {
        const float *p = m + gr * ld + c;
        if (VEC) {
          if (c + 3 < ld) t = *(const float4 *)p;
          else if (c < ld) { t.x = p[0]; if (c + 1 < ld) t.y = p[1]; if (c + 2 < ld) t.z = p[2]; }
        } else {
          if (c < ld) t.x = p[0];
          if (c + 1 < ld) t.y = p[1];
          if (c + 2 < ld) t.z = p[2];
          if (c + 3 < ld) t.w = p[3];
        }
      }
      v[i] = t;
    }
  }
  // s: LDS tile with row stride LDS_LD floats
  template <int LDS_LD>
  __device__ __forceinline__ void store(float *s) const {
    constexpr int QPR = W / 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int q = (int)threadIdx.x + i * 256;
      const int r = q / QPR, c = 4 * (q % QPR);
      float *d = s + r * LDS_LD + c;
      d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
    }
  }
};

__device__ __forceinline__ float4 relu_mask(float4 g, float4 y) {
  g.x = y.x > 0.f ? g.x : 0.f; g.y = y.y > 0.f ? g.y : 0.f;
  g.z = y.z > 0.f ? g.z : 0.f; g.w = y.w > 0.f ? g.w : 0.f;
  return g;
}

// ------------------------------------------------------------------------------ forward
// y[B, N] = act(x[B, K] @ w[N, K]^T + b);  NB = ceil(N / 32); K walked in chunks of KC = 32 * KB
// with the next chunk's x tile in flight (registers) while the MFMAs run on the current one.
template <int NB, int KB, bool VEC>
__global__ __launch_bounds__(256) void k_linear_fwd(const float *__restrict__ x, const float *__restrict__ w,
                                                    const float *__restrict__ b, int64_t B, int K, int N, int act,
                                                    float *__restrict__ y) {
  constexpr int KC = 32 * KB, LD = KC + 1;
  __shared__ float sX[BM * LD];
  __shared__ float sW[NB * 32 * LD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * NB * 32;   // small batches split the fan-out over blockIdx.y to fill the chip
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;

  TileRegs<KC, VEC> rx;
  TileRegs<KC, VEC, NB * 32> rw;     // the weight chunk [fan-out, KC] takes the same route
  rx.load(x, B, K, row0, 0);
  rw.load(w, N, K, n0, 0);
  for (int k0 = 0; k0 < K; k0 += KC) {
    __syncthreads();                     // readers of the previous chunk are done
    rx.template store<LD>(sX);
    rw.template store<LD>(sW);
    __syncthreads();
    if (k0 + KC < K) {
      rx.load(x, B, K, row0, k0 + KC);
      rw.load(w, N, K, n0, k0 + KC);
    }
    // fixed trip count (the tiles are zero-padded) so that the operand reads of the next
    // k-steps are issued ahead of the MFMAs that consume them
#pragma unroll 8
    for (int kk = 0; kk < KC; kk += 2) {
      const float a = sX[(wave * 32 + (lane & 31)) * LD + kk + (lane >> 5)];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const float bb = sW[(nb * 32 + (lane & 31)) * LD + kk + (lane >> 5)];
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[nb], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int col = n0 + nb * 32 + (lane & 31);
    const float bias = col < N ? b[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t gr = row0 + wave * 32 + cd_row(r, lane);
      if (gr < B && col < N) {
        float v = acc[nb][r] + bias;
        if (act == 1) v = v > 0.f ? v : 0.f;
        y[gr * N + col] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------ backward
// g = dy * (y > 0 if relu);  dx[B, K] = g @ w;  dw[N, K] = g^T @ x;  db[N] = sum_rows g.
// Grid: x = row slab, y = chunk of KCB = 32 * KB input columns.  A block walks its slab as a
// stream of steps (tile, g-chunk of GC = 32 * GCB columns); the operands of step s+1 (dy, y and,
// at a tile's first chunk, x) are loaded into registers while the MFMAs of step s run out of
// LDS.  Per tile it accumulates dx (per wave: its 32 rows), per slab dw (per wave: its rows'
// contribution to the whole [N, KCB] chunk, in registers across the tiles).  The four waves' dw
// partials are added through LDS at the end and written to ws_dw[slab]; k_linear_bwd_reduce sums
// the slabs.
template <int NB, int KB, bool VECN, bool VECK>
__global__ __launch_bounds__(256) void k_linear_bwd(const float *__restrict__ dy, const float *__restrict__ y,
                                                    const float *__restrict__ x, const float *__restrict__ w,
                                                    int64_t B, int K, int N, int act, int tiles_per_slab,
                                                    float *__restrict__ dx, float *__restrict__ ws_dw,
                                                    float *__restrict__ ws_db, int dbg) {
  // dbg: ablation switches for tools/mlp_probe.py (0 in production): 1 no bias sums, 2 no dx MFMAs,
  // 4 no dw MFMAs, 8 no operand refetch, 16 no dx stores
  constexpr int KCB = 32 * KB, LDX = KCB + 1, LDW = KCB + 1;
  constexpr int GCB = NB < 2 ? NB : 2;       // 32-column blocks of g per LDS chunk
  constexpr int GC = 32 * GCB, LDG = GC + 1;
  constexpr int NCH = NB / GCB;
  constexpr int RED = NB * 32 * KCB;         // floats of the dw chunk
  constexpr int SG = BM * LDG > RED ? BM * LDG : RED;
  __shared__ float sG[SG];                   // g chunk; re-used for the cross-wave dw reduction
  __shared__ float sX[BM * LDX];
  __shared__ float sW[NB * 32 * LDW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slab = blockIdx.x;
  const int k0 = blockIdx.y * KCB;
  const int64_t ntiles = (B + BM - 1) / BM;
  const int64_t tile_lo = (int64_t)slab * tiles_per_slab;
  int64_t tile_hi = tile_lo + tiles_per_slab;
  if (tile_hi > ntiles) tile_hi = ntiles;
  const int nsteps = (int)(tile_hi - tile_lo) * NCH;

  {
    TileRegs<KCB, VECK, NB * 32> rw;
    rw.load(w, N, K, 0, k0);
    rw.template store<LDW>(sW);
  }
  f32x16 accW[NB][KB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) accW[nb][kb][r] = 0.f;
  f32x16 accX[KB];
  float dbacc = 0.f;   // thread t sums column t of g (blockIdx.y == 0 only)

  TileRegs<GC, VECN> rg, ry;
  TileRegs<KCB, VECK> rx;
  auto fetch = [&](int step) {   // operands of `step` -> registers
    const int64_t row0 = (tile_lo + step / NCH) * BM;
    const int nc = (step % NCH) * GC;
    rg.load(dy, B, N, row0, nc);
    if (act == 1) ry.load(y, B, N, row0, nc);
    if (step % NCH == 0) rx.load(x, B, K, row0, k0);
  };
  if (nsteps > 0) fetch(0);

  for (int tstep = 0; tstep < nsteps; tstep += NCH) {
    const int64_t row0 = (tile_lo + tstep / NCH) * BM;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) accX[kb][r] = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int nc = ch * GC;
      __syncthreads();           // the previous step's readers of sG (and sX) are done
      if (act == 1) {
#pragma unroll
        for (int i = 0; i < TileRegs<GC, VECN>::NV; ++i) rg.v[i] = relu_mask(rg.v[i], ry.v[i]);
      }
      rg.template store<LDG>(sG);
      if (ch == 0) rx.template store<LDX>(sX);
      __syncthreads();
      if (tstep + ch + 1 < nsteps && !(dbg & 8)) fetch(tstep + ch + 1);   // in flight behind the MFMAs below
      if (blockIdx.y == 0 && !(dbg & 1)) {     // bias gradient: column sums of g
        const int t = threadIdx.x;
        if (t >= nc && t < nc + GC && t < N) {
          float s = 0.f;
          for (int r = 0; r < BM; ++r) s += sG[r * LDG + (t - nc)];
          dbacc += s;
        }
      }
      if (dx && !(dbg & 2)) {    // dx[rows of this wave, KCB] += g[rows, chunk] @ w[chunk, KCB]
#pragma unroll 8
        for (int n2 = 0; n2 < GC; n2 += 2) {
          const float a = sG[(wave * 32 + (lane & 31)) * LDG + n2 + (lane >> 5)];
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) {
            const float bb = sW[(nc + n2 + (lane >> 5)) * LDW + kb * 32 + (lane & 31)];
            accX[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, accX[kb], 0, 0, 0);
          }
        }
      }
      // dw[chunk, KCB] += g[rows of this wave, chunk]^T @ x[rows of this wave, KCB]
      if (!(dbg & 4))
#pragma unroll 8
      for (int r2 = 0; r2 < 32; r2 += 2) {
        const int row = wave * 32 + r2 + (lane >> 5);
#pragma unroll
        for (int gb = 0; gb < GCB; ++gb) {
          const float a = sG[row * LDG + gb * 32 + (lane & 31)];
#pragma unroll
          for (int kb = 0; kb < KB; ++kb) {
            const float bb = sX[row * LDX + kb * 32 + (lane & 31)];
            accW[ch * GCB + gb][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, accW[ch * GCB + gb][kb], 0, 0, 0);
          }
        }
      }
    }
    if (dx && !(dbg & 16)) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int col = k0 + kb * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t gr = row0 + wave * 32 + cd_row(r, lane);
          if (gr < B && col < K) dx[gr * K + col] = accX[kb][r];
        }
      }
    }
  }

  // ---- add the four waves' dw partials in wave order through LDS, then write the slab's partial
  for (int wv = 0; wv < 4; ++wv) {
    __syncthreads();
    if (wave == wv) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int idx = (nb * 32 + cd_row(r, lane)) * KCB + kb * 32 + (lane & 31);
            sG[idx] = wv == 0 ? accW[nb][kb][r] : sG[idx] + accW[nb][kb][r];
          }
    }
  }
  __syncthreads();
  float *out = ws_dw + (int64_t)slab * N * K;
  for (int e = threadIdx.x; e < RED; e += 256) {
    const int n = e / KCB, c = e % KCB;
    if (n < N && k0 + c < K) out[(int64_t)n * K + k0 + c] = sG[e];
  }
  if (blockIdx.y == 0 && (int)threadIdx.x < N) ws_db[(int64_t)slab * N + threadIdx.x] = dbacc;
}

// dw[i] = sum_s ws_dw[s][i], db likewise.  A block owns 16 consecutive outputs; its 16 thread
// groups each sum every 16th slab (in slab order), the 16 partials are added in group order.
__global__ __launch_bounds__(256) void k_linear_bwd_reduce(const float *__restrict__ ws_dw,
                                                           const float *__restrict__ ws_db, int nslabs, int64_t nk,
                                                           int N, float *__restrict__ dw, float *__restrict__ db) {
  __shared__ float part[16][17];
  const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + e;
  float s = 0.f;
  if (i < nk) {
    for (int sl = grp; sl < nslabs; sl += 16) s += ws_dw[(int64_t)sl * nk + i];
  } else if (i < nk + N) {
    for (int sl = grp; sl < nslabs; sl += 16) s += ws_db[(int64_t)sl * N + (i - nk)];
  }
  part[grp][e] = s;
  __syncthreads();
  if (grp == 0) {
    float t = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 16; ++g2) t += part[g2][e];
    if (i < nk) dw[i] = t;
    else if (i < nk + N) db[i - nk] = t;
  }
}

// ==========================================================================================
// Whole-network kernels for MLPs whose layers are all at most 64 wide (fan-in and fan-out):
// Linear -> ReLU -> ... -> Linear, 1..3 Linear layers.  One launch runs the forward of every
// layer for a 128-row tile with the activations handed from layer to layer through LDS (they
// are also written out once, for the backward); one launch runs the whole backward for a slab
// of tiles: per layer dW += g^T a, db += colsum g, g <- (g W) * (a > 0), with g and the layer
// input a in LDS and every layer's dW partial in registers across the slab's tiles.  Compared
// with one launch pair per layer this removes the per-launch latency floor (~10 us each at
// these sizes) and the re-reads of g between layers.  Every layer is treated as 64x64 with zero
// padding (NB = KB = 2): padded weight rows/columns are zero, so padded activations are zero.
struct MlpArgs {
  int32_t depth;             // number of Linear layers, 1..3
  int32_t dims[4];           // dims[0] = fan-in of the net, dims[l+1] = fan-out of layer l (all <= 64)
  const float *w[3];         // [dims[l+1], dims[l]] row-major
  const float *b[3];
  float *act[3];             // forward: out; backward: in.  act[l] = output of layer l, [B, dims[l+1]]
  int32_t vec[4];            // vec[l]: rows of width dims[l] are 16 B-loadable (dims % 4 == 0, aligned bases)
};

constexpr int MAXG = 4;      // networks per grouped launch (blockIdx.y)
struct MlpGroup {
  MlpArgs net[MAXG];
  const float *x[MAXG];
  const float *dy[MAXG];     // backward only
  float *ws_part[MAXG];      // backward only: [slabs][nparams[g]]
  int32_t nparams[MAXG];
};

constexpr int LDA = 65;      // LDS row stride of every 128 x 64 / 64 x 64 tile

template <bool VEC, int ROWS>
__device__ __forceinline__ void stage64(float *s, const float *__restrict__ m, int64_t rows, int ld, int64_t row0) {
  TileRegs<64, VEC, ROWS> r;
  r.load(m, rows, ld, row0, 0);
  r.template store<LDA>(s);
}

__global__ __launch_bounds__(256) void k_mlp_fwd(int64_t B, MlpGroup grp) {
  const MlpArgs &a = grp.net[blockIdx.y];       // blockIdx.y: which network of the group
  const float *__restrict__ x = grp.x[blockIdx.y];
  __shared__ float sA[BM * LDA];     // the current layer's input; overwritten in place by its output
  __shared__ float sW[64 * LDA];     // 50 KB in all: three workgroups per CU, so tiles overlap each other's latency
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row0 = (int64_t)blockIdx.x * BM;
  if (a.vec[0]) stage64<true, BM>(sA, x, B, a.dims[0], row0);
  else stage64<false, BM>(sA, x, B, a.dims[0], row0);
  for (int l = 0; l < a.depth; ++l) {
    const int K = a.dims[l], N = a.dims[l + 1];
    if (a.vec[l]) stage64<true, 64>(sW, a.w[l], N, K, 0);   // sW is free: a barrier follows every MFMA loop
    else stage64<false, 64>(sW, a.w[l], N, K, 0);
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    const int kmax = (K + 1) & ~1;
    for (int kk = 0; kk < kmax; kk += 2) {
      const float av = sA[(wave * 32 + (lane & 31)) * LDA + kk + (lane >> 5)];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float bv = sW[(nb * 32 + (lane & 31)) * LDA + kk + (lane >> 5)];
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[nb], 0, 0, 0);
      }
    }
    __syncthreads();                      // every wave is done reading sA and sW
    const bool relu = l + 1 < a.depth;
    float *out = a.act[l];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int col = nb * 32 + (lane & 31);
      const float bias = col < N ? a.b[l][col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = wave * 32 + cd_row(r, lane);
        float v = acc[nb][r] + bias;
        if (relu) v = v > 0.f ? v : 0.f;
        sA[lr * LDA + col] = v;             // next layer's operand (padded columns are exactly 0)
        const int64_t gr = row0 + lr;
        if (gr < B && col < N) out[gr * N + col] = v;
      }
    }
  }
}

// ws_part[slab] holds the slab's parameter-gradient partials, densely: for l = 0..depth-1:
// dW_l [dims[l+1] * dims[l]] then db_l [dims[l+1]].
__global__ __launch_bounds__(256) void k_mlp_bwd(int64_t B, MlpGroup grp, int tiles_per_slab) {
  const MlpArgs &a = grp.net[blockIdx.y];
  const float *__restrict__ x = grp.x[blockIdx.y];
  const float *__restrict__ dy = grp.dy[blockIdx.y];
  float *__restrict__ ws_part = grp.ws_part[blockIdx.y];
  const int nparams = grp.nparams[blockIdx.y];
  __shared__ float sG[BM * LDA];     // gradient w.r.t. the current layer's output (masked)
  __shared__ float sAct[BM * LDA];   // the current layer's input activation
  __shared__ float sW[64 * LDA];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slab = blockIdx.x;
  const int64_t ntiles = (B + BM - 1) / BM;
  const int64_t tile_lo = (int64_t)slab * tiles_per_slab;
  int64_t tile_hi = tile_lo + tiles_per_slab;
  if (tile_hi > ntiles) tile_hi = ntiles;

  f32x16 accW[3][2][2];
#pragma unroll
  for (int l = 0; l < 3; ++l)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[l][nb][kb][r] = 0.f;
  float dbacc[3] = {0.f, 0.f, 0.f};   // thread t: column t & 63, rows of group t >> 6
  const int D = a.depth;

  for (int64_t tile = tile_lo; tile < tile_hi; ++tile) {
    const int64_t row0 = tile * BM;
    __syncthreads();                   // the previous tile's readers of sG / sAct are done
    if (a.vec[D]) stage64<true, BM>(sG, dy, B, a.dims[D], row0);
    else stage64<false, BM>(sG, dy, B, a.dims[D], row0);
    // layers from the last to the first; `li` is a compile-time constant so that accW[li] stays in registers
    auto layer = [&](auto li_c) {
      constexpr int li = decltype(li_c)::value;
      const int l = D - 1 - li;
      if (l < 0) return;
      const int K = a.dims[l], N = a.dims[l + 1];
      const float *ain = l > 0 ? a.act[l - 1] : x;
      if (a.vec[l]) stage64<true, BM>(sAct, ain, B, K, row0);
      else stage64<false, BM>(sAct, ain, B, K, row0);
      if (l > 0) {
        if (a.vec[l]) stage64<true, 64>(sW, a.w[l], N, K, 0);
        else stage64<false, 64>(sW, a.w[l], N, K, 0);
      }
      __syncthreads();
      {   // db_l: column sums of g
        const int c = threadIdx.x & 63, grp = threadIdx.x >> 6;
        float sum = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) sum += sG[(grp * 32 + r) * LDA + c];
        dbacc[li] += sum;
      }
      // dW_l += g^T a   (rows of this wave)
      const int nmax_b = (N + 31) >> 5, kmax_b = (K + 31) >> 5;   // skip all-zero 32-blocks
      for (int r2 = 0; r2 < 32; r2 += 2) {
        const int row = wave * 32 + r2 + (lane >> 5);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          if (nb < nmax_b) {
            const float av = sG[row * LDA + nb * 32 + (lane & 31)];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
              if (kb < kmax_b) {
                const float bv = sAct[row * LDA + kb * 32 + (lane & 31)];
                accW[li][nb][kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accW[li][nb][kb], 0, 0, 0);
              }
            }
          }
        }
      }
      if (l > 0) {   // g <- (g @ W_l) * (a > 0)
        f32x16 accX[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) accX[kb][r] = 0.f;
        const int nmax = (N + 1) & ~1;
        for (int n2 = 0; n2 < nmax; n2 += 2) {
          const float av = sG[(wave * 32 + (lane & 31)) * LDA + n2 + (lane >> 5)];
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            const float bv = sW[(n2 + (lane >> 5)) * LDA + kb * 32 + (lane & 31)];
            accX[kb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, accX[kb], 0, 0, 0);
          }
        }
        __syncthreads();               // every reader of the old g is done
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          const int col = kb * 32 + (lane & 31);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int lr = wave * 32 + cd_row(r, lane);
            sG[lr * LDA + col] = sAct[lr * LDA + col] > 0.f ? accX[kb][r] : 0.f;
          }
        }
        __syncthreads();               // new g complete; sAct / sW free for the next layer
      }
    };
    layer(std::integral_constant<int, 0>{});
    layer(std::integral_constant<int, 1>{});
    layer(std::integral_constant<int, 2>{});
  }

  // ---- per layer: add the four waves' dW partials in wave order through LDS, write the slab's partial
  float *part = ws_part + (int64_t)slab * nparams;
  auto flush = [&](auto li_c) {
    constexpr int li = decltype(li_c)::value;
    const int l = D - 1 - li;
    if (l < 0) return;
    const int K = a.dims[l], N = a.dims[l + 1];
    int loff = 0;                      // offset of layer l in the dense parameter order
    for (int j = 0; j < l; ++j) loff += a.dims[j + 1] * a.dims[j] + a.dims[j + 1];
    for (int wv = 0; wv < 4; ++wv) {
      __syncthreads();
      if (wave == wv) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int idx = (nb * 32 + cd_row(r, lane)) * 64 + kb * 32 + (lane & 31);
              sG[idx] = wv == 0 ? accW[li][nb][kb][r] : sG[idx] + accW[li][nb][kb][r];
            }
      }
    }
    // bias partials of the four row groups -> sAct[0..255], summed in group order below
    sAct[threadIdx.x] = dbacc[li];
    __syncthreads();
    for (int e = threadIdx.x; e < N * K; e += 256) part[loff + e] = sG[(e / K) * 64 + (e % K)];
    if ((int)threadIdx.x < N)
      part[loff + N * K + threadIdx.x] =
          ((sAct[threadIdx.x] + sAct[64 + threadIdx.x]) + sAct[128 + threadIdx.x]) + sAct[192 + threadIdx.x];
  };
  flush(std::integral_constant<int, 0>{});
  flush(std::integral_constant<int, 1>{});
  flush(std::integral_constant<int, 2>{});
}

// out[i] = sum over slabs of ws_part[slab][i], 16 outputs per block, slabs in a fixed order.  The
// sums go either to one dense array or, parameter by parameter, to caller-given destinations
// (e.g. the .grad views of a flat gradient buffer), segment j covering [start[j], start[j+1]).
struct GradDst {
  int32_t nseg;              // 0: dense output
  int32_t start[7];
  float *dst[6];
};

struct ReduceGroup {
  const float *ws_part[MAXG];
  float *grads[MAXG];
  int32_t nparams[MAXG];
  GradDst gd[MAXG];
};

__global__ __launch_bounds__(256) void k_mlp_bwd_reduce(int nslabs, ReduceGroup rg) {
  const float *__restrict__ ws_part = rg.ws_part[blockIdx.y];
  float *__restrict__ grads = rg.grads[blockIdx.y];
  const int nparams = rg.nparams[blockIdx.y];
  const GradDst &gd = rg.gd[blockIdx.y];
  __shared__ float part[16][17];
  const int e = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + e;
  float s = 0.f;
  if (i < nparams)
    for (int sl = grp; sl < nslabs; sl += 16) s += ws_part[(int64_t)sl * nparams + i];
  part[grp][e] = s;
  __syncthreads();
  if (grp == 0 && i < nparams) {
    float t = 0.f;
#pragma unroll
    for (int g2 = 0; g2 < 16; ++g2) t += part[g2][e];
    if (gd.nseg == 0) {
      grads[i] = t;
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j)
        if (j < gd.nseg && i >= gd.start[j] && i < gd.start[j + 1]) gd.dst[j][i - gd.start[j]] = t;
    }
  }
}

constexpr int64_t WS_CAP_BYTES = 128ll << 20;
constexpr int MAX_SLABS = 256;

int n_slabs(int64_t B, int K, int N) {
  const int64_t ntiles = trl_cdiv(B, BM);
  int64_t s = ntiles < MAX_SLABS ? ntiles : MAX_SLABS;
  const int64_t per = (int64_t)N * K * 4;
  const int64_t cap = WS_CAP_BYTES / per;
  if (s > cap) s = cap < 1 ? 1 : cap;
  return (int)s;
}

int check_dims(const char *fn, int64_t B, int K, int N, int act) {
  TRL_REQUIRE(B > 0 && B < (1ll << 31), "%s: B=%lld out of range", fn, (long long)B);
  TRL_REQUIRE(K > 0 && K <= (1 << 20), "%s: fan-in K=%d out of range", fn, K);
  TRL_REQUIRE(N > 0 && N <= 256, "%s: fan-out N=%d must be in [1, 256]", fn, N);
  TRL_REQUIRE(act == 0 || act == 1, "%s: act must be 0 (none) or 1 (relu)", fn);
  return TRL_OK;
}

}  // namespace

extern "C" int trl_linear_fwd(const float *x, const float *w, const float *b, int64_t B, int32_t K, int32_t N,
                              int32_t act, float *y, void *stream) {
  const char *fn = "trl_linear_fwd";
  int rc = check_dims(fn, B, K, N, act);
  if (rc) return rc;
  TRL_REQUIRE(x && w && b && y, "%s: null pointer", fn);
  hipStream_t st = (hipStream_t)stream;
  // a workgroup owns 128 rows x NBv * 32 columns; with few row tiles (small batches: the off-policy
  // learners' 32-256 rows) the fan-out is split one 32-column block per workgroup so that the
  // launch covers more than a handful of CUs
  int NBv = N <= 32 ? 1 : (N <= 64 ? 2 : (N <= 128 ? 4 : 8));
  if (trl_cdiv(B, BM) < 32) NBv = 1;
  const int KBv = (K > 32 && NBv <= 4) ? 2 : 1;
  const dim3 grid((unsigned)trl_cdiv(B, BM), (unsigned)trl_cdiv(N, NBv * 32)), block(256);
  const bool vec = K % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0;
#define TRL_FWD_CASE(nb, kb)                                                                                     \
  if (NBv == nb && KBv == kb) {                                                                                  \
    if (vec)                                                                                                     \
      hipLaunchKernelGGL((k_linear_fwd<nb, kb, true>), grid, block, 0, st, x, w, b, B, (int)K, (int)N, (int)act, y); \
    else                                                                                                         \
      hipLaunchKernelGGL((k_linear_fwd<nb, kb, false>), grid, block, 0, st, x, w, b, B, (int)K, (int)N, (int)act, y); \
  }
  TRL_FWD_CASE(1, 1) TRL_FWD_CASE(1, 2) TRL_FWD_CASE(2, 1) TRL_FWD_CASE(2, 2) TRL_FWD_CASE(4, 1) TRL_FWD_CASE(4, 2)
  TRL_FWD_CASE(8, 1)
#undef TRL_FWD_CASE
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

extern "C" int64_t trl_linear_bwd_workspace_bytes(int64_t B, int32_t K, int32_t N) {
  if (B <= 0 || K <= 0 || N <= 0 || N > 256) return 0;
  const int s = n_slabs(B, K, N);
  return (int64_t)s * ((int64_t)N * K + N) * 4;
}

extern "C" int trl_linear_bwd(const float *dy, const float *y, const float *x, const float *w, int64_t B, int32_t K,
                              int32_t N, int32_t act, float *dx, float *dw, float *db, void *ws, void *stream) {
  const char *fn = "trl_linear_bwd";
  int rc = check_dims(fn, B, K, N, act);
  if (rc) return rc;
  TRL_REQUIRE(dy && x && w && dw && db && ws, "%s: null pointer", fn);
  TRL_REQUIRE(act == 0 || y != nullptr, "%s: relu backward needs the layer output y", fn);
  hipStream_t st = (hipStream_t)stream;
  const int64_t ntiles = trl_cdiv(B, BM);
  const int tps = (int)trl_cdiv(ntiles, n_slabs(B, K, N));
  const int slabs = (int)trl_cdiv(ntiles, tps);   // <= n_slabs(): every slab has at least one tile
  float *ws_dw = (float *)ws;
  float *ws_db = ws_dw + (int64_t)slabs * N * K;
  const int NBv = N <= 32 ? 1 : (N <= 64 ? 2 : (N <= 128 ? 4 : 8));
  const int KBv = (NBv <= 4 && K > 32) ? 2 : 1;
  const dim3 grid((unsigned)slabs, (unsigned)trl_cdiv(K, 32 * KBv)), block(256);
  TRL_REQUIRE(grid.y <= 65535, "%s: fan-in %d needs too many column chunks", fn, K);
  int dbg = 0;
#ifdef TRL_TUNING   // ablation switches (wrong results, timing only): diagnostic build, tools/mlp_probe.py
  if (const char *t = getenv("TRL_MLP_DBG")) dbg = atoi(t);
#endif
  const bool vn = N % 4 == 0 && (uintptr_t)dy % 16 == 0 && (y == nullptr || (uintptr_t)y % 16 == 0);
  const bool vk = K % 4 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)w % 16 == 0;
#define TRL_BWD_LAUNCH(nb, kb, a, c)                                                                               \
  hipLaunchKernelGGL((k_linear_bwd<nb, kb, a, c>), grid, block, 0, st, dy, y, x, w, B, (int)K, (int)N, (int)act, tps, \
                     dx, ws_dw, ws_db, dbg)
#define TRL_BWD_CASE(nb, kb)                                     \
  if (NBv == nb && KBv == kb) {                                  \
    if (vn && vk) TRL_BWD_LAUNCH(nb, kb, true, true);            \
    else if (vn) TRL_BWD_LAUNCH(nb, kb, true, false);            \
    else if (vk) TRL_BWD_LAUNCH(nb, kb, false, true);            \
    else TRL_BWD_LAUNCH(nb, kb, false, false);                   \
  }
  TRL_BWD_CASE(1, 1) TRL_BWD_CASE(1, 2) TRL_BWD_CASE(2, 1) TRL_BWD_CASE(2, 2) TRL_BWD_CASE(4, 1) TRL_BWD_CASE(4, 2)
  TRL_BWD_CASE(8, 1)
#undef TRL_BWD_CASE
#undef TRL_BWD_LAUNCH
  TRL_CHECK_LAUNCH(fn);
  const int64_t nk = (int64_t)N * K;
  hipLaunchKernelGGL(k_linear_bwd_reduce, dim3((unsigned)trl_cdiv(nk + N, 16)), dim3(256), 0, st,
                     (const float *)ws_dw, (const float *)ws_db, slabs, nk, (int)N, dw, db);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

namespace {
int fill_mlp_args(const char *fn, const trl_mlp_desc *net, MlpArgs *a, int *nparams) {
  TRL_REQUIRE(net != nullptr, "%s: null network description", fn);
  TRL_REQUIRE(net->depth >= 1 && net->depth <= 3, "%s: depth %d not in [1, 3]", fn, net->depth);
  a->depth = net->depth;
  int np = 0;
  for (int l = 0; l <= net->depth; ++l) {
    TRL_REQUIRE(net->dims[l] >= 1 && net->dims[l] <= 64, "%s: width dims[%d]=%d must be in [1, 64]", fn, l,
                net->dims[l]);
    a->dims[l] = net->dims[l];
  }
  for (int l = 0; l < net->depth; ++l) {
    TRL_REQUIRE(net->w[l] && net->b[l], "%s: null weight/bias of layer %d", fn, l);
    a->w[l] = net->w[l];
    a->b[l] = net->b[l];
    np += net->dims[l + 1] * net->dims[l] + net->dims[l + 1];
  }
  *nparams = np;
  return TRL_OK;
}
inline bool rows16(int width, const void *p) { return width % 4 == 0 && (uintptr_t)p % 16 == 0; }
}  // namespace

extern "C" int trl_mlp_fwd_group(const trl_mlp_desc *nets, int32_t G, const float *const *xs, int64_t B,
                                 float *const *acts, void *stream) {
  const char *fn = "trl_mlp_fwd_group";
  TRL_REQUIRE(nets && xs && acts, "%s: null pointer", fn);
  TRL_REQUIRE(G >= 1 && G <= MAXG, "%s: group of %d networks not in [1, %d]", fn, G, MAXG);
  TRL_REQUIRE(B > 0 && B < (1ll << 31), "%s: B=%lld out of range", fn, (long long)B);
  MlpGroup grp{};
  for (int g = 0; g < G; ++g) {
    MlpArgs &a = grp.net[g];
    int np = 0;
    int rc = fill_mlp_args(fn, &nets[g], &a, &np);
    if (rc) return rc;
    TRL_REQUIRE(xs[g] != nullptr, "%s: xs[%d] is null", fn, g);
    grp.x[g] = xs[g];
    for (int l = 0; l < a.depth; ++l) {
      float *out = acts[3 * g + l];
      TRL_REQUIRE(out != nullptr, "%s: acts[%d][%d] is null", fn, g, l);
      a.act[l] = out;
      a.vec[l] = (l == 0 ? rows16(a.dims[0], xs[g]) : rows16(a.dims[l], acts[3 * g + l - 1])) &&
                 rows16(a.dims[l], a.w[l]);
    }
  }
  hipLaunchKernelGGL(k_mlp_fwd, dim3((unsigned)trl_cdiv(B, BM), (unsigned)G), dim3(256), 0, (hipStream_t)stream, B,
                     grp);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

extern "C" int trl_mlp_fwd(const trl_mlp_desc *net, const float *x, int64_t B, float *const *acts, void *stream) {
  TRL_REQUIRE(net && acts, "trl_mlp_fwd: null pointer");
  float *a3[3] = {nullptr, nullptr, nullptr};
  for (int l = 0; l < net->depth && l < 3; ++l) a3[l] = acts[l];
  return trl_mlp_fwd_group(net, 1, &x, B, a3, stream);
}

static int64_t mlp_slabs(int64_t B) {
  const int64_t ntiles = trl_cdiv(B, BM);
  return ntiles < MAX_SLABS ? ntiles : MAX_SLABS;
}

extern "C" int64_t trl_mlp_bwd_workspace_bytes(const trl_mlp_desc *net, int64_t B) {
  if (!net || net->depth < 1 || net->depth > 3 || B <= 0) return 0;
  int np = 0;
  for (int l = 0; l < net->depth; ++l) np += net->dims[l + 1] * net->dims[l] + net->dims[l + 1];
  return mlp_slabs(B) * (int64_t)((np + 3) / 4 * 4) * 4;
}

extern "C" int trl_mlp_bwd_group(const trl_mlp_desc *nets, int32_t G, const float *const *xs,
                                 const float *const *acts, const float *const *dys, int64_t B, float *const *grads,
                                 float *const *param_grads, void *ws, void *stream) {
  const char *fn = "trl_mlp_bwd_group";
  TRL_REQUIRE(nets && xs && acts && dys && ws && (grads || param_grads), "%s: null pointer", fn);
  TRL_REQUIRE(G >= 1 && G <= MAXG, "%s: group of %d networks not in [1, %d]", fn, G, MAXG);
  TRL_REQUIRE(B > 0 && B < (1ll << 31), "%s: B=%lld out of range", fn, (long long)B);
  const int64_t ntiles = trl_cdiv(B, BM);
  const int tps = (int)trl_cdiv(ntiles, mlp_slabs(B));
  const int slabs = (int)trl_cdiv(ntiles, tps);
  MlpGroup grp{};
  ReduceGroup rg{};
  float *wsp = (float *)ws;
  int np_max = 0;
  for (int g = 0; g < G; ++g) {
    MlpArgs &a = grp.net[g];
    int np = 0;
    int rc = fill_mlp_args(fn, &nets[g], &a, &np);
    if (rc) return rc;
    TRL_REQUIRE(xs[g] && dys[g], "%s: xs[%d] / dys[%d] is null", fn, g, g);
    grp.x[g] = xs[g];
    grp.dy[g] = dys[g];
    grp.nparams[g] = np;
    grp.ws_part[g] = wsp;
    rg.ws_part[g] = wsp;
    rg.nparams[g] = np;
    wsp += mlp_slabs(B) * (int64_t)((np + 3) / 4 * 4);
    np_max = np > np_max ? np : np_max;
    for (int l = 0; l + 1 < a.depth; ++l) {
      TRL_REQUIRE(acts[3 * g + l] != nullptr, "%s: acts[%d][%d] is null", fn, g, l);
      a.act[l] = (float *)acts[3 * g + l];
    }
    for (int l = 0; l < a.depth; ++l)
      a.vec[l] = (l == 0 ? rows16(a.dims[0], xs[g]) : rows16(a.dims[l], acts[3 * g + l - 1])) &&
                 rows16(a.dims[l], a.w[l]);
    a.vec[a.depth] = rows16(a.dims[a.depth], dys[g]);
    GradDst &gd = rg.gd[g];
    if (param_grads) {
      gd.nseg = 2 * a.depth;
      int off = 0;
      for (int l = 0; l < a.depth; ++l) {
        float *dw = param_grads[6 * g + 2 * l], *db = param_grads[6 * g + 2 * l + 1];
        TRL_REQUIRE(dw && db, "%s: null gradient destination of net %d layer %d", fn, g, l);
        gd.start[2 * l] = off;
        gd.dst[2 * l] = dw;
        off += a.dims[l + 1] * a.dims[l];
        gd.start[2 * l + 1] = off;
        gd.dst[2 * l + 1] = db;
        off += a.dims[l + 1];
      }
      gd.start[gd.nseg] = off;
    } else {
      TRL_REQUIRE(grads[g] != nullptr, "%s: grads[%d] is null", fn, g);
      rg.grads[g] = grads[g];
    }
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_mlp_bwd, dim3((unsigned)slabs, (unsigned)G), dim3(256), 0, st, B, grp, tps);
  TRL_CHECK_LAUNCH(fn);
  hipLaunchKernelGGL(k_mlp_bwd_reduce, dim3((unsigned)trl_cdiv(np_max, 16), (unsigned)G), dim3(256), 0, st, slabs, rg);
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}

extern "C" int trl_mlp_bwd(const trl_mlp_desc *net, const float *x, const float *const *acts, const float *dy,
                           int64_t B, float *grads, float *const *param_grads, void *ws, void *stream) {
  TRL_REQUIRE(net && acts, "trl_mlp_bwd: null pointer");
  const float *a3[3] = {nullptr, nullptr, nullptr};
  for (int l = 0; l + 1 < net->depth && l < 3; ++l) a3[l] = acts[l];
  float *pg[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (param_grads)
    for (int j = 0; j < 2 * net->depth && j < 6; ++j) pg[j] = param_grads[j];
  return trl_mlp_bwd_group(net, 1, &x, a3, &dy, B, grads ? &grads : nullptr, param_grads ? pg : nullptr, ws, stream);
}
