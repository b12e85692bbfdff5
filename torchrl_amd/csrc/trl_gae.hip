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
// Structure: a workgroup owns E consecutive envs and walks T backwards in tiles of TT rows.
// The recurrence is serial in t, so the work is split by role:
//   - mover lanes: load the reward/value tile (16 B per lane, coalesced along the env axis),
//                  form delta off the dependent chain and stage delta and v in LDS;
//   - E lanes:     run the chain  gae = delta + c * gae  over the tile out of LDS
//                  (2 dependent VALU ops per step), writing gae back in place;
//   - mover lanes: ret/adv from gae and the staged value tile, 16 B stores.
// k_gae_pipe (the fast path) gives each role waves of its own and software-pipelines them
// over LDS rings: at stage s two DMA waves stream tile s+PF straight into LDS, the worker
// waves stage tile s and write out tile s-2, and the chain wave scans tile s-1; one barrier
// per stage.  In reference mode the chain wave leaves only one value per 8 steps in LDS and the workers expand
// the tile a stage later (template flag CKPT, below).  k_gae_tiled is the plain, unpipelined form for what the
// fast path does not take (N % 4 != 0, unaligned bases).
// HBM traffic is the algorithmic minimum: r and v read once (the v[t+1] operand re-reads the
// same lines through L1/L2), ret and adv written once = 16 B per (t, env) step.
//
// Layouts.  [T, N] (env index contiguous) is what one env step over all envs appends and the parity
// mode; there a workgroup's E = 16 envs are a 64 B segment of every row.  The BLOCKED rollout layout
// [N / E][T][E] keeps a workgroup's E envs x T steps contiguous, so that its reward / value / returns /
// advantage streams are flat (every LDS-DMA instruction moves 1 KiB of consecutive bytes, every tile
// is one 4 KiB run): same kernel, the planes are addressed as base(group) + t * row_stride.
#include <stdlib.h>

#include "trl_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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


// ---------------------------------------------------------------------------------------
// Pipelined kernel (reference mode).  Block = 448 lanes = 7 waves in three roles:
//   wave 0      chain    E active lanes scan tile s-1 out of LDS
//   waves 1-2   DMA      stream tile s+PF of reward and value HBM -> LDS with LDS-DMA
//                        (global_load_lds, 16 B per lane, 1 KiB per instruction, no VGPRs),
//                        PF tiles in flight; nothing else sits in these waves' vector-memory
//                        queue, so one counted s_waitcnt vmcnt retires exactly the tile the
//                        workers read next.  Loads are never predicated (clamped addresses),
//                        which keeps the instruction count per stage fixed.
//   waves 3-6   workers  tile s: delta from the raw planes -> padded scan buffers;
//                        tile s-2: ret = gae + v, adv = ret - v, 16 B stores
// One barrier per stage.  A tile is TT = 2048 / E rows = 8 KiB per f32 plane.
// LDS-DMA visibility rule (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the
// wait that retires it"): the DMA waves wait for tile s+1 at the END of stage s, before the
// barrier; the workers read it in stage s+1.
// MASKED (extension): a third plane, done, rides the same ring; the workers fold (1 - done) into
// delta and the chain wave reads the tile's done values straight from the raw ring (the slot of
// the tile being scanned is the one kept alive as the next tile's halo):
//   delta = (r + (gamma * v[t+1]) * (1 - d)) - v;   gae = delta + (gamma_lambda * (1 - d)) * gae
// CKPT (reference mode only): the chain wave keeps the running gae in registers and leaves ONE value per 8 steps in
// LDS (a checkpoint) instead of writing every step back: an LDS store costs the issuing wave ~26 cycles whatever
// its width (tools/chain_probe.hip: the chain is 10.9 cycles per step on the VALU, 6.5 more with the write-backs),
// and the chain wave is the critical path.  One stage later the worker lanes, which have the issue slots to spare,
// EXPAND the tile: lane (env, group) re-runs the group's 8 steps from the checkpoint above it -- same operations
// in the same order, so the same bits -- and puts the 8 values where the write-out stage expects them.  The
// pipeline is then one stage deeper (stage / scan / expand / write out).
template <int E, int PF, bool MASKED, int U = 2, bool CKPT = false>
__global__ __launch_bounds__(448) void k_gae_pipe(const float *__restrict__ reward,
                                                  const float *__restrict__ value,
                                                  const float *__restrict__ done,
                                                  const float *__restrict__ bootstrap, int T, int N,
                                                  float gamma, float gl, int out_nt, int ngroups,
                                                  int64_t in_gs, int in_rs, int64_t out_gs, int out_rs,
                                                  float *__restrict__ returns,
                                                  float *__restrict__ adv, int dbg_in) {
  // element (t, env) of an input plane is at  grp * in_gs + t * in_rs + (env - grp * E)  floats:
  // [T, N]: in_gs = E, in_rs = N;  blocked [N / E][t_stride][E]: in_gs = t_stride * E, in_rs = E.
  // The row-strided outputs (out_nt == 0) likewise with out_gs / out_rs.
#ifdef TRL_TUNING
  // ablation switches for tools/gae_sweep.py (diagnostic build only; results are then wrong):
  // 1 = no chain, 2 = no output stores, 4 = no staging pass, 8 = no DMA; policy variants (results stay right):
  // 16 = non-temporal output stores, 32 = non-temporal input loads (both set by the blocked entry point);
  // 256 = cycle stamps of the chain wave
  const int dbg = dbg_in;
#else
  const int dbg = dbg_in & 48;   // the production library honours the two cache-policy bits only
#endif
  constexpr int QPR = E / 4;        // quads (float4) per tile row
  constexpr int RPU = 256 / QPR;    // rows covered by one pass of the 256 worker lanes
  constexpr int TT = U * RPU;       // rows per tile: U passes of the worker lanes (U = 1: half-size tiles,
                                    // twice the stages, half the pipeline fill and drain)
  constexpr int PLANE = TT * E;     // 2048 floats
  constexpr int LD = E + 1;         // row stride of the staged value tile [t][env] (odd: the env-major
                                    // writer's column reads spread over the banks)
  constexpr int TTP = TT + 4;       // row stride of the scan buffer, which is TRANSPOSED: [env][t], so
                                    // that a chain lane reads/writes 4 steps per ds_read/write_b128;
                                    // TTP % 64 == 4 (E 16, 32) or 36 (E 64) keeps 16 lanes on 16 bank quads
  constexpr int NS = CKPT ? 4 : 3;  // scan ring: staged / being scanned / (CKPT: being expanded /) being written out
  constexpr int NR = PF + 2;        // raw ring: PF in flight + being read + halo of the previous tile
  constexpr int RPI = 256 / E;      // rows one 1 KiB DMA instruction covers
  constexpr int NP = MASKED ? 3 : 2;                // raw planes per tile: reward, value(, done)
  constexpr int SLOT = NP * PLANE;
  constexpr int IPW = 2 * U;                        // DMA instructions per plane per tile per wave
  constexpr int LPT = IPW * NP;                     // loads per tile per DMA wave
  static_assert(!(CKPT && MASKED), "checkpointed chain: reference mode only");
  constexpr int CPS = TT / 8 + 1;                   // checkpoints per env per tile (+ the tile's carry-in); odd
  // one array: see cdna_hip_programming.md on a second __shared__ object beside a DMA ring
  __shared__ __attribute__((aligned(16))) float lds[NR * SLOT + NS * (E * TTP + TT * LD) + (CKPT ? NS * E * CPS : 1)];
  float *s_raw = lds;                               // [NR][NP][PLANE]  (reward plane, value plane, done plane)
  float *s_d = lds + NR * SLOT;                     // [NS][E][TTP]    delta (-> gae in place unless CKPT)
  float *s_v = s_d + NS * E * TTP;                  // [NS][TT][LD]
  float *s_cp = s_v + NS * TT * LD;                 // [NS][E][CPS]    CKPT: gae after each group of 8 steps

  // XCD-aware group order: blocks b and b + 8 share an XCD (its L2), so give each XCD a
  // contiguous run of env groups; neighbours that share a 128 B line then share an L2.
  int grp = blockIdx.x;
  if ((ngroups & 7) == 0) grp = (blockIdx.x & 7) * (ngroups >> 3) + (blockIdx.x >> 3);
  const int env0 = grp * E;

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int ntiles = (T + TT - 1) / TT;
  const int nstages = ntiles + (CKPT ? 3 : 2);
  // tile j (j = 0 is the LAST TT rows) covers t in [t_lo(j), t_hi(j))
  auto t_hi = [&](int j) { return T - j * TT; };
  auto t_lo = [&](int j) { const int lo = T - (j + 1) * TT; return lo < 0 ? 0 : lo; };

  if (wave == 1 || wave == 2) {
    // ================================================================== DMA waves
    const int dw = wave - 1;                       // owns instructions [dw*IPW, (dw+1)*IPW) of each plane
    const int q = lane % QPR, rsub = lane / QPR;
    // clamped: a quad beyond N (ragged last group of a [T, N] rollout) reads envs 0..3 instead
    const int64_t qoff = env0 + 4 * q < N ? (int64_t)grp * in_gs + 4 * q : 0;
    auto issue = [&](int j) {
      const int jc = j < ntiles ? j : ntiles - 1;
      const int lo = t_lo(jc), hi = t_hi(jc);
      float *slot = s_raw + (j % NR) * SLOT;
#pragma unroll
      for (int i = 0; i < IPW; ++i) {
        const int ins = dw * IPW + i;
        int t = lo + ins * RPI + rsub;
        t = t < hi ? t : hi - 1;
        const int64_t g = (int64_t)t * in_rs + qoff;
        if (dbg & 32) {   // tuning build: non-temporal policy on the input stream (aux = 2)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reward + g),
                                           (__attribute__((address_space(3))) void *)(slot + ins * 256), 16, 0, 2);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(value + g),
                                           (__attribute__((address_space(3))) void *)(slot + PLANE + ins * 256), 16,
                                           0, 2);
          if (MASKED)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(done + g),
                                             (__attribute__((address_space(3))) void *)(slot + 2 * PLANE + ins * 256),
                                             16, 0, 2);
        } else {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reward + g),
                                           (__attribute__((address_space(3))) void *)(slot + ins * 256), 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(value + g),
                                           (__attribute__((address_space(3))) void *)(slot + PLANE + ins * 256), 16,
                                           0, 0);
          if (MASKED)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(done + g),
                                             (__attribute__((address_space(3))) void *)(slot + 2 * PLANE + ins * 256),
                                             16, 0, 0);
        }
      }
    };
    // all but the PF - 1 youngest tiles of this wave have landed
    auto wait_tile = [&]() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT * (PF - 1)) : "memory"); };
    static_assert(LPT * (PF - 1) <= 63, "vmcnt is a 6-bit counter");
    static_assert(PF >= 1 && PF <= 5, "prefetch depth");
#pragma unroll
    for (int j = 0; j < PF; ++j)
      if (!(dbg & 8)) issue(j);
    // tile 0 landed (LPT loads per tile per wave; PF - 1 younger tiles may stay in flight)
    wait_tile();
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < nstages; ++s) {
      if (!(dbg & 8)) issue(s + PF);
      // tile s + 1 landed; tiles s+2 .. s+PF stay in flight across the barrier
      wait_tile();
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain before the LDS is released
  } else if (wave >= 3) {
    // ================================================================== workers
    const int m = (int)threadIdx.x - 192;            // [0, 256)
    const int q = m % QPR, row0 = m / QPR;
    const int envq = env0 + 4 * q;
    const bool qlive = envq < N;
    float4 boot4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (qlive) boot4 = *(const float4 *)(bootstrap + envq);
    __syncthreads();                                 // pairs with the DMA waves' prologue barrier
    for (int s = 0; s < nstages; ++s) {
      if (s < ntiles && !(dbg & 4)) {   // ---- stage tile s: delta and v into the padded scan buffers
        const int lo = t_lo(s), rows = t_hi(s) - lo;
        const float *raw_r = s_raw + (s % NR) * SLOT, *raw_v = raw_r + PLANE, *raw_m = raw_r + 2 * PLANE;
        const float *halo = s_raw + ((s + NR - 1) % NR) * SLOT + PLANE;   // tile s-1, row 0
        float *sd = s_d + (s % NS) * E * TTP, *sv = s_v + (s % NS) * TT * LD;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int tl = row0 + u * RPU;
          if (tl < rows) {
            const float4 r = *(const float4 *)(raw_r + tl * E + 4 * q);
            const float4 v = *(const float4 *)(raw_v + tl * E + 4 * q);
            float4 n;
            if (tl + 1 < rows) n = *(const float4 *)(raw_v + (tl + 1) * E + 4 * q);
            else if (s > 0) n = *(const float4 *)(halo + 4 * q);
            else n = boot4;
            const float r4[4] = {r.x, r.y, r.z, r.w}, v4[4] = {v.x, v.y, v.z, v.w};
            const float n4[4] = {n.x, n.y, n.z, n.w};
            float m4[4] = {0.f, 0.f, 0.f, 0.f};
            if (MASKED) {
              const float4 m = *(const float4 *)(raw_m + tl * E + 4 * q);
              m4[0] = m.x; m4[1] = m.y; m4[2] = m.z; m4[3] = m.w;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              float gv = __fmul_rn(gamma, n4[c]);
              if (MASKED) gv = __fmul_rn(gv, __fsub_rn(1.0f, m4[c]));
              sd[(4 * q + c) * TTP + tl] = __fsub_rn(__fadd_rn(r4[c], gv), v4[c]);
              sv[tl * LD + 4 * q + c] = v4[c];
            }
          }
        }
      }
      if constexpr (CKPT) {   // ---- expand tile s-2: checkpoints -> gae of every step, in place of the deltas
        const int je = s - 2;
        constexpr int GPT = TT / 8;                      // groups per env per tile; E * GPT <= 256 lanes
        if (je >= 0 && je < ntiles && m < E * GPT && !(dbg & 2)) {
          const int rows = t_hi(je) - t_lo(je);
          const int e = m / GPT, g8 = 8 * (m % GPT);
          if (g8 < rows) {
            float *sd_e = s_d + (je % NS) * E * TTP + e * TTP + g8;
            float gae = s_cp[(je % NS) * E * CPS + e * CPS + (g8 >> 3) + 1];
            const f32x4 dl = *(const f32x4 *)sd_e, dh = *(const f32x4 *)(sd_e + 4);
            f32x4 ol, oh;
#pragma unroll
            for (int i = 3; i >= 0; --i) {               // rows beyond a partial tile's end are not stepped
              const float nx = __fadd_rn(dh[i], __fmul_rn(gl, gae));
              gae = g8 + 4 + i < rows ? nx : gae;
              oh[i] = gae;
            }
#pragma unroll
            for (int i = 3; i >= 0; --i) {
              const float nx = __fadd_rn(dl[i], __fmul_rn(gl, gae));
              gae = g8 + i < rows ? nx : gae;
              ol[i] = gae;
            }
            *(f32x4 *)(sd_e + 4) = oh;
            *(f32x4 *)sd_e = ol;
          }
        }
      }
      const int jo = s - (CKPT ? 3 : 2);   // ---- write out tile s-2 (CKPT: s-3, expanded in the previous stage)
      if (jo >= 0 && jo < ntiles && !(dbg & 2)) {
        const int lo = t_lo(jo), hi = t_hi(jo);
        const float *sd = s_d + (jo % NS) * E * TTP, *sv = s_v + (jo % NS) * TT * LD;
        if (!out_nt) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int tl = row0 + u * RPU;
            if (qlive && lo + tl < hi) {
              float rt[4], ad[4];
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const float v = sv[tl * LD + 4 * q + c];
                rt[c] = __fadd_rn(sd[(4 * q + c) * TTP + tl], v);
                ad[c] = __fsub_rn(rt[c], v);
              }
              const int64_t g = (int64_t)grp * out_gs + (int64_t)(lo + tl) * out_rs + 4 * q;
              const f32x4 r4 = {rt[0], rt[1], rt[2], rt[3]}, a4 = {ad[0], ad[1], ad[2], ad[3]};
              if (dbg & 16) {   // tuning build: non-temporal output stores
                __builtin_nontemporal_store(r4, (f32x4 *)(returns + g));
                if (adv) __builtin_nontemporal_store(a4, (f32x4 *)(adv + g));
              } else {
                *(f32x4 *)(returns + g) = r4;
                if (adv) *(f32x4 *)(adv + g) = a4;
              }
            }
          }
        } else {
          // env-major output: a lane takes 4 consecutive t of one env -> one 16 B store
          constexpr int TQ = TT / 4;  // t-quads per env per tile; E * TQ == 256 * U
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int i = m + u * 256;
            const int e = i / TQ, tq = i % TQ;
            const int tl = 4 * tq, t = lo + tl;
            if (env0 + e < N && t < hi) {
              float rt[4], ad[4];
              const float4 g4 = *(const float4 *)(sd + e * TTP + tl);
              const float gq[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const float v = sv[(tl + c) * LD + e];
                rt[c] = __fadd_rn(gq[c], v);
                ad[c] = __fsub_rn(rt[c], v);
              }
              const int64_t g = (int64_t)(env0 + e) * T + t;
              if (t + 4 <= hi && (g & 3) == 0) {
                *(float4 *)(returns + g) = make_float4(rt[0], rt[1], rt[2], rt[3]);
                if (adv) *(float4 *)(adv + g) = make_float4(ad[0], ad[1], ad[2], ad[3]);
              } else {
                for (int c = 0; c < 4 && t + c < hi; ++c) {
                  returns[g + c] = rt[c];
                  if (adv) adv[g + c] = ad[c];
                }
              }
            }
          }
        }
      }
      // LDS traffic only is waited for: the output stores stay in flight across the barrier
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
    // ================================================================== chain wave
    float gae = 0.0f;
    __builtin_amdgcn_s_setprio(3);                   // the serial chain is the critical path: win issue arbitration
#ifdef TRL_TUNING   // dbg & 256: cycle stamps of the chain wave (tools/gae_ablate.py); they go to adv[4 * grp ..]
    long long st_scan = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_first = 0;
#endif
    __syncthreads();                                 // prologue barrier
    for (int s = 0; s < nstages; ++s) {
      const int jc = s - 1;
#ifdef TRL_TUNING
      const long long st_a = __builtin_amdgcn_s_memtime();
      if (s == 1) st_first = st_a;
#endif
      if (jc >= 0 && jc < ntiles && lane < E && !(dbg & 1)) {
        const int rows = t_hi(jc) - t_lo(jc);
        float *sd = s_d + (jc % NS) * E * TTP + lane * TTP;   // this env's row of the transposed tile
        const float *sm = s_raw + (jc % NR) * SLOT + 2 * PLANE + lane;   // MASKED: done[tl] at sm[tl * E]
        int top = rows;
        float *cp = s_cp + (jc % NS) * E * CPS + lane * CPS;   // CKPT: this env's checkpoints of the tile
        if (CKPT) cp[(rows + 7) >> 3] = gae;                    // the tile's carry-in, above its top group
        for (; top & 7; --top) {   // ragged head of a partial tile, one step at a time
          const float c = MASKED ? __fmul_rn(gl, __fsub_rn(1.0f, sm[(top - 1) * E])) : gl;
          gae = __fadd_rn(sd[top - 1], __fmul_rn(c, gae));
          if (!CKPT) sd[top - 1] = gae;
        }
        if (CKPT && (rows & 7)) cp[top >> 3] = gae;             // below the partial top group
        if (MASKED) {
          // per-step coefficient gamma_lambda * (1 - done): formed off the dependent chain from
          // the raw done plane, one group of 8 steps ahead
          auto fetch_m = [&](f32x4 (&d)[2], float (&c)[8], int g) {
            d[0] = *(const f32x4 *)(sd + g);
            d[1] = *(const f32x4 *)(sd + g + 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = sm[(g + i) * E];
          };
          auto run_m = [&](const f32x4 (&d)[2], const float (&m)[8], int g) {
            float c[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = __fmul_rn(gl, __fsub_rn(1.0f, m[i]));
            f32x4 lo4, hi4;
            hi4[3] = gae = __fadd_rn(d[1][3], __fmul_rn(c[7], gae));
            hi4[2] = gae = __fadd_rn(d[1][2], __fmul_rn(c[6], gae));
            hi4[1] = gae = __fadd_rn(d[1][1], __fmul_rn(c[5], gae));
            hi4[0] = gae = __fadd_rn(d[1][0], __fmul_rn(c[4], gae));
            lo4[3] = gae = __fadd_rn(d[0][3], __fmul_rn(c[3], gae));
            lo4[2] = gae = __fadd_rn(d[0][2], __fmul_rn(c[2], gae));
            lo4[1] = gae = __fadd_rn(d[0][1], __fmul_rn(c[1], gae));
            lo4[0] = gae = __fadd_rn(d[0][0], __fmul_rn(c[0], gae));
            *(f32x4 *)(sd + g + 4) = hi4;
            *(f32x4 *)(sd + g) = lo4;
          };
          f32x4 da[2], db[2];
          float ma[8], mb[8];
          int g = top - 8;                       // top is a multiple of 8 here
          if (g >= 0) fetch_m(da, ma, g);
          while (g >= 0) {
            fetch_m(db, mb, g >= 8 ? g - 8 : 0);   // clamped: the last prefetch is discarded
            run_m(da, ma, g);
            g -= 8;
            if (g < 0) break;
            fetch_m(da, ma, g >= 8 ? g - 8 : 0);
            run_m(db, mb, g);
            g -= 8;
          }
        } else {
        // Full groups of 8 steps, branch-free, two register sets in ping-pong: the next group's
        // deltas are fetched from LDS (2 x ds_read_b128) while the current group runs down the
        // dependent chain (2 VALU ops per step) and goes back as 2 x ds_write_b128.
        auto fetch = [&](f32x4 (&d)[2], int g) {
          d[0] = *(const f32x4 *)(sd + g);
          d[1] = *(const f32x4 *)(sd + g + 4);
        };
        auto run = [&](const f32x4 (&d)[2], int g) {
          f32x4 lo4, hi4;   // one IR vector store each, so they stay ds_write_b128
          hi4[3] = gae = __fadd_rn(d[1][3], __fmul_rn(gl, gae));
          hi4[2] = gae = __fadd_rn(d[1][2], __fmul_rn(gl, gae));
          hi4[1] = gae = __fadd_rn(d[1][1], __fmul_rn(gl, gae));
          hi4[0] = gae = __fadd_rn(d[1][0], __fmul_rn(gl, gae));
          lo4[3] = gae = __fadd_rn(d[0][3], __fmul_rn(gl, gae));
          lo4[2] = gae = __fadd_rn(d[0][2], __fmul_rn(gl, gae));
          lo4[1] = gae = __fadd_rn(d[0][1], __fmul_rn(gl, gae));
          lo4[0] = gae = __fadd_rn(d[0][0], __fmul_rn(gl, gae));
          if constexpr (CKPT) {
            cp[g >> 3] = gae;           // one value per group; the workers re-run the group's steps from the one above
          } else {
            *(f32x4 *)(sd + g + 4) = hi4;
            *(f32x4 *)(sd + g) = lo4;
          }
        };
        if (rows == TT) {
          // full tile: straight-line code, no loop bookkeeping on the chain's issue slots;
          // three register sets rotate so that each fetch is two groups ahead of its use
          constexpr int NG = TT / 8;
          f32x4 d[3][2];
          fetch(d[0], TT - 8);
          if (NG > 1) fetch(d[1], TT - 16);
#pragma unroll
          for (int k = 0; k < NG; ++k) {
            if (k + 2 < NG) fetch(d[(k + 2) % 3], TT - 8 * (k + 3));
            run(d[k % 3], TT - 8 * (k + 1));
          }
        } else {
          f32x4 da[2], db[2];
          int g = top - 8;                       // top is a multiple of 8 here
          if (g >= 0) fetch(da, g);
          while (g >= 0) {
            fetch(db, g >= 8 ? g - 8 : 0);       // clamped: the last prefetch is discarded
            run(da, g);
            g -= 8;
            if (g < 0) break;
            fetch(da, g >= 8 ? g - 8 : 0);
            run(db, g);
            g -= 8;
          }
        }
        }   // !MASKED
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef TRL_TUNING
      if (dbg & 256) st_scan += __builtin_amdgcn_s_memtime() - st_a;
#endif
      __builtin_amdgcn_s_barrier();
    }
#ifdef TRL_TUNING
    if ((dbg & 256) && lane == 0 && adv) {
      const long long st_end = __builtin_amdgcn_s_memtime();
      float *o = adv + 4 * grp;
      o[0] = (float)st_scan;                 // cycles inside the per-tile scan sections
      o[1] = (float)(st_end - st_t0);        // whole life of the wave after its prologue
      o[2] = (float)(st_first - st_t0);      // until the first tile could be scanned
      o[3] = (float)(st_end - st_first);
    }
#endif
  }
}

// 1: the checkpointed chain (k_gae_pipe<..., CKPT = true>) is what reference-mode launches run
#define TRL_GAE_CKPT_DEFAULT 1

struct PipeArgs {
  const float *reward, *value, *done, *bootstrap;
  int T, N;
  float gamma, gl;
  int out_nt;
  int64_t in_ts, out_ts;   // blocked layouts: rows allocated per env group (0: the plane is [T, N])
  float *returns, *adv;
  int dbg;
  bool ckpt;               // checkpointed chain (reference mode only)
};

template <int E, int PF, bool MASKED, int U = 2, bool CKPT = false>
int launch_pipe(const PipeArgs &a, hipStream_t st) {
  const int ngroups = (int)trl_cdiv(a.N, E);
  const int64_t in_gs = a.in_ts ? a.in_ts * E : E, out_gs = a.out_ts ? a.out_ts * E : E;
  const int in_rs = a.in_ts ? E : a.N, out_rs = a.out_ts ? E : a.N;
  hipLaunchKernelGGL((k_gae_pipe<E, PF, MASKED, U, CKPT>), dim3(ngroups), dim3(448), 0, st, a.reward, a.value, a.done,
                     a.bootstrap, a.T, a.N, a.gamma, a.gl, a.out_nt, ngroups, in_gs, in_rs, out_gs, out_rs, a.returns,
                     a.adv, a.dbg);
  TRL_CHECK_LAUNCH("trl_gae_reverse_scan");
  return TRL_OK;
}

// E (envs per workgroup), PF (tiles in flight) and U (tile rows = U * 256 * 4 / E) as run-time choices,
// 0 = the default heuristic.  Production calls pass zeros; the diagnostic build's sweep passes the rest.
int dispatch_pipe(int E, int PF, int U, const PipeArgs &a, hipStream_t st) {
  const bool masked = a.done != nullptr;
  if (E == 0) E = a.N > 256 * 32 ? 64 : (a.N > 256 * 16 ? 32 : 16);
  if (PF == 0) PF = 2;   // deeper prefetch measured no faster on [T, N], cold or cache-resident (tools/gae_sweep.py)
  // half-size tiles (64 rows at E = 16): twice the stages, half the pipeline fill and drain;
  // 36.6 -> 32.5 us cold at T = 2048, N = 4096
  if (U == 0) U = E == 16 ? 1 : 2;
  if (a.ckpt && !masked) {
#define TRL_CKPT_CASE(e, pf, u) \
  if (E == e && PF == pf && U == u) return launch_pipe<e, pf, false, u, true>(a, st);
    TRL_CKPT_CASE(16, 2, 1) TRL_CKPT_CASE(16, 3, 1) TRL_CKPT_CASE(8, 2, 1) TRL_CKPT_CASE(8, 3, 1)
    TRL_CKPT_CASE(32, 2, 2) TRL_CKPT_CASE(64, 2, 2)
#ifdef TRL_TUNING
    TRL_CKPT_CASE(16, 4, 1) TRL_CKPT_CASE(16, 2, 2) TRL_CKPT_CASE(16, 3, 2)
#endif
#undef TRL_CKPT_CASE
  }
#define TRL_PIPE_CASE(e, pf, m, u) \
  if (E == e && PF == pf && masked == m && U == u) return launch_pipe<e, pf, m, u>(a, st);
  // production set
  TRL_PIPE_CASE(16, 2, false, 1) TRL_PIPE_CASE(32, 2, false, 2) TRL_PIPE_CASE(64, 2, false, 2)
  TRL_PIPE_CASE(16, 2, true, 1) TRL_PIPE_CASE(32, 2, true, 2) TRL_PIPE_CASE(64, 2, true, 2)
  TRL_PIPE_CASE(16, 3, false, 1) TRL_PIPE_CASE(16, 4, false, 1)
  TRL_PIPE_CASE(8, 2, false, 1) TRL_PIPE_CASE(8, 3, false, 1) TRL_PIPE_CASE(8, 2, true, 1)
#ifdef TRL_TUNING
  TRL_PIPE_CASE(16, 2, false, 2) TRL_PIPE_CASE(16, 3, false, 2) TRL_PIPE_CASE(16, 4, false, 2)
  TRL_PIPE_CASE(16, 5, false, 1) TRL_PIPE_CASE(8, 4, false, 1)
  TRL_PIPE_CASE(32, 3, false, 2) TRL_PIPE_CASE(32, 4, false, 2) TRL_PIPE_CASE(64, 3, false, 2) TRL_PIPE_CASE(64, 4, false, 2)
  TRL_PIPE_CASE(16, 2, true, 2)
#endif
#undef TRL_PIPE_CASE
  trl_set_error("trl_gae: unsupported tuning E=%d PF=%d U=%d masked=%d", E, PF, U, (int)masked);
  return TRL_EINVAL;
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

bool al16(const void *p) { return (uintptr_t)p % 16 == 0; }

int check_common(const char *fn, const float *reward, const float *value, const float *done, const float *bootstrap,
                 int64_t T, int64_t N, int32_t mode, const float *returns) {
  TRL_REQUIRE(reward && value && bootstrap && returns, "%s: null pointer", fn);
  TRL_REQUIRE(mode == TRL_GAE_REFERENCE || mode == TRL_GAE_MASKED, "%s: bad mode %d", fn, mode);
  TRL_REQUIRE(mode != TRL_GAE_MASKED || done != nullptr, "%s: masked mode needs done", fn);
  TRL_REQUIRE(T >= 0 && N >= 0 && T < (1ll << 31) && N < (1ll << 31) && T * N < (1ll << 40),
              "%s: T=%lld N=%lld out of range", fn, (long long)T, (long long)N);
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
  int rc = check_common(fn, reward, value, done, bootstrap, T, N, mode, returns);
  if (rc) return rc;
  TRL_REQUIRE(out_layout == TRL_LAYOUT_TN || out_layout == TRL_LAYOUT_NT, "%s: bad out_layout %d", fn,
              out_layout);
  if (T == 0 || N == 0) return TRL_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool masked = mode == TRL_GAE_MASKED;
  const bool aligned = N % 4 == 0 && al16(reward) && al16(value) && al16(bootstrap) && al16(returns) &&
                       (adv == nullptr || al16(adv));
  if (aligned && (!masked || al16(done))) {
    // Envs per workgroup: 16 (a 64 B row segment) while that gives at most one workgroup per
    // CU (the pipelined kernel holds most of a CU's LDS), then wider segments.
    int E = 0, PF = 0, dbg = 0, U = 0, ckpt = TRL_GAE_CKPT_DEFAULT;
#ifdef TRL_TUNING   // diagnostic build only (tools/gae_sweep.py); the production library never reads the environment
    if (const char *tune = getenv("TRL_GAE_TUNE")) sscanf(tune, "%d,%d,%d,%d,%d", &E, &PF, &dbg, &U, &ckpt);
#endif
    const PipeArgs a{reward, value, masked ? done : nullptr, bootstrap, (int)T, (int)N, gamma, gamma_lambda,
                     out_layout == TRL_LAYOUT_NT, 0, 0, returns, adv, dbg, ckpt != 0};
    return dispatch_pipe(E, PF, U, a, st);
  }
  return launch<16>(masked, reward, value, done, bootstrap, (int)T, (int)N, gamma, gamma_lambda,
                    out_layout, returns, adv, st);
}

extern "C" int trl_gae_reverse_scan_blocked(const float *reward, const float *value, const float *done,
                                            const float *bootstrap, int64_t T, int64_t N, int32_t block_envs,
                                            int64_t in_t_stride, double gamma_d, double lmbda_d, int32_t mode,
                                            int32_t out_layout, int64_t out_t_stride, float *returns, float *adv,
                                            void *stream) {
  const char *fn = "trl_gae_reverse_scan_blocked";
  const float gamma = (float)gamma_d;
  const float gamma_lambda = (float)(gamma_d * lmbda_d);
  int rc = check_common(fn, reward, value, done, bootstrap, T, N, mode, returns);
  if (rc) return rc;
  TRL_REQUIRE(block_envs == 8 || block_envs == 16, "%s: block_envs=%d must be 8 or 16", fn, block_envs);
  TRL_REQUIRE(N % block_envs == 0, "%s: N=%lld must be a multiple of block_envs=%d", fn, (long long)N, block_envs);
  TRL_REQUIRE(out_layout == TRL_LAYOUT_BLOCKED || out_layout == TRL_LAYOUT_NT, "%s: bad out_layout %d", fn, out_layout);
  if (in_t_stride == 0) in_t_stride = T;
  if (out_t_stride == 0) out_t_stride = T;
  TRL_REQUIRE(in_t_stride >= T && in_t_stride < (1ll << 31), "%s: in_t_stride=%lld < T=%lld", fn,
              (long long)in_t_stride, (long long)T);
  TRL_REQUIRE(out_layout != TRL_LAYOUT_BLOCKED || (out_t_stride >= T && out_t_stride < (1ll << 31)),
              "%s: out_t_stride=%lld < T=%lld", fn, (long long)out_t_stride, (long long)T);
  TRL_REQUIRE(al16(reward) && al16(value) && al16(bootstrap) && al16(returns) && (adv == nullptr || al16(adv)) &&
                  (mode != TRL_GAE_MASKED || al16(done)),
              "%s: every plane must be 16-byte aligned", fn);
  if (T == 0 || N == 0) return TRL_OK;
  const bool masked = mode == TRL_GAE_MASKED;
  int PF = 0, dbg = 0, U = 0, ckpt = TRL_GAE_CKPT_DEFAULT;
#ifdef TRL_TUNING
  if (const char *tune = getenv("TRL_GAE_TUNE")) {
    int e_ignored = 0;
    sscanf(tune, "%d,%d,%d,%d,%d", &e_ignored, &PF, &dbg, &U, &ckpt);
  }
#endif
  if (U == 0) U = 1;
  // flat per-group streams: three tiles in flight, and the non-temporal policy on the once-read inputs and the
  // once-written outputs (tools/gae_sweep.py --blocked: 30.3 us plain -> 27.0 us, T = 2048, N = 4096, cold; on the
  // 64 B row segments of [T, N] neither changes anything, so that path keeps PF = 2 and the default policy)
  if (PF == 0) PF = masked ? 2 : 3;
#ifdef TRL_TUNING
  if (!getenv("TRL_GAE_TUNE")) dbg = 48;
#else
  dbg = 48;
#endif
  const PipeArgs a{reward, value, masked ? done : nullptr, bootstrap, (int)T, (int)N, gamma, gamma_lambda,
                   out_layout == TRL_LAYOUT_NT, in_t_stride, out_layout == TRL_LAYOUT_BLOCKED ? out_t_stride : 0,
                   returns, adv, dbg, ckpt != 0};
  return dispatch_pipe(block_envs, PF, U, a, (hipStream_t)stream);
}

#ifdef TRL_TUNING
// ---- diagnostic build only: the flat copy the scan is measured against (tools/copy_probe.hip's best setting).
// Both input planes to both output planes in ONE launch, fully coalesced, a persistent grid striding over the
// whole range, UN x 16 B in flight per lane, non-temporal: what a read-plus-write pass over the scan's bytes can
// reach on this part when it does nothing else.  bench.py times it over the same cold buffer sets.
namespace {
template <int UN>
__global__ __launch_bounds__(256) void k_flat_copy(const f32x4 *__restrict__ a, const f32x4 *__restrict__ b,
                                                   f32x4 *__restrict__ oa, f32x4 *__restrict__ ob, size_t nq) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += stride * UN) {
    f32x4 va[UN], vb[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const size_t j = i + u * stride;
      if (j < nq) {
        va[u] = __builtin_nontemporal_load(a + j);
        vb[u] = __builtin_nontemporal_load(b + j);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const size_t j = i + u * stride;
      if (j < nq) {
        __builtin_nontemporal_store(va[u], oa + j);
        __builtin_nontemporal_store(vb[u], ob + j);
      }
    }
  }
}
}  // namespace

extern "C" int trl_tune_flat_copy(const float *a, const float *b, float *oa, float *ob, int64_t n, int32_t grid,
                                  void *stream) {
  const char *fn = "trl_tune_flat_copy";
  TRL_REQUIRE(a && b && oa && ob && n > 0 && n % 4 == 0 && al16(a) && al16(b) && al16(oa) && al16(ob), "%s: bad arguments", fn);
  hipLaunchKernelGGL(k_flat_copy<4>, dim3(grid > 0 ? grid : 256), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)a,
                     (const f32x4 *)b, (f32x4 *)oa, (f32x4 *)ob, (size_t)(n / 4));
  TRL_CHECK_LAUNCH(fn);
  return TRL_OK;
}
#endif
