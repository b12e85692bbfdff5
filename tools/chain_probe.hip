// Diagnostic (not part of libtrl_hip.so): what the GAE scan's dependent chain
//   gae = delta + c * gae        (v_mul_f32 then v_add_f32, one rounding each, no FMA)
// costs per step on one wavefront, with and without the LDS traffic that feeds it, and what a second
// busy wave on the same SIMD does to it.  One wave of interest per workgroup, one workgroup per CU,
// T = 2048 steps in tiles of 64 like k_gae_pipe's chain wave.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/chain_probe.hip -o tools/bin/chain_probe && tools/bin/chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TT = 64, TTP = TT + 4, E = 16;

// MODE bits: 1 = LDS reads, 2 = LDS writes, 4 = all 64 lanes active (else 16), 8 = a busy VALU wave on the same
// SIMD (wave 4 of the workgroup), 16 = the same busy wave but on another SIMD (wave 1), 32 = interleave the LDS
// instructions between the VALU pairs (sched_group_barrier) instead of where the compiler puts them
template <int MODE>
__global__ __launch_bounds__(320) void k_chain(float *out, long long *cycles, int ntiles, float gl, const float *seed) {
  __shared__ __attribute__((aligned(16))) float sd[3 * E * 4 * TTP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 3 * E * 4 * TTP; i += blockDim.x) sd[i] = 1e-3f * (float)(i % 97);
  __syncthreads();
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    const int nl = (MODE & 4) ? 64 : 16;
    if (lane < nl) {
      float gae = 0.f;
      // run-time values, so that the LDS-free modes cannot be folded away
      const f32x4 k0 = *(const f32x4 *)(seed + 8 * lane), k1 = *(const f32x4 *)(seed + 8 * lane + 4);
      const long long c0 = __builtin_amdgcn_s_memtime();
      for (int tile = 0; tile < ntiles; ++tile) {
        float *row = sd + (tile % 3) * (E * 4 * TTP) + lane * TTP;   // a ring of three tiles, like k_gae_pipe's scan ring
        if (MODE & 256) {
          // reads as k_gae_pipe does (two groups ahead), results exported one step per ds_write_addtid_b32:
          // no address register, one data register, LDS address = M0 + offset + 4 * lane, i.e. a [t][env] tile
          const unsigned m0 = (unsigned)((tile % 3) * (E * 4 * TTP) * 4);
          f32x4 d[3][2];
          auto fetch = [&](f32x4(&x)[2], int g) {
            x[0] = *(const f32x4 *)(row + g);
            x[1] = *(const f32x4 *)(row + g + 4);
          };
#define ADDTID(val, t) asm volatile("s_mov_b32 m0, %1\n\tds_write_addtid_b32 %0 offset:%2" ::"v"(val), "s"(m0), "n"((t) * 64) : "memory", "m0")
          auto run = [&](const f32x4(&x)[2], int g) {
            gae = __fadd_rn(x[1][3], __fmul_rn(gl, gae)); ADDTID(gae, 7);
            gae = __fadd_rn(x[1][2], __fmul_rn(gl, gae)); ADDTID(gae, 6);
            gae = __fadd_rn(x[1][1], __fmul_rn(gl, gae)); ADDTID(gae, 5);
            gae = __fadd_rn(x[1][0], __fmul_rn(gl, gae)); ADDTID(gae, 4);
            gae = __fadd_rn(x[0][3], __fmul_rn(gl, gae)); ADDTID(gae, 3);
            gae = __fadd_rn(x[0][2], __fmul_rn(gl, gae)); ADDTID(gae, 2);
            gae = __fadd_rn(x[0][1], __fmul_rn(gl, gae)); ADDTID(gae, 1);
            gae = __fadd_rn(x[0][0], __fmul_rn(gl, gae)); ADDTID(gae, 0);
          };
          constexpr int NG = TT / 8;
          fetch(d[0], TT - 8);
          fetch(d[1], TT - 16);
#pragma unroll
          for (int k = 0; k < NG; ++k) {
            if (k + 2 < NG) fetch(d[(k + 2) % 3], TT - 8 * (k + 3));
            run(d[k % 3], TT - 8 * (k + 1));
          }
          continue;
        }
        if (MODE & 64) {
          // the whole tile's deltas first (16 x ds_read_b128 = 64 registers), then the chain with only the
          // write-backs between its groups (or after it, MODE & 128)
          f32x4 x[TT / 4], y[TT / 4];
#pragma unroll
          for (int i = TT / 4 - 1; i >= 0; --i) x[i] = *(const f32x4 *)(row + 4 * i);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = TT / 4 - 1; i >= 0; --i) {
            y[i][3] = gae = __fadd_rn(x[i][3], __fmul_rn(gl, gae));
            y[i][2] = gae = __fadd_rn(x[i][2], __fmul_rn(gl, gae));
            y[i][1] = gae = __fadd_rn(x[i][1], __fmul_rn(gl, gae));
            y[i][0] = gae = __fadd_rn(x[i][0], __fmul_rn(gl, gae));
            if (!(MODE & 128)) *(f32x4 *)(row + 4 * i) = y[i];
          }
          if (MODE & 128) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = TT / 4 - 1; i >= 0; --i) *(f32x4 *)(row + 4 * i) = y[i];
          }
          __builtin_amdgcn_sched_barrier(0);
          continue;
        }
        f32x4 d[3][2];
        auto fetch = [&](f32x4(&x)[2], int g) {
          if (MODE & 1) {
            x[0] = *(const f32x4 *)(row + g);
            x[1] = *(const f32x4 *)(row + g + 4);
          } else {
            x[0] = k0;
            x[1] = k1;
          }
        };
        auto run = [&](const f32x4(&x)[2], int g) {
          f32x4 lo4, hi4;
          hi4[3] = gae = __fadd_rn(x[1][3], __fmul_rn(gl, gae));
          hi4[2] = gae = __fadd_rn(x[1][2], __fmul_rn(gl, gae));
          if (MODE & 32) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0), __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          hi4[1] = gae = __fadd_rn(x[1][1], __fmul_rn(gl, gae));
          hi4[0] = gae = __fadd_rn(x[1][0], __fmul_rn(gl, gae));
          if (MODE & 32) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0), __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          lo4[3] = gae = __fadd_rn(x[0][3], __fmul_rn(gl, gae));
          lo4[2] = gae = __fadd_rn(x[0][2], __fmul_rn(gl, gae));
          if (MODE & 32) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0), __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          lo4[1] = gae = __fadd_rn(x[0][1], __fmul_rn(gl, gae));
          lo4[0] = gae = __fadd_rn(x[0][0], __fmul_rn(gl, gae));
          if (MODE & 32) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0), __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          if (MODE & 2) {
            *(f32x4 *)(row + g + 4) = hi4;
            *(f32x4 *)(row + g) = lo4;
          }
        };
        constexpr int NG = TT / 8;
        fetch(d[0], TT - 8);
        fetch(d[1], TT - 16);
#pragma unroll
        for (int k = 0; k < NG; ++k) {
          if (k + 2 < NG) fetch(d[(k + 2) % 3], TT - 8 * (k + 3));
          run(d[k % 3], TT - 8 * (k + 1));
        }
        if (!(MODE & 2)) row[0] = gae;   // keep the chain alive
      }
      out[blockIdx.x * 64 + lane] = gae;
      if (lane == 0) cycles[blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
    }
  } else if (((MODE & 8) && wave == 4) || ((MODE & 16) && wave == 1)) {
    // a busy VALU wave (what a worker wave of the real kernel is to the chain wave)
    float a = (float)lane, b = 1.0001f;
    for (int i = 0; i < ntiles * 64 / 4; ++i) {   // comparable in length to the chain wave's work
      a = a * b + 0.5f;
      b = b * 0.99999f + 1e-6f;
    }
    if (a == 123.456f) out[0] = a + b;
  }
}

template <int MODE>
void run(const char *name, float *out, int ntiles) {
  static long long *cyc = nullptr;
  static float *seed = nullptr;
  if (!cyc) {
    CK(hipMalloc(&cyc, 256 * sizeof(long long)));
    CK(hipMalloc(&seed, 64 * 8 * 4));
    float h[64 * 8];
    for (int i = 0; i < 64 * 8; ++i) h[i] = 0.01f * (float)(i % 37);
    CK(hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice));
  }
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  const int reps = 20;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_chain<MODE>, dim3(256), dim3(320), 0, 0, out, cyc, ntiles, 0.9405f, seed);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_chain<MODE>, dim3(256), dim3(320), 0, 0, out, cyc, ntiles, 0.9405f, seed);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms * 1e3 / reps;
  long long h[256];
  CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  for (int i = 0; i < 256; ++i)
    for (int j = i + 1; j < 256; ++j)
      if (h[j] < h[i]) { long long t = h[i]; h[i] = h[j]; h[j] = t; }
  printf("%-64s %7.2f us/launch   chain wave: %7lld cycles (median) = %5.2f cycles/step\n", name, us, h[128],
         (double)h[128] / (ntiles * 64));
}

int main() {
  float *out;
  CK(hipMalloc(&out, 256 * 64 * 4));
  const int nt = 32;   // 2048 steps
  run<0>("VALU chain only, 16 lanes", out, nt);
  run<4>("VALU chain only, 64 lanes", out, nt);
  run<1>("+ LDS reads, 16 lanes", out, nt);
  run<2>("+ LDS writes, 16 lanes", out, nt);
  run<3>("+ LDS reads and writes, 16 lanes (k_gae_pipe's chain)", out, nt);
  run<7>("+ LDS reads and writes, 64 lanes", out, nt);
  run<3 + 32>("+ LDS reads and writes interleaved by sched_group_barrier", out, nt);
  run<3 + 256>("reads as now, results by ds_write_addtid_b32 (one per step), 16 lanes", out, nt);
  run<3 + 256 + 4>("reads as now, results by ds_write_addtid_b32, 64 lanes", out, nt);
  run<3 + 64>("all reads of a tile first, writes between groups, 16 lanes", out, nt);
  run<3 + 64 + 128>("all reads first, all writes last, 16 lanes", out, nt);
  run<3 + 64 + 4>("all reads of a tile first, writes between groups, 64 lanes", out, nt);
  run<3 + 64 + 8>("all reads first + busy VALU wave on the SAME SIMD", out, nt);
  run<3 + 8>("chain + LDS, busy VALU wave on the SAME SIMD", out, nt);
  run<3 + 16>("chain + LDS, busy VALU wave on ANOTHER SIMD", out, nt);
  run<0 + 8>("VALU chain only, busy VALU wave on the SAME SIMD", out, nt);
  run<3 + 8 + 32>("interleaved + busy wave on the same SIMD", out, nt);
  // launch floor
  run<0>("(2 tiles: launch floor)", out, 2);
  return 0;
}
