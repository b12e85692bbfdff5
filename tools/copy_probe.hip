// Diagnostic (not part of libtrl_hip.so): what a read-plus-write pass over the GAE kernel's
// bytes (two [T, N] f32 planes in, two out, T = 2048, N = 4096 -> 134 MB) can reach on this part
// when it does nothing else, for several access geometries.  The launches of one variant are
// captured into a HIP graph and rotate over enough buffer sets that every launch finds its
// inputs cold in HBM, exactly as bench.py times k_gae_pipe.
//
//   hipcc -O3 --offload-arch=gfx950 tools/copy_probe.hip -o tools/bin/copy_probe && tools/bin/copy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if (e != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// flat: fully coalesced grid-stride float4 copy of both planes in one launch, UN loads in flight per lane
template <int UN, bool NT>
__global__ __launch_bounds__(256) void k_flat(const f4 *__restrict__ a, const f4 *__restrict__ b, f4 *__restrict__ oa,
                                              f4 *__restrict__ ob, size_t nq) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nq; i += stride * UN) {
    f4 va[UN], vb[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const size_t j = i + u * stride;
      if (j < nq) {
        va[u] = NT ? __builtin_nontemporal_load(a + j) : a[j];
        vb[u] = NT ? __builtin_nontemporal_load(b + j) : b[j];
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const size_t j = i + u * stride;
      if (j < nq) {
        if (NT) {
          __builtin_nontemporal_store(va[u], oa + j);
          __builtin_nontemporal_store(vb[u], ob + j);
        } else {
          oa[j] = va[u];
          ob[j] = vb[u];
        }
      }
    }
  }
}

// column: the GAE kernel's geometry.  A workgroup owns E consecutive envs (a 4*E-byte segment of every
// row) and walks the rows from the last to the first, RB rows per step per pass of its 256 lanes, UN
// passes in flight.  No LDS, no chain: what the geometry alone costs.
template <int E, int UN>
__global__ __launch_bounds__(256) void k_column(const float *__restrict__ a, const float *__restrict__ b,
                                                float *__restrict__ oa, float *__restrict__ ob, int T, int N) {
  constexpr int QPR = E / 4, RPP = 256 / QPR;
  // XCD-aware: blocks b and b + 8 run on the same XCD; give them neighbouring segments
  const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = (nb >= 8 && nb % 8 == 0) ? xcd * (nb / 8) + slot : blockIdx.x;
  const int q = threadIdx.x % QPR, r = threadIdx.x / QPR;
  const size_t col = (size_t)grp * E + 4 * q;
  for (int t0 = T - RPP * UN; t0 >= 0; t0 -= RPP * UN) {
    f4 va[UN], vb[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const size_t off = (size_t)(t0 + u * RPP + r) * N + col;
      va[u] = *(const f4 *)(a + off);
      vb[u] = *(const f4 *)(b + off);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const size_t off = (size_t)(t0 + u * RPP + r) * N + col;
      *(f4 *)(oa + off) = va[u];
      *(f4 *)(ob + off) = vb[u];
    }
  }
}

// column geometry + a flat-geometry prefetch wave per workgroup: wave 4 streams this workgroup's 1/256 share
// of every 16-row block (1 KiB contiguous per plane) into a dead register, running ahead of the four
// copy waves, so that their 64 B-segment reads find the lines already in the memory-side cache.
// MODE 0: plain loads, 1: nt loads.  AHEAD: the prefetcher is throttled to at most this many loads in flight.
template <int E, int UN, int MODE>
__global__ __launch_bounds__(320) void k_column_pf(const float *__restrict__ a, const float *__restrict__ b,
                                                   float *__restrict__ oa, float *__restrict__ ob, int T, int N) {
  constexpr int QPR = E / 4, RPP = 256 / QPR;
  const int nb = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = (nb >= 8 && nb % 8 == 0) ? xcd * (nb / 8) + slot : blockIdx.x;
  __shared__ __attribute__((aligned(16))) float dead[256];
  if (threadIdx.x >= 256) {   // prefetch wave: rows from the last block down, 16 rows per step
    const int lane = threadIdx.x - 256;
    const int rows_per_blk = (nb * 1024) / (N * 4);          // rows covered by nb x 1 KiB
    const size_t seg = (size_t)blockIdx.x * 256 + lane * 4;  // float offset inside a block
    for (int t0 = T - rows_per_blk; t0 >= 0; t0 -= rows_per_blk) {
      const size_t off = (size_t)t0 * N + seg;
      // LDS-DMA: the data lands in a dead 1 KiB of LDS, no register is written behind the compiler's back
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a + off),
                                       (__attribute__((address_space(3))) void *)dead, 16, 0, MODE == 1 ? 2 : 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(b + off),
                                       (__attribute__((address_space(3))) void *)dead, 16, 0, MODE == 1 ? 2 : 0);
      asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const int q = threadIdx.x % QPR, r = threadIdx.x / QPR;
  const size_t col = (size_t)grp * E + 4 * q;
  for (int t0 = T - RPP * UN; t0 >= 0; t0 -= RPP * UN) {
    f4 va[UN], vb[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const size_t off = (size_t)(t0 + u * RPP + r) * N + col;
      va[u] = *(const f4 *)(a + off);
      vb[u] = *(const f4 *)(b + off);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const size_t off = (size_t)(t0 + u * RPP + r) * N + col;
      *(f4 *)(oa + off) = va[u];
      *(f4 *)(ob + off) = vb[u];
    }
  }
}

struct Set {
  float *a, *b, *oa, *ob;
};

template <class L>
float graph_us(hipStream_t st, int nsets, int reps, L launch) {
  for (int s = 0; s < nsets; ++s) launch(s);
  CK(hipStreamSynchronize(st));
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int r = 0; r < reps; ++r)
    for (int s = 0; s < nsets; ++s) launch(s);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  return ms * 1e3f / (reps * nsets);
}

int main() {
  const int T = 2048, N = 4096;
  const size_t n = (size_t)T * N, bytes = n * 4;
  const int nsets = 6, reps = 8;
  std::vector<Set> sets(nsets);
  for (auto &s : sets) {
    CK(hipMalloc(&s.a, bytes));
    CK(hipMalloc(&s.b, bytes));
    CK(hipMalloc(&s.oa, bytes));
    CK(hipMalloc(&s.ob, bytes));
    CK(hipMemset(s.a, 1, bytes));
    CK(hipMemset(s.b, 2, bytes));
  }
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const double algo = 16.0 * n;
  auto report = [&](const char *name, float us) {
    printf("%-44s %7.2f us  %6.0f GB/s  %.3f of 8 TB/s\n", name, us, algo / us / 1e3, algo / us / 1e3 / 8000.0);
    fflush(stdout);
  };
  const size_t nq = n / 4;
  char nm[128];
#define FLAT(UN, NT, G)                                                                                          \
  {                                                                                                              \
    float us = graph_us(st, nsets, reps, [&](int s) {                                                            \
      hipLaunchKernelGGL((k_flat<UN, NT>), dim3(G), dim3(256), 0, st, (const f4 *)sets[s].a, (const f4 *)sets[s].b, \
                         (f4 *)sets[s].oa, (f4 *)sets[s].ob, nq);                                                \
    });                                                                                                          \
    snprintf(nm, sizeof nm, "flat un=%d nt=%d grid=%d", UN, (int)NT, G);                                         \
    report(nm, us);                                                                                              \
  }
  for (int G : {256, 512, 8192}) {
    FLAT(1, false, G)
    FLAT(2, false, G)
    FLAT(4, false, G)
    FLAT(4, true, G)
  }
  {  // two launches, one per plane (what two torch copy_ calls do)
    float us = graph_us(st, nsets, reps, [&](int s) {
      CK(hipMemcpyAsync(sets[s].oa, sets[s].a, bytes, hipMemcpyDeviceToDevice, st));
      CK(hipMemcpyAsync(sets[s].ob, sets[s].b, bytes, hipMemcpyDeviceToDevice, st));
    });
    report("2 x hipMemcpyAsync D2D", us);
  }
#define COL(E, UN)                                                                                                   \
  {                                                                                                                  \
    float us = graph_us(st, nsets, reps, [&](int s) {                                                                \
      hipLaunchKernelGGL((k_column<E, UN>), dim3(N / E), dim3(256), 0, st, sets[s].a, sets[s].b, sets[s].oa, sets[s].ob, \
                         T, N);                                                                                      \
    });                                                                                                              \
    snprintf(nm, sizeof nm, "column E=%d (%d WGs) un=%d", E, N / E, UN);                                             \
    report(nm, us);                                                                                                  \
  }
  COL(16, 1) COL(16, 2) COL(16, 4) COL(16, 8)
  COL(32, 1) COL(32, 2) COL(32, 4) COL(32, 8)
#define COLPF(E, UN, MODE)                                                                                           \
  {                                                                                                                  \
    float us = graph_us(st, nsets, reps, [&](int s) {                                                                \
      hipLaunchKernelGGL((k_column_pf<E, UN, MODE>), dim3(N / E), dim3(320), 0, st, sets[s].a, sets[s].b, sets[s].oa, \
                         sets[s].ob, T, N);                                                                          \
    });                                                                                                              \
    snprintf(nm, sizeof nm, "column+prefetch E=%d un=%d mode=%d", E, UN, MODE);                                      \
    report(nm, us);                                                                                                  \
  }
  COLPF(16, 1, 0) COLPF(16, 2, 0) COLPF(16, 4, 0) COLPF(16, 1, 1) COLPF(16, 2, 1) COLPF(16, 4, 1)
  for (auto &s : sets) {
    CK(hipFree(s.a));
    CK(hipFree(s.b));
    CK(hipFree(s.oa));
    CK(hipFree(s.ob));
  }
  return 0;
}
