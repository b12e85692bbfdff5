// Cost of a device-wide barrier inside one launch (all workgroups resident: grid <= CUs), against the ~7 us a
// dependent kernel boundary costs on this stack: a sense barrier on two words of device memory, agent-scope
// release / acquire around it (L2 write-back + invalidate across the 8 XCDs), bounded spins.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/grid_barrier_probe tools/grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#ifdef RELAXED
#define ORD_REL __ATOMIC_RELAXED
#define ORD_ACQ __ATOMIC_RELAXED
#define ORD_AR __ATOMIC_RELAXED
#else
#define ORD_REL __ATOMIC_RELEASE
#define ORD_ACQ __ATOMIC_ACQUIRE
#define ORD_AR __ATOMIC_ACQ_REL
#endif
#ifndef SLEEP
#define SLEEP 2
#endif

// bar layout (32-dword = 128 B lines): line 0 root count, line 1 generation, lines 2 .. 2 + NG - 1 group counts
constexpr int NG = 16;
__device__ __forceinline__ void grid_barrier(unsigned *bar, unsigned &gen, unsigned nwg) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned want = gen + 1;
    const unsigned g = blockIdx.x % NG;
    const unsigned gsize = nwg / NG + (g < nwg % NG ? 1 : 0);
    const unsigned ngroups = nwg < NG ? nwg : NG;
    bool last = false;
    if (__hip_atomic_fetch_add(bar + 32 * (2 + g), 1u, ORD_REL, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1) {
      __hip_atomic_store(bar + 32 * (2 + g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = __hip_atomic_fetch_add(bar, 1u, ORD_AR, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1;
    }
    if (last) {
      __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(bar + 32, want, ORD_REL, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      for (int spin = 0; spin < (1 << 22); ++spin) {
        if ((int)(__hip_atomic_load(bar + 32, ORD_ACQ, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) break;
        __builtin_amdgcn_s_sleep(SLEEP);
      }
    }
    gen = want;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_probe(unsigned *bar, float *buf, int nbar, int work) {
  __shared__ unsigned gen_s;
  unsigned gen = 0;
  if (threadIdx.x == 0) gen = __hip_atomic_load(bar + 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int n = gridDim.x * 256;
  const int me = blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  for (int b = 0; b < nbar; ++b) {
    // a phase's worth of dependent traffic: read what another workgroup wrote in the previous phase
    if (work) {
      const int src = (me + 256 * 37 * (b + 1)) % n;
#ifdef RELAXED
      // the exchanged data itself goes through agent-scope accesses (sc1: coherent across the XCDs' L2s), so the
      // barrier needs no cache write-back / invalidate, only the stores' completion
      acc += __hip_atomic_load(buf + (b & 1) * n + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(buf + ((b + 1) & 1) * n + me, acc + 1.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0) lgkmcnt(0) expcnt(0): the store has left the CU
#else
      acc += buf[(b & 1) * n + src];
      buf[((b + 1) & 1) * n + me] = acc + 1.f;
#endif
    }
    grid_barrier(bar, gen, gridDim.x);
  }
  if (work && acc == -1.f) buf[0] = acc;
  (void)gen_s;
}

int main() {
  unsigned *bar;
  float *buf;
  hipMalloc(&bar, 128 * 32);
  hipMemset(bar, 0, 128 * 32);
  hipMalloc(&buf, 2 * 256 * 256 * 4);
  hipMemset(buf, 0, 2 * 256 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int nwg : {64, 128, 256})
    for (int work : {0, 1})
      for (int nbar : {1, 11, 101}) {
        hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(256), 0, 0, bar, buf, nbar, work);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_probe, dim3(nwg), dim3(256), 0, 0, bar, buf, nbar, work);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("wgs %3d work %d barriers %3d: %8.2f us per launch\n", nwg, work, nbar, ms * 1000.f / 20);
      }
  // correctness of the exchange: after nbar phases every element is nbar-deep
  hipMemset(buf, 0, 2 * 256 * 256 * 4);
  hipLaunchKernelGGL(k_probe, dim3(256), dim3(256), 0, 0, bar, buf, 10, 1);
  float h[4];
  hipMemcpy(h, buf, 16, hipMemcpyDeviceToHost);
  printf("after 10 phases buf[0..3] = %g %g %g %g (expect a value that only depends on the phase count: all equal)\n", h[0], h[1], h[2], h[3]);
  return 0;
}
