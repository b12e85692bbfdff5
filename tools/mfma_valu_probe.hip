// Does v_mfma_f32_32x32x2_f32 (exact f32, 64 cycles per SIMD) leave the SIMD's vector issue free for VALU work, as the
// bf16 MFMAs do?  One wave per SIMD: NV independent v_fma_f32 (or NL ds_read_b128) behind each MFMA, cycles per MFMA
// from s_memtime.  Two waves per SIMD (512 threads): one wave issues only MFMAs, its partner only VALU.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_valu_probe tools/mfma_valu_probe.hip && /tmp/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int NV, int NL>
__global__ __launch_bounds__(256) void k_one(float *out, unsigned long long *cyc, int iters, float seed) {
  __shared__ __align__(16) float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * i;
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = seed + j;
  float a = seed + threadIdx.x, b = seed * 2.f;
  float4 l[4] = {};
  const float *lp = lds + (threadIdx.x & 63) * 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] = MFMA(a, b, acc[j]);
#pragma unroll
      for (int k = 0; k < NV; ++k) v[(j * NV + k) & 15] = __builtin_fmaf(v[(j * NV + k) & 15], a, b);
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f t = *(const volatile v4f *)(lp + 256 * ((j * NL + k) & 7));
        l[k & 3].x += t.x;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  for (int j = 0; j < 4; ++j) s += l[j].x;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// 512 threads: waves 0-3 MFMA only, waves 4-7 VALU only (NV fma per "slot" of the partner's MFMA)
template <int NV>
__global__ __launch_bounds__(512) void k_two(float *out, unsigned long long *cyc, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = seed + j;
  float a = seed + threadIdx.x, b = seed * 2.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = MFMA(a, b, acc[j]);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < NV; ++k) v[(j * NV + k) & 15] = __builtin_fmaf(v[(j * NV + k) & 15], a, b);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// 512 threads: waves 0-3 stream MFMAs (or idle, ALONE); the partner runs 256 v_fma per round as NC independent
// dependency chains: what a dependent VALU instruction costs beside the partner's MFMA stream
template <int NC, bool ALONE>
__global__ __launch_bounds__(512) void k_chain(float *out, unsigned long long *cyc, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = seed + j;
  float a = seed + threadIdx.x, b = seed * 2.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (!ALONE)
      for (int it = 0; it < 24 * iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = MFMA(a, b, acc[j]);
      }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 256; ++k) v[k % NC] = __builtin_fmaf(v[k % NC], a, b);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// the same beside a stream of v_mfma_f32_16x16x4_f32 (8 passes, 32 cycles, the same FLOP rate): does the partner get
// its VALU slots per MFMA or per cycle?
template <int NC>
__global__ __launch_bounds__(512) void k_chain16(float *out, unsigned long long *cyc, int iters, float seed) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const int wave = threadIdx.x >> 6;
  f4v acc[4];
  for (int j = 0; j < 4; ++j) acc[j] = f4v{0.f, 0.f, 0.f, 0.f};
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = seed + j;
  float a = seed + threadIdx.x, b = seed * 2.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    for (int it = 0; it < 48 * iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 256; ++k) v[k % NC] = __builtin_fmaf(v[k % NC], a, b);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  for (int j = 0; j < 16; ++j) s += v[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// 512 threads: waves 0-3 stream MFMAs; the partner runs NP (compare, select) pairs per slot -- v_cmp writes VCC, a
// scalar register pair -- or, MODE 1, the same selection by integer arithmetic (no scalar register involved)
template <int NP, int MODE>
__global__ __launch_bounds__(512) void k_cmp(float *out, unsigned long long *cyc, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float v[16], h[16];
  for (int j = 0; j < 16; ++j) { v[j] = seed + j; h[j] = seed * (j - 8) + threadIdx.x * 1e-3f; }
  float a = seed + threadIdx.x, b = seed * 2.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if (MODE < 2)
      for (int it = 0; it < 8 * iters; ++it) {      // (long enough that the partner always runs beside the stream)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = MFMA(a, b, acc[j]);
      }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const int q = (j * NP + k) & 15;
          if ((MODE & 1) == 0) {
            v[q] = h[q] > 0.f ? v[q] + a : 0.f;
          } else {
            const int m = (0 - __float_as_int(h[q])) >> 31;
            v[q] = __int_as_float(__float_as_int(v[q] + a) & m);
          }
          asm volatile("" : "+v"(h[q]));
        }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int j = 0; j < 16; ++j) s += v[j] + h[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// 512 threads: waves 0-3 (or 4-7 when SWAP) stream MFMAs, the partner waves run NW ds_write_b32 + NR ds_read_b128 per
// slot (the reads are consumed): does LDS traffic of the partner proceed beside the MFMA stream?
template <int NWR, int NRD, bool SWAP>
__global__ __launch_bounds__(512) void k_lds(float *out, unsigned long long *cyc, int iters, float seed) {
  __shared__ __align__(16) float lds[8 * 2304];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float *mine = lds + wave * 2304;
  for (int i = lane; i < 2304; i += 64) mine[i] = seed * i;
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = seed + threadIdx.x, b = seed * 2.f, v = seed;
  const bool mf = SWAP ? wave >= 4 : wave < 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mf) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = MFMA(a, b, acc[j]);
    }
  } else {
    typedef float v4f __attribute__((ext_vector_type(4)));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < NWR; ++k) mine[k * 36 + (lane & 31) + 144 * (lane >> 5)] = v + k;
#pragma unroll
      for (int k = 0; k < NRD; ++k) {
        const v4f t = *(const v4f *)(mine + (lane & 31) * 36 + 4 * (lane >> 5) + 8 * (k & 3));
        v += t.x;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = v;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// 512 threads, both waves of a SIMD issue MFMAs.  Waves 0-3: phases of MA dependent MFMAs (two accumulators) and
// VA v_fma; waves 4-7: phases of MB independent MFMAs (four accumulators) and VB v_fma.  Work is fixed; the wall
// time of the pair tells whether one wave's MFMAs fill the other's VALU phases.
template <int MA, int VA, int MB, int VB, int PRIO>
__global__ __launch_bounds__(512) void k_mix(float *out, unsigned long long *cyc, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float v[16];
  for (int j = 0; j < 16; ++j) v[j] = seed + j;
  float a = seed + threadIdx.x, b = seed * 2.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    for (int it = 0; it < iters; ++it) {
      if (PRIO) __builtin_amdgcn_s_setprio(PRIO);      // matrix phase: this wave's (rare) issues go first
#pragma unroll
      for (int j = 0; j < MA; ++j) acc[j & 1] = MFMA(a, b, acc[j & 1]);
      if (PRIO) __builtin_amdgcn_s_setprio(0);         // vector phase: yield to the partner's MFMA issues
#pragma unroll
      for (int k = 0; k < VA; ++k) v[k & 15] = __builtin_fmaf(v[k & 15], a, b);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
#pragma unroll
      for (int j = 0; j < MB; ++j) acc[j & 3] = MFMA(a, b, acc[j & 3]);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
      for (int k = 0; k < VB; ++k) v[k & 15] = __builtin_fmaf(v[k & 15], a, b);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int j = 0; j < 16; ++j) s += v[j];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// which SIMD does wave w of a 512-thread workgroup run on?  (HW_ID bits 5:4)
__global__ __launch_bounds__(512) void k_simd(unsigned *out) {
  const unsigned id = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NV, int NL>
void run_one(float *out, unsigned long long *cyc, unsigned long long *h, int iters) {
  hipLaunchKernelGGL((k_one<NV, NL>), dim3(256), dim3(256), 0, 0, out, cyc, iters, 1e-9f);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k_one<NV, NL>), dim3(256), dim3(256), 0, 0, out, cyc, iters, 1e-9f);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, cyc, 1024 * 8, hipMemcpyDeviceToHost));
  double s = 0;
  for (int i = 0; i < 1024; ++i) s += (double)h[i];
  printf("one wave/SIMD: %2d v_fma + %d ds_read_b128 (each waited for) per MFMA: %7.1f cycles per MFMA\n", NV, NL, s / 1024 / iters / 4);
}

template <int NV>
void run_two(float *out, unsigned long long *cyc, unsigned long long *h, int iters) {
  hipLaunchKernelGGL((k_two<NV>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1e-9f);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k_two<NV>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1e-9f);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, cyc, 2048 * 8, hipMemcpyDeviceToHost));
  double sm = 0, sv = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) (w < 4 ? sm : sv) += (double)h[b * 8 + w];
  printf("two waves/SIMD, equal rounds (the partner's average includes its solo tail after the MFMA wave has finished; see the chain rows): MFMA wave %7.1f cycles per MFMA; partner with %2d v_fma per slot: %7.1f cycles per slot (%.1f per v_fma)\n",
         sm / 1024 / iters / 4, NV, sv / 1024 / iters / 4, sv / 1024 / iters / 4 / NV);
}

template <int NC, bool ALONE>
void run_chain(float *out, unsigned long long *cyc, unsigned long long *h, int iters) {
  hipLaunchKernelGGL((k_chain<NC, ALONE>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1e-9f);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k_chain<NC, ALONE>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1e-9f);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, cyc, 2048 * 8, hipMemcpyDeviceToHost));
  double sv = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 4; w < 8; ++w) sv += (double)h[b * 8 + w];
  printf("chain: %s: v_fma in %2d independent chains: %6.2f cycles per instruction\n", ALONE ? "alone                " : "beside an MFMA stream",
         NC, sv / 1024 / iters / 256);
}

template <int NC>
void run_chain16(float *out, unsigned long long *cyc, unsigned long long *h, int iters) {
  hipLaunchKernelGGL((k_chain16<NC>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1e-9f);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k_chain16<NC>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1e-9f);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, cyc, 2048 * 8, hipMemcpyDeviceToHost));
  double sv = 0, sm = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) (w < 4 ? sm : sv) += (double)h[b * 8 + w];
  printf("chain: beside a 16x16x4 MFMA stream (%5.1f cycles per MFMA): v_fma in %2d independent chains: %6.2f cycles per instruction\n",
         sm / 1024 / (48.0 * iters) / 4, NC, sv / 1024 / iters / 256);
}

template <int NP, int MODE>
void run_cmp(float *out, unsigned long long *cyc, unsigned long long *h, int iters) {
  hipLaunchKernelGGL((k_cmp<NP, MODE>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1e-9f);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k_cmp<NP, MODE>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1e-9f);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, cyc, 2048 * 8, hipMemcpyDeviceToHost));
  double sm = 0, sv = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) (w < 4 ? sm : sv) += (double)h[b * 8 + w];
  printf("cmp: %s; partner with %2d %s: %7.1f cycles per group of %d\n", MODE < 2 ? "beside an MFMA stream" : "alone                ",
         NP, (MODE & 1) ? "(add, sub, ashr, and)" : "(add, v_cmp, v_cndmask)", sv / 1024 / iters / 4, NP);
}

template <int NWR, int NRD, bool SWAP>
void run_lds(float *out, unsigned long long *cyc, unsigned long long *h, int iters) {
  hipLaunchKernelGGL((k_lds<NWR, NRD, SWAP>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1e-9f);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k_lds<NWR, NRD, SWAP>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1e-9f);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, cyc, 2048 * 8, hipMemcpyDeviceToHost));
  double sm = 0, sv = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) ((SWAP ? w >= 4 : w < 4) ? sm : sv) += (double)h[b * 8 + w];
  printf("lds (MFMA stream in waves %s): MFMA wave %7.1f cycles per MFMA; partner with %2d ds_write_b32 + %2d ds_read_b128 per slot: %7.1f cycles per slot\n",
         SWAP ? "4-7" : "0-3", sm / 1024 / iters / 4, NWR, NRD, sv / 1024 / iters / 4);
}

template <int MA, int VA, int MB, int VB, int PRIO>
void run_mix(float *out, unsigned long long *cyc, unsigned long long *h, int iters) {
  hipLaunchKernelGGL((k_mix<MA, VA, MB, VB, PRIO>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1e-9f);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k_mix<MA, VA, MB, VB, PRIO>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1e-9f);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, cyc, 2048 * 8, hipMemcpyDeviceToHost));
  double sa = 0, sb = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) (w < 4 ? sa : sb) += (double)h[b * 8 + w];
  sa /= 1024.0 * iters; sb /= 1024.0 * iters;
  printf("mix (setprio %d in matrix phases): A %3d MFMA + %3d v_fma per round: %8.0f cycles; B %3d MFMA + %3d v_fma: %8.0f cycles; MFMA pipe need %d, A alone %d, B alone %d\n",
         PRIO, MA, VA, sa, MB, VB, sb, 64 * (MA + MB), 64 * MA + 4 * VA, 64 * MB + 4 * VB);
}

int main() {
  float *out;
  unsigned long long *cyc, *h = (unsigned long long *)malloc(2048 * 8);
  CK(hipMalloc(&out, 256 * 512 * 4));
  CK(hipMalloc(&cyc, 2048 * 8));
  const int iters = 2000;
  run_one<0, 0>(out, cyc, h, iters);
  run_one<2, 0>(out, cyc, h, iters);
  run_one<4, 0>(out, cyc, h, iters);
  run_one<8, 0>(out, cyc, h, iters);
  run_one<12, 0>(out, cyc, h, iters);
  run_one<16, 0>(out, cyc, h, iters);
  run_one<24, 0>(out, cyc, h, iters);
  run_one<0, 1>(out, cyc, h, iters);
  run_one<0, 2>(out, cyc, h, iters);
  run_one<0, 4>(out, cyc, h, iters);
  run_one<4, 2>(out, cyc, h, iters);
  run_two<4>(out, cyc, h, iters);
  run_two<8>(out, cyc, h, iters);
  run_two<16>(out, cyc, h, iters);
  run_two<32>(out, cyc, h, iters);
  run_chain<1, true>(out, cyc, h, 200);
  run_chain<1, false>(out, cyc, h, 200);
  run_chain<2, true>(out, cyc, h, 200);
  run_chain<2, false>(out, cyc, h, 200);
  run_chain<4, false>(out, cyc, h, 200);
  run_chain<8, false>(out, cyc, h, 200);
  run_chain<16, true>(out, cyc, h, 200);
  run_chain<16, false>(out, cyc, h, 200);
  run_chain16<1>(out, cyc, h, 200);
  run_chain16<16>(out, cyc, h, 200);
  run_cmp<16, 0>(out, cyc, h, 1000);
  run_cmp<16, 2>(out, cyc, h, 1000);
  run_cmp<16, 1>(out, cyc, h, 1000);
  run_cmp<16, 3>(out, cyc, h, 1000);
  run_lds<8, 0, false>(out, cyc, h, 2000);
  run_lds<8, 0, true>(out, cyc, h, 2000);
  run_lds<0, 4, false>(out, cyc, h, 2000);
  run_lds<0, 4, true>(out, cyc, h, 2000);
  run_lds<8, 2, false>(out, cyc, h, 2000);
  run_lds<8, 2, true>(out, cyc, h, 2000);
  run_mix<64, 0, 64, 0, 0>(out, cyc, h, 200);
  run_mix<64, 0, 64, 0, 2>(out, cyc, h, 200);
  run_mix<64, 1000, 64, 0, 0>(out, cyc, h, 200);
  run_mix<64, 1000, 64, 0, 2>(out, cyc, h, 200);
  run_mix<64, 1000, 64, 1000, 0>(out, cyc, h, 200);
  run_mix<64, 1000, 64, 1000, 2>(out, cyc, h, 200);
  run_mix<68, 600, 128, 400, 0>(out, cyc, h, 200);
  run_mix<68, 600, 128, 400, 2>(out, cyc, h, 200);
  run_mix<132, 700, 64, 300, 0>(out, cyc, h, 200);
  run_mix<132, 700, 64, 300, 2>(out, cyc, h, 200);
  run_mix<16, 250, 16, 250, 0>(out, cyc, h, 400);
  run_mix<16, 250, 16, 250, 2>(out, cyc, h, 400);
  {
    unsigned *d, hsim[64];
    CK(hipMalloc(&d, 64 * 4));
    hipLaunchKernelGGL(k_simd, dim3(8), dim3(512), 0, 0, d);
    CK(hipMemcpy(hsim, d, 64 * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < 8; ++b) {
      printf("workgroup %d: SIMD of waves 0..7:", b);
      for (int w = 0; w < 8; ++w) printf(" %u", hsim[b * 8 + w]);
      printf("\n");
    }
  }
  return 0;
}
