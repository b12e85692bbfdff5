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
  printf("one wave/SIMD: %2d v_fma + %d ds_read_b128 per MFMA: %7.1f cycles per MFMA\n", NV, NL, s / 1024 / iters / 4);
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
  printf("two waves/SIMD: MFMA wave %7.1f cycles per MFMA; partner with %2d v_fma per slot: %7.1f cycles per slot (%.1f per v_fma)\n",
         sm / 1024 / iters / 4, NV, sv / 1024 / iters / 4, sv / 1024 / iters / 4 / NV);
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
