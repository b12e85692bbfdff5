// Diagnostic probe (not part of the library): what would a 64 -> 64 layer of the whole-step kernel cost per 32-sample
// strip if the exact-f32 product v_mfma_f32_32x32x2_f32 (64 cycles, 4096 FLOP) were replaced by the three-way bf16
// split of both operands -- x = hi + mid + lo, each part 8 mantissa bits, exactly -- and the six largest of the nine
// partial products on v_mfma_f32_32x32x16_bf16 (f32 accumulation; numerically this is f32-equivalent: max error
// 1.2e-7 of sum|a||b| over K = 64 against 2.2e-7 for plain f32, numpy emulation in DESIGN.md section 8).
// The weights' parts would be split once per launch; the activations come out of the accumulators and must be split
// on the VALU for every strip, which is the open cost this probe measures.
//
//   hipcc -O3 --offload-arch=gfx950 -o bf16x6_probe tools/bf16x6_probe.hip && ./bf16x6_probe
//
// Modes: 0 exact f32 (64 MFMAs 32x32x2 per layer-strip, weights from LDS);  1 bf16x6 (48 MFMAs 32x32x16 + split);
//        2 bf16x6 without the split (MFMAs + LDS reads only);  3 the split alone.
// One wave per SIMD (256 lanes per workgroup, one workgroup per CU) and two (two workgroups per CU).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);     \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

// the high halves of two floats as one dword of two bf16 (truncation): lo 16 bits from a, hi 16 bits from b
__device__ __forceinline__ unsigned pack_hi(float a, float b) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ float trunc16(float x) { return __uint_as_float(__float_as_uint(x) & 0xFFFF0000u); }

// eight f32 -> three bf16x8 operands (hi, mid, lo), exactly: x = hi + mid + lo
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 (&part)[3]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = x[2 * p], b = x[2 * p + 1];
    part[0][p] = pack_hi(a, b);
    const f32x2 r = f32x2{a, b} - f32x2{trunc16(a), trunc16(b)};
    part[1][p] = pack_hi(r[0], r[1]);
    const f32x2 l = r - f32x2{trunc16(r[0]), trunc16(r[1])};
    part[2][p] = pack_hi(l[0], l[1]);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_probe(float *__restrict__ out, const float *__restrict__ w, int R) {
  // f32: [blk 2][kq 8][lane 64][4 floats]  = 16 KB;  bf16: [part 3][blk, kb 8][lane 64][16 B] = 24 KB
  __shared__ u32x4 sw[1536];
  for (int i = threadIdx.x; i < 1536; i += 256) sw[i] = ((const u32x4 *)w)[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 acc[2];
  float x[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = 0.01f * (float)((lane + i) & 15);
  int off = lane;
  for (int it = 0; it < R; ++it) {
    asm volatile("" : "+v"(off));      // the weights are re-read from LDS for every strip, as in the real kernel
    acc[0] = f32x16{0};
    acc[1] = f32x16{0};
    if constexpr (MODE == 0) {
      u32x4 wq[17];      // the next group's weights are read while this group's MFMAs run
      wq[0] = sw[off];
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int blk = g >> 3, kq = g & 7;
        if (g + 1 < 16) wq[g + 1] = sw[(g + 1) * 64 + off];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(wq[g][j]), x[4 * kq + j], acc[blk], 0, 0, 0);
      }
    } else {
      u32x4 xp[4][3];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        if constexpr (MODE == 2) {      // no split: the raw registers stand in for the parts
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) xp[kb][p][j] = __float_as_uint(x[8 * kb + 2 * j + (p & 1)]);
        } else {
          const float x8[8] = {x[8 * kb], x[8 * kb + 1], x[8 * kb + 2], x[8 * kb + 3],
                               x[8 * kb + 4], x[8 * kb + 5], x[8 * kb + 6], x[8 * kb + 7]};
          split8(x8, xp[kb]);
        }
      }
      if constexpr (MODE == 3) {        // the split alone: fold the parts into the accumulators on the VALU
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[kb & 1][4 * p + j] += __uint_as_float(xp[kb][p][j]);
      } else {
        u32x4 wp[9][3];      // the next group's weights are read while this group's MFMAs run
#pragma unroll
        for (int p = 0; p < 3; ++p) wp[0][p] = sw[p * 512 + off];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          const int blk = g >> 2, kb = g & 3;
          if (g + 1 < 8) {
#pragma unroll
            for (int p = 0; p < 3; ++p) wp[g + 1][p] = sw[(p * 8 + g + 1) * 64 + off];
          }
          // the six largest partial products, smallest first: (w part, x part)
          constexpr int T[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
#pragma unroll
          for (int t = 0; t < 6; ++t)
            acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wp[g][T[t][0]]),
                                                               __builtin_bit_cast(bf16x8, xp[kb][T[t][1]]), acc[blk],
                                                               0, 0, 0);
        }
      }
    }
    // bias + ReLU, and the accumulators are the next strip's activations (kept bounded)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      x[i] = fminf(fmaxf(acc[0][i] + 0.5f, 0.0f), 2.0f);
      x[16 + i] = fminf(fmaxf(acc[1][i] + 0.5f, 0.0f), 2.0f);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += x[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
double run(int grid, int R, float *out, const float *w) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, out, w, 16);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a, 0));
  hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, out, w, R);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, a, b));
  CHECK(hipEventDestroy(a));
  CHECK(hipEventDestroy(b));
  return (double)ms * 1e6 / R;      // ns per layer-strip per wave (all waves run side by side)
}

int main() {
  int dev = 0, cus = 0, mhz = 0;
  CHECK(hipGetDevice(&dev));
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  CHECK(hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, dev));
  float *out, *w;
  CHECK(hipMalloc(&out, (size_t)2 * cus * 256 * sizeof(float)));
  std::vector<float> hw(6144);
  for (int i = 0; i < 6144; ++i) hw[i] = 0.001f * (float)((i * 37) % 61 - 30);
  CHECK(hipMalloc(&w, 6144 * sizeof(float)));
  CHECK(hipMemcpy(w, hw.data(), 6144 * sizeof(float), hipMemcpyHostToDevice));
  const int R = 20000;
  const char *names[4] = {"exact f32: 64 x mfma_32x32x2_f32", "bf16x6: split + 48 x mfma_32x32x16_bf16",
                          "bf16x6 without the split", "the split alone (VALU)"};
  printf("%d CUs, %d MHz; one 64 -> 64 layer for a 32-sample strip (262,144 FLOP), ns per layer-strip per wave\n", cus, mhz / 1000);
  for (int wps = 1; wps <= 2; ++wps) {
    const int grid = cus * wps;
    double t[4] = {run<0>(grid, R, out, w), run<1>(grid, R, out, w), run<2>(grid, R, out, w), run<3>(grid, R, out, w)};
    for (int m = 0; m < 4; ++m)
      printf("  %d wave(s) per SIMD  %-42s %8.1f ns  (%6.0f cycles at %d MHz; chip: %6.1f TFLOP/s f32-equivalent)\n", wps,
             names[m], t[m], t[m] * mhz / 1e6, mhz / 1000, m == 3 ? 0.0 : 262144.0 * grid * 4 / t[m] / 1e3);
  }
  CHECK(hipFree(out));
  CHECK(hipFree(w));
  return 0;
}
