// Is r = x - top16(x) computed as v_dot2c_f32_bf16((hi0, hi1), (-1, 0), x0) exact?  (One instruction per element where the
// mask + packed subtraction of the bf16x6 operand split take 1.5.)  Compares both remainders bit for bit over random,
// tiny and huge inputs:  hipcc --offload-arch=gfx950 -O3 -o tools/dot2_split_probe tools/dot2_split_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *x, float *ra, float *rb, int n, const unsigned int *mul) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * t + 1 >= n) return;
  const float a = x[2 * t], b = x[2 * t + 1];
  const unsigned int p = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
  const bf2 h = __builtin_bit_cast(bf2, p);
  const bf2 m0 = __builtin_bit_cast(bf2, mul[0]), m1 = __builtin_bit_cast(bf2, mul[1]);      // (-1, 0), (0, -1): from memory, not folded into an inline constant
  ra[2 * t] = a - __uint_as_float(__float_as_uint(a) & 0xFFFF0000u);
  ra[2 * t + 1] = b - __uint_as_float(__float_as_uint(b) & 0xFFFF0000u);
  rb[2 * t] = __builtin_amdgcn_fdot2_f32_bf16(h, m0, a, false);
  rb[2 * t + 1] = __builtin_amdgcn_fdot2_f32_bf16(h, m1, b, false);
}
int main() {
  const int n = 1 << 22;
  std::vector<float> x(n);
  srand(1);
  for (int i = 0; i < n; ++i) {
    unsigned int u = ((unsigned)rand() << 16) ^ (unsigned)rand();
    const int kind = i & 3;
    if (kind == 1) u = (u & 0x807FFFFFu) | ((unsigned)(1 + rand() % 30) << 23);       // tiny exponents
    if (kind == 2) u = (u & 0x807FFFFFu) | ((unsigned)(224 + rand() % 30) << 23);     // huge exponents
    if (kind == 3) u = (u & 0x807FFFFFu) | ((unsigned)(100 + rand() % 60) << 23);     // the usual range
    if ((u & 0x7F800000u) == 0x7F800000u) u &= 0xBFFFFFFFu;                          // no inf / nan
    memcpy(&x[i], &u, 4);
  }
  float *dx, *da, *db;
  hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  unsigned int hm[2] = {0x0000bf80u, 0xbf800000u}, *dm;
  (void)hipMalloc(&dm, 8); (void)hipMemcpy(dm, hm, 8, hipMemcpyHostToDevice);
  k<<<n / 2 / 256, 256>>>(dx, da, db, n, dm);
  std::vector<float> a(n), b(n);
  hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
  long bad = 0, bad_normal = 0;
  for (int i = 0; i < n; ++i) {
    if (memcmp(&a[i], &b[i], 4)) {
      ++bad;
      unsigned int ua; memcpy(&ua, &a[i], 4);
      if ((ua & 0x7F800000u) != 0) ++bad_normal;
      if (bad <= 5) printf("x=%a  sub=%a  dot2=%a\n", x[i], a[i], b[i]);
    }
  }
  printf("%ld of %d remainders differ (%ld of them with a normal-range exact remainder)\n", bad, n, bad_normal);
  return 0;
}
