// mfma16_check.hip -- stand-alone: the lane layout of v_mfma_f32_16x16x32_bf16 as conv_fast.hip assumes it
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {  // A (16,32), B (16 n,32 k) row-major; D (16,16)
  const int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)A[(l & 15) * 32 + 8 * (l >> 4) + j];
    b[j] = (__bf16)B[(l & 15) * 32 + 8 * (l >> 4) + j];
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
int main() {
  float hA[512], hB[512], hD[256], *dA, *dB, *dD;
  for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double s = 0; for (int kk = 0; kk < 32; ++kk) s += (double)hA[i * 32 + kk] * hB[j * 32 + kk];
    worst = fmax(worst, fabs(s - hD[i * 16 + j]));
  }
  printf("mfma_f32_16x16x32_bf16 layout check: max |err| = %g (expect 0)\n", worst);
  return 0;
}
