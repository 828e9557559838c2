#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ void k(const float* A, const float* B, float* D) {  // A (32,32), B (32 n,32 k); D (32,32)
  const int lane = threadIdx.x;
  const int l15 = lane & 15, q15 = l15 >> 2;
  const int piq = q15 == 1 ? 3 : (q15 == 3 ? 1 : q15);
  const int rho15 = 4 * piq + (l15 & 3);
  const int lh = lane >> 5;
  f32x4 acc4[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
      a[e] = (__bf16)A[(16 * i + rho15) * 32 + 8 * (lane >> 4) + e];
      b[e] = (__bf16)B[(16 * j + rho15) * 32 + 8 * (lane >> 4) + e];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  f32x16 acc;
  const int qa = lh, qb = 2 + lh;
  const int pqa = qa == 1 ? 3 : (qa == 3 ? 1 : qa), pqb = qb == 1 ? 3 : (qb == 3 ? 1 : qb);
  const int addr_a = (rho15 + 16 * pqa) * 4, addr_b = (rho15 + 16 * pqb) * 4;
  const bool right = (lane >> 4) & 1;
  __shared__ float sc[32 * 33];
  if (MODE == 2) {
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 4; ++e) {
      const int mrow = (lane >> 4) * 4 + e, mcol = lane & 15;  // MFMA indices
      const int qr = mrow >> 2, qc = mcol >> 2;
      const int trow = 16 * i + 4 * (qr == 1 ? 3 : (qr == 3 ? 1 : qr)) + (mrow & 3);
      const int tcol = 16 * j + 4 * (qc == 1 ? 3 : (qc == 3 ? 1 : qc)) + (mcol & 3);
      sc[trow * 33 + tcol] = acc4[i][j][e];
    }
    __syncthreads();
    for (int r = 0; r < 16; ++r) acc[r] = sc[((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + (lane & 31)];
  } else {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int bi = r >> 3;
    const int addr = ((r >> 2) & 1) ? addr_b : addr_a;
    float x0 = acc4[bi][0][r & 3], x1 = acc4[bi][1][r & 3];
    if (MODE == 0) {
      const int v0 = __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, x0));
      const int v1 = __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, x1));
      acc[r] = __builtin_bit_cast(float, right ? v1 : v0);
    } else {
      const float v0 = __shfl(x0, addr >> 2, 64), v1 = __shfl(x1, addr >> 2, 64);
      acc[r] = right ? v1 : v0;
    }
  }
  }
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + (lane & 31)] = acc[r];
}
int main() {
  float hA[1024], hB[1024], hD[1024], *dA, *dB, *dD;
  for (int i = 0; i < 1024; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
  hipMalloc(&dA, 4096); hipMalloc(&dB, 4096); hipMalloc(&dD, 4096);
  hipMemcpy(dA, hA, 4096, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 4096, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 3; ++mode) {
  if (mode == 0) k<0><<<1, 64>>>(dA, dB, dD); else if (mode == 1) k<1><<<1, 64>>>(dA, dB, dD); else k<2><<<1, 64>>>(dA, dB, dD);
  hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
  double worst = 0; int bad = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double s = 0; for (int kk = 0; kk < 32; ++kk) s += (double)hA[i * 32 + kk] * hB[j * 32 + kk];
    if (fabs(s - hD[i * 32 + j]) > 1e-3) { if (bad < 3) printf("bad (%d,%d): got %g want %g\n", i, j, hD[i*32+j], s); ++bad; }
    worst = fmax(worst, fabs(s - hD[i * 32 + j]));
  }
  printf("mode %d (0 bpermute, 1 shfl, 2 LDS): max |err| = %g, bad %d\n", mode, worst, bad);
  }
  return 0;
}
