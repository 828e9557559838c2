// mfma_valu_kinds.hip -- stand-alone: which instruction kinds take time from the fp32 matrix pipe?
// mfma_valu_mix.hip showed v_fma_f32 beside v_mfma_f32_32x32x2_f32 does not add FLOPs: each one costs the MFMA stream
// 2.4-3.8 cycles.  Here the filler is an integer add, a move, a 64-bit add, a select, a multiply, a transcendental, a
// scalar add, an LDS read: NV of them between two MFMAs, 4 waves per SIMD.  Reported: cycles per MFMA (64 = free).
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define KERNEL(NAME, ASM, ...)                                                                              \
  template <int NV>                                                                                          \
  __global__ __launch_bounds__(256) void NAME(float* out, int iters, float a0, float b0) {                   \
    __shared__ float lds[1024];                                                                              \
    lds[threadIdx.x] = a0;                                                                                   \
    __syncthreads();                                                                                         \
    f32x16 acc;                                                                                              \
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;                                                               \
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;                                        \
    unsigned u[4] = {threadIdx.x, threadIdx.x + 1, threadIdx.x + 2, threadIdx.x + 3};                        \
    unsigned long long w[4] = {threadIdx.x, threadIdx.x + 1ull, threadIdx.x + 2ull, threadIdx.x + 3ull};     \
    float f[4] = {a, b, a + b, a - b};                                                                       \
    unsigned la = (threadIdx.x & 63) * 16;                                                                   \
    f32x4 q[4] = {};                                                                                             \
    int sc = iters;                                                                                          \
    for (int it = 0; it < iters; ++it) {                                                                     \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                       \
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));                  \
        _Pragma("unroll") for (int j = 0; j < NV; ++j) { ASM; }                                              \
      }                                                                                                      \
    }                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)");                                                                    \
    float s = 0.f;                                                                                           \
    for (int e = 0; e < 16; ++e) s += acc[e];                                                                \
    for (int j = 0; j < 4; ++j) s += u[j] + (float)w[j] + f[j] + q[j].x + q[j].y + q[j].z + q[j].w;           \
    s += sc;                                                                                                 \
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;                                              \
  }

KERNEL(k_none, asm volatile(""))
KERNEL(k_add_u32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[j & 3]) : "v"(u[(j + 1) & 3])))
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1" : "=v"(u[j & 3]) : "v"(la)))
KERNEL(k_add_u64, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(w[j & 3]) : "v"(w[(j + 1) & 3])))
KERNEL(k_cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[j & 3]) : "v"(la)))
KERNEL(k_mul_f32, asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[j & 3]) : "v"(b)))
KERNEL(k_max_f32, asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[j & 3]) : "v"(b)))
KERNEL(k_mul_lo, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[j & 3]) : "v"(la)))
KERNEL(k_exp, asm volatile("v_exp_f32 %0, %0" : "+v"(f[j & 3])))
KERNEL(k_salu, asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc)))
KERNEL(k_dsread, asm volatile("ds_read_b128 %0, %1" : "=v"(q[j & 3]) : "v"(la)))
KERNEL(k_dswrite, asm volatile("ds_write_b128 %0, %1" : : "v"(la), "v"(q[j & 3])))

template <class K>
void run(const char* name, K kern, int nv, float* d) {
  const int grid = 256 * 4, iters = 1500;
  hipEvent_t s, e;
  (void)hipEventCreate(&s);
  (void)hipEventCreate(&e);
  kern<<<grid, 256>>>(d, 200, 1.f, 2.f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(s);
  kern<<<grid, 256>>>(d, iters, 1.f, 2.f);
  (void)hipEventRecord(e);
  (void)hipEventSynchronize(e);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, s, e);
  const double n_mfma = (double)grid * 4 * iters * 16;
  const double tf = n_mfma * 4096.0 / ms / 1e9;
  // cycles per MFMA per SIMD at the clock that gives 64 for the bare loop (155.5 TFLOP/s)
  std::printf("%-10s x %2d : MFMA %6.1f TFLOP/s  -> %5.1f cyc/MFMA, %4.2f cyc per filler\n", name, nv, tf, 64.0 * 155.5 / tf,
              nv ? (64.0 * 155.5 / tf - 64.0) / nv : 0.0);
}

#define RUN3(K) run(#K, K<2>, 2, d); run(#K, K<4>, 4, d); run(#K, K<8>, 8, d);

int main() {
  float* d;
  (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run("k_none", k_none<0>, 0, d);
  RUN3(k_add_u32)
  RUN3(k_mov)
  RUN3(k_add_u64)
  RUN3(k_cndmask)
  RUN3(k_mul_f32)
  RUN3(k_max_f32)
  RUN3(k_mul_lo)
  RUN3(k_exp)
  RUN3(k_salu)
  RUN3(k_dsread)
  RUN3(k_dswrite)
  return 0;
}
