// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on MI355X (one wave per SIMD, independent
// accumulators, operands in registers).  Build: hipcc -O3 --offload-arch=gfx950 mfma_f64_peak.hip -o mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256, 1) void k(double *out, int iters, double a0, double b0) {
  v4f64 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks, int waves_per_block) {
  double *out;
  hipMalloc(&out, sizeof(double) * 256 * 4096);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, iters, 1.0, 0.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 2048.0 * NACC * iters * blocks * waves_per_block;
    double per_simd_cycles = (ms * 1e-3) * 2.4e9 / ((double)NACC * iters * ((blocks * waves_per_block + 1023) / 1024));
    printf("NACC=%d blocks=%d waves/blk=%d: %.2f ms  %.2f TFLOP/s  (%.1f cycles@2.4GHz per MFMA per SIMD)\n", NACC,
           blocks, waves_per_block, ms, flops / (ms * 1e-3) / 1e12, per_simd_cycles);
  }
  hipFree(out);
}

int main() {
  run<1>(256, 4);
  run<4>(256, 4);
  run<16>(256, 4);
  run<16>(512, 4);   // 2 waves per SIMD
  run<16>(1, 1);     // one wave alone: issue interval at idle clocks
  return 0;
}
