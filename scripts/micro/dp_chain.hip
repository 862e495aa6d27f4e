// dp_chain.hip -- how long does a DEPENDENT f64 add / mul / fma take on gfx950, and how many independent chains hide it?
// One workgroup of WAVES waves; cycles from s_memtime (shader clock) around 4096 operations per chain.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/micro/dp_chain.hip -o scripts/micro/dp_chain
// Measured on MI355X (round 3): one chain 7.1 cycles per step for add, mul and fma alike; 2 chains 12.6 per step (6.3 per
// operation), 4 chains 18.1 (4.5 per operation), 8 chains 41.6 (5.2); a dependent add fed by an independent multiply 7.5 per
// pair; eight waves per workgroup (two per SIMD) each keep the one-wave figures: a SIMD issues f64 from two waves at the rate
// one wave reaches alone, so the sequential sums of the exact kernels cost ~7 cycles per term when nothing else is in the way.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int CHAINS, int OP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void chain_kernel(double seed, unsigned long long *cycles, double *sink) {
  double a[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) a[c] = seed + c + threadIdx.x;
  const double m = seed * 1.0000001;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (OP == 0) a[c] = __dadd_rn(a[c], m);
        if (OP == 1) a[c] = __dmul_rn(a[c], m);
        if (OP == 2) a[c] = __fma_rn(a[c], m, m);
        if (OP == 3) a[c] = __dadd_rn(a[c], __dmul_rn(m + u, m));  // one dependent add + one independent multiply
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += a[c];
  sink[threadIdx.x] = s;
  if (threadIdx.x == 0) *cycles = t1 - t0;
}

template <int CHAINS, int OP, int WAVES>
void run(const char *name) {
  unsigned long long *d, h = 0;
  double *sink;
  (void)hipMalloc(&d, 8);
  (void)hipMalloc(&sink, 8 * 64 * WAVES);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((chain_kernel<CHAINS, OP, WAVES>), dim3(1), dim3(64 * WAVES), 0, 0, 1.0 + rep, d, sink);
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  }
  const double per = (double)h / (256.0 * 16.0 * CHAINS);
  printf("%-26s chains %d waves/WG %d: %8llu cycles, %6.2f cycles per operation per wave, %6.2f per step of a chain\n", name, CHAINS,
         WAVES, h, per, per * CHAINS);
  (void)hipFree(d);
  (void)hipFree(sink);
}

int main() {
  run<1, 0, 1>("v_add_f64 dependent");
  run<2, 0, 1>("v_add_f64 dependent");
  run<4, 0, 1>("v_add_f64 dependent");
  run<8, 0, 1>("v_add_f64 dependent");
  run<1, 1, 1>("v_mul_f64 dependent");
  run<4, 1, 1>("v_mul_f64 dependent");
  run<1, 2, 1>("v_fma_f64 dependent");
  run<4, 2, 1>("v_fma_f64 dependent");
  run<8, 2, 1>("v_fma_f64 dependent");
  run<1, 3, 1>("add(dep) + mul(indep)");
  run<4, 3, 1>("add(dep) + mul(indep)");
  run<1, 0, 4>("v_add_f64 dependent");
  run<1, 0, 8>("v_add_f64 dependent");
  run<4, 0, 8>("v_add_f64 dependent");
  run<8, 2, 8>("v_fma_f64 dependent");
  return 0;
}
