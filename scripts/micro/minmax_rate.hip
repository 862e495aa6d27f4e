// minmax_rate.hip -- issue cost of the compare-exchange of a sorting network on gfx950: f64 min/max against u32 / u64-as-two-u32
// keys, one wave and two waves per SIMD.  16 independent compare-exchanges per layer (as in a 32-input Batcher network).
// build: hipcc -O3 --offload-arch=gfx950 scripts/micro/minmax_rate.hip -o scripts/micro/minmax_rate
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T>
__device__ __forceinline__ void ce(T &a, T &b) {
  const T lo = a < b ? a : b, hi = a < b ? b : a;
  a = lo;
  b = hi;
}
__device__ __forceinline__ void ce(double &a, double &b) {
  const double lo = __builtin_fmin(a, b), hi = __builtin_fmax(a, b);
  a = lo;
  b = hi;
}

template <typename T, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void ce_kernel(unsigned seed, unsigned long long *cycles, T *sink) {
  T v[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) v[k] = (T)((threadIdx.x * 2654435761u + k * 40503u + seed) >> 3);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int j = 16; j >= 1; j >>= 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int a = (i / j) * 2 * j + (i % j);
        ce(v[a], v[a + j]);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  T s = 0;
#pragma unroll
  for (int k = 0; k < 32; ++k) s += v[k];
  sink[threadIdx.x] = s;
  if (threadIdx.x == 0) *cycles = t1 - t0;
}

template <typename T, int WAVES>
void run(const char *name) {
  unsigned long long *d, h = 0;
  T *sink;
  (void)hipMalloc(&d, 8);
  (void)hipMalloc(&sink, sizeof(T) * 64 * WAVES);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((ce_kernel<T, WAVES>), dim3(1), dim3(64 * WAVES), 0, 0, 1u + rep, d, sink);
    (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  }
  printf("%-8s waves/WG %d: %8llu cycles = %6.2f per compare-exchange per wave\n", name, WAVES, h, (double)h / (64.0 * 5 * 16));
  (void)hipFree(d);
  (void)hipFree(sink);
}

int main() {
  run<double, 1>("f64");
  run<double, 4>("f64");
  run<double, 8>("f64");
  run<unsigned, 1>("u32");
  run<unsigned, 4>("u32");
  run<unsigned, 8>("u32");
  run<float, 1>("f32");
  run<float, 8>("f32");
  run<unsigned long long, 1>("u64");
  run<unsigned long long, 8>("u64");
  return 0;
}
