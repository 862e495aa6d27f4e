// What the FIRST kernel that touches freshly hipMalloc'ed device memory pays (the 30 - 100 ms "stalls" of a build that needs a new
// 250 MB block for its CSR: bench.py, value_hard_data.allocator_inside_each -- hipMalloc itself returns in 0.3 ms).
// build: hipcc --offload-arch=gfx950 -O2 scripts/micro/first_touch.hip -o scripts/micro/first_touch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void fill(uint4 *p, size_t n16, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(v, v, v, v);
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; hipStreamCreate(&s);
  void *warm; hipMalloc(&warm, 1 << 20); hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, s, (uint4 *)warm, (size_t)(1 << 16), 1u); hipStreamSynchronize(s);
  for (size_t mb : {16, 64, 256, 1024, 256, 256}) {
    const size_t bytes = mb << 20;
    void *p = nullptr;
    double t0 = now();
    hipMalloc(&p, bytes);
    double t1 = now();
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, s, (uint4 *)p, bytes / 16, 2u); hipStreamSynchronize(s);
    double t2 = now();
    hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, s, (uint4 *)p, bytes / 16, 3u); hipStreamSynchronize(s);
    double t3 = now();
    hipFree(p);
    double t4 = now();
    printf("%5zu MB: hipMalloc %7.3f ms  first fill %8.3f ms  second fill %7.3f ms  hipFree %7.3f ms\n", mb, t1 - t0, t2 - t1, t3 - t2, t4 - t3);
  }
  return 0;
}
