// Probe: does the instruction offset of global_load_lds_dwordx4 (saddr form) also shift the LDS
// destination?  Loads one 1 KB row with offset:1024 into LDS with M0 = 0 and dumps the first 4 KB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const char* g, unsigned* out) {
  extern __shared__ unsigned sm[];
  for (int i = threadIdx.x; i < 2048; i += 64) sm[i] = 0xdeadbeefu;
  __syncthreads();
  unsigned lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)sm;
  unsigned sl = __builtin_amdgcn_readfirstlane(lds);
  unsigned voff = threadIdx.x * 16;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024" :: "s"(sl), "v"(voff), "s"(g) : "memory", "m0");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += 64) out[i] = sm[i];
}
int main() {
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;
  char* g; unsigned* o;
  hipMalloc(&g, 16384); hipMalloc(&o, 8192);
  hipMemcpy(g, h.data(), 16384, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 8192, 0, g, o);
  std::vector<unsigned> r(2048);
  hipMemcpy(r.data(), o, 8192, hipMemcpyDeviceToHost);
  for (int i = 0; i < 2048; i += 128) printf("lds dword %4d: %08x %08x\n", i, r[i], r[i + 1]);
  return 0;
}
