// i8_block_test_rate.hip -- what bounds the block loop of the int8 first pass (mfma_prefilter_i8.hip, one k step) on gfx950:
// two v_mfma_i32_32x32x32_i8 per 64 x 32 block against the 16 v_max3_i32 + compare + branch of its coarse test.  One workgroup
// on one CU, 2 or 4 waves per SIMD, no memory traffic (mode 4 adds the block's ds_read_b128).  Cycles are s_memtime ticks.
//   mode 0: the MFMAs alone           mode 1: the test alone (on registers the compiler cannot see through)
//   mode 2: MFMAs, then the test of the same block (one accumulator set; the product kernel at four waves per SIMD)
//   mode 3: MFMAs of block b, then the test of block b - 1 (two accumulator sets; the product kernel at two waves per SIMD)
//   mode 4: mode 2 + one ds_read_b128 per block, requested three blocks ahead
//   mode 5: mode 2 with the test on HALF the values (what a test of 8 v_max3 would cost: not a correct filter, a price)
//   mode 6: mode 2 with an s_sleep between the MFMAs and the test (does a waiting wave hold the issue port?)
// build: hipcc -O3 --offload-arch=gfx950 scripts/micro/i8_block_test_rate.hip -o scripts/micro/i8_block_test_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int v16i32 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void opaque(v16i32 &x) {
  asm volatile("" : "+v"(x));
}
__device__ __forceinline__ int max32(const v16i32 &x0, const v16i32 &x1) {
  int ma = max(max(x0[0], x0[1]), x0[2]), mb = max(max(x0[8], x0[9]), x0[10]);
  int mc = max(max(x1[0], x1[1]), x1[2]), md = max(max(x1[8], x1[9]), x1[10]);
  ma = max(max(ma, x0[3]), x0[4]);
  mb = max(max(mb, x0[11]), x0[12]);
  mc = max(max(mc, x1[3]), x1[4]);
  md = max(max(md, x1[11]), x1[12]);
  ma = max(max(ma, x0[5]), x0[6]);
  mb = max(max(mb, x0[13]), x0[14]);
  mc = max(max(mc, x1[5]), x1[6]);
  md = max(max(md, x1[13]), x1[14]);
  ma = max(max(ma, x0[7]), x0[15]);
  mc = max(max(mc, x1[7]), x1[15]);
  ma = max(max(ma, mb), mc);
  return max(ma, md);
}
__device__ __forceinline__ int max16(const v16i32 &x0) {
  int ma = max(max(x0[0], x0[1]), x0[2]), mb = max(max(x0[8], x0[9]), x0[10]);
  ma = max(max(ma, x0[3]), x0[4]);
  mb = max(max(mb, x0[11]), x0[12]);
  ma = max(max(ma, x0[5]), x0[6]);
  mb = max(max(mb, x0[13]), x0[14]);
  ma = max(max(ma, x0[7]), x0[15]);
  return max(ma, mb);
}

constexpr int kIters = 4096;

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void block_loop_kernel(int seed, int rhs, unsigned long long *cycles, int *sink) {
  __shared__ i32x4 frag[64 * 8];
  const int lane = threadIdx.x & 63;
  i32x4 a0, a1, b;
  v16i32 nth0, nth1, xa0, xa1, xb0, xb1;
  for (int k = 0; k < 4; ++k) {
    a0[k] = (threadIdx.x * 2654435761u + k * 40503u + seed) & 0x03030303;
    a1[k] = (threadIdx.x * 40503u + k * 2654435761u + seed) & 0x03030303;
    b[k] = (threadIdx.x * 97u + k * 13u + seed) & 0x03030303;
  }
  for (int r = 0; r < 16; ++r) nth0[r] = -(1 << 20) - r, nth1[r] = -(1 << 20) - 2 * r;  // nothing reaches its threshold
  xa0 = xa1 = xb0 = xb1 = nth0;
  for (int i = threadIdx.x; i < 64 * 8; i += blockDim.x) frag[i] = b;
  __syncthreads();
  const unsigned f_addr = (unsigned)(size_t)(__attribute__((address_space(3))) void *)frag + lane * 16u;
  i32x4 fa[4] = {b, b, b, b};
  if (MODE == 4) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(fa[0]) : "v"(f_addr) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(fa[1]) : "v"(f_addr) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(fa[2]) : "v"(f_addr) : "memory");
  }
  unsigned bits = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < kIters; it += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 4) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[(u + 3) & 3]) : "v"(f_addr), "n"(((u + 3) & 7) * 1024) : "memory");
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[u & 3]));
      }
      if (MODE != 4) asm volatile("" : "+v"(b));  // (a new operand as far as the compiler knows: the MFMAs stay in the loop)
      const i32x4 bb = MODE == 4 ? fa[u & 3] : b;
      if (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 5 || MODE == 6) {
        xa0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bb, nth0, 0, 0, 0);
        xa1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bb, nth1, 0, 0, 0);
      }
      if (MODE == 3) {
        if (u & 1) {
          xb0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bb, nth0, 0, 0, 0);
          xb1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bb, nth1, 0, 0, 0);
        } else {
          xa0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, bb, nth0, 0, 0, 0);
          xa1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, bb, nth1, 0, 0, 0);
        }
      }
      if (MODE == 0) {
        opaque(xa0);
        opaque(xa1);
      }
      if (MODE == 6) __builtin_amdgcn_s_sleep(1);
      if (MODE == 1) {
        opaque(xa0);
        opaque(xa1);
      }
      if (MODE != 0) {
        int m;
        if (MODE == 3) m = (u & 1) ? max32(xa0, xa1) : max32(xb0, xb1);  // the block before
        else if (MODE == 5) m = max16(xa0) | (xa1[0] & 1);
        else m = max32(xa0, xa1);
        if (__any(m >= rhs)) bits |= 1u << u;
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  int s = bits;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += xa0[k] + xa1[k] + xb0[k] + xb1[k];
  sink[threadIdx.x] = s;
  if (lane == 0) {  // first start to last end over the workgroup's waves (the oldest wave wins every arbitration: its own time says little)
    atomicMin(cycles, t0);
    atomicMax(cycles + 1, t1);
  }
}

template <int MODE, int WAVES>
void run(const char *name) {
  unsigned long long *d, h = 0, hh[2], init[2] = {~0ull, 0ull};
  int *sink;
  (void)hipMalloc(&d, 16);
  (void)hipMalloc(&sink, sizeof(int) * 64 * WAVES);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0.f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemcpy(d, init, 16, hipMemcpyHostToDevice);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((block_loop_kernel<MODE, WAVES>), dim3(1), dim3(64 * WAVES), 0, 0, 1 + rep, 0, d, sink);
    (void)hipEventRecord(e1, 0);
    (void)hipMemcpy(hh, d, 16, hipMemcpyDeviceToHost);
    h = hh[1] - hh[0];
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  const double per_wave = (double)h / kIters, per_simd = per_wave / (WAVES / 4.0);
  printf("%-44s %2d waves/SIMD: %7.1f cycles per block and wave = %6.1f per block on a SIMD (MFMA pipe: 64)  [%.3f ms]\n", name, WAVES / 4,
         per_wave, per_simd, ms);
  (void)hipFree(d);
  (void)hipFree(sink);
}

int main() {
  run<0, 4>("0 MFMAs alone");
  run<0, 8>("0 MFMAs alone");
  run<0, 16>("0 MFMAs alone");
  run<1, 4>("1 test alone");
  run<1, 8>("1 test alone");
  run<1, 16>("1 test alone");
  run<2, 4>("2 MFMAs + test of the same block");
  run<2, 8>("2 MFMAs + test of the same block");
  run<2, 16>("2 MFMAs + test of the same block");
  run<3, 4>("3 MFMAs + test of the block before");
  run<3, 8>("3 MFMAs + test of the block before");
  run<3, 16>("3 MFMAs + test of the block before");
  run<4, 8>("4 as 2 + ds_read_b128 per block");
  run<4, 16>("4 as 2 + ds_read_b128 per block");
  run<5, 8>("5 as 2, half the values tested");
  run<5, 16>("5 as 2, half the values tested");
  run<6, 16>("6 as 2 + s_sleep 1 before the test");
  return 0;
}
