// mfma_join_f16_flow.h -- the masked f16 join, four 128-column stages in flight and NO stage barrier (round 6).
// mfma_join_f16_deep.h with its per-stage __syncthreads replaced by two counters per buffer in LDS: `landed` (a wave adds 1 when
// its pieces of the stage are in LDS) and `done` (a wave adds 1 when it has read the stage for the last time).  A wave starts a
// stage when all eight have landed it and refills a buffer when all eight are done with its previous stage: a wave that drew few
// blocks (or none to list) runs up to three stages ahead of the one that drew many, instead of waiting for it at every stage.

__device__ __forceinline__ uint32_t ls_load(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
typedef float ls_f4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void ls_write128(uint32_t addr, const ls_f4 &v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

constexpr uint32_t kFlQueue = 216;
constexpr int kFlBufs = 4;
constexpr size_t kFlBytes = (size_t)kFlBufs * 32768 + (size_t)kFlBufs * 128 * sizeof(float) + (size_t)8 * kFlQueue * 12 + (size_t)8 * 256 * sizeof(float) + 256;

__global__ __launch_bounds__(512, 2) void pair_join_f16_flow_kernel(const f16x8 *__restrict__ P, uint32_t T2, uint32_t I_begin, uint32_t I_count, uint32_t W,
                                                                    uint32_t n, const float *__restrict__ theta, uint32_t cap, uint32_t *__restrict__ cnt,
                                                                    uint32_t *__restrict__ cand, uint32_t *__restrict__ unit_counter,
                                                                    unsigned long long *__restrict__ stamps, const uint32_t *__restrict__ perm,
                                                                    const unsigned long long *__restrict__ mask, const uint32_t *__restrict__ unit_list, int dbg,
                                                                    const uint32_t *__restrict__ tokens) {
  constexpr int KS = 8;
  constexpr int kTileVecs = KS * 4 * 64;  // a stage: one 128-row tile of the image
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f16x8 *bufB = reinterpret_cast<f16x8 *>(smem);
  float *colthr = reinterpret_cast<float *>(smem + (size_t)kFlBufs * 32768);
  uint2 *queue_all = reinterpret_cast<uint2 *>(colthr + kFlBufs * 128);
  float *queue_s_all = reinterpret_cast<float *>(queue_all + 8 * kFlQueue);
  float *dump_all = queue_s_all + 8 * kFlQueue;
  // flow[8 b + w]: the token wave w's load of buffer b's stage ends with (its generation + 1: a DMA of 4 bytes behind the wave's pieces,
  // which land in issue order); flow[32 + b]: how many waves are done with buffer b, all generations (monotonic)
  uint32_t *flow = reinterpret_cast<uint32_t *>(dump_all + 8 * 256);
  __shared__ uint32_t s_unit;
  __shared__ uint32_t s_nlist[2];
  __shared__ __attribute__((aligned(16))) unsigned char s_step[128];
  __shared__ unsigned long long s_bits[128];

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, lc = lane & 31;
  auto lanes_below = [&](unsigned long long m) -> uint32_t {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  };
  const uint32_t n_norm = (T2 - 1) / 2;
  const uint32_t n_cblocks = (T2 + W - 1) / W;
  const uint32_t total_units = n_cblocks * I_count;
  const unsigned long long kHalf0 = 0x0F0F0F0F0F0F0F0Full;

  uint2 *queue = queue_all + (size_t)wv * kFlQueue;
  float *queue_s = queue_s_all + (size_t)wv * kFlQueue;
  float *ex = dump_all + (size_t)wv * 256;
  uint32_t qn = 0;
  unsigned long long n_blk = 0, n_pass = 0, n_sparse = 0, n_pre = 0;
  if (tid < 40) flow[tid] = 0u;
  const uint32_t flow_addr = lds_addr(flow);
  uint32_t gv = 0;  // stages consumed so far (all units): stage g lives in buffer g & 3, generation g >> 2
  auto flow_wait = [&](uint32_t word, uint32_t target) {
    while ((uint32_t)__builtin_amdgcn_readfirstlane((int)ls_load(flow_addr + 4u * word)) < target) __builtin_amdgcn_s_sleep(1);
  };
  auto landed_wait = [&](uint32_t buf, uint32_t expect) {  // the eight waves' tokens of this buffer all carry the stage's generation
    for (;;) {
      const uint32_t t = ls_load(flow_addr + 32u * buf + 4u * (uint32_t)(lane & 7));
      if (!__any(t != expect)) break;
      __builtin_amdgcn_s_sleep(1);
    }
  };
  auto flow_add = [&](uint32_t word) {
    if (lane == 0) asm volatile("ds_add_u32 %0, %1" ::"v"(flow_addr + 4u * word), "v"(1u) : "memory");
  };

  auto flush_queue = [&]() {
    lds_sync();
    constexpr int R = kFlushRounds;
    const uint32_t fq_addr = lds_addr(queue) + 8u * (uint32_t)lane, fs_addr = lds_addr(queue_s) + 4u * (uint32_t)lane;
    for (uint32_t k0 = 0; k0 * 64u < qn; k0 += R) {
      uint32_t ta[R], tb[R], ts[R], pa[R], pb[R];
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const uint32_t e = (k0 + (uint32_t)k) * 64u + lane;
        ts[k] = 0;
        ta[k] = tb[k] = 0;
        if (e < qn) {
          unsigned long long vxy;
          uint32_t vs;
          asm volatile(
              "ds_read_b64 %0, %2\n\t"
              "ds_read_b32 %1, %3\n\t"
              "s_waitcnt lgkmcnt(0)"
              : "=&v"(vxy), "=&v"(vs)
              : "v"(fq_addr + (k0 + (uint32_t)k) * 512u), "v"(fs_addr + (k0 + (uint32_t)k) * 256u)
              : "memory");
          const uint32_t vx = (uint32_t)vxy, vy = (uint32_t)(vxy >> 32);
          if (vx != vy) {
            ta[k] = perm ? perm[vx] : vx;
            tb[k] = perm ? perm[vy] : vy;
            ts[k] = vs;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        pa[k] = pb[k] = 0;
        if (ts[k] & 1u) pa[k] = atomicAdd(cnt + ta[k], 1u);
        if (ts[k] & 2u) pb[k] = atomicAdd(cnt + tb[k], 1u);
      }
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const uint32_t sc = ts[k] & ~3u;
        if ((ts[k] & 1u) && pa[k] < cap) reinterpret_cast<uint2 *>(cand)[(size_t)ta[k] * cap + pa[k]] = make_uint2(tb[k], sc);
        if ((ts[k] & 2u) && pb[k] < cap) reinterpret_cast<uint2 *>(cand)[(size_t)tb[k] * cap + pb[k]] = make_uint2(ta[k], sc);
      }
    }
    lds_sync();
    qn = 0;
  };

  // the listing of one tile: pair_join_f16_kernel's emit (MODE 0 behind the mask) -- the sign-bit word, the prefix sums, the sparse
  // form through the dump, else the 16-step walk; a block too dense for the queue goes register by register
  auto list_tile = [&](const v16f32 &acc, const v16f32 &nth, uint32_t row_group0, uint32_t colbase, float tc, bool dead) -> bool {
    const uint32_t col = colbase + lc, row0 = row_group0 + 4 * lh;
    const float z = col < n ? 0.f : __builtin_inff();
    const uint32_t q_addr = lds_addr(queue), s_addr = lds_addr(queue_s);
    auto cmp_ge = [](float a, float b) -> unsigned long long {
      unsigned long long m;
      asm("v_cmp_ge_f32 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
      return m;
    };
    auto pick = [](unsigned long long m, uint32_t v) -> uint32_t {
      uint32_t r;
      asm("v_cndmask_b32 %0, 0, %2, %1" : "=v"(r) : "s"(m), "v"(v));
      return r;
    };
    uint32_t wd = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d2 = (acc[r] - nth[r]) - tc;
      wd = __builtin_amdgcn_alignbit(wd, __float_as_uint(d2), 31);
      wd = __builtin_amdgcn_alignbit(wd, __float_as_uint(acc[r]), 31);
    }
    uint32_t pass = col < n ? ~wd : 0u;
    if (dead) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (nth[r] == -__builtin_inff()) pass &= ~(3u << (2 * (15 - r)));
    }
    if (!__any(pass != 0u)) return false;
    const uint32_t c = (uint32_t)__builtin_popcount((pass | (pass >> 1)) & 0x55555555u);
    uint32_t inc = c;
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xF, 0xF, true);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xF, 0xF, true);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xF, 0xF, true);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xF, 0xF, true);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x142, 0xA, 0xF, true);
    inc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)inc, 0x143, 0xC, 0xF, true);
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    if (tot <= kFlQueue) {
      if (qn && qn + tot > 64u * (uint32_t)kFlushRounds) flush_queue();
      const uint32_t o = qn + inc - c;
      uint32_t qa = q_addr + 8u * o, sa = s_addr + 4u * o;
      const unsigned long long hm = __ballot(c != 0u);
      if ((uint32_t)__builtin_popcountll(hm) <= 16u && !__any(c > 4u)) {
        const uint32_t dump = lds_addr(ex) + 64u * lanes_below(hm);
        if (c != 0u) {
          ls_f4 v0, v1, v2, v3;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v0[r] = acc[r] - nth[r];
            v1[r] = acc[4 + r] - nth[4 + r];
            v2[r] = acc[8 + r] - nth[8 + r];
            v3[r] = acc[12 + r] - nth[12 + r];
          }
          ls_write128<0>(dump, v0);
          ls_write128<16>(dump, v1);
          ls_write128<32>(dump, v2);
          ls_write128<48>(dump, v3);
        }
        uint32_t pp = (pass | (pass >> 1)) & 0x55555555u;
        uint32_t rk[4], fk[4];
        float vk[4];
        bool hk[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          hk[k] = pp != 0u;
          const uint32_t b = hk[k] ? (uint32_t)__builtin_ctz(pp) : 0u;
          pp &= pp - 1u;
          rk[k] = 15u - (b >> 1);
          fk[k] = (pass >> b) & 3u;
          asm volatile("ds_read_b32 %0, %1" : "=v"(vk[k]) : "v"(dump + 4u * rk[k]) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vk[0]), "+v"(vk[1]), "+v"(vk[2]), "+v"(vk[3])::"memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (hk[k]) {
            const uint32_t sc = (__float_as_uint(vk[k] * (1.f / kProductScale)) & ~3u) | fk[k];
            const uint32_t row = row0 + (rk[k] & 3u) + 8u * (rk[k] >> 2);
            asm volatile(
                "ds_write2_b32 %0, %2, %3 offset1:1\n\t"
                "ds_write_b32 %1, %4\n\t"
                "v_add_u32 %0, 8, %0\n\t"
                "v_add_u32 %1, 4, %1"
                : "+v"(qa), "+v"(sa)
                : "v"(row), "v"(col), "v"(sc)
                : "memory");
          }
        }
        qn += tot;
        n_sparse += 1;
        lds_sync();
        return true;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t f = (pass >> (2 * (15 - r))) & 3u;
        if (f) {
          const float vr = acc[r] - nth[r];
          const uint32_t sc = (__float_as_uint(vr * (1.f / kProductScale)) & ~3u) | f;
          const uint32_t row = row0 + (r & 3) + 8 * (r >> 2);
          asm volatile(
              "ds_write2_b32 %0, %2, %3 offset1:1\n\t"
              "ds_write_b32 %1, %4\n\t"
              "v_add_u32 %0, 8, %0\n\t"
              "v_add_u32 %1, 4, %1"
              : "+v"(qa), "+v"(sa)
              : "v"(row), "v"(col), "v"(sc)
              : "memory");
        }
      }
      qn += tot;
      lds_sync();
      return true;
    }
    uint32_t resume = 0;
    bool listed = false;
    for (;;) {
      bool full = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (full || (uint32_t)r < resume) continue;
        const float vr = acc[r] - nth[r];
        // (a row that never lists has an accumulator of -inf and acc - nth = NaN: neither compare is true)
        const unsigned long long m1 = cmp_ge(acc[r], z), m2 = cmp_ge(vr, tc), mm = m1 | m2;
        if (mm == 0) continue;
        const uint32_t cnt_r = (uint32_t)__builtin_popcountll(mm);
        if (qn + cnt_r > kFlQueue) {
          full = true;
          resume = (uint32_t)r;
          continue;
        }
        const uint32_t o = qn + lanes_below(mm);
        const uint32_t sb = (__float_as_uint(vr * (1.f / kProductScale)) & ~3u) | pick(m1, 1u) | pick(m2, 2u);
        const unsigned long long rc = ((unsigned long long)col << 32) | (row0 + (r & 3) + 8 * (r >> 2));
        unsigned long long saved;
        asm volatile(
            "s_and_saveexec_b64 %0, %1\n\t"
            "ds_write_b64 %2, %3\n\t"
            "ds_write_b32 %4, %5\n\t"
            "s_mov_b64 exec, %0"
            : "=&s"(saved)
            : "s"(mm), "v"(q_addr + 8u * o), "v"(rc), "v"(s_addr + 4u * o), "v"(sb)
            : "memory");
        qn += cnt_r;
        listed = true;
      }
      if (!full) break;
      flush_queue();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_sync();
    return listed;
  };


  const uint32_t step_addr = lds_addr(s_step), bits_addr = lds_addr(s_bits);
  for (;;) {
    __syncthreads();
    if (tid == 0) s_unit = atomicAdd(unit_counter, 1u);
    __syncthreads();
    uint32_t u = s_unit;
    if (unit_list && unit_list[0] * 2u <= total_units) {
      if (u >= unit_list[0]) break;
      u = unit_list[1 + u];
    } else if (u >= total_units) break;
    const uint32_t c = u / I_count, I = I_begin + (u - c * I_count);
    const uint32_t B0 = c * W, Wc = min(W, T2 - B0);
    uint32_t p_lo, p_hi;
    {
      const bool antipodal = (T2 % 2 == 0) && (I < T2 / 2) && T2 > 1;
      const uint32_t L = 1 + n_norm + (antipodal ? 1u : 0u);
      const uint32_t delta = B0 >= I ? B0 - I : B0 + T2 - I;
      if (delta + Wc <= T2) {
        p_lo = delta;
        p_hi = min(delta + Wc, L);
      } else {
        p_lo = 0;
        p_hi = min(delta + Wc - T2, L);
      }
      if (p_hi <= p_lo) continue;
    }
    uint32_t steps = (p_hi - p_lo) * 2;  // 256-column stages of the mask
    if (tid < 128) {
      unsigned long long b = 0;
      if ((uint32_t)tid < steps) {
        uint32_t S = I + p_lo + ((uint32_t)tid >> 1);
        S = S >= T2 ? S - T2 : S;
        b = mask[(size_t)(I - I_begin) * 2 * T2 + 2 * S + ((uint32_t)tid & 1u)];
      }
      const unsigned long long have = __ballot(b != 0);
      if (lane == 0) s_nlist[wv] = (uint32_t)__builtin_popcountll(have);
      __syncthreads();
      if (b != 0) {
        const uint32_t at = (wv == 1 ? s_nlist[0] : 0u) + lanes_below(have);
        s_step[at] = (unsigned char)tid;
        s_bits[at] = b;
      }
    } else {
      __syncthreads();
    }
    __syncthreads();
    steps = s_nlist[0] + s_nlist[1];
    if (steps == 0) continue;

    // the wave's 64 rows: two groups of 32, A fragments and (minus) thresholds in registers for the whole unit
    f16x8 a0[KS], a1[KS];
    {
      const size_t T = (size_t)4 * I + (wv >> 1);
      const int s0 = 2 * (wv & 1);
      const f16x8 *a_src = P + T * kTileVecs + (size_t)s0 * 64 + lane;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        a0[ks] = a_src[(size_t)ks * 256];
        a1[ks] = a_src[(size_t)ks * 256 + 64];
      }
    }
    const uint32_t row_base0 = I * kSuperH + wv * 64 + 4 * lh, row_base1 = row_base0 + 32;
    v16f32 nth0, nth1;
    float tmax = -__builtin_inff();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float t0 = thr_scaled(theta[row_base0 + (r & 3) + 8 * (r >> 2)]);
      const float t1 = thr_scaled(theta[row_base1 + (r & 3) + 8 * (r >> 2)]);
      nth0[r] = -t0;
      nth1[r] = -t1;
      if (t0 < __builtin_inff()) tmax = __builtin_fmaxf(tmax, t0);
      if (t1 < __builtin_inff()) tmax = __builtin_fmaxf(tmax, t1);
    }
    asm volatile("" : "+v"(nth0), "+v"(nth1));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tmax = __builtin_fmaxf(tmax, __shfl_xor(tmax, o, 64));
    tmax = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tmax)));
    bool dead = false;
    {
      bool d0 = false;
#pragma unroll
      for (int r = 0; r < 16; ++r) d0 = d0 || nth0[r] == -__builtin_inff() || nth1[r] == -__builtin_inff();
      dead = __any(d0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the loads above: from here on the wave's vector-memory count is its DMA's and its flushes')

    // the visits: half stages (li, h) that hold a flagged block for anybody, in order
    auto step_of = [&](uint32_t li) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)(ls_load(step_addr + (li & ~3u)) >> (8u * (li & 3u)))) & 0xFFu; };
    auto bits_of = [&](uint32_t li) -> unsigned long long {
      unsigned long long b;
      asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(b) : "v"(bits_addr + 8u * li) : "memory");
      return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    };
    auto tile_of = [&](uint32_t li, uint32_t h) -> uint32_t {
      const uint32_t step = step_of(li);
      uint32_t S = I + p_lo + (step >> 1);
      S = S >= T2 ? S - T2 : S;
      return 4 * S + 2 * (step & 1u) + h;
    };
    auto next_visit = [&](uint32_t &li, uint32_t &h) -> bool {  // the first visit at or behind (li, h); false: none left
      for (; li < steps; ++li, h = 0) {
        const unsigned long long b = bits_of(li);
        for (; h < 2; ++h)
          if (b & (kHalf0 << (4 * h))) return true;
      }
      return false;
    };
    // the load of a stage: 32 pieces of 1 KB, four a wave, and the wave's sixteenth of the 128 column thresholds: FIVE DMA
    // instructions per wave and stage (what the vmcnt waits below count)
    auto stage = [&](uint32_t tile, int buf, uint32_t gen) {
      const f16x8 *src = P + (size_t)tile * kTileVecs + lane;
      f16x8 *dst = bufB + (size_t)buf * kTileVecs;
#pragma unroll
      for (int p = 0; p < 4; ++p)
        __builtin_amdgcn_global_load_lds((gbl_void *)(src + (wv * 4 + p) * 64), (lds_void *)(dst + (wv * 4 + p) * 64), 16, 0, 0);
      // (lanes 0 - 15 of the wave: 64 bytes)
      if (lane < 16)
        __builtin_amdgcn_global_load_lds((gbl_void *)(theta + (size_t)tile * kTileRows + wv * 16 + lane), (lds_void *)(colthr + buf * 128 + wv * 16), 4, 0, 0);
      if (lane == 0)
        __builtin_amdgcn_global_load_lds((gbl_void *)(tokens + (((uint32_t)gen + 1u) & 0xFFFFu)), (lds_void *)(flow + buf * 8 + wv), 4, 0, 0);
    };
    uint32_t li = 0, h = 0;
    next_visit(li, h);  // (steps > 0: there is one)
    // the issue cursor runs three visits ahead
    uint32_t li_i = li, h_i = h;
    int ahead = 0;  // stages issued and not yet consumed, the current one included
    bool more_i = true;
    for (int k = 0; k < 3 && more_i; ++k) {
      const uint32_t g = gv + (uint32_t)k;
      flow_wait(32u + (g & 3u), 8u * (g >> 2));  // (everybody is done with the buffer's previous stage: true behind the unit's barrier)
      stage(tile_of(li_i, h_i), (int)(g & 3u), g >> 2);
      ++ahead;
      ++h_i;
      more_i = next_visit(li_i, h_i);
    }
    for (;;) {
      const int buf = (int)(gv & 3u);
      // stage gv + 3 into the buffer stage gv - 1 lived in, once all eight waves are done with it
      if (more_i) {
        const uint32_t g = gv + 3u;
        flow_wait(32u + (g & 3u), 8u * (g >> 2));
        stage(tile_of(li_i, h_i), (int)(g & 3u), g >> 2);
        ++ahead;
        ++h_i;
        more_i = next_visit(li_i, h_i);
      }
      landed_wait((uint32_t)buf, ((gv >> 2) + 1u) & 0xFFFFu);  // all eight waves' pieces of this stage have landed
      const uint32_t tile0 = tile_of(li, h);
      const unsigned long long bits = bits_of(li);
      const uint32_t step = step_of(li);
      const bool diag = (p_lo + (step >> 1)) == 0;
      const uint32_t nib = (uint32_t)(bits >> (8 * wv + 4 * h)) & 0xFu;
      const uint32_t b_addr = lds_addr(bufB + (size_t)buf * kTileVecs) + (uint32_t)lane * 16u;
      const uint32_t tc_addr = lds_addr(colthr + buf * 128) + (uint32_t)lc * 4u;
#pragma nounroll
      for (int nn = 0; nn < 4; ++nn) {
        if (!((nib >> nn) & 1u)) continue;
        const uint32_t blk = b_addr + (uint32_t)(nn * 64 * 16);
        const uint32_t colbase = tile0 * kTileRows + nn * 32;
        float tcr;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(tcr) : "v"(tc_addr + (uint32_t)nn * 128u) : "memory");
        const float tc = (diag || colbase + lc >= n) ? __builtin_inff() : thr_scaled(tcr);
        auto half = [&](const f16x8(&a)[KS], const v16f32 &nth, uint32_t row_base) {
          f16x8 fa[4];
          lds_read16<0>(fa[0], blk);
          lds_read16<1 * 4 * 64 * 16>(fa[1], blk);
          lds_read16<2 * 4 * 64 * 16>(fa[2], blk);
          v16f32 x;
          static_for<0, KS>([&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            constexpr int ahead_k = ks + 3;
            if constexpr (ahead_k < KS) lds_read16<ahead_k * 4 * 64 * 16>(fa[ahead_k & 3], blk);
            constexpr int pending = (KS - 1 - ks) < 3 ? (KS - 1 - ks) : 3;
            lds_wait1<pending>(fa[ks & 3]);
            x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], fa[ks & 3], ks == 0 ? nth : x, 0, 0, 0);
          });
          n_blk += 1;
          if (dbg & 2) return;
          // most blocks hold no pair: one maximum per lane settles those (pair_join_f16_kernel's pretest)
          if (!__any(max16(x) >= __builtin_fminf(0.f, tc - tmax))) return;
          n_pre += 1;
          n_pass += list_tile(x, nth, row_base - 4 * lh, colbase, tc, dead) ? 1 : 0;
        };
        half(a0, nth0, row_base0);
        half(a1, nth1, row_base1);
      }
      // the end of this stage for the wave: it will not read the buffer again
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      flow_add(32u + (uint32_t)buf);
      --ahead;
      ++gv;
      ++h;
      if (!next_visit(li, h)) break;
    }
  }
  flush_queue();
  if (lane == 0) {
    if (n_blk) atomicAdd(stamps + 5, n_blk);
    if (n_pass) atomicAdd(stamps + 7, n_pass);
    if (n_sparse) atomicAdd(stamps + 9, n_sparse);
    if (n_pre) atomicAdd(stamps + 1, n_pre);
  }
}
