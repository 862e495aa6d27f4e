// mfma_join_f16_lister.h -- the masked f16 join with LISTER WAVES (round 6; rows of 128 values behind the first pass).
// Included by mfma_join_f16.hip inside its anonymous namespace (it uses that file's helpers: lds_read16, lds_wait1, static_for,
// thr_scaled, max16, lds_sync, lds_addr, kFlushRounds, the image layout).
//
// What bounds pair_join_f16_kernel<8, 0, masked> was measured this round (scripts/join_phases.py): without its listing 1.8 ms, with
// it 3.3 -- and neither the appends (3.3 ms without a single global atomic) nor the instruction count of the queue writes (a form
// with a third fewer changed nothing) is the cost: every wave lists its own blocks, a wave that lists is late at the stage barrier,
// and the other seven wait.  Here the two jobs have waves of their own.  A workgroup is still 8 waves on a 512-row super tile, but
//   * waves 0 - 3 are PRODUCERS: each holds the A fragments of 128 rows (four groups of 32) in registers and contracts the flagged
//     blocks of a 128-column stage, two row groups per fragment read; a block whose largest product reaches the smallest threshold
//     of its row group or its lane's column threshold (one v_max3 chain and a compare: everything else of the listing is gone from
//     this wave) is DUMPED -- 16 products and the column threshold per lane, 4.25 KB -- into a ring of four slots in LDS;
//   * waves 4 - 7 are LISTERS, one per producer, on the same SIMD (waves w and w + 4 share one): a lister takes a tile from its
//     producer's ring, subtracts the row thresholds, forms the sign-bit word, and queues / appends the pairs exactly as the listing
//     of pair_join_f16_kernel does.  Its vector instructions run under the producer's MFMAs instead of in front of them.
// A lister goes to a stage barrier only when its producer has finished the stage (a counter in LDS, written behind the last tile),
// so a full ring never meets a lister that waits at a barrier; it has drained its ring by then (the tiles refer to the unit's table
// of row thresholds).  Stages are 128 columns (one tile of the image, 32 KB, two in flight): the LDS the 256-column stages held
// is the rings'.  Same lists as pair_join_f16_kernel up to the last bits of a listed score (the accumulator starts at 0 here, at
// minus the row threshold there; cos_delta_f16_accum covers either).

constexpr int kLsSlots = 4;
constexpr uint32_t kLsSlotBytes = 4096 + 256;
constexpr uint32_t kLsQueue = 216;
constexpr size_t kLsBytes = (size_t)2 * 32768 + 2 * 128 * sizeof(float) + (size_t)4 * kLsSlots * kLsSlotBytes + (size_t)4 * kLsSlots * 8 + 16 * sizeof(uint32_t) +
                            512 * sizeof(float) + (size_t)4 * kLsQueue * 12 + (size_t)4 * 256 * sizeof(float);

__device__ __forceinline__ uint32_t ls_load(uint32_t addr) {
  uint32_t v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void ls_store(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
typedef float ls_f4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void ls_read128(ls_f4 &dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void ls_write128(uint32_t addr, const ls_f4 &v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

__global__ __launch_bounds__(512, 2) void pair_join_f16_lister_kernel(const f16x8 *__restrict__ P, uint32_t T2, uint32_t I_begin, uint32_t I_count, uint32_t W,
                                                                      uint32_t n, const float *__restrict__ theta, uint32_t cap, uint32_t *__restrict__ cnt,
                                                                      uint32_t *__restrict__ cand, uint32_t *__restrict__ unit_counter,
                                                                      unsigned long long *__restrict__ stamps, const uint32_t *__restrict__ perm,
                                                                      const unsigned long long *__restrict__ mask, const uint32_t *__restrict__ unit_list, int dbg) {
  constexpr int KS = 8;
  constexpr int kTileVecs = KS * 4 * 64;  // a stage: one 128-row tile of the image
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f16x8 *bufB = reinterpret_cast<f16x8 *>(smem);
  float *colthr = reinterpret_cast<float *>(smem + 2 * 32768);
  unsigned char *ring = smem + 2 * 32768 + 2 * 128 * sizeof(float);
  unsigned char *hdr = ring + (size_t)4 * kLsSlots * kLsSlotBytes;               // 8 B per slot: column base, row group | diagonal << 8
  uint32_t *ctl = reinterpret_cast<uint32_t *>(hdr + (size_t)4 * kLsSlots * 8);  // per pair: [0] tiles produced, [1] consumed, [2] stages finished
  float *rowthr = reinterpret_cast<float *>(ctl + 16);                           // the unit's 512 scaled row thresholds
  uint2 *queue_all = reinterpret_cast<uint2 *>(rowthr + 512);
  float *queue_s_all = reinterpret_cast<float *>(queue_all + 4 * kLsQueue);
  float *dump_all = queue_s_all + 4 * kLsQueue;
  __shared__ uint32_t s_unit;
  __shared__ uint32_t s_nlist[2];
  __shared__ unsigned char s_step[128];
  __shared__ unsigned long long s_bits[128];

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lh = lane >> 5, lc = lane & 31;
  const bool producer = wv < 4;
  const int w = wv & 3;  // the pair
  const uint32_t ctl_addr = lds_addr(ctl) + 16u * (uint32_t)w;
  const uint32_t ring_addr = lds_addr(ring) + (uint32_t)w * kLsSlots * kLsSlotBytes;
  const uint32_t hdr_addr = lds_addr(hdr) + (uint32_t)w * kLsSlots * 8u;
  if (tid < 16) ctl[tid] = 0u;
  auto lanes_below = [&](unsigned long long m) -> uint32_t {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
  };
  const uint32_t n_norm = (T2 - 1) / 2;
  const uint32_t n_cblocks = (T2 + W - 1) / W;
  const uint32_t total_units = n_cblocks * I_count;
  const unsigned long long kHalf0 = 0x0F0F0F0F0F0F0F0Full;

  uint2 *queue = queue_all + (size_t)w * kLsQueue;
  float *queue_s = queue_s_all + (size_t)w * kLsQueue;
  float *ex = dump_all + (size_t)w * 256;
  uint32_t qn = 0;
  uint32_t n_tiles = 0;    // producer: tiles published; lister: tiles consumed
  uint32_t n_stages = 0;   // producer: stages finished; lister: stages it has gone to the barrier for
  uint32_t seen_tail = 0;  // producer: the lister's count as last read
  unsigned long long n_blk = 0, n_pass = 0, n_sparse = 0, n_pre = 0;

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
    if (tot <= kLsQueue) {
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
        if (qn + cnt_r > kLsQueue) {
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
    rowthr[tid] = thr_scaled(theta[(size_t)I * kSuperH + tid]);
    __syncthreads();
    steps = s_nlist[0] + s_nlist[1];
    if (steps == 0) continue;

    // the half stages (128 columns) that hold a flagged block for anybody, in order: visit v = (li, h)
    auto tile_of = [&](uint32_t li, uint32_t h) -> uint32_t {
      const uint32_t step = __builtin_amdgcn_readfirstlane((uint32_t)s_step[li]);
      uint32_t S = I + p_lo + (step >> 1);
      S = S >= T2 ? S - T2 : S;
      return 4 * S + 2 * (step & 1u) + h;
    };
    auto bits_of = [&](uint32_t li) -> unsigned long long {
      const unsigned long long b = s_bits[li];
      return ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    };
    auto next_visit = [&](uint32_t &li, uint32_t &h) -> bool {  // the first visit at or behind (li, h); false: none left
      for (; li < steps; ++li, h = 0) {
        const unsigned long long b = bits_of(li);
        for (; h < 2; ++h)
          if (b & (kHalf0 << (4 * h))) return true;
      }
      return false;
    };

    if (producer) {
      // ---- producer: 128 rows = tile 4 I + w of the image, four groups of 32 ----
      f16x8 a[4][KS];
      {
        const f16x8 *a_src = P + ((size_t)4 * I + w) * kTileVecs + lane;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) a[g][ks] = a_src[(size_t)ks * 256 + g * 64];
      }
      float tmin[4];  // the smallest finite scaled threshold of each row group (wave-uniform)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float t = rowthr[128 * w + 32 * g + lc];
        t = t < __builtin_inff() ? t : __builtin_inff();
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) t = __builtin_fminf(t, __shfl_xor(t, o, 64));
        tmin[g] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t)));
      }
      auto stage = [&](uint32_t tile, int buf) {  // 32 pieces of 1 KB: eight a producer; the 128 column thresholds: producer 3
        const f16x8 *src = P + (size_t)tile * kTileVecs + lane;
        f16x8 *dst = bufB + (size_t)buf * kTileVecs;
#pragma unroll
        for (int p = 0; p < 8; ++p)
          __builtin_amdgcn_global_load_lds((gbl_void *)(src + (w * 8 + p) * 64), (lds_void *)(dst + (w * 8 + p) * 64), 16, 0, 0);
        if (w == 3) {
          __builtin_amdgcn_global_load_lds((gbl_void *)(theta + (size_t)tile * kTileRows + lane), (lds_void *)(colthr + buf * 128), 4, 0, 0);
          __builtin_amdgcn_global_load_lds((gbl_void *)(theta + (size_t)tile * kTileRows + 64 + lane), (lds_void *)(colthr + buf * 128 + 64), 4, 0, 0);
        }
      };
      uint32_t li = 0, h = 0;
      next_visit(li, h);  // (steps > 0: there is one)
      stage(tile_of(li, h), 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      int buf = 0;
      for (;;) {
        const uint32_t tile0 = tile_of(li, h);
        const unsigned long long bits = bits_of(li);
        const uint32_t step = __builtin_amdgcn_readfirstlane((uint32_t)s_step[li]);
        const bool diag = (p_lo + (step >> 1)) == 0;
        uint32_t li_n = li, h_n = h + 1;
        const bool more = next_visit(li_n, h_n);
        if (more) stage(tile_of(li_n, h_n), buf ^ 1);
        const uint32_t b_addr = lds_addr(bufB + (size_t)buf * kTileVecs) + (uint32_t)lane * 16u;
        const uint32_t tc_addr = lds_addr(colthr + buf * 128) + (uint32_t)lc * 4u;
#pragma nounroll
        for (int q = 0; q < 2; ++q) {  // row groups 2 q, 2 q + 1: one byte of the mask (the 64 rows a wave of the first pass took)
          const uint32_t nib = (uint32_t)(bits >> (8 * (2 * w + q) + 4 * h)) & 0xFu;
          if (nib == 0u) continue;
#pragma nounroll
          for (int nn = 0; nn < 4; ++nn) {
            if (!((nib >> nn) & 1u)) continue;
            const uint32_t blk = b_addr + (uint32_t)(nn * 64 * 16);
            f16x8 fa[4];
            lds_read16<0>(fa[0], blk);
            lds_read16<1 * 4 * 64 * 16>(fa[1], blk);
            lds_read16<2 * 4 * 64 * 16>(fa[2], blk);
            v16f32 x0, x1, zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.f;
            static_for<0, KS>([&](auto ks_c) {
              constexpr int ks = decltype(ks_c)::value;
              constexpr int ahead = ks + 3;
              if constexpr (ahead < KS) lds_read16<ahead * 4 * 64 * 16>(fa[ahead & 3], blk);
              constexpr int pending = (KS - 1 - ks) < 3 ? (KS - 1 - ks) : 3;
              lds_wait1<pending>(fa[ks & 3]);
              if (q == 0) {
                x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][ks], fa[ks & 3], ks == 0 ? zero : x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][ks], fa[ks & 3], ks == 0 ? zero : x1, 0, 0, 0);
              } else {
                x0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2][ks], fa[ks & 3], ks == 0 ? zero : x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[3][ks], fa[ks & 3], ks == 0 ? zero : x1, 0, 0, 0);
              }
            });
            n_blk += 2;
            const uint32_t colbase = tile0 * kTileRows + nn * 32;
            float tcr;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(tcr) : "v"(tc_addr + (uint32_t)nn * 128u) : "memory");
            const float tc = (diag || colbase + lc >= n) ? __builtin_inff() : thr_scaled(tcr);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const v16f32 &x = e ? x1 : x0;
              const int g = 2 * q + e;
              // can a pair of this block reach its row's or its column's threshold?  (necessary: the largest product of a lane's
              // 16 rows against the smallest threshold of the group and the lane's column threshold)
              if (!__any(max16(x) >= __builtin_fminf(tmin[g], tc)) || (dbg & 2)) continue;  // (dbg 2: nothing is dumped)
              n_pre += 1;
              while (n_tiles - seen_tail >= (uint32_t)kLsSlots) {  // the ring is full: wait for the lister
                __builtin_amdgcn_s_sleep(1);
                seen_tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)ls_load(ctl_addr + 4u));
              }
              const uint32_t slot = n_tiles % (uint32_t)kLsSlots;
              const uint32_t base = ring_addr + slot * kLsSlotBytes;
              ls_f4 v0, v1, v2, v3;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v0[r] = x[r];
                v1[r] = x[4 + r];
                v2[r] = x[8 + r];
                v3[r] = x[12 + r];
              }
              const uint32_t ta = base + (uint32_t)lane * 16u;
              ls_write128<0>(ta, v0);
              ls_write128<1024>(ta, v1);
              ls_write128<2048>(ta, v2);
              ls_write128<3072>(ta, v3);
              asm volatile("ds_write_b32 %0, %1 offset:4096" ::"v"(base + (uint32_t)lane * 4u), "v"(tc) : "memory");
              if (lane == 0) {
                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" ::"v"(hdr_addr + slot * 8u), "v"(colbase), "v"((uint32_t)g | (diag ? 0x100u : 0u)) : "memory");
              }
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              n_tiles += 1;
              if (lane == 0) ls_store(ctl_addr, n_tiles);
            }
          }
        }
        n_stages += 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) ls_store(ctl_addr + 8u, n_stages);  // behind the stage's last tile
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next stage has landed
        __syncthreads();
        if (!more) break;
        li = li_n;
        h = h_n;
        buf ^= 1;
      }
    } else {
      // ---- lister: the tiles of producer w, stage by stage ----
      uint32_t li = 0, h = 0;
      next_visit(li, h);
      __syncthreads();  // (the producers' first stage)
      for (;;) {
        uint32_t li_n = li, h_n = h + 1;
        const bool more = next_visit(li_n, h_n);
        n_stages += 1;
        for (;;) {
          // (the stage counter first: the producer writes a stage's tiles, then its counter -- a counter that has reached this stage
          // and a ring that is empty behind it mean the stage is done)
          const uint32_t done = (uint32_t)__builtin_amdgcn_readfirstlane((int)ls_load(ctl_addr + 8u));
          const uint32_t head = (uint32_t)__builtin_amdgcn_readfirstlane((int)ls_load(ctl_addr));
          if (n_tiles != head) {
            const uint32_t slot = n_tiles % (uint32_t)kLsSlots;
            const uint32_t base = ring_addr + slot * kLsSlotBytes;
            ls_f4 s0, s1, s2, s3, t0, t1, t2, t3;
            float tc;
            uint32_t colbase, gd;
            const uint32_t ta = base + (uint32_t)lane * 16u;
            ls_read128<0>(s0, ta);
            ls_read128<1024>(s1, ta);
            ls_read128<2048>(s2, ta);
            ls_read128<3072>(s3, ta);
            asm volatile("ds_read_b32 %0, %1 offset:4096" : "=v"(tc) : "v"(base + (uint32_t)lane * 4u) : "memory");
            unsigned long long hd;
            asm volatile("ds_read_b64 %0, %1" : "=v"(hd) : "v"(hdr_addr + slot * 8u) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(tc), "+v"(hd)::"memory");
            colbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)hd);
            gd = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(hd >> 32));
            n_tiles += 1;
            if (lane == 0) ls_store(ctl_addr + 4u, n_tiles);  // the slot is the producer's again
            const uint32_t g = gd & 0xFFu;
            // the thresholds of the lane's 16 rows: rows 4 lh + (r & 3) + 8 (r >> 2) of the group
            const uint32_t th_addr = lds_addr(rowthr) + (uint32_t)(128 * w + 32 * (int)g + 4 * lh) * 4u;
            ls_read128<0>(t0, th_addr);
            ls_read128<32>(t1, th_addr);
            ls_read128<64>(t2, th_addr);
            ls_read128<96>(t3, th_addr);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3)::"memory");
            v16f32 acc, nth;
            bool d0 = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              nth[r] = -t0[r];
              nth[4 + r] = -t1[r];
              nth[8 + r] = -t2[r];
              nth[12 + r] = -t3[r];
              acc[r] = s0[r] - t0[r];
              acc[4 + r] = s1[r] - t1[r];
              acc[8 + r] = s2[r] - t2[r];
              acc[12 + r] = s3[r] - t3[r];
              d0 = d0 || t0[r] == __builtin_inff() || t1[r] == __builtin_inff() || t2[r] == __builtin_inff() || t3[r] == __builtin_inff();
            }
            const bool dead = __any(d0);
            if (!(dbg & 4)) n_pass += list_tile(acc, nth, I * kSuperH + 128u * (uint32_t)w + 32u * g, colbase, tc, dead) ? 1 : 0;  // (dbg 4: tiles consumed, not listed)
            continue;
          }
          if (done >= n_stages) break;
          __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (!more) break;
        li = li_n;
        h = h_n;
      }
    }
  }
  if (!producer) flush_queue();
  if (lane == 0) {
    if (n_blk) atomicAdd(stamps + 5, n_blk);
    if (n_pass) atomicAdd(stamps + 7, n_pass);
    if (n_sparse) atomicAdd(stamps + 9, n_sparse);
    if (n_pre) atomicAdd(stamps + 1, n_pre);
  }
}
