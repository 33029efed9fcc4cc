// mjpl_fused.h -- ONE launch for the float32 filter of an edge batch (included at the end of mjpl_filter.h).
//
// k_filter_endpoints_pw -> k_filter_items_pw are two persistent grids with a kernel boundary between them: the
// endpoint grid's second, one-third-full round of tiles (4 096 tiles on 3 072 wave slots at config-3 size) and
// the item grid's last round drain the chip twice per batch, each kernel stages its table copy again, and the
// items travel through HBM (8 B written and read per waypoint, plus the step fractions).  Three engines taking
// batches in turns (bench.py round 3) showed what that costs: 0.243 ms per batch on one stream, 0.169 ms with the
// drains filled by another batch's kernels.  This kernel fills them with the batch's own work.
//
// A workgroup is resident for the whole launch (one per CU: twelve waves, three per SIMD) and owns a WORK POOL in
// its LDS.  A wave that takes an endpoint tile (64 edges: waypoint count by the reference's recurrence, float32
// check of the endpoint QB -- the statements of k_filter_endpoints_pw) appends one ENTRY (edge, interior
// waypoints K, step fraction) per surviving edge to the pool; a wave that takes an item tile claims the next 64
// waypoints of the pool -- consecutive waypoints of consecutive entries, so the lanes of a wave still see
// neighbouring poses -- and checks them (the statements of k_filter_items_pw).  Both kinds of tile go through ONE
// call of the per-configuration check, so the kernel holds one copy of a model's straight-line code.  Endpoint
// tiles are dealt statically for the first round (tile = wave number) and through sharded device counters after
// it; items never leave the workgroup that produced them, so nothing is handed from one workgroup to another
// inside the launch: no cross-XCD visibility question arises (MI355X_MICROARCH.md, inter-workgroup visibility),
// and a workgroup never waits for one that is not resident.  What crosses waves of a workgroup is ordered by the
// pool's lock: a workgroup-scope release (every store of the holder has left the CU) before the unlock, an
// acquire behind the lock -- so the producer's `valid[e] = 1` is in the XCD's L2 before a consumer's `valid[e] = 0`.
//
// What is left for the launch behind it (k_tail) is unchanged: undecided geom pairs, undecided whole edges, and
// the walking list -- here only edges with more than `kmax` interior waypoints.
#pragma once

namespace mjpl {

constexpr int kFusedWaves = 12;   // waves of a workgroup (one workgroup per CU at three waves per SIMD)
constexpr int kFusedShards = 8;   // device counters the dynamic endpoint tiles are dealt from
constexpr int kStatusFusedTimeout = 4;

struct FusedArgs {
  const int *ip; int nip;
  const float *fp; int nfp;
  const double *QA, *QB; long long E; int layout;
  float tol; double step;
  uint8_t *valid; int32_t *first_bad;
  int *status, *ulist, *ucount;
  UndecidedConfigs uc;
  int *llist, *lcount;
  int *claim; int gen;           // per-edge claim words of the undecided-edge list (see k_filter_items)
  double *tstep;                 // [E] scratch: an edge's step fraction between its count and its pool entry
  int *item_count, *surv_count;  // statistics: kItemRegions counters each, kCounterStride apart
  int *tiles;                    // kFusedShards counters, kCounterStride apart: dynamic endpoint tiles
  int *zero_next;
  int kmax;                      // edges with more interior waypoints take the walking list
  int ring;                      // entries of a workgroup's pool (>= 64 per wave)
  int policy;                    // bit 0: item tiles before further endpoint tiles (default: endpoint tiles first)
  // diagnostic builds (-DMJPL_FUSED_DEBUG; null otherwise): eight 64-bit words per wave of the grid --
  // endpoint tiles, item tiles, polls while waiting; clocks in endpoint tiles, item tiles, waiting, the lock, in all
  unsigned long long *dbg;
};

// control words of a workgroup's pool (LDS)
enum : int { RG_LOCK = 0, RG_HEAD, RG_TAIL, RG_OFF, RG_PENDING, RG_RESERVED, RG_EXHAUSTED, RG_WORDS = 8 };

__host__ __device__ constexpr size_t fused_wave_bytes(int nplan, int nsave, size_t qbytes) {
  return wave_slice_bytes(nplan, nsave, sizeof(float), qbytes) + 2 * 64 * sizeof(int);  // + (edge, index) of the wave's items
}
inline size_t fused_lds_bytes(int nwaves, int nplan, int nsave, size_t ntab, bool mbox, int ring) {
  const size_t q = mbox ? WaveQueue<float, true>::bytes() : WaveQueue<float, false>::bytes();
  return (size_t)nwaves * fused_wave_bytes(nplan, nsave, q) + ((ntab * sizeof(float) + 7) & ~(size_t)7) +
         (size_t)ring * (sizeof(double) + 2 * sizeof(int)) + RG_WORDS * sizeof(int);
}
// the float64 walking rows of the waypoint count lie over a wave's slice
inline bool fused_fits(int nplan, int nsave, bool mbox) {
  const size_t q = mbox ? WaveQueue<float, true>::bytes() : WaveQueue<float, false>::bytes();
  return (size_t)2 * nplan * 64 * sizeof(double) <= wave_slice_bytes(nplan, nsave, sizeof(float), q);
}

template <class Spec, int MAXS, bool WBOX, bool MBOX, int NW>
__global__ void __launch_bounds__(NW * 64, (kMinWaves<Spec, MAXS>))
k_edges_fused(FusedArgs a) {
  static_assert(kQueued<float, MAXS>, "the fused kernel serves the queued interpreter");
  extern __shared__ double smem[];
  zero_counters(a.zero_next);
  const int nplan = StaticNplan<Spec>::value ? StaticNplan<Spec>::value : a.ip[H_NPLAN];
  const int nsave = a.ip[H_NSAVE];
  const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
  const int R = a.ring;
  // ---- LDS: [wave slices, each followed by the wave's item table | table copy | pool entries | pool control]
  const size_t qbytes = WaveQueue<float, MBOX>::bytes();
  WaveLds<float, MBOX> w;
  w.bytes = wave_slice_bytes(nplan, nsave, sizeof(float), qbytes);
  const size_t wbytes = w.bytes + 2 * 64 * sizeof(int);
  w.base = reinterpret_cast<char *>(smem) + (size_t)wv * wbytes;
  w.col = reinterpret_cast<float *>(w.base);
  w.save = reinterpret_cast<float *>(w.base + (((size_t)nplan * 64 * sizeof(float) + 7) & ~(size_t)7));
  w.qmem = reinterpret_cast<char *>(w.save) + (((size_t)nsave * 7 * 64 * sizeof(float) + 7) & ~(size_t)7);
  int *w_edge = reinterpret_cast<int *>(w.base + w.bytes), *w_idx = w_edge + 64;
  char *shared = reinterpret_cast<char *>(smem) + (size_t)NW * wbytes;
  w.ltab = reinterpret_cast<float *>(shared);
  for (int k = threadIdx.x; k < a.nfp; k += blockDim.x) w.ltab[k] = a.fp[k];
  double *r_ts = reinterpret_cast<double *>(shared + (((size_t)a.nfp * sizeof(float) + 7) & ~(size_t)7));
  int *r_edge = reinterpret_cast<int *>(r_ts + R), *r_K = r_edge + R;
  volatile int *ctl = r_K + R;
  // endpoint tiles: the first round is dealt statically -- wave v of workgroup b takes tile v * grid + b, so a
  // launch with fewer tiles than waves spreads them over the workgroups -- the rest through the shard counters
  const long long ntile = (a.E + 63) >> 6;
  const long long nstatic = ntile < (long long)NW * gridDim.x ? ntile : (long long)NW * gridDim.x;
  const int ndyn = (int)(ntile - nstatic);
  if (threadIdx.x < RG_WORDS) {
    // (the static tiles' room in the pool is set aside from the start: up to 64 entries each)
    int nstat_wg = 0;
    for (int v = 0; v < NW; v++) nstat_wg += ((long long)v * gridDim.x + blockIdx.x < nstatic) ? 1 : 0;
    ctl[threadIdx.x] = threadIdx.x == RG_EXHAUSTED ? (ndyn == 0 ? 1 : 0) : (threadIdx.x == RG_RESERVED ? 64 * nstat_wg : 0);
  }
  __syncthreads();  // the table copy and the pool's control words; from here on a wave is on its own

  auto lock = [&]() {
    if (lane == 0)
      while (atomicCAS(const_cast<int *>(&ctl[RG_LOCK]), 0, 1) != 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  };
  auto unlock = [&]() {
    // (release: every store of this wave -- pool entries in LDS, verdict bytes on their way to L2 -- has been
    // performed before another wave can see the lock free)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(const_cast<int *>(&ctl[RG_LOCK]), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_wave_barrier();
  };
  auto shard_tiles = [&](int q) { return q < ndyn ? (ndyn - q + kFusedShards - 1) / kFusedShards : 0; };
  // the next dynamic endpoint tile, or -1 when every shard is empty
  auto dequeue = [&]() -> long long {
    int q = (int)(blockIdx.x % kFusedShards);
    for (;;) {
      int j = 0;
      if (lane == 0) j = atomicAdd(a.tiles + q * kCounterStride, 1);
      j = __builtin_amdgcn_readfirstlane(j);
      if (j < shard_tiles(q)) return nstatic + q + (long long)kFusedShards * j;
      int left = 0;  // this shard is empty: look at all of them at once (a counter only grows)
      if (lane < kFusedShards)
        left = shard_tiles(lane) - __hip_atomic_load(a.tiles + lane * kCounterStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long m = __ballot(left > 0);
      if (m == 0ull) return -1;
      const int s = q + 1 == kFusedShards ? 0 : q + 1;
      const unsigned long long hi = m >> s;
      q = hi ? s + (int)__builtin_ctzll(hi) : (int)__builtin_ctzll(m);
    }
  };

  const bool fits = (size_t)2 * nplan * 64 * sizeof(double) <= w.bytes;  // (the launcher has checked)
  double *qe = reinterpret_cast<double *>(w.base) + lane;
  double *qx = qe + (size_t)nplan * 64;
  float *qw = w.col + lane;
  long long my_static = (long long)wv * gridDim.x + blockIdx.x;
  if (my_static >= nstatic) my_static = -1;
  int stat_items = 0, stat_surv = 0;
  int spins = 0;
#ifdef MJPL_FUSED_DEBUG
  unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long dg_t0 = wall_clock64();
#define MJPL_DG(k, v) dg[k] += (v)
#define MJPL_DG_NOW() wall_clock64()
#else
#define MJPL_DG(k, v)
#define MJPL_DG_NOW() 0ull
#endif
  enum : int { ACT_WAIT = 0, ACT_EXIT, ACT_EP, ACT_EP_DYN, ACT_IT };

  for (;;) {
    // ---- what next?  Decided (and, for an item tile, claimed) under the pool's lock.
    int act = ACT_WAIT;
    int ed = -1, idx = 0;   // item tile: this lane's waypoint
    double ts = 0.0;
    const unsigned long long dg_a = MJPL_DG_NOW();
    (void)dg_a;
    lock();
    {
      const int head = ctl[RG_HEAD], tail = ctl[RG_TAIL], pending = ctl[RG_PENDING], reserved = ctl[RG_RESERVED];
      const bool exhausted = ctl[RG_EXHAUSTED] != 0;
      const bool room = R - (tail - head) - reserved >= 64;  // an endpoint tile adds up to 64 entries
      int take = 0;
      if (my_static >= 0) {
        act = ACT_EP;
      } else if (a.policy & 1) {
        if (pending >= 64) take = 64;
        else if (!exhausted && room) act = ACT_EP_DYN;
        else if (pending > 0 && exhausted) take = pending;
      } else {
        if (!exhausted && room) act = ACT_EP_DYN;
        else if (pending >= 64) take = 64;
        else if (pending > 0 && exhausted) take = pending;
      }
      if (act == ACT_EP_DYN) {
        if (lane == 0) ctl[RG_RESERVED] = reserved + 64;
      } else if (act == ACT_EP) {
        // (reserved at the start)
      } else if (take > 0) {
        // claim the next `take` waypoints of the pool: entries head, head + 1, ... (the first `off` waypoints of
        // entry `head` were claimed before); every entry still holds at least one, so 64 entries cover the claim
        act = ACT_IT;
        const int off = ctl[RG_OFF];
        const int navail = tail - head < 64 ? tail - head : 64;
        int slot = head + lane;
        slot = slot >= R ? slot - R : slot;  // (head < R is kept below; lane < 64 <= R)
        int kj = lane < navail ? r_K[slot] - (lane == 0 ? off : 0) : 0;
        int cum = kj;  // inclusive prefix sum over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int up = __shfl_up(cum, o);
          if (lane >= o) cum += up;
        }
        // lane = waypoint number `lane` of the claim: its entry is the first whose prefix exceeds it
        int lo = 0, hi = navail;
#pragma unroll
        for (int it = 0; it < 7; it++) {  // (every lane takes part in every shuffle)
          const int mid = (lo + hi) >> 1;
          const int c = __shfl(cum, mid < 63 ? mid : 63);
          const bool go = lo < hi, below = c <= lane;
          lo = (go && below) ? mid + 1 : lo;
          hi = (go && !below) ? mid : hi;
        }
        const int ent = lo < navail ? lo : (navail > 0 ? navail - 1 : 0);
        const int before_raw = __shfl(cum, ent > 0 ? ent - 1 : 0);
        const int before = ent > 0 ? before_raw : 0;
        const bool mine = lane < take;
        int es = head + ent;
        es = es >= R ? es - R : es;
        ed = mine ? r_edge[es] : -1;
        ts = mine ? r_ts[es] : 0.0;
        idx = lane - before + (ent == 0 ? off : 0) + 1;
        // the pool's new head: entries whose prefix is within the claim are used up
        const int adv = (int)__builtin_popcountll(__ballot(lane < navail && cum <= take));
        const int cum_adv = __shfl(cum, adv > 0 ? adv - 1 : 0);
        if (lane == 0) {
          int nh = head + adv, nt = tail;
          if (nh >= R) { nh -= R; nt -= R; }  // (head stays below R; tail - head is what counts)
          ctl[RG_HEAD] = nh;
          ctl[RG_TAIL] = nt;
          ctl[RG_OFF] = adv > 0 ? take - cum_adv : off + take;
          ctl[RG_PENDING] = pending - take;
        }
      } else if (act == ACT_WAIT && exhausted && reserved == 0 && pending == 0) {
        act = ACT_EXIT;
      }
    }
    unlock();
    const unsigned long long dg_b = MJPL_DG_NOW();
    (void)dg_b;
    MJPL_DG(6, dg_b - dg_a);
    if (act == ACT_EXIT) break;
    if (act != ACT_WAIT) spins = 0;
    if (act == ACT_WAIT) {
      __builtin_amdgcn_s_sleep(64);
      MJPL_DG(2, 1);
      MJPL_DG(5, MJPL_DG_NOW() - dg_b);
      if (++spins > (1 << 22)) {  // (seconds: report, do not hang)
        if (lane == 0) atomicOr(a.status, kStatusFusedTimeout);
        break;
      }
      continue;
    }
    long long tile = -1;
    if (act == ACT_EP) {
      tile = my_static;
      my_static = -1;
    } else if (act == ACT_EP_DYN) {
      tile = dequeue();
      if (tile < 0) {  // nothing left anywhere: give the reservation back, tell the workgroup
        lock();
        if (lane == 0) {
          ctl[RG_RESERVED] = ctl[RG_RESERVED] - 64;
          ctl[RG_EXHAUSTED] = 1;
        }
        unlock();
        continue;
      }
    }
    const bool ep = tile >= 0;
    // ---- the tile's configurations -> this wave's binary32 columns
    const long long i = ep ? tile * 64 + lane : (long long)(ed >= 0 ? ed : 0);
    bool active = ep ? i < a.E : ed >= 0;
    bool finite = true;
    int K = 0;
    if (ep) {
      for_rows(a.QA, a.QB, a.E, i, nplan, a.layout, active, [&](int, double x, double y) {
        finite = finite && (fabs(x) <= 1.79769313486231570815e+308) && (fabs(y) <= 1.79769313486231570815e+308);
      });
      // the waypoint COUNT first (see k_filter_endpoints): float64 rows laid over the wave's slice
      if (fits) {
        load_columns(qe, 64, a.QB, a.E, i, nplan, a.layout, active);
        double tsw;
        K = count_waypoints_ts(a.ip, a.QA, a.E, i, a.step, a.layout, active && finite, qe, 64, qx, 64, a.kmax, nplan, tsw);
        if (K > 0) a.tstep[i] = tsw;  // (parked in memory across the check: two registers less to keep alive)
        wave_lds_fence();
      } else {
        K = -1;
      }
      load_columns(qw, 64, a.QB, a.E, i, nplan, a.layout, active);
      active = active && finite;
    } else {
      w_edge[lane] = ed >= 0 ? ed : 0;
      w_idx[lane] = idx;
      // the waypoint in closed form (see count_waypoints): QA + min(idx * step / |QB - QA|, 1) (QB - QA),
      // rounded to binary32 as the check would round it anyway
      double tt = active ? (double)idx * ts : 0.0;
      tt = tt < 1.0 ? tt : 1.0;
      for_rows(a.QA, a.QB, a.E, i, nplan, a.layout, true, [&](int k, double x, double y) {
        qw[k * 64] = active ? (float)fma(tt, y - x, x) : 0.0f;
      });
    }
    wave_lds_fence();
    // ---- ONE call site of the per-configuration check for both kinds of tile
    const EdgeSource src = {ep ? a.QB : a.QA, a.QB, a.E, a.layout, ep ? 0.0 : a.step, nullptr};
    const int code = check_wave<MAXS, WBOX, MBOX, Spec>(a.ip, a.fp, w, active, a.tol, ep ? i : (long long)lane, a.uc,
                                                        ep ? (const int *)nullptr : w_edge, ep ? (const int *)nullptr : w_idx, src);
    wave_lds_fence();
    if (!ep) {
      if (active && code != V_NONE) {
        if (code == V_CONTACT) {
          a.valid[ed] = 0;
          if (a.first_bad) atomicMin(reinterpret_cast<unsigned *>(a.first_bad) + ed, (unsigned)idx);
        } else if (atomicExch(&a.claim[ed], a.gen) != a.gen) {
          a.ulist[atomicAdd(a.ucount, 1)] = ed;  // (the exact edge kernel redoes the whole edge, once)
        }
      }
      stat_items += (int)__builtin_popcountll(__ballot(active));
      MJPL_DG(1, 1);
      MJPL_DG(4, MJPL_DG_NOW() - dg_b);
      continue;
    }
    // ---- endpoint tile: verdicts, then the survivors' entries
    bool survive = active && code != V_CONTACT;
    if (i < a.E) {
      if (!finite) {
        a.valid[i] = 0;
        if (a.first_bad) a.first_bad[i] = -2;
        atomicOr(a.status, kStatusNonFinite);
      } else if (code == V_CONTACT) {
        a.valid[i] = 0;
        if (a.first_bad) a.first_bad[i] = 0;
      } else if (code == V_UNSURE) {  // (queued interpreter: the whole edge goes to the exact edge kernel)
        a.ulist[atomicAdd(a.ucount, 1)] = (int)i;
        survive = false;
      } else {
        a.valid[i] = 1;  // so far; item tiles and the pair re-check may clear it
        if (a.first_bad) a.first_bad[i] = -1;
      }
    }
    if (survive && K < 0) a.llist[atomicAdd(a.lcount, 1)] = (int)i;  // too long (or too many columns): walking kernel
    const bool entry = survive && K > 0;
    const unsigned long long me = __ballot(entry);
    const int nent = (int)__builtin_popcountll(me);
    int total = entry ? K : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    const double tse = entry ? a.tstep[i] : 0.0;
    stat_surv += (int)__builtin_popcountll(__ballot(survive));
    lock();
    {
      const int tail = ctl[RG_TAIL];
      if (entry) {
        int slot = tail + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(me >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)me, 0u));
        slot = slot >= R ? slot - R : slot;
        slot = slot >= R ? slot - R : slot;  // (tail < 2 R: head < R and at most R entries)
        r_edge[slot] = (int)i;
        r_K[slot] = K;
        r_ts[slot] = tse;
      }
      if (lane == 0) {
        ctl[RG_TAIL] = tail + nent;
        ctl[RG_PENDING] = ctl[RG_PENDING] + total;
        ctl[RG_RESERVED] = ctl[RG_RESERVED] - 64;
      }
    }
    unlock();
    MJPL_DG(0, 1);
    MJPL_DG(3, MJPL_DG_NOW() - dg_b);
  }
#ifdef MJPL_FUSED_DEBUG
  if (a.dbg && lane == 0) {
    dg[7] = wall_clock64() - dg_t0;
    unsigned long long *row = a.dbg + ((size_t)blockIdx.x * NW + wv) * 8;
    for (int k = 0; k < 8; k++) row[k] = dg[k];
  }
#endif
  // statistics of the launch (mjpl_filter_last_items / _last_interior_edges)
  if (lane == 0) {
    const int region = (int)(blockIdx.x % kItemRegions);
    if (stat_items) atomicAdd(a.item_count + region * kCounterStride, stat_items);
    if (stat_surv) atomicAdd(a.surv_count + region * kCounterStride, stat_surv);
  }
}

// grid of the fused kernel: what the device holds at once, never more workgroups than endpoint tiles
template <class K>
inline unsigned fused_grid(K kernel, int block, size_t lds, long long ntile) {
  static thread_local const void *last_k = nullptr;
  static thread_local size_t last_lds = 0;
  static thread_local int last_dev = -1, resident = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (last_k != reinterpret_cast<const void *>(kernel) || last_lds != lds || last_dev != dev) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kernel), block, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    resident = per_cu * cus;
    last_k = reinterpret_cast<const void *>(kernel); last_lds = lds; last_dev = dev;
  }
  return (unsigned)(ntile < 1 ? 1 : (ntile < resident ? ntile : resident));
}

}  // namespace mjpl
