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
constexpr int kFusedMaxPool = 4096;  // entries of a workgroup's pool at most (two rounds of the 64-way search)
constexpr int kStatusFusedTimeout = 4;

struct FusedArgs {
  const int *ip; int nip;
  const float *fp; int nfp;      // the float32 tables (k_edges_fused)
  const double *dp; int ndp;     // the float64 tables (k_edges_fused_f64)
  const double *QA, *QB; long long E; int layout;
  float tol; double step;
  uint8_t *valid; int32_t *first_bad;
  int *status, *ulist, *ucount;
  UndecidedConfigs uc;
  int *llist, *lcount;
  int *claim; int gen;           // per-edge claim words of the undecided-edge list (see k_filter_items)
  double *tstep;                 // [E] scratch: an edge's step fraction between its count and its pool entry
  int *item_count, *surv_count;  // statistics: kItemRegions counters each, kCounterStride apart
  int *cert_count;               // ... and one: surviving edges the certificate spared the waypoint checks
  int cert;                      // 1: endpoint tiles of a two-round launch try the edge certificate (generated checks with kCert)
  int *zero_next;
  long long tile0, tile1;        // this launch serves endpoint tiles [tile0, tile1) of the batch (64 edges each)
  int kmax;                      // edges with more interior waypoints take the walking list
  int pool;                      // slots of a workgroup's ring of entries (>= 64 per wave)
  int policy;                    // bit 0: item tiles before further endpoint tiles (default: endpoint tiles first)
  // != 0: the endpoint is an item, too -- an endpoint tile only counts waypoints, and every edge enters the pool with
  // checks 0 .. K.  Costs the interior checks of the edges whose endpoint is in contact (never made otherwise) and
  // saves a batch that does not fill the chip the latency of one of its two rounds of checks.
  int single;
  // diagnostic builds (-DMJPL_FUSED_DEBUG; null otherwise): eight 64-bit words per wave of the grid --
  // endpoint tiles, item tiles, polls while waiting; clocks in endpoint tiles, item tiles, waiting, deciding, in all
  unsigned long long *dbg;
};

// control words of a workgroup's pool (LDS)
// RG_HEAD (64 bits, 8-byte aligned): [waypoints claimed so far | number of the first entry that still holds an
// unclaimed waypoint]; RG_NEED + w: the first entry wave w is still reading (kFusedIdle: none)
enum : int { RG_HEAD = 0, RG_LOCK = 2, RG_ENT, RG_COMMIT, RG_NEXT, RG_PRODUCED, RG_RESERVED, RG_NEED = 8, RG_WORDS = 8 + 16 };
constexpr int kFusedIdle = 0x7fffffff;

__host__ __device__ constexpr size_t fused_wave_bytes(int nplan, int nsave, size_t qbytes, bool cert = false) {
  // + (edge, index) of the wave's items; cert (a library built with the edge certificate): + |QB - QA| per planning
  // column of an endpoint tile's lanes, as binary16
  return wave_slice_bytes(nplan, nsave, sizeof(float), qbytes) + 2 * 64 * sizeof(int) +
         (cert ? (((size_t)nplan * 64 * sizeof(_Float16) + 7) & ~(size_t)7) : 0);
}
constexpr size_t kFusedEntryBytes = sizeof(double) + 2 * sizeof(int);  // step fraction, edge, first waypoint number
inline size_t fused_lds_bytes(int nwaves, int nplan, int nsave, size_t ntab, bool mbox, int pool, bool cert = false) {
  const size_t q = mbox ? WaveQueue<float, true>::bytes() : WaveQueue<float, false>::bytes();
  return (size_t)nwaves * fused_wave_bytes(nplan, nsave, q, cert) + ((ntab * sizeof(float) + 7) & ~(size_t)7) +
         (size_t)pool * kFusedEntryBytes + RG_WORDS * sizeof(int);
}
template <class Spec> struct SpecCert { static constexpr bool value = Spec::kCert; };
template <> struct SpecCert<void> { static constexpr bool value = false; };
// the float64 walking rows of the waypoint count lie over a wave's slice
inline bool fused_fits(int nplan, int nsave, bool mbox) {
  const size_t q = mbox ? WaveQueue<float, true>::bytes() : WaveQueue<float, false>::bytes();
  return (size_t)2 * nplan * 64 * sizeof(double) <= wave_slice_bytes(nplan, nsave, sizeof(float), q);
}

// How the pool works.  A workgroup OWNS the endpoint tiles tile0 + b, tile0 + b + grid, ... (b its number): nothing is
// dealt between workgroups, so every workgroup carries the same number of edges (+-64) and their waypoints -- a greedy
// device-wide queue of endpoint tiles was measured first: whoever takes an endpoint tile also takes its ~2.5 item tiles
// of later work, and the workgroups that were quick at the start ended 40 % after the slow ones.  Its waves take the
// tiles in order (an LDS counter).  The survivors of a tile become ENTRIES [edge, number of its first waypoint in the
// pool's running count, step fraction], numbered as they come and kept in a ring of `pool` slots; they are appended
// under a lock that only producers take (sixteen times per workgroup at config-3 size).  `commit` = waypoints
// entered so far.  Consumers take no lock: the pool's HEAD is one 64-bit word [waypoints claimed, first entry that
// still holds an unclaimed one]; a wave reads it, reads the first numbers of the 64 entries from there on (every entry
// holds at least one waypoint, so they cover a claim; the slot behind the last entry holds the number the next one
// will start at, so an entry's end is always the next slot's number), works out where a claim of up to 64 waypoints
// ends, and moves
// the head there with one compare-and-swap -- which fails, and is tried again, exactly when another wave moved it
// first.  Its lanes then find their entries among those 64 by a binary search through the wave's registers.
// A slot of the ring is written again only when no wave can still read it: a consumer publishes the number of the
// first entry of its claim (RG_NEED + wave) BEFORE its compare-and-swap and withdraws it after its last read; a
// producer that wants room reads the head FIRST and the published numbers after it, and everything below the
// smallest of them is free (a consumer whose publication the producer missed read the head before the producer
// did, so its entries are not below the producer's bound -- or its compare-and-swap fails).  A wave starts an
// endpoint tile only with 64 free slots set aside for it (RG_RESERVED), so a commit never waits; when the ring is
// full it takes an item tile instead.  With a ring that holds all entries of the launch none of this is needed
// and the reservation is skipped (`a.pool >= 64 * tiles of the workgroup`).

// A workgroup's pool and one wave's view of it (every member function is called by whole waves).
struct WorkPool {
  double *r_ts;
  int *r_edge, *r_first;
  volatile int *ctl;
  int R, ntw, lane, wv, nw;
  bool reuse;
  int my_first;  // the wave's first endpoint tile (ordinal), dealt statically
  enum : int { LEAVE = 0, WAIT, RETRY, TILE, ITEMS };

  static __host__ __device__ constexpr size_t bytes(int pool) { return (size_t)pool * kFusedEntryBytes + RG_WORDS * sizeof(int); }

  // carve the pool at `mem` (8-byte aligned) and set its control words; the caller's workgroup barrier publishes them
  __device__ __forceinline__ void init(char *mem, int pool, long long tile0, long long tile1, int nwaves) {
    R = pool;
    nw = nwaves;
    lane = (int)(threadIdx.x & 63);
    wv = (int)(threadIdx.x >> 6);
    r_ts = reinterpret_cast<double *>(mem);
    r_edge = reinterpret_cast<int *>(r_ts + R);
    r_first = r_edge + R;
    ctl = r_first + R;
    // this workgroup's endpoint tiles: ordinal n -> tile0 + b + n * grid
    const long long span = tile1 - tile0 - (long long)blockIdx.x;
    ntw = span > 0 ? (int)((span + gridDim.x - 1) / gridDim.x) : 0;
    // (the first `nwaves` tiles: one per wave, their room in the ring set aside from the start)
    if (threadIdx.x < RG_WORDS)
      ctl[threadIdx.x] = threadIdx.x == RG_NEXT ? (ntw < nw ? ntw : nw)
                       : threadIdx.x == RG_RESERVED ? 64 * (ntw < nw ? ntw : nw)
                       : threadIdx.x >= RG_NEED ? kFusedIdle : 0;
    if (threadIdx.x == 0) r_first[0] = 0;  // (first[ENT] == COMMIT from the start)
    reuse = (long long)ntw * 64 + 1 > (long long)R;  // slots are written more than once in this launch
    my_first = wv < ntw ? wv : -1;
  }
  __device__ __forceinline__ long long tile_of(int n, long long tile0) const { return tile0 + blockIdx.x + (long long)n * gridDim.x; }

  __device__ __forceinline__ void lock() const {
    if (lane == 0)
      while (atomicCAS(const_cast<int *>(&ctl[RG_LOCK]), 0, 1) != 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  __device__ __forceinline__ void unlock() const {
    // (release: every store of this wave -- pool entries in LDS, verdict bytes on their way to L2 -- has been
    // performed before another wave can see the lock free)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(const_cast<int *>(&ctl[RG_LOCK]), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_wave_barrier();
  }

  // What next for this wave?  TILE: endpoint tile `tile_n` (ordinal in the workgroup).  ITEMS: waypoints
  // [.., + take) of the pool, this lane's in (ed, idx, ts): edge, 1-based position among the edge's items, step
  // fraction; ed < 0 on the lanes beyond `take`.  WAIT: nothing to take yet, look again later.  RETRY: another wave
  // was quicker, look again at once.  LEAVE: the workgroup's work is done.
  __device__ __forceinline__ int next(int policy, int &tile_n, int &take, int &ed, int &idx, double &ts) {
    tile_n = -1;
    take = 0;
    ed = -1;
    idx = 0;
    ts = 0.0;
    if (my_first >= 0) {
      tile_n = my_first;
      my_first = -1;
      return TILE;
    }
    // (read in this order: `produced` == ntw means `commit` is final)
    const int produced = __hip_atomic_load(const_cast<int *>(&ctl[RG_PRODUCED]), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned long long head = __hip_atomic_load(reinterpret_cast<unsigned long long *>(const_cast<int *>(&ctl[RG_HEAD])),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int commit = __hip_atomic_load(const_cast<int *>(&ctl[RG_COMMIT]), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int claimed = (int)(unsigned)(head >> 32), he = (int)(unsigned)head;
    const int next_tile = ctl[RG_NEXT];
    const int avail = commit - claimed;
    const bool tiles_left = next_tile < ntw;
    bool want_tile = tiles_left && !((policy & 1) && avail >= 64);
    if (want_tile && reuse) {
      // room for the tile's entries (up to 64)?  Entries below the head's and below every entry a wave is still
      // reading are free.  (head first, the publications after it: see above)
      lock();
      const unsigned long long hd = __hip_atomic_load(reinterpret_cast<unsigned long long *>(const_cast<int *>(&ctl[RG_HEAD])),
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      int lowest = lane < nw ? ctl[RG_NEED + lane] : kFusedIdle;
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        const int other = __shfl_xor(lowest, o);
        lowest = other < lowest ? other : lowest;
      }
      lowest = __builtin_amdgcn_readfirstlane(lowest);
      const int free_below = lowest < (int)(unsigned)hd ? lowest : (int)(unsigned)hd;
      const int reserved = ctl[RG_RESERVED];
      const bool room = R - (ctl[RG_ENT] - free_below) - reserved >= 65;  // (64 entries and the slot behind them)
      if (room && lane == 0) ctl[RG_RESERVED] = reserved + 64;
      unlock();
      want_tile = room;
    }
    if (want_tile) {
      int n = 0;
      if (lane == 0) n = atomicAdd(const_cast<int *>(&ctl[RG_NEXT]), 1);
      n = __builtin_amdgcn_readfirstlane(n);
      if (n < ntw) {
        tile_n = n;
        return TILE;
      }
      if (reuse) {  // (another wave was quicker)
        lock();
        if (lane == 0) ctl[RG_RESERVED] = ctl[RG_RESERVED] - 64;
        unlock();
      }
      return RETRY;
    }
    if (avail >= 64 || (avail > 0 && (!tiles_left || reuse))) {
      // (with no tile left to take -- or no room for one -- a partial claim beats waiting for the producers; sharing a
      // workgroup's last waypoints out evenly, so that all its waves end together, was measured: part-filled tiles
      // cost nearly what full ones do -- 40 % more tiles, 9 % more time per batch)
      const int want = avail < 64 ? avail : 64;
      const int h = claimed;
      // the first numbers of the entries behind the head's: g[j] = first[he + 1 + j] (the end of entry he + j)
      if (lane == 0) ctl[RG_NEED + wv] = he;  // BEFORE the compare-and-swap
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      const int nent = ctl[RG_ENT];  // (read after `commit`: covers every waypoint below it)
      const int hslot = he % R;
      int gs = hslot + 1 + lane;
      gs = gs >= R ? gs - R : gs;
      gs = gs >= R ? gs - R : gs;
      const int ge = he + 1 + lane;
      // (slot `nent` holds the number the next entry will start at -- the end of the last one: the producers keep
      // first[ENT] == COMMIT.  `nent` may be newer than `commit`: the numbers beyond it are real, and the claim
      // ends at `commit` whatever lies behind)
      const int g = ge <= nent ? r_first[gs] : 0x7fffffff;
      const int f0 = r_first[hslot];  // (he < nent: a waypoint is pending)
      const int adv = (int)__builtin_popcountll(__ballot(g <= h + want));  // entries used up by this claim
      int got = 0;
      if (lane == 0)
        got = atomicCAS(reinterpret_cast<unsigned long long *>(const_cast<int *>(&ctl[RG_HEAD])), head,
                        ((unsigned long long)(unsigned)(h + want) << 32) | (unsigned)(he + adv)) == head ? 1 : 0;
      if (!__builtin_amdgcn_readfirstlane(got)) {  // (another wave was quicker: look again)
        if (lane == 0) ctl[RG_NEED + wv] = kFusedIdle;
        return RETRY;
      }
      // this lane's entry: he + (number of j with g[j] <= its waypoint number)
      const int s = h + lane;
      int jl = 0, jh = 64;  // invariant: g[jl - 1] <= s < g[jh - 1]   (g[-1] = f0 <= h)
      int fl = f0;
#pragma unroll
      for (int it = 0; it < 6; it++) {  // (every lane takes part in every shuffle)
        const int mid = (jl + jh) >> 1;  // 1 .. 63
        const int gm = __shfl(g, mid - 1);
        const bool le = gm <= s;
        jl = le ? mid : jl;
        fl = le ? gm : fl;
        jh = le ? jh : mid;
      }
      if (lane < want) {
        int es = hslot + jl;
        es = es >= R ? es - R : es;
        ed = r_edge[es];
        ts = r_ts[es];
        idx = s - fl + 1;
      }
      wave_lds_fence();  // (the reads above before the withdrawal)
      if (lane == 0) ctl[RG_NEED + wv] = kFusedIdle;
      take = want;
      return ITEMS;
    }
    return (produced >= ntw && avail == 0) ? LEAVE : WAIT;
  }

  // An endpoint tile is through: its surviving lanes (`entry`) enter the pool with K items each.
  __device__ __forceinline__ void commit(bool entry, int K, long long i, double tse) {
    const unsigned long long me = __ballot(entry);
    const int nent = (int)__builtin_popcountll(me);
    int incl = entry ? K : 0;  // this lane's first waypoint number, relative to the tile's: an exclusive prefix sum
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o);
      if (lane >= o) incl += up;
    }
    const int total = __shfl(incl, 63);
    lock();
    {
      const int e0 = ctl[RG_ENT], f0 = ctl[RG_COMMIT];
      if (entry) {
        int slot = e0 % R + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(me >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)me, 0u));
        slot = slot >= R ? slot - R : slot;
        r_edge[slot] = (int)i;
        r_first[slot] = f0 + incl - K;
        r_ts[slot] = tse;
      }
      if (lane == 0) {  // the slot behind the last entry: where the next one will start (first[ENT] == COMMIT)
        int send = e0 % R + nent;
        send = send >= R ? send - R : send;
        r_first[send] = f0 + total;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the entries (and this tile's verdict bytes) before the counts
      if (lane == 0) {
        ctl[RG_ENT] = e0 + nent;
        __hip_atomic_store(const_cast<int *>(&ctl[RG_COMMIT]), f0 + total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (reuse) ctl[RG_RESERVED] = ctl[RG_RESERVED] - 64;
        __hip_atomic_store(const_cast<int *>(&ctl[RG_PRODUCED]), ctl[RG_PRODUCED] + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    unlock();
  }
};

#ifdef MJPL_FUSED_DEBUG
#define MJPL_DG_DECL unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long dg_t0 = wall_clock64()
#define MJPL_DG(k, v) dg[k] += (v)
#define MJPL_DG_NOW() wall_clock64()
#define MJPL_DG_FLUSH(a, nwaves)                                                          \
  if ((a).dbg && (threadIdx.x & 63) == 0) {                                               \
    dg[7] = wall_clock64() - dg_t0;                                                       \
    unsigned long long *row = (a).dbg + ((size_t)blockIdx.x * (nwaves) + (threadIdx.x >> 6)) * 8; \
    for (int k = 0; k < 8; k++) row[k] = dg[k];                                           \
  }
#else
#define MJPL_DG_DECL
#define MJPL_DG(k, v)
#define MJPL_DG_NOW() 0ull
#define MJPL_DG_FLUSH(a, nwaves)
#endif

template <class Spec, int MAXS, bool WBOX, bool MBOX, int NW>
__global__ void __launch_bounds__(NW * 64, (kMinWaves<Spec, MAXS>))
k_edges_fused(FusedArgs a) {
  static_assert(kQueued<float, MAXS>, "the fused kernel serves the queued interpreter");
  extern __shared__ double smem[];
  zero_counters(a.zero_next);
  const int nplan = StaticNplan<Spec>::value ? StaticNplan<Spec>::value : a.ip[H_NPLAN];
  const int nsave = a.ip[H_NSAVE];
  const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
  // ---- LDS: [wave slices, each followed by the wave's item table | table copy | pool entries | pool control]
  const size_t qbytes = WaveQueue<float, MBOX>::bytes();
  WaveLds<float, MBOX> w;
  w.bytes = wave_slice_bytes(nplan, nsave, sizeof(float), qbytes);
  constexpr bool kCertBuild = SpecCert<Spec>::value;
  const size_t wbytes = fused_wave_bytes(nplan, nsave, qbytes, kCertBuild);
  w.base = reinterpret_cast<char *>(smem) + (size_t)wv * wbytes;
  w.col = reinterpret_cast<float *>(w.base);
  w.save = reinterpret_cast<float *>(w.base + (((size_t)nplan * 64 * sizeof(float) + 7) & ~(size_t)7));
  w.qmem = reinterpret_cast<char *>(w.save) + (((size_t)nsave * 7 * 64 * sizeof(float) + 7) & ~(size_t)7);
  int *w_edge = reinterpret_cast<int *>(w.base + w.bytes), *w_idx = w_edge + 64;
  double *wts = reinterpret_cast<double *>(w_edge);  // an endpoint tile's step fractions: the same 512 bytes (8-byte aligned: the slice is)
  _Float16 *wadq = reinterpret_cast<_Float16 *>(w_idx + 64) + lane;  // [nplan][64]: |QB - QA| of the lane's edge, rounded UP to binary16 (the certificate; as binary32 the twelve waves' rows no longer fit the LDS beside the pool)
  // The edge certificate.  An endpoint tile of a two-round launch checks QB with every bounding cull widened
  // by what the pair can move while the planning joints go from QB back to QA, and every candidate's narrowphase routine
  // also says whether the pair is clear by that much: an edge all of whose pairs are never enters the pool -- none of its
  // waypoints can be in contact (planning/utils.py:188-216 would find them all free).  A certificate only ever says
  // "free"; verdicts and first-bad indices are what they were.
  const bool cert_tile = kCertBuild && a.cert != 0 && !a.single;
  int stat_cert = 0;
  char *shared = reinterpret_cast<char *>(smem) + (size_t)NW * wbytes;
  w.ltab = reinterpret_cast<float *>(shared);
  for (int k = threadIdx.x; k < a.nfp; k += blockDim.x) w.ltab[k] = a.fp[k];
  WorkPool pool;
  pool.init(shared + (((size_t)a.nfp * sizeof(float) + 7) & ~(size_t)7), a.pool, a.tile0, a.tile1, NW);
  __syncthreads();  // the table copy and the pool's control words; from here on a wave is on its own

  const bool fits = (size_t)2 * nplan * 64 * sizeof(double) <= w.bytes;  // (the launcher has checked)
  double *qe = reinterpret_cast<double *>(w.base) + lane;
  double *qx = qe + (size_t)nplan * 64;
  float *qw = w.col + lane;
  int stat_items = 0, stat_surv = 0;
  int spins = 0;
  MJPL_DG_DECL;

  for (;;) {
    // ---- what next?
    const unsigned long long dg_a = MJPL_DG_NOW();
    (void)dg_a;
    int tile_n, take, ed, idx;
    double ts;
    const int kind = pool.next(a.policy, tile_n, take, ed, idx, ts);
    if (kind == WorkPool::LEAVE) break;
    if (kind == WorkPool::RETRY) continue;
    if (kind == WorkPool::WAIT) {  // nothing to take (the last endpoint tiles of the workgroup are still being checked), or a lost race
      __builtin_amdgcn_s_sleep(32);
      MJPL_DG(2, 1);
      MJPL_DG(5, MJPL_DG_NOW() - dg_a);
      if (++spins > (1 << 22)) {  // (seconds: report, do not hang)
        if (lane == 0) atomicOr(a.status, kStatusFusedTimeout);
        break;
      }
      continue;
    }
    spins = 0;
    const bool ep = kind == WorkPool::TILE;
    const unsigned long long dg_b = MJPL_DG_NOW();
    (void)dg_b;
    MJPL_DG(6, dg_b - dg_a);
    // ---- the tile's configurations -> this wave's binary32 columns
    const long long tile = ep ? pool.tile_of(tile_n, a.tile0) : -1;
    const long long i = ep ? tile * 64 + lane : (long long)(ed >= 0 ? ed : 0);
    bool active = ep ? i < a.E : ed >= 0;
    bool finite = true;
    int K = 0;
    if (ep) {
      if (fits) {
        // ONE pass over the edge's rows: QA -> the walking row, QB -> the end row (float64, laid over the wave's
        // slice), then the waypoint COUNT by the reference's recurrence (see k_filter_endpoints), then the end row
        // becomes the check's binary32 columns in place: column k lies inside float64 row k / 2, which has been
        // read by the time it is written (eight columns at a time: all reads of a group before its writes)
        bool at_end = true;
        for_rows(a.QA, a.QB, a.E, i, nplan, a.layout, active, [&](int k, double x, double y) {
          finite = finite && (fabs(x) <= 1.79769313486231570815e+308) && (fabs(y) <= 1.79769313486231570815e+308);
          qx[k * 64] = x;
          qe[k * 64] = y;
          at_end = at_end && (x == y);
          if constexpr (kCertBuild) wadq[k * 64] = (_Float16)((float)fabs(y - x) * 1.002f);  // (no less than |y - x|: the conversions round by 2^-24 and 2^-11; beyond 65 504: +inf, never certified)
        });
        double tsw;
        K = count_waypoints_walk(a.ip, a.step, active && finite, at_end, qe, 64, qx, 64, a.kmax, nplan, tsw);
        wts[lane] = tsw;  // (parked across the check -- two registers less to keep alive -- in the wave's item table, which an endpoint tile does not use; round 4 parked it in HBM: 2 MB written and read per launch)
        wave_lds_fence();
        for (int k0 = 0; k0 < nplan; k0 += 8) {
          double v[8];
#pragma unroll
          for (int j = 0; j < 8; j++) v[j] = k0 + j < nplan ? qe[(k0 + j) * 64] : 0.0;
          wave_lds_fence();
#pragma unroll
          for (int j = 0; j < 8; j++)
            if (k0 + j < nplan) qw[(k0 + j) * 64] = finite ? (float)v[j] : 0.0f;  // (a NaN / inf row never reaches the check, not even on a lane that only keeps company: a NaN passes the plane culls' negated compare)
          wave_lds_fence();
        }
      } else {  // too many columns for the float64 rows: every surviving edge takes the walking list
        for_rows(a.QA, a.QB, a.E, i, nplan, a.layout, active, [&](int, double x, double y) {
          finite = finite && (fabs(x) <= 1.79769313486231570815e+308) && (fabs(y) <= 1.79769313486231570815e+308);
        });
        K = -1;
        wts[lane] = 0.0;  // (the walking list redoes the edge; its one item here is check 0, the endpoint, whatever the step)
        load_columns(qw, 64, a.QB, a.E, i, nplan, a.layout, active && finite);
      }
      active = active && finite;
    } else {
      if (a.single) idx -= 1;  // (an edge's items are its checks 0 .. K: 0 is the endpoint)
      w_edge[lane] = ed >= 0 ? ed : 0;
      w_idx[lane] = idx;
      // the waypoint in closed form (see count_waypoints): QA + min(idx * step / |QB - QA|, 1) (QB - QA),
      // rounded to binary32 as the check would round it anyway; check 0 of a single-round launch: QB itself
      double tt = active ? (double)idx * ts : 0.0;
      tt = tt < 1.0 ? tt : 1.0;
      const bool at_end = a.single && idx == 0;
      for_rows(a.QA, a.QB, a.E, i, nplan, a.layout, true, [&](int k, double x, double y) {
        qw[k * 64] = active ? (at_end ? (float)y : (float)fma(tt, y - x, x)) : 0.0f;
      });
    }
    wave_lds_fence();
    // ---- ONE call site of the per-configuration check for both kinds of tile
    const EdgeSource src = {ep ? a.QB : a.QA, a.QB, a.E, a.layout, ep ? 0.0 : a.step, nullptr, a.single};
    int code = V_NONE;
    if (!(ep && a.single))  // (an endpoint tile of a single-round launch only counts)
      code = check_wave<MAXS, WBOX, MBOX, Spec>(a.ip, a.fp, w, active, a.tol, ep ? i : (long long)lane, a.uc,
                                                ep ? (const int *)nullptr : w_edge, ep ? (const int *)nullptr : w_idx, src,
                                                (ep && fits && cert_tile) ? wadq : (const _Float16 *)nullptr);
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
    const bool certified = active && code == V_CLEAR;  // (free, and every pair clear of contact by what it can move along the edge)
    if (code == V_CLEAR) code = V_NONE;
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
    if (survive && K < 0 && !certified) a.llist[atomicAdd(a.lcount, 1)] = (int)i;  // too long (or too many columns): walking kernel
    if (a.single) K = K < 0 ? 1 : K + 1;  // (the endpoint is check 0 of the edge's items; of a walking edge the only one)
    const bool entry = survive && K > 0 && !certified;
    stat_cert += (int)__builtin_popcountll(__ballot(certified && survive && K != 0));
    const double tse = entry ? wts[lane] : 0.0;
    stat_surv += (int)__builtin_popcountll(__ballot(survive));
    pool.commit(entry, K, i, tse);
    MJPL_DG(0, 1);
    MJPL_DG(3, MJPL_DG_NOW() - dg_b);
  }
  MJPL_DG_FLUSH(a, NW);
  // statistics of the launch (mjpl_filter_last_items / _last_interior_edges)
  if (lane == 0) {
    const int region = (int)(blockIdx.x % kItemRegions);
    if (stat_items) atomicAdd(a.item_count + region * kCounterStride, stat_items);
    if (stat_surv) atomicAdd(a.surv_count + region * kCounterStride, stat_surv);
    if (stat_cert) atomicAdd(a.cert_count, stat_cert);
  }
}

// ---- the float64 path with the same launch shape (filter off: mjpl_set_filter(e, 0), or a model the filter refuses) ----
// k_check_edges gives every edge a lane for all of its checks: a wave lives as long as its longest edge, lanes whose
// edge ended at its endpoint idle through every later waypoint, and with 64 unrelated configurations per wave most
// pairs' narrowphase runs for a lane or two.  Here the float64 checks go through the pool: the endpoints of a tile
// (exact verdicts: nothing is left undecided), then the interior waypoints of the survivors as items -- consecutive
// waypoints of an edge on neighbouring lanes, which pass the same bounding culls, so that the interpreter's "some lane
// needs this routine" runs it for many lanes at once.  A waypoint is the reference's own: rebuilt by the recurrence
// from QA (k steps for waypoint k), never the closed form the float32 filter may test -- so edges of more than
// kFusedF64Kmax interior waypoints take the walking kernel (k_check_edges over the walking list), whose lane carries
// the waypoint along.  Verdicts are those of k_check_edges bit for bit: same statements on the same values.
constexpr int kFusedF64Waves = 8;   // 186 .. 256 VGPRs: two waves per SIMD
constexpr int kFusedF64Kmax = 31;
// QUEUED: the float64 check through the candidate queues (run_config_queued<double>: culls lane per configuration,
// narrowphase with full lanes) instead of the immediate interpreter; a wave's queues follow its rows.
__host__ __device__ constexpr size_t fused_f64_wave_bytes(int nplan, int nsave, bool queued = false) {
  const size_t rows = (size_t)nplan * 64 * sizeof(double), saves = (size_t)nsave * 7 * 64 * sizeof(double);
  return rows + (saves > rows ? saves : rows) + (queued ? WaveQueue<double, false>::bytes() : 0);  // [columns / end row | pose saves / walking row | queues]
}
inline size_t fused_f64_lds_bytes(int nwaves, int nplan, int nsave, int pool, bool queued = false) {
  return (size_t)nwaves * fused_f64_wave_bytes(nplan, nsave, queued) + WorkPool::bytes(pool);
}

// XF: void = the interpreting statement; else a library built with MJPL_SPEC_F64=1 brings the check as straight-line code
// (struct ExactFull, mjpl_amd/specialise.py: generate_full_exact -- the interpreter's float64 FK with the constants folded in,
// its culls with the partners' rows as literals, its pushes, drains and routines): the same candidates reach the same
// routines, the same verdicts.  An experiment the default libraries do not carry: it runs no faster than the interpreter
// (0.4497 against 0.4473 ms per 262 144 edges: a float64 literal costs two scalar moves where the interpreter's one
// sixteen-value scalar load brings four partners' rows; profiles/README.md, round 5).
template <int MAXS, bool WBOX, bool MBOX, int NW, bool QUEUED = false, class XF = void>
__global__ void __launch_bounds__(NW * 64)
k_edges_fused_f64(FusedArgs a) {
  static_assert(!QUEUED || (!MBOX && MAXS <= 16), "the queued float64 check serves the slot files of 4 / 8 / 16, no moving boxes");
  static_assert(std::is_void<XF>::value || QUEUED, "a generated float64 check goes through the candidate queues");
  extern __shared__ double smem[];
  zero_counters(a.zero_next);
  const int nplan = a.ip[H_NPLAN], nsave = a.ip[H_NSAVE];
  const int lane = threadIdx.x & 63, wv = (int)(threadIdx.x >> 6);
  const size_t wbytes = fused_f64_wave_bytes(nplan, nsave, QUEUED);
  double *col = reinterpret_cast<double *>(reinterpret_cast<char *>(smem) + (size_t)wv * wbytes) + lane;  // [nplan][64]
  double *aux = col + (size_t)nplan * 64;  // pose saves during a check, the walking / end row between checks
  WaveQueue<double, false> wq;
  if constexpr (QUEUED) wq.carve(reinterpret_cast<char *>(smem) + (size_t)(wv + 1) * wbytes - WaveQueue<double, false>::bytes());
  PatchSink sink = {};  // (nothing is handed on: the exact check decides everything itself)
  WorkPool pool;
  pool.init(reinterpret_cast<char *>(smem) + (size_t)NW * wbytes, a.pool, a.tile0, a.tile1, NW);
  __syncthreads();
  const int *perm = a.ip + a.ip[H_OFF_PERM];
  FkOut none = {};
  int stat_items = 0, stat_surv = 0, spins = 0;
  MJPL_DG_DECL;
  for (;;) {
    const unsigned long long dg_a = MJPL_DG_NOW();
    (void)dg_a;
    int tile_n, take, ed, idx;
    double ts;
    const int kind = pool.next(a.policy, tile_n, take, ed, idx, ts);
    if (kind == WorkPool::LEAVE) break;
    if (kind == WorkPool::RETRY) continue;
    if (kind == WorkPool::WAIT) {
      __builtin_amdgcn_s_sleep(32);
      MJPL_DG(2, 1);
      MJPL_DG(5, MJPL_DG_NOW() - dg_a);
      if (++spins > (1 << 22)) {
        if (lane == 0) atomicOr(a.status, kStatusFusedTimeout);
        break;
      }
      continue;
    }
    spins = 0;
    const bool ep = kind == WorkPool::TILE;
    const unsigned long long dg_b = MJPL_DG_NOW();
    (void)dg_b;
    MJPL_DG(6, dg_b - dg_a);
    const long long i = ep ? pool.tile_of(tile_n, a.tile0) * 64 + lane : (long long)(ed >= 0 ? ed : 0);
    bool active = ep ? i < a.E : ed >= 0;
    bool finite = true;
    int K = 0;
    if (ep) {
      // QB -> the columns (it is check 0 and the end of the walk), QA -> the walking row; the waypoint count
      bool at_end = true;
      for_rows(a.QA, a.QB, a.E, i, nplan, a.layout, active, [&](int k, double x, double y) {
        finite = finite && (fabs(x) <= 1.79769313486231570815e+308) && (fabs(y) <= 1.79769313486231570815e+308);
        aux[k * 64] = x;
        col[k * 64] = y;
        at_end = at_end && (x == y);
      });
      double tsw;
      K = count_waypoints_walk(a.ip, a.step, active && finite, at_end, col, 64, aux, 64, a.kmax, nplan, tsw);
      if (!finite)  // (never a NaN / inf row in a check, not even on a lane that keeps company: see k_filter_endpoints)
        for (int k = 0; k < nplan; k++) col[k * 64] = 0.0;
      active = active && finite;
    } else {
      // waypoint idx of edge ed by the reference's recurrence: idx steps from QA towards QB (planning/utils.py:182-185,
      // the statements of edge_body); the end row sits in `aux`, the walking waypoint in the columns
      for_rows(a.QA, a.QB, a.E, i, nplan, a.layout, true, [&](int k, double x, double y) {
        col[k * 64] = x;
        aux[k * 64] = y;
      });
      for (int n = 0; __ballot(active && n < idx) != 0ull; n++) {
        if (active && n < idx) {
          double sq = 0;
          for (int k = 0; k < nplan; k++) {
            const int c = perm[k];
            const double d = aux[c * 64] - col[c * 64];
            sq = sq + d * d;
          }
          const double mag = sqrt(sq);
          const double sm = a.step < mag ? a.step : mag;
          for (int k = 0; k < nplan; k++) {
            const double d = aux[k * 64] - col[k * 64];
            col[k * 64] = col[k * 64] + (d / mag) * sm;
          }
        }
      }
    }
    wave_lds_fence();
    bool hit;
    if constexpr (!std::is_void<XF>::value)
      hit = XF::run(a.dp, col, 64, aux, 64, active, wq, (int)i, sink) == V_CONTACT;
    else if constexpr (QUEUED)  // (the tables the drains gather from: the float64 tables in global memory, 23 KB, cache resident)
      hit = run_config_queued<double, MAXS, WBOX, false>((IP)a.ip, (DP)a.dp, a.dp, col, 64, aux, 64, active, 0.0, wq, (int)i, sink) == V_CONTACT;
    else
      hit = run_config<double, MAXS, false, WBOX, MBOX>((IP)a.ip, (DP)a.dp, col, 64, aux, 64, active, 0.0, none, i) == V_CONTACT;
    wave_lds_fence();
    if (!ep) {
      if (active && hit) {
        a.valid[ed] = 0;
        if (a.first_bad) atomicMin(reinterpret_cast<unsigned *>(a.first_bad) + ed, (unsigned)idx);
      }
      stat_items += (int)__builtin_popcountll(__ballot(active));
      MJPL_DG(1, 1);
      MJPL_DG(4, MJPL_DG_NOW() - dg_b);
      continue;
    }
    const bool survive = active && !hit;
    if (i < a.E) {
      if (!finite) {
        a.valid[i] = 0;
        if (a.first_bad) a.first_bad[i] = -2;
        atomicOr(a.status, kStatusNonFinite);
      } else {
        a.valid[i] = hit ? 0 : 1;  // (a survivor's: so far; its item tiles or the walking kernel may clear it)
        if (a.first_bad) a.first_bad[i] = hit ? 0 : -1;
      }
    }
    if (survive && K < 0) a.llist[atomicAdd(a.lcount, 1)] = (int)i;  // long (or degenerate): the walking kernel
    stat_surv += (int)__builtin_popcountll(__ballot(survive));
    pool.commit(survive && K > 0, K, i, 0.0);
    MJPL_DG(0, 1);
    MJPL_DG(3, MJPL_DG_NOW() - dg_b);
  }
  MJPL_DG_FLUSH(a, NW);
  if (lane == 0) {
    const int region = (int)(blockIdx.x % kItemRegions);
    if (stat_items) atomicAdd(a.item_count + region * kCounterStride, stat_items);
    if (stat_surv) atomicAdd(a.surv_count + region * kCounterStride, stat_surv);
  }
}

// Launch the fused kernel over the whole batch: the grid is what the device holds at once, never more workgroups
// than endpoint tiles (a launch of a few tiles spreads them: one per workgroup, whose other waves take its waypoints).
template <class K>
inline hipError_t fused_launch(K kernel, int nwaves, size_t lds, FusedArgs a, hipStream_t st) {
  static thread_local const void *last_k = nullptr;
  static thread_local size_t last_lds = 0;
  static thread_local int last_dev = -1, resident = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (last_k != reinterpret_cast<const void *>(kernel) || last_lds != lds || last_dev != dev) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kernel), nwaves * 64, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    resident = per_cu * cus;
    last_k = reinterpret_cast<const void *>(kernel); last_lds = lds; last_dev = dev;
  }
  const long long ntile = (a.E + 63) / 64;
  a.tile0 = 0;
  a.tile1 = ntile;
  hipLaunchKernelGGL(kernel, dim3((unsigned)(ntile < resident ? ntile : resident)), dim3(nwaves * 64), lds, st, a);
  return hipGetLastError();
}

}  // namespace mjpl
