#!/usr/bin/env python3
"""Headline benchmark: validated RRT edges/sec on the Franka 7-DoF 16-obstacle scene.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (the edge kernel: endpoint + interior waypoints of
every edge, FK + narrowphase per waypoint) over one batch of E synthetic edges that is
already resident in HBM.  Weak scaling: every rank owns its own E edges (independent units,
no data-path collective).  Rank 0 prints ONE JSON line.

The timed K steps are launched through mjpl_time_edges_stages_dev: K back-to-back launches on the
engine's own stream between two HIP events, every fourth launch also carrying one event after each
of its kernels; roofline.achieved is the longest kernel's algorithmic bytes over its mean duration
from those events.

N > 1: one process per GPU.  `python bench.py --gpus N` starts its N rank processes itself -- children,
before anything in this process has touched a GPU -- and `torch.distributed.run` may start them just as
well (RANK / LOCAL_RANK / WORLD_SIZE in the environment).  Either way PyTorch is not imported: rank 0
publishes the library's 128-byte ncclUniqueId through a file, every rank attaches an RCCL communicator
to its engine (mjpl_comm_init), and the barrier and the max over ranks of the elapsed time are
all-gathers of eight bytes per rank on that communicator (mjpl_allgather_dev).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "validated RRT edges/sec, Franka 7-DoF 16-geom scene, at 1/2/4/8 MI355X"
EDGES_PER_GPU = 262144       # BASELINE.json configs[2]
EPS, STEP = 0.05, 0.01       # rrt.py:30 epsilon; 4 interior waypoints + endpoint per edge
BYTES_PER_EDGE = 113         # 2 x 7 x 8 B read + 1 B verdict written
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# MI355X_MICROARCH.md "Wave scheduling": a wave64 VALU instruction issues over 2 cycles on a SIMD-32,
# 4 SIMDs per CU, 256 CUs, 2.4 GHz max clock -> wave-instructions per second the chip can issue
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2.0
CLOCK_WARM_STEPS = 300       # untimed launches before anything is measured (~60 ms: the clocks)
SHORT_REGION_STEPS = 64      # a timed region shorter than this carries no HIP events (they come from untimed launches)
IN_TURNS_MIN_STEPS = 600     # the three-engine `in_turns` figure is only taken in runs at least this long
FP64_VECTOR_PEAK = 78.6e12   # MI355X_MICROARCH.md: FP64 vector 78.6 TFLOP/s


class World:
    """This process's place among the ranks of one node.  What every workload needs from the other ranks -- a
    barrier either side of the timed region and the slowest rank's time -- travels over a Unix-domain socket
    that rank 0 opens (no data-path collective, no GPU runtime involved: BASELINE's north_star shards the
    edges, and the ranks never exchange any).  The planner workload has one real exchange step per round:
    for it `attach(eng, rccl=True)` also opens the library's RCCL communicator, whose ncclUniqueId travels
    over the same socket."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.distributed = self.world > 1
        self.eng = None
        self.rccl = False
        self._peers = None   # rank 0: sockets of ranks 1 .. W-1, by rank
        self._sock = None    # other ranks: the socket to rank 0
        self._listener = None

    @property
    def device(self) -> int:
        """The GPU of this rank: its local rank -- modulo the devices there are when MJPL_BENCH_SHARE_GPU=1
        (several ranks on one GPU: exercises the launch, the rendezvous and the timing protocol on a
        one-GPU box; never a measurement)."""
        if os.environ.get("MJPL_BENCH_SHARE_GPU") == "1":
            from mjpl_amd import engine
            return self.local_rank % max(1, engine.device_count())
        return self.local_rank

    def _rendezvous_path(self):
        p = os.environ.get("MJPL_BENCH_RDZV")
        if p:
            return p
        # started by torch.distributed.run (or any launcher that sets RANK / WORLD_SIZE): all ranks of one
        # launch share their parent process and the MASTER_PORT, two launches do not
        return os.path.join(os.environ.get("TMPDIR", "/tmp"),
                            "mjpl_bench_%s_%s_%d.sock" % (os.environ.get("MASTER_PORT", "0"),
                                                          os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.getppid()))

    @staticmethod
    def _recv(sock, n):
        buf = b""
        while len(buf) < n:
            part = sock.recv(n - len(buf))
            if not part:
                raise ConnectionError("bench.py: a rank closed the rendezvous socket")
            buf += part
        return buf

    def connect(self):
        """Rank 0 listens, the others connect (retrying until it does) and say who they are."""
        if not self.distributed or self._peers is not None or self._sock is not None:
            return
        import socket
        path = self._rendezvous_path()
        if self.rank == 0:
            try:
                os.remove(path)
            except OSError:
                pass
            self._listener = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            self._listener.bind(path)
            self._listener.listen(self.world)
            self._listener.settimeout(600)
            self._peers = {}
            while len(self._peers) < self.world - 1:
                c, _ = self._listener.accept()
                c.settimeout(600)
                self._peers[int.from_bytes(self._recv(c, 4), "little")] = c
            try:
                os.remove(path)  # (everybody is in: the name is not needed any more)
            except OSError:
                pass
        else:
            t0 = time.time()
            while True:
                sk = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    sk.connect(path)
                    break
                except OSError:
                    sk.close()
                    if time.time() - t0 > 600:
                        sys.exit(f"bench.py rank {self.rank}: rank 0 never opened {path}")
                    time.sleep(0.01)
            sk.settimeout(600)
            sk.sendall(self.rank.to_bytes(4, "little"))
            self._sock = sk

    def exchange(self, payload: bytes) -> list:
        """All ranks' payloads (equal lengths), in rank order; also a barrier."""
        if not self.distributed:
            return [payload]
        self.connect()
        if self.rank == 0:
            parts = [payload] + [self._recv(self._peers[r], len(payload)) for r in range(1, self.world)]
            blob = b"".join(parts)
            for r in range(1, self.world):
                self._peers[r].sendall(blob)
            return parts
        self._sock.sendall(payload)
        blob = self._recv(self._sock, len(payload) * self.world)
        return [blob[k * len(payload):(k + 1) * len(payload)] for k in range(self.world)]

    def attach(self, eng, rccl: bool = False):
        """Meet the other ranks; with rccl=True also open the library's communicator on `eng` (collective)."""
        self.eng = eng
        self.connect()
        if rccl and not self.rccl:
            from mjpl_amd import engine
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            uid = self.exchange(engine.comm_unique_id() if self.rank == 0 else bytes(128))[0]
            eng.comm_init(uid, self.rank, self.world)
            self.rccl = True
        self.gather(0.0)

    def gather(self, x: float) -> np.ndarray:
        """All ranks' values of x, in rank order (also a barrier)."""
        return np.array([np.frombuffer(b, dtype=np.float64)[0] for b in self.exchange(np.float64(x).tobytes())])

    def barrier(self):
        if self.eng is not None:
            self.eng.sync()
        self.gather(0.0)

    def close(self):
        if self.eng is not None:
            self.barrier()
            if self.rccl:
                self.eng.comm_destroy()
                self.rccl = False
        for sk in ([self._sock] if self._sock else []) + list((self._peers or {}).values()) + ([self._listener] if self._listener else []):
            try:
                sk.close()
            except OSError:
                pass
        self._sock = self._peers = self._listener = None


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N rank processes as children of this
    one (which never touches a GPU), wait for them, pass rank 0's JSON line through."""
    import subprocess
    import tempfile
    rdzv = os.path.join(tempfile.gettempdir(), "mjpl_bench_%d_%d.sock" % (os.getpid(), int(time.time() * 1e3) & 0xFFFFFF))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MJPL_BENCH_RDZV=rdzv,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    live = list(procs)
    while live:  # (a rank that dies must not leave the others waiting at the rendezvous)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = max(rc, abs(code))
                for q in live:
                    q.terminate()
        time.sleep(0.05)
    try:
        os.remove(rdzv)
    except OSError:
        pass
    return rc


def make_edges(model, qidx, n, seed):
    """Seeded synthetic edges: q_a uniform in the joint ranges,
    direction ~ normalised N(0, I), ||q_b - q_a|| = eps, clipped to the joint ranges."""
    rng = np.random.default_rng(seed)
    lo, hi = model.jnt_range[qidx, 0], model.jnt_range[qidx, 1]
    qa = rng.uniform(lo, hi, size=(n, len(qidx)))
    d = rng.normal(size=qa.shape)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    qb = np.clip(qa + EPS * d, lo, hi)
    return qa, qb


def host_cpu():
    """CPU model string and the affinity of this process (SURVEY.md 8d 'CPU baseline timing')."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        aff = sorted(os.sched_getaffinity(0))
    except AttributeError:
        aff = list(range(os.cpu_count() or 1))
    runs, lo = [], None
    for c in aff + [None]:  # compress to ranges
        if lo is None:
            lo = prev = c
        elif c is not None and c == prev + 1:
            prev = c
        else:
            runs.append(f"{lo}-{prev}" if prev != lo else f"{lo}")
            lo = prev = c
    # a container may own fewer CPUs than it can see: the cgroup quota bounds what threads can use
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", ):
        try:
            q, per = open(path).read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
        except (OSError, ValueError):
            pass
    if quota is None:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    usable = len(aff) if quota is None else max(1, min(len(aff), int(quota + 0.5)))
    return model, usable, ",".join(runs), (None if quota is None else f"cgroup cpu quota {quota:g} CPUs of {len(aff)} visible")


def cpu_baseline(model, qidx, base, qa, qb):
    """The oracle (a port, not the reference) timed on this box's host cores on a bounded
    sample of the same edges; reported, never the target."""
    from oracle import pyoracle
    orc = pyoracle.Oracle(model, planning_qidx=qidx, qpos_base=base)
    cpu, cores, aff, quota = host_cpu()
    pilot = min(4096, len(qa))
    orc.valid_edges(qa[:64], qb[:64], STEP, nthreads=1)
    t0 = time.perf_counter()
    orc.valid_edges(qa[:pilot], qb[:pilot], STEP, nthreads=1)
    per_edge = (time.perf_counter() - t0) / pilot
    n = len(qa)
    orc.valid_edges(qa[:n], qb[:n], STEP, nthreads=cores)  # starts the worker pool, warms the caches
    # bounded sample: `reps` passes over this rank's batch with every core this process may use
    # busy (one thread per CPU of the cgroup quota: more threads only get throttled), about
    # 20 core-seconds of oracle work and never less than ~1.5 s of wall time
    reps = max(1, int(round(max(20.0 / (per_edge * n), 1.5 * cores / (per_edge * n)))))
    t0 = time.perf_counter()
    for _ in range(reps):
        v = orc.valid_edges(qa[:n], qb[:n], STEP, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": reps * n / dt, "unit": "edges/s", "cores": cores, "kind": "port",
            "cpu_model": cpu, "affinity": aff, "cpu_quota": quota, "single_thread_edges_per_s": 1.0 / per_edge,
            "sample": f"{reps} passes over the {n} edges of rank 0 ({reps * n * per_edge:.0f} core-seconds of "
                      f"single-thread work, {dt:.2f} s wall), oracle/libmjpl_oracle.so (gcc -O2 "
                      f"-ffp-contract=off), persistent pool of {cores} threads, 64-edge chunks"}, v, n


def profile_record(kernel, E, layout, filt=True, spec=True):
    """Counters of the committed rocprofv3 PMC passes of this same command (profiles/pmc.json,
    written by tools/pmc_collect.py): HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950
    correction of MI355X_MICROARCH.md "HBM") and wave-level instruction counts, keyed by the kernel AND
    the engine configuration it was captured under (float32 filter on / off, specialised kernels on /
    off): a record never speaks for another configuration's run.  {} if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc.json")) as f:
            return json.load(f).get("%s_%d_%s_f%d_s%d" % (kernel, E, layout, int(filt), int(spec)), {})
    except (OSError, ValueError):
        return {}


def kernel_time_this_run(eng, step, steps=32, mode=1):
    """(ms of the dominant kernel per step, launches per step) of `steps` untimed steps of THIS run: option "kernel_timer"
    (include/mjpl_hip.h) puts a HIP event on either side of every launch of the workload's dominant kernel, on the stream it
    is launched on.  mode 1: the configuration filter / projection rows / IK rows / the planner's generating kernel; 2:
    the nearest-neighbour scans."""
    eng.set_option("kernel_timer", mode)
    for _ in range(steps):
        step()
    ms, n = eng.get_option("kernel_timer_ms"), eng.get_option("kernel_timer_launches")
    eng.set_option("kernel_timer", 0)
    return (ms / steps, n / steps) if n > 0 and ms >= 0 else (None, None)


def issue_roofline(key, kernel, hbm, step_ms, launches_per_step=1.0, this_run=None):
    """`roofline` of a line whose dominant kernel is bound by vector-instruction issue: achieved = the kernel's wave-level
    vector instructions per step (SQ_INSTS_VALU per launch of the committed counter pass of this same command,
    profiles/pmc.json[key], x the launches per step) / the kernel's duration per step.  `this_run` = (ms per step, launches
    per step) from kernel_time_this_run: the duration is THIS run's, between HIP events (round 6; before, the committed
    trace's average -- a line could print a kernel longer than its step).  The record carries the digest of the kernel
    sources it was captured under; another digest is said so (`counters_stale`).  The HBM figure north_star asks for rides
    along as `hbm`.  No record: the HBM object alone, saying so."""
    rec = profile_key(key)
    if not rec.get("SQ_INSTS_VALU") or not rec.get("avg_ns"):
        return dict(hbm, note=f"no committed counter pass for {kernel} ({key} in profiles/pmc.json): the HBM figure only; the path is "
                              "issue / latency bound (SURVEY.md 8d)")
    iv = float(rec["SQ_INSTS_VALU"])
    if this_run and this_run[0]:
        ms_step, launches_per_step = float(this_run[0]), float(this_run[1])
        source = "HIP events around the kernel's launches in 32 untimed steps of THIS run (option kernel_timer)"
    else:
        ms_step = float(rec["avg_ns"]) * 1e-6 * launches_per_step
        source = "average duration of the kernel in the committed kernel trace (the counters' own profiling round), not this run"
    achieved = iv * launches_per_step / (ms_step * 1e-3)
    out = {"bound": "valu_issue", "kernel": kernel, "achieved": achieved, "peak": VALU_ISSUE_PEAK, "unit": "wave-instructions/s",
           "frac": achieved / VALU_ISSUE_PEAK, "traffic": rec.get("hbm_bytes_per_launch"),
           "wave_insts_valu_per_launch": iv, "kernel_ms_source": source,
           "launches_per_step": launches_per_step, "kernel_ms_per_step": ms_step, "step_ms_this_run": step_ms,
           "waves_per_launch": rec.get("SQ_WAVES"), "counters_source": rec.get("source"), "hbm": hbm}
    if step_ms < ms_step <= step_ms * 1.05:
        # (a step that IS its one kernel: between events the kernel reads a few percent longer than the un-instrumented step
        #  -- the events' own cost; the step's time is the kernel's)
        out["kernel_ms_between_events"] = ms_step
        ms_step = step_ms
        achieved = iv * launches_per_step / (ms_step * 1e-3)
        out.update(kernel_ms_per_step=ms_step, achieved=achieved, frac=achieved / VALU_ISSUE_PEAK)
    elif ms_step > step_ms:  # (a kernel cannot outlast the step it is part of: the figure is withheld rather than printed)
        out.update(frac=None, achieved=None, inconsistent=f"kernel {ms_step:.4f} ms per step > step {step_ms:.4f} ms")
    try:
        from mjpl_amd import build as _build
        stamp = "%016x" % _build.src_stamp()
        if rec.get("src_stamp") and rec["src_stamp"] != stamp:
            out["counters_stale"] = f"the counters were captured under kernel sources {rec['src_stamp']}, this build is {stamp}"
    except Exception:  # noqa: BLE001
        pass
    if rec.get("SQ_THREAD_CYCLES_VALU") and rec.get("SQ_ACTIVE_INST_VALU"):
        out["live_lanes_per_valu_instruction"] = float(rec["SQ_THREAD_CYCLES_VALU"]) / float(rec["SQ_ACTIVE_INST_VALU"])
    if rec.get("SQ_WAIT_ANY") and rec.get("SQ_WAVE_CYCLES"):
        out["waiting_fraction_of_wave_cycles"] = float(rec["SQ_WAIT_ANY"]) / float(rec["SQ_WAVE_CYCLES"])
    out["note"] = ("peak = 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md, Wave scheduling); "
                   "float64 Newton chains: what binds is the issue rate and the latency of one wave's dependent instructions, not HBM")
    return out


def profile_key(key):
    try:
        with open(os.path.join(ROOT, "profiles", "pmc.json")) as f:
            return json.load(f).get(key, {})
    except (OSError, ValueError):
        return {}


def flop_record():
    """The oracle's static operation count of one step of the headline workload (profiles/flops.json,
    written by tools/count_flops.py from oracle/libmjpl_oracle_count.so); {} if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "flops.json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def aggregate(world, n_units, steps, elapsed):
    """Whole-job rate: the units all ranks processed / the slowest rank's time."""
    slowest = float(world.gather(elapsed).max())
    return n_units * world.world * steps / slowest, slowest


def bench_configs(args, world):
    """BASELINE configs[1], per GPU: N = 65 536 Franka-P configurations, self-collision + floor,
    q ~ U[jnt_range] (default_rng(1 + rank)), verdicts checked against the oracle."""
    from mjpl_amd import engine, scenes
    model = scenes.franka_p(obstacles=False)
    qidx = scenes.planning_index(model, scenes.FRANKA_ARM_JOINTS)
    base = model.keyframe("home").qpos.copy()
    eng = engine.Engine(model, device=world.device)
    eng.set_planning(qidx, base)
    world.attach(eng)
    N = 65536
    Q = np.random.default_rng(1 + world.rank).uniform(model.jnt_range[qidx, 0], model.jnt_range[qidx, 1], size=(N, len(qidx)))
    h = np.ascontiguousarray(Q.T)
    dq, dv = eng.alloc(h.nbytes).upload(h), eng.alloc(N)
    if args.warmup > 0:
        eng.time_configs_dev(dq.ptr, N, engine.SOA, dv.ptr, args.warmup)
    this_run = kernel_time_this_run(eng, lambda: eng.time_configs_dev(dq.ptr, N, engine.SOA, dv.ptr, 1))
    world.barrier()
    t0 = time.perf_counter()
    ms = eng.time_configs_dev(dq.ptr, N, engine.SOA, dv.ptr, args.steps)
    world.barrier()
    elapsed = time.perf_counter() - t0
    value, slowest = aggregate(world, N, args.steps, elapsed)
    valid = dv.download(np.uint8, N)
    if world.rank == 0:
        out = {"metric": "validated configurations/sec, Franka-P self-collision (BASELINE configs[1])",
               "value": value, "unit": "configs/s", "n_gpus": world.world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": slowest / args.steps * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None,
               "dtype": "f32-filter+f64-exact" if eng.info()["filter_enabled"] else "f64", "data": "synthetic",
               "config": {"workload": f"configs[1]: Franka-P 7-DoF, self-collision + floor, {N} configurations/launch/GPU",
                          "valid_fraction": float(valid.mean()), "step_ms_hip_events": float(np.mean(ms)),
                          "parallelism": f"configuration-sharded x{world.world}, no data-path collective"},
               "roofline": issue_roofline(
                   f"k_filter_configs_configs{N}", "k_filter_configs",
                   {"bound": "hbm", "achieved": 57 * N / (float(np.mean(ms)) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": 57 * N / (float(np.mean(ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "traffic": None, "algorithmic_bytes_per_configuration": 57},
                   float(np.mean(ms)), this_run=this_run)}
        if not args.no_cpu_baseline:
            from oracle import pyoracle
            orc = pyoracle.Oracle(model, planning_qidx=qidx, qpos_base=base)
            cores = host_cpu()[1]  # (the cores this process may use: cgroup quota, affinity)
            t0 = time.perf_counter()
            v = orc.valid_configs(Q, nthreads=cores)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": N / dt, "unit": "configs/s", "cores": cores, "kind": "port",
                                   "sample": f"one pass over rank 0's {N} configurations, {cores} pthreads"}
            if not np.array_equal(v.astype(np.uint8), valid):
                sys.exit("bench.py: GPU verdicts differ from the CPU oracle")
        _flush_c_stdio()
        print(json.dumps(out), flush=True)
    world.close()
    return 0


def bench_next_rows(args, world):
    """One GPU's share of BASELINE configs[3] (PoseConstraint projections of 1 M / 8 samples) or
    configs[4] (128k / 8 IK seeds -> FK -> collision filter) on every rank; device-resident inputs for
    the projection, host buffers (PCIe included) for the IK seeds.  Independent units: no data-path
    collective.  Rank 0's results are checked against the oracle on a sample."""
    import mjpl_amd as mjpl
    from mjpl_amd import scenes
    from oracle import pyoracle
    model = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    q_home = model.keyframe("home").qpos.copy()
    cc = mjpl.CollisionConstraint(model, device=world.device)
    eng = cc.engine
    world.attach(eng)
    lo, hi = model.jnt_range[:, 0], model.jnt_range[:, 1]
    rng = np.random.default_rng(4 + world.rank)
    common = {"n_gpus": world.world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
              "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
    sharded = f"sharded x{world.world}, no data-path collective"
    if args.workload == "pose":
        frame = mjpl.site_pose(model, q_home, "ee_site", engine=eng)
        pc = mjpl.PoseConstraint(model, "ee_site", frame, roll=(-0.1, 0.1), pitch=(-0.1, 0.1), q_step=0.5, engine=eng)
        n = 131072
        Q_old = np.clip(q_home + rng.normal(scale=0.01, size=(n, model.nq)), lo, hi)
        d = rng.normal(size=(n, model.nq))
        d[:, 7:] = 0
        Q = np.clip(Q_old + 0.3 * d / np.linalg.norm(d, axis=1, keepdims=True), lo, hi)
        dqo, dq = eng.alloc(Q_old.nbytes).upload(Q_old), eng.alloc(Q.nbytes).upload(Q)
        dout, dok, dit = eng.alloc(Q.nbytes), eng.alloc(n), eng.alloc(4 * n)
        for _ in range(max(args.warmup, 1)):
            pc._proj.apply_dev(dqo.ptr, dq.ptr, n, dout.ptr, dok.ptr, dit.ptr)
        this_run = kernel_time_this_run(eng, lambda: pc._proj.apply_dev(dqo.ptr, dq.ptr, n, dout.ptr, dok.ptr, dit.ptr))
        eng.sync()
        world.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pc._proj.apply_dev(dqo.ptr, dq.ptr, n, dout.ptr, dok.ptr, dit.ptr)
        eng.sync()
        world.barrier()
        elapsed = time.perf_counter() - t0
        value, slowest = aggregate(world, n, args.steps, elapsed)
        out = None
        if world.rank == 0:
            out_q, ok, iters = dout.download(np.float64, n * model.nq).reshape(n, -1), dok.download(np.uint8, n), dit.download(np.int32, n)
            inv = frame.inverse()
            po = pyoracle.PoseOracle(model, "ee_site", (inv.wxyz_xyz[:4], inv.wxyz_xyz[4:]),
                                     [(-np.inf, np.inf)] * 3 + [(-0.1, 0.1)] * 2 + [(-np.inf, np.inf)], q_step=0.5)
            k = 8192
            cores = host_cpu()[1]
            t0 = time.perf_counter()
            ref, rok, rit = po.apply_batch(Q_old[:k], Q[:k], nthreads=cores)
            dtc = time.perf_counter() - t0
            same = (rok == ok[:k].astype(bool)) & (rit == iters[:k])
            if same.mean() < 0.99 or np.abs(ref[same] - out_q[:k][same]).max() > 1e-8:
                sys.exit("bench.py: GPU projections differ from the CPU oracle")
            row_bytes = 3 * 8 * model.nq + 1
            out = dict(common, metric="PoseConstraint projections/sec, Franka-P ee roll/pitch +-0.1 (one GPU's share of configs[3] per rank)",
                       value=value, unit="rows/s", ms_per_step=slowest / args.steps * 1e3,
                       config={"workload": f"{n} rows per launch per GPU, {np.abs(iters).mean():.1f} projection steps per row on average",
                               "accepted_fraction": float(ok.mean()), "parallelism": "row-" + sharded},
                       roofline=issue_roofline(
                           f"k_pose_apply_rows_pose{n}", "k_pose_apply_rows",
                           {"bound": "hbm", "achieved": n * row_bytes * args.steps / elapsed / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": n * row_bytes * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, "traffic": None,
                            "algorithmic_bytes_per_row": row_bytes},
                           slowest / args.steps * 1e3, this_run=this_run),
                       cpu_baseline={"value": k / dtc, "unit": "rows/s", "cores": cores, "kind": "port",
                                     "sample": f"first {k} rows of rank 0, {cores} pthreads"})
    else:
        solver = mjpl.HipIKSolver(model, joints, [], seed=3 + world.rank, num_seeds=16384, iterations=200, engine=eng)
        q_t = mjpl.random_config(model, q_home, joints, 5, [mjpl.JointLimitConstraint(model), cc])
        target = mjpl.site_pose(model, q_t, "ee_site", engine=eng)
        Q0 = solver._seeds(q_home, np.random.default_rng(3 + world.rank))

        def once():
            Qs, ok, its, err = eng.ik_solve("ee_site", target.translation(), target.rotation().wxyz, Q0, solver.movable,
                                            iterations=200, restarts=8, restart_seed=11)
            return Qs, ok, its, cc.valid_configs(Qs[ok])

        for _ in range(max(args.warmup, 1)):
            once()
        this_run = kernel_time_this_run(eng, lambda: eng.ik_solve("ee_site", target.translation(), target.rotation().wxyz, Q0, solver.movable,
                                                                 iterations=200, restarts=8, restart_seed=11), steps=8)
        world.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            Qs, ok, its, free = once()
        world.barrier()
        elapsed = time.perf_counter() - t0
        value, slowest = aggregate(world, len(Q0), args.steps, elapsed)
        out = None
        if world.rank == 0:
            orc = pyoracle.Oracle(model)
            sample = np.flatnonzero(ok)[:2048]
            if not np.array_equal(orc.valid_configs(Qs[sample], nthreads=8).astype(bool), free[:len(sample)]):
                sys.exit("bench.py: collision filter differs from the CPU oracle")
            # CPU baseline: the oracle's statement of the same iteration on a bounded sample of the seeds
            cpu, cores, aff, quota = host_cpu()
            k = 2048
            t0 = time.perf_counter()
            Qc, okc, itc, _ = pyoracle.ik_solve_batch(model, "ee_site", target.translation(), target.rotation().wxyz, Q0[:k],
                                                      solver.movable, iterations=200, restarts=8, restart_seed=11, nthreads=cores)
            freec = orc.valid_configs(Qc[okc], nthreads=cores)
            dtc = time.perf_counter() - t0
            if abs(float(okc.mean()) - float(ok[:k].mean())) > 0.05:
                sys.exit("bench.py: GPU and CPU IK converge on different fractions of the same seeds")
            cpu_ik = {"value": k / dtc, "unit": "seeds/s", "cores": cores, "kind": "port", "cpu_model": cpu, "affinity": aff,
                      "cpu_quota": quota, "sample": f"first {k} seeds of rank 0, {cores} threads, converged {okc.mean():.3f} "
                                                    f"(GPU on the same rows {ok[:k].mean():.3f}), {len(freec)} collision checks"}
            out = dict(common, metric="IK seeds/sec: damped-least-squares seeds -> FK -> collision filter (one GPU's share of configs[4] per rank)",
                       value=value, unit="seeds/s", ms_per_step=slowest / args.steps * 1e3,
                       config={"workload": f"{len(Q0)} seeds per launch per GPU, <= 200 iterations, host buffers (PCIe included)",
                               "converged_fraction": float(ok.mean()), "collision_free_of_converged": float(free.mean()),
                               "mean_iterations": float(its.mean()), "parallelism": "seed-" + sharded},
                       roofline=issue_roofline(
                           f"k_ik_solve_rows_ik{len(Q0)}", "k_ik_solve_rows",
                           {"bound": "hbm", "achieved": len(Q0) * 2 * 8 * model.nq * args.steps / elapsed / 1e9, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": len(Q0) * 2 * 8 * model.nq * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, "traffic": None,
                            "algorithmic_bytes_per_seed": 2 * 8 * model.nq},
                           slowest / args.steps * 1e3, this_run=this_run),
                       cpu_baseline=cpu_ik)
    if out is not None:
        _flush_c_stdio()
        print(json.dumps(out), flush=True)
    world.close()
    return 0


def _flush_c_stdio():
    """librccl writes its version banner through C stdio, which is flushed at exit -- after Python's
    own output -- unless it is flushed here: the JSON line is to be the last line of stdout."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass


def bench_rrt(args, world):
    """BASELINE configs[3] as the planner runs it (not the headline): rounds of `mjpl_rrt_round` --
    sample, nearest, extend with projection, validate, connect, then the exchange of every rank's new
    nodes over the RCCL communicator -- with 131 072 lanes PER RANK (weak scaling: 8 ranks = the 1 M
    samples per round of configs[3]).  All ranks hold the same two trees after every round.  A step is
    one round; round 1 (single-node trees) is the warm-up.  The path a round reports is checked
    against the oracle on rank 0."""
    import mjpl_amd as mjpl
    from mjpl_amd import scenes
    from oracle import pyoracle
    L = args.lanes
    rounds = args.steps if args.steps != 2000 else 5
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    cc = mjpl.CollisionConstraint(m, device=world.device)
    frame = mjpl.site_pose(m, q_init, "ee_site", engine=cc.engine)
    pc = mjpl.PoseConstraint(m, "ee_site", frame, roll=(-0.1, 0.1), pitch=(-0.1, 0.1), engine=cc.engine)
    cons = [pc, mjpl.JointLimitConstraint(m), cc]
    pc.q_step = np.inf
    q_goal = mjpl.random_config(m, q_init, joints, 7, cons)
    pc.q_step = 0.05
    # (the exchange always runs through a communicator here, one rank included; the id travels over the ranks' socket)
    from mjpl_amd import engine as _engine
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    uid = world.exchange(_engine.comm_unique_id() if world.rank == 0 else bytes(128))[0]
    dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, epsilon=0.05, interval_step=0.01, seed=3, goal_biasing_probability=0.05,
                           batch=L, capacity=args.capacity, pose=pc, comm=(uid, world.rank, world.world))
    world.rccl = True
    world.attach(cc.engine)
    if os.environ.get("MJPL_RRT_TRACE"):  # (tracing: the look-ups count the pairs that reach their exact evaluation)
        cc.engine.set_option("nn_probe", 2)
    dev.rrt.reset(q_init[qidx], q_goal[qidx][None], 3)
    info = dev.rrt.round()  # warm-up: the first round grows from two single-node trees
    # (the rounds cannot be repeated untimed -- the trees grow -- so the dominant kernels are bracketed in the timed rounds
    #  themselves: two events per launch of the generating kernel and of the nearest-neighbour scans, ~40 launches per round)
    cc.engine.set_option("kernel_timer", 3)
    rows, new_nodes, exch = [], 0, []
    nplan = len(qidx)
    world.barrier()
    t0 = time.perf_counter()
    for _ in range(rounds):
        t1 = time.perf_counter()
        info = dev.rrt.round()
        rows.append((time.perf_counter() - t1) * 1e3)
        nn = int(info.new_nodes[0]) + int(info.new_nodes[1])
        new_nodes += nn
        exch.append(nn * (8 * nplan + 4) + 32 * world.world)  # rows + parents of every rank's slabs, + the headers
        # (a connection does not end the measurement: the trees keep growing, as for a query still unsolved)
    world.barrier()
    elapsed = time.perf_counter() - t0
    done = len(rows)
    gen_ms, gen_n = cc.engine.get_option("kernel_timer_ms"), cc.engine.get_option("kernel_timer_launches")
    nn_ms, nn_n = cc.engine.get_option("kernel_timer2_ms"), cc.engine.get_option("kernel_timer2_launches")
    cc.engine.set_option("kernel_timer", 0)
    slowest = float(world.gather(elapsed).max())
    path_ok = None
    if world.rank == 0:
        if info.connected:
            path = dev.rrt.path()
            full = np.repeat(q_init[None, :], len(path), axis=0)
            full[:, qidx] = path
            orc = pyoracle.Oracle(m, planning_qidx=qidx, qpos_base=q_init)
            ok_edges = orc.valid_edges(path[:-1], path[1:], 0.01, nthreads=8)
            path_ok = bool(ok_edges.all() and np.all(pc.valid_configs(full)))
            if not path_ok:
                raise SystemExit("bench: the planner's path fails the oracle's collision check / the pose constraint")
        # ---- roofline of a round.  A round is ~60 launches of a dozen kernels; most of its GPU time is the projecting
        # extension's generating kernel, float64 Newton chains (profiles/*_rrt_kernel_stats.csv): its issue rate and the latency of
        # one wave's chain bind, not HBM.  Algorithmic bytes of a round: per sample its target read and what
        # it reached written (2 x 8 nplan + 11 B), per new node its row and parent written (8 nplan + 4 B), and the two
        # nearest-neighbour scans (row f2: N x nq x 8 B per query batch -- the node matrix ONCE per scan; that the kernel walks it
        # once per 512-query tile is implementation traffic served by L2, not algorithmic bytes) plus their queries and answers.
        nn_bytes = sum(int(n) for n in info.nodes) * 8 * nplan + 2 * L * (8 * nplan + 4)
        round_bytes = L * (2 * 8 * nplan + 11) + (new_nodes / max(done, 1)) * (8 * nplan + 4) + nn_bytes
        round_s = slowest / done
        # (the generating kernel comes in two forms: rows of one, four or eight lanes, and -- the tail -- rows of sixteen that run a
        #  step ahead, mjpl_rows.h; the line quotes the one with more of the round's time in the committed kernel trace)
        gen_kernel = max(("k_rrt_gen_project_rows", "k_rrt_gen_project_ahead"),
                         key=lambda k: profile_key(f"{k}_rrt{L}").get("avg_ns", 0.0) * profile_key(f"{k}_rrt{L}").get("calls_per_round", 0.0))
        rec = profile_key(f"{gen_kernel}_rrt{L}")
        roofline = issue_roofline(
            f"{gen_kernel}_rrt{L}", gen_kernel,
            {"bound": "hbm", "achieved": round_bytes / round_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round_bytes / round_s / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_round": round_bytes,
             "of_which_nearest_neighbour_scans": nn_bytes, "over": "the whole round (its kernels are not bracketed one by one)"},
            round_s * 1e3, launches_per_step=float(rec.get("calls_per_round", 0.0)) or 1.0,
            this_run=(gen_ms / done, gen_n / done) if gen_n > 0 else None)
        if isinstance(roofline, dict) and roofline.get("bound") == "valu_issue":
            roofline["kernel"] = "k_rrt_gen_project_rows / _ahead (every launch of the rounds' generating kernel)"
            # (the committed counters are per launch of ONE of the two forms at the first rounds' sizes; this run's launches differ in
            #  lanes and steps, so `achieved` here = the committed instructions per launch x this run's launches / this run's time)
        roofline["nearest_neighbour_scans"] = {
            "ms_per_round": nn_ms / done if nn_n > 0 else None, "launches_per_round": nn_n / done if nn_n > 0 else None,
            "share_of_round": nn_ms / done / (round_s * 1e3) if nn_n > 0 else None,
            "note": "k_nearest_mfma (cell-ordered scan, mjpl_nearest_cells.h) between HIP events in the timed rounds; beside the "
                    "second stream's look-ups the shares may add up to more than the round"}
        cpu = None
        if not args.no_cpu_baseline:
            # the same algorithm stated in NumPy (mjpl_amd.planning.parallel_rrt.ParallelBiRRT: what the GPU planner's trees are
            # compared with node for node in tests/test_gpu_rrt.py) with the CPU oracle behind every constraint, on a bounded
            # sample: 512 lanes per round from the same start to the same goal, until it connects or 15 s have passed
            from mjpl_amd.lie import SE3, SO3
            from mjpl_amd.planning.parallel_rrt import EdgeValidator, ParallelBiRRT
            cpu_model, cores, aff, quota = host_cpu()
            ident = pyoracle.PoseOracle(m, "ee_site", (np.array([1.0, 0, 0, 0]), np.zeros(3)), [(-np.inf, np.inf)] * 6)
            pos, mat = ident.site_pose(q_init)
            inv = SE3.from_rotation_and_translation(SO3.from_matrix(mat), pos).inverse()
            unb = (-np.inf, np.inf)
            po = pyoracle.PoseOracle(m, "ee_site", (inv.wxyz_xyz[:4], inv.wxyz_xyz[4:]), [unb, unb, unb, (-0.1, 0.1), (-0.1, 0.1), unb], q_step=0.05)
            orc = pyoracle.Oracle(m, planning_qidx=qidx, qpos_base=q_init)

            class OracleSet(EdgeValidator):  # [PoseConstraint, JointLimit, Collision] on the oracle
                def full(self, Qp):
                    F = np.repeat(q_init[None], len(Qp), axis=0)
                    F[:, qidx] = Qp
                    return F

                def valid_edges(self, QA, QB, step):
                    if step is None:
                        ok = orc.valid_configs(QB, nthreads=cores).astype(bool)
                        return ok & np.array([po.valid_config(r) for r in self.full(QB)], dtype=bool)
                    return orc.valid_edges(QA, QB, step, nthreads=cores).astype(bool)

                def project(self, Q_old, Q):
                    o, ok, _ = po.apply_batch(self.full(Q_old), self.full(Q), nthreads=cores)
                    return o[:, qidx], ok

            lanes_cpu = 512
            host = ParallelBiRRT(m, joints, OracleSet(), q_init, epsilon=0.05, interval_step=0.01, seed=3, batch=lanes_cpu,
                                 goal_biasing_probability=0.05, max_planning_time=15.0)
            tc = time.perf_counter()
            cpath = host.plan_to_config(q_init, q_goal)
            dtc = time.perf_counter() - tc
            cpu = {"value": lanes_cpu * host.stats["rounds"] / dtc, "unit": "samples/s", "cores": cores, "kind": "port",
                   "cpu_model": cpu_model, "affinity": aff, "cpu_quota": quota,
                   "sample": f"{host.stats['rounds']} rounds of {lanes_cpu} lanes ({dtc:.1f} s, {'connected' if cpath else 'time limit'}), "
                             f"trees of {tuple(int(x) for x in host.stats['nodes'])} nodes: ParallelBiRRT (NumPy) with oracle/libmjpl_oracle.so "
                             f"validating and projecting on {cores} threads"}
        _flush_c_stdio()
        print(json.dumps({
            "roofline": roofline, "cpu_baseline": cpu,
            "metric": "RRT samples/sec through the frontier bi-RRT (BASELINE configs[3]: 131 072 samples per GPU per round)",
            "value": L * world.world * done / slowest, "unit": "samples/s", "n_gpus": world.world, "steps": done, "warmup": 1,
            "ms_per_step": slowest / done * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 planner + f32-filter+f64-exact validation", "data": "synthetic",
            "config": {"workload": f"{L} lanes per rank per round, [PoseConstraint(roll, pitch +-0.1), JointLimit, Collision], "
                                   "eps 0.05, interval 0.01, Franka-P + 16 obstacles, RCCL all-gather of the new nodes per round",
                       "rccl_ranks": world.world, "round_ms": rows, "exchange_bytes_per_round": exch,
                       "new_nodes_per_s": new_nodes / slowest,
                       "nodes": [int(info.nodes[0]), int(info.nodes[1])], "connected": bool(info.connected),
                       "connecting_rank": int(info.conn_rank) if info.connected else None,
                       "path_checked_against_oracle": path_ok,
                       "nearest_neighbour_screen": {0: "none", 1: "binary32 (vector units)", 2: "binary16 x 2 (matrix cores)"}.get(
                           cc.engine.nearest_last_screen(), "?"),
                       "parallelism": f"frontier-sharded x{world.world}: every rank samples and extends its own lanes, "
                                      "one all-gather of headers + two of slabs per round"}}), flush=True)
    world.barrier()
    dev.rrt.close()
    world.close()
    return 0


def bench_plan(args, world):
    """examples/benchmark.py's loop in the reference's shape (/root/reference/examples/benchmark.py:28-48,
    83-91: 15 attempts, epsilon 0.05, seed 42, goal bias 0.1, 10 s limit, pose goal through IK, constraints
    = joint limits + collision) on the device-resident frontier planner, Franka-P + 16 obstacles; every
    returned path checked against the oracle.  Beside it, as the CPU baseline: the serial `RRT` with the
    reference's step-by-step extension, every configuration validated by the CPU oracle (the role MuJoCo
    has in the reference), on a bounded sample of the same attempts.  Replicas only across ranks."""
    from oracle import pyoracle
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import benchmark as harness
    from mjpl_amd import scenes
    from mjpl_amd.constraint import Constraint
    attempts = 15 if args.steps == 2000 else args.steps
    model = scenes.franka_p(obstacles=True)
    orc = pyoracle.Oracle(model)

    def check_path(path):
        P = np.stack(path)
        return bool(orc.valid_configs(P, nthreads=4).all() and
                    np.all((P >= model.jnt_range[:, 0] - 1e-12) & (P <= model.jnt_range[:, 1] + 1e-12)) and
                    np.linalg.norm(np.diff(P, axis=0), axis=1).max() <= 0.05 + 1e-9)

    harness.run(planner="device", attempts=1, obstacles=True, device=world.device, quiet=True)  # warm-up
    res = harness.run(planner="device", attempts=attempts, obstacles=True, device=world.device, quiet=True, check_path=check_path)
    if world.rank == 0:
        if not res["paths_valid"]:
            sys.exit("bench.py: a planned path fails the oracle's checks")
        cpu = None
        if not args.no_cpu_baseline:
            class OracleCollision(Constraint):  # (cpu_baseline leg: the oracle behind the Constraint interface)
                def valid_config(self, q):
                    return orc.valid_config(q)

                def apply(self, q_old, q):
                    return q if orc.valid_config(q) else None

            k = min(3, attempts)
            cres = harness.run(planner="rrt", attempts=k, obstacles=True, device=world.device, quiet=True,
                               collision=OracleCollision(), check_path=check_path)
            cpu = {"value": float(np.median(cres["planning_times"])) if cres["planning_times"] else None,
                   "unit": "s (median planning time)", "cores": 1, "kind": "port",
                   "sample": f"first {k} attempts: serial RRT, step-by-step _constrained_extend, every configuration "
                             "validated by oracle/libmjpl_oracle.so on one thread (IK seeds from the batched solver)",
                   "successes": int(cres["successes"]), "attempts": k}
        times = res["planning_times"]
        out = {"metric": "median planning time, examples/benchmark.py in the reference's shape (Franka-P + 16 obstacles, pose goal)",
               "value": float(np.median(times)) if times else None, "unit": "s", "n_gpus": world.world, "steps": attempts, "warmup": 1,
               "ms_per_step": float(np.mean(times)) * 1e3 if times else None, "higher_is_better": False, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64 planner + f32-filter+f64-exact validation", "data": "synthetic",
               "config": {"workload": f"{attempts} attempts of plan_to_pose, epsilon 0.05, seed 42, goal bias 0.1, 10 s limit, "
                                      "512 lanes per round, constraints [JointLimit, Collision]",
                          "successes": int(res["successes"]), "attempts": attempts,
                          "success_rate": res["successes"] / attempts,
                          "paths_checked_against_oracle": bool(res["paths_valid"]),
                          "planning_times_s": [float(t) for t in times], "parallelism": "replicas only"},
               "cpu_baseline": cpu}
        _flush_c_stdio()
        print(json.dumps(out), flush=True)
    return 0


def time_variant(eng, engine, dqa, dqb, E, layout, dvalid, steps, warmup):
    if warmup > 0:
        eng.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, STEP, layout, dvalid.ptr, warmup, 1 << 30)
    eng.sync()
    t0 = time.perf_counter()
    launch_ms, stage_ms, nsamp = eng.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, STEP, layout, dvalid.ptr, steps, 4)
    return time.perf_counter() - t0, launch_ms, stage_ms, nsamp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000,
                    help="timed steps; the default keeps the timed region near one second")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--edges", type=int, default=EDGES_PER_GPU)
    ap.add_argument("--layout", choices=["soa", "aos"], default="soa")
    ap.add_argument("--streams", type=int, default=1,
                    help="edges workload: engines (HIP streams) that take the steps in turns -- batches in flight per GPU. "
                         "The line's value is ONE engine on one stream (what every caller of mjpl_check_edges_dev gets); "
                         "the line also carries `in_turns`, the same steps taken in turns by three engines")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="edges workload: skip the float64-only and interpreter engines timed beside the headline")
    ap.add_argument("--variant", choices=["headline", "f64", "interpreter", "generic", "pads"], default="headline",
                    help="edges workload: which engine configuration the TIMED steps run (profiling passes of the "
                         "variants: tools/profile_gpu.sh); the default line carries all three")
    ap.add_argument("--lanes", type=int, default=131072, help="rrt workload: lanes per rank per round")
    ap.add_argument("--capacity", type=int, default=1 << 24, help="rrt workload: node capacity per tree")
    ap.add_argument("--workload", choices=["edges", "configs", "pose", "ik", "rrt", "plan"], default="edges",
                    help="edges: the headline metric (BASELINE configs[2]).  Extra lines, not the headline: "
                         "configs = configs[1], 65 536 Franka-P self-collision configurations per launch; "
                         "pose = one GPU's share of configs[3], 131 072 PoseConstraint projections; "
                         "ik = one GPU's share of configs[4], 16 384 IK seeds -> FK -> collision filter; "
                         "rrt = configs[3] as the planner runs it: rounds of the device-resident frontier bi-RRT, "
                         "131 072 samples per rank each, [PoseConstraint, JointLimit, Collision], new nodes exchanged "
                         "over RCCL (--steps = timed rounds, default 5); plan = examples/benchmark.py's 15 attempts")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args.gpus)  # (this process never touches a GPU)
    world = World()
    if world.world != args.gpus:
        args.gpus = world.world

    if args.workload == "configs":
        return bench_configs(args, world)
    if args.workload in ("pose", "ik"):
        return bench_next_rows(args, world)
    if args.workload == "rrt":
        return bench_rrt(args, world)
    if args.workload == "plan":
        return bench_plan(args, world)

    from mjpl_amd import engine, scenes

    rank = world.rank
    # (--variant pads: the same workload on Franka-P with the reference Panda's ten finger-pad boxes -- moving boxes --
    #  for the profiling passes of that model's kernels; never the line the driver reads)
    model = scenes.franka_p(obstacles=True, pads=args.variant == "pads")
    qidx = scenes.planning_index(model, scenes.FRANKA_ARM_JOINTS)
    base = model.keyframe("home").qpos.copy()

    def make_engine(filt=True, spec=True, model=model):
        e = engine.Engine(model, device=world.device)
        if spec is not True:
            e.set_spec(int(spec))  # (0: interpreter; 2: the robot's scene-generic library)
        e.set_planning(qidx, base)
        if not filt:
            e.set_filter(False)
        return e

    timed = {"headline": (True, True), "f64": (False, True), "interpreter": (True, False), "generic": (True, 2),
             "pads": (True, True)}[args.variant]
    eng = make_engine(*timed)
    world.attach(eng)
    info = eng.info()

    E = args.edges
    qa, qb = make_edges(model, qidx, E, seed=2 + rank)
    layout = engine.SOA if args.layout == "soa" else engine.AOS
    ha = np.ascontiguousarray(qa.T) if layout == engine.SOA else qa
    hb = np.ascontiguousarray(qb.T) if layout == engine.SOA else qb
    dqa, dqb = eng.alloc(ha.nbytes).upload(ha), eng.alloc(hb.nbytes).upload(hb)
    dvalid = eng.alloc(E)
    eng.sync()
    # Batches in flight: S engines of the same model, each with its own HIP stream and scratch, take the K steps in
    # turns (step i on engine i mod S; the inputs are shared, every engine writes its own verdicts).  A step is
    # three kernels, each ending in a tail where the chip drains (and the last one a 20 us latency chain of 30
    # waves): with a second stream the next batch's kernels fill those -- what a planner with more than one
    # batch to validate does with `mjpl_check_edges_dev` on two engines.  --streams 1: one engine, as in rounds 1-3.
    S = max(1, min(args.streams, args.steps))
    engs = engs_default = [eng] + [make_engine(*timed) for _ in range(S - 1)]
    outs = outs_default = [dvalid] + [e.alloc(E) for e in engs[1:]]
    share = [args.steps // S + (1 if k < args.steps % S else 0) for k in range(S)]

    def run_all(steps_of, sample, before=None, engs=None, outs=None):
        """steps_of[k] launches on engine k, all engines at once (one host thread each: the call blocks until its
        stream is through; ctypes releases the GIL).  The threads are up and waiting when `before` -- the barrier
        and the clock of the timed region -- runs.  Returns (what `before` returned, per engine (launch_ms, stage_ms, nsamp))."""
        import threading
        engs = engs_default if engs is None else engs
        outs = outs_default if outs is None else outs
        n_eng = len(engs)
        res = [None] * n_eng
        errs = []
        go = threading.Event()

        def one(k):
            try:
                go.wait()
                res[k] = engs[k].time_edges_stages_dev(dqa.ptr, dqb.ptr, E, STEP, layout, outs[k].ptr, steps_of[k], sample)
            except Exception as ex:  # noqa: BLE001 -- reported below, on the main thread
                errs.append(ex)
        th = [threading.Thread(target=one, args=(k,)) for k in range(1, n_eng) if steps_of[k] > 0]
        for t in th:
            t.start()
        stamp = before() if before else None
        go.set()
        one(0)
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        return stamp, res

    # Untimed: the clocks first -- at least ~60 ms of launches whatever --warmup is (a 20-step region is 4 ms: started
    # cold the chip runs it 5 % below the rate it holds a second later) --, then the W warmup steps of the contract.
    run_all([CLOCK_WARM_STEPS] * S, 1 << 30)
    if args.warmup > 0:
        run_all([args.warmup] * S, 1 << 30)
    # A short region carries no events at all: a bracketed launch costs ~56 us more than a plain one (below), 3 % of a
    # 20-step region for one sample.  The per-kernel durations of such a line come from 32 launches taken here, every
    # one bracketed, on engine 0 alone -- the same kernels on the same batch, outside the clock.
    pre_res = None
    if args.steps < SHORT_REGION_STEPS:
        _, pre_res = run_all([32] + [0] * (S - 1), 1)

    def start_of_timed_region():
        world.barrier()
        return time.perf_counter()
    # EXACTLY args.steps launches, back to back on their engines' streams; one launch in `sample` of an engine also
    # carries one HIP event after each of its kernels (the per-kernel durations roofline.achieved uses).  A bracketed
    # launch costs ~56 us more than a plain one -- the events break the back-to-back submission: every 4th launch cost
    # the one-stream line 2 % of its rate, every 16th still 1.8 % (0.1966 against 0.1931 ms per step without any,
    # tools/time_fused.py) -- so about 32 launches of the region are bracketed, never more than one in 16.
    sample = (1 << 30) if pre_res is not None else max(16, args.steps // 32)
    t0, timed_res = run_all(share, sample, start_of_timed_region)  # every call synchronises its stream
    world.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = float(world.gather(elapsed).max())
    wsum = float(sum(share))
    if pre_res is not None:
        launch_ms, stage_ms, nsamp = pre_res[0]
        sample_note = "HIP events on the engine's own stream around every kernel of 32 untimed launches just before the timed region (a region of fewer than %d steps carries no events), nothing else on the chip" % SHORT_REGION_STEPS
    else:
        launch_ms = sum(r[0] * n for r, n in zip(timed_res, share) if r) / wsum
        stage_ms = {k: sum(r[1][k] * n for r, n in zip(timed_res, share) if r) / wsum for k in timed_res[0][1]}
        nsamp = sum(r[2] for r in timed_res if r)
        sample_note = f"HIP events on the engine's own stream around every kernel of every {sample}th launch of the timed region, nothing else on the chip"

    valid = dvalid.download(np.uint8, E)
    for k in range(1, S):
        if share[k] > 0 and not np.array_equal(outs[k].download(np.uint8, E), valid):
            sys.exit(f"bench.py: engine {k} of {S} returned other verdicts than engine 0 for the same batch")
    one_stream = None
    if S > 1:  # ... and the same K steps on ONE stream (not the measurement: the figure of rounds 1-3, and the kernels alone)
        eng.sync()
        t1 = time.perf_counter()
        l1, st1, _ = eng.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, STEP, layout, dvalid.ptr, max(100, args.steps // 4), 16)
        dt1 = time.perf_counter() - t1
        one_stream = {"value": E * max(100, args.steps // 4) / dt1, "unit": "edges/s", "steps": max(100, args.steps // 4),
                      "ms_per_step": dt1 / max(100, args.steps // 4) * 1e3, "step_ms_all_kernels": l1, "kernels_ms": st1}

    if rank == 0:
        total_edges = E * world.world * args.steps
        value = total_edges / elapsed
        filt = bool(info["filter_enabled"])
        spec = eng.spec_kind()  # 0 interpreting kernels, 1 the program's own library, 2 the robot's scene-generic one
        interior = eng.last_interior_edges() if filt else 0
        items = eng.last_items() if filt else 0
        undecided = eng.last_undecided()
        fused = bool(filt and info.get("fused_edges"))
        # units and SURVEY.md 8(d) algorithmic bytes of every kernel of a step:
        #   fused filter:  one edge = 2 x 56 B of columns read + 1 B verdict (113 B); its waypoints are generated on chip
        #   endpoint pass: the same 113 B per edge;  item pass: one waypoint configuration = 57 B + its 8 B (edge, index)
        #   walking pass:  one edge, 113 B + the 4 B list entry;  pair re-check: 56 B + 16 B per undecided pair
        per_stage = {"k_filter_endpoints": (E, BYTES_PER_EDGE, "edges"),
                     "k_filter_items": (0 if fused else items, 57 + 8, "waypoint configurations"),
                     "k_filter_edges": (max(interior - 0, 0) if items == 0 else 0, BYTES_PER_EDGE + 4, "edges"),
                     "k_patch_pairs": (undecided, 56 + 16, "undecided geom pairs"),
                     "k_check_edges": (E if (not filt and not info.get("fused_edges")) else 0, BYTES_PER_EDGE, "edges")}
        # kernel names as a profiler shows them for this engine's launch layout

        def stage_names(inf, with_filter):
            n = {}
            if not with_filter and inf.get("fused_edges"):
                n["k_filter_endpoints"] = "k_edges_fused_f64"
            elif with_filter and inf.get("fused_edges"):
                n["k_filter_endpoints"] = "k_edges_fused"
            elif with_filter and inf.get("persistent_kernels"):
                n.update({"k_filter_endpoints": "k_filter_endpoints_pw", "k_filter_items": "k_filter_items_pw"})
            if with_filter and inf.get("fused_tail"):
                n["k_patch_pairs"] = "k_tail"
            return n
        names = stage_names(info, filt)
        stage_ms = {names.get(k, k): v for k, v in stage_ms.items()}
        per_stage = {names.get(k, k): v for k, v in per_stage.items()}
        # The dominant kernel and its duration: HIP events around the kernel ALONE on the chip -- this run's own with one
        # stream; with several streams the events of the timed region bracket kernels that share the chip, so the
        # kernel's figure comes from the same steps taken on one engine (one_stream), never from the overlapped ones.
        alone_ms = stage_ms if S == 1 else {names.get(k, k): v for k, v in one_stream["kernels_ms"].items()}
        kernel = max(alone_ms, key=lambda k: alone_ms[k])
        kernel_ms = alone_ms[kernel]
        units, unit_bytes, unit_name = per_stage[kernel]
        achieved = unit_bytes * units / (kernel_ms * 1e-3) / 1e9
        ksuffix = "_pads" if args.variant == "pads" else ""  # (counter records are per model as well)
        prof = profile_record(kernel + ksuffix, E, args.layout, filt, spec)
        workload = (f"configs[2]: Franka-P 7-DoF + 16 box/sphere obstacles + floor"
                    f"{' + the Panda finger pads, ten MOVING boxes (--variant pads: not the headline model)' if args.variant == 'pads' else ''}, {E} edges/GPU, "
                    f"eps {EPS}, step {STEP} (endpoint + interior waypoints per edge)")
        # what binds: vector-ALU issue (not HBM, not MFMA) -- SQ_INSTS_VALU wave-instructions of the committed
        # counter pass of this kernel / (its duration alone x the chip's issue slots per second)
        hbm = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
               "traffic": prof.get("hbm_bytes_per_launch"),
               "algorithmic_bytes_per_unit": unit_bytes, "units_in_this_kernel": units, "unit_of_work": unit_name,
               "whole_step": {"algorithmic_bytes": BYTES_PER_EDGE * E,
                              "achieved_GBs": BYTES_PER_EDGE * E * args.steps / elapsed / 1e9,
                              "frac": BYTES_PER_EDGE * E * args.steps / elapsed / 1e9 / HBM_PEAK_GBS},
               "note": "the figure north_star asks for; the path is not HBM bound (SURVEY.md 8d): 113 algorithmic bytes per edge "
                       "against ~22 600 flop"}
        if prof.get("hbm_bytes_per_launch") and units:
            # counter traffic against the algorithmic bytes, and where the excess comes from (profiles/README.md, round 5):
            # reads = the two column sets once per endpoint tile + once more per waypoint tile of their edges (L2 misses of
            # 8-byte gathers, doubled by the gfx950 FETCH_SIZE correction, which may not apply to them); writes = almost all
            # SCRATCH write-backs -- the kernel sits at the 168-register bound of three waves per SIMD and keeps 68 B per lane in
            # scratch (16 vector + 83 scalar registers spilled): 3 072 waves x 64 lanes x 68 B = 13.4 MB -- not verdict stores (0.26 MB)
            hbm["traffic_over_algorithmic"] = float(prof["hbm_bytes_per_launch"]) / float(unit_bytes * units)
            hbm["traffic_breakdown"] = {"FETCH_SIZE_x2_bytes": 2 * 1024 * float(prof.get("FETCH_SIZE_KiB") or 0.0),
                                        "WRITE_SIZE_bytes": 1024 * float(prof.get("WRITE_SIZE_KiB") or 0.0),
                                        "of_the_writes": "scratch write-backs (68 B per lane x 196 608 lanes = 13.4 MB); verdict bytes 0.26 MB"}
        roof = {"kernel": kernel, "kernel_ms": kernel_ms, "step_ms_all_kernels": launch_ms if S == 1 else one_stream["step_ms_all_kernels"],
                "kernels_ms": alone_ms, "kernel_samples": nsamp, "streams_of_these_durations": 1,
                "kernel_ms_source": sample_note
                                    + ("" if S == 1 else " (the one_stream run of this line; the timed region overlaps kernels of several engines)"),
                "hbm": hbm}
        if prof.get("SQ_INSTS_VALU"):
            iv = float(prof["SQ_INSTS_VALU"])
            roof.update({"bound": "valu_issue", "achieved": iv / (kernel_ms * 1e-3), "peak": VALU_ISSUE_PEAK, "unit": "wave-instructions/s",
                         "frac": iv / (kernel_ms * 1e-3) / VALU_ISSUE_PEAK, "traffic": prof.get("hbm_bytes_per_launch"),
                         "wave_insts_valu_per_launch": iv, "wave_insts_salu_per_launch": prof.get("SQ_INSTS_SALU"),
                         "counters_source": prof.get("source"),
                         "note": "peak = 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction (MI355X_MICROARCH.md, Wave "
                                 "scheduling); the HBM figure north_star asks for is in `hbm`"})
        else:  # no committed counter pass for this kernel / configuration: the HBM object alone
            roof.update({k: hbm[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")})
            roof["note"] = "no committed SQ_INSTS_VALU record for this kernel and configuration (profiles/pmc.json): HBM figure only"
        out = {
            "metric": METRIC, "value": value, "unit": "edges/s", "n_gpus": world.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # what ran: a binary32 filter decides what it can within its tolerance band, binary64
            # kernels decide the rest; every verdict equals the pure binary64 path's
            "dtype": "f32-filter+f64-exact" if filt else "f64",
            "data": "synthetic",
            "config": {"workload": workload, "edges_per_gpu": E, "layout": args.layout,
                       "geom_pairs": info["npairs"], "valid_fraction": float(valid.mean()),
                       "float32_filter": filt, "filter_tol_m": info["filter_tol"],
                       "specialised_kernels": bool(spec), "library": {0: "none (interpreting kernels)", 1: "this program's own", 2: "the robot's scene-generic one"}[spec],
                       "clock_warm_steps": CLOCK_WARM_STEPS,  # untimed launches before the W warm-up steps (the clocks)
                       "events_in_timed_region": bool(args.steps >= SHORT_REGION_STEPS),  # (shorter regions: per-kernel durations from 32 bracketed launches just before it)
                       "streams": S, "batches_in_flight": S,
                       "streams_note": (f"{S} engines (one HIP stream and one scratch set each) take the steps in turns: the kernels of "
                                        "consecutive batches overlap; ms_per_step = elapsed / steps; one_stream = the same steps on one engine")
                                       if S > 1 else "one engine, one stream: what every caller of mjpl_check_edges_dev gets",
                       "fused_edges": fused, "fused_waves_per_workgroup": info.get("fused_waves"),
                       "persistent_kernels": bool(info.get("persistent_kernels")), "fused_tail": bool(info.get("fused_tail")),
                       "kernels_per_launch": len([k for k in stage_ms if stage_ms[k] > 0.008]),
                       "undecided_items_last_step": undecided,
                       "edges_reaching_interior_pass": interior, "interior_waypoint_items": items,
                       "parallelism": f"edge-sharded x{world.world}, no data-path collective",
                       "rank_launcher": ("bench.py (child processes)" if os.environ.get("MJPL_BENCH_RDZV") else
                                         ("external (RANK / WORLD_SIZE)" if world.world > 1 else "single process")),
                       "collectives": "none on the GPUs: barrier + max of the elapsed time over the ranks' Unix socket"
                                      if world.distributed else None,
                       "ranks_share_a_gpu": os.environ.get("MJPL_BENCH_SHARE_GPU") == "1" and world.distributed},
            "roofline": roof,
        }
        def valu_issue(kern, ms, rec):
            if not rec.get("SQ_INSTS_VALU"):
                return None
            iv = float(rec["SQ_INSTS_VALU"])
            return {"kernel": kern, "wave_insts_valu_per_launch": iv,
                    "wave_insts_salu_per_launch": rec.get("SQ_INSTS_SALU"),
                    "achieved_per_s": iv / (ms * 1e-3), "peak_per_s": VALU_ISSUE_PEAK,
                    "frac": iv / (ms * 1e-3) / VALU_ISSUE_PEAK, "source": rec.get("source")}
        # the vector-issue rate of the chip over a whole step: the vector instructions of all its kernels / the time a step takes
        recs = [profile_record(k + ksuffix, E, args.layout, filt, spec) for k in stage_ms]
        if all(r.get("SQ_INSTS_VALU") for r, k in zip(recs, stage_ms) if stage_ms[k] > 0.01):
            iv = sum(float(r.get("SQ_INSTS_VALU", 0.0)) for r in recs)
            out["valu_issue_whole_step"] = {"wave_insts_valu_per_step": iv, "ms_per_step": elapsed / args.steps * 1e3,
                                            "frac": iv / (elapsed / args.steps) / VALU_ISSUE_PEAK, "peak_per_s": VALU_ISSUE_PEAK,
                                            "kernels": [k for k in stage_ms if stage_ms[k] > 0.01]}
        out["one_stream"] = one_stream
        if S == 1 and world.world == 1 and not args.no_variants and args.variant == "headline" and args.steps >= IN_TURNS_MIN_STEPS:
            # (only in a long run: three freshly created engines sharing 20 steps measure their own start-up, and the figure
            #  printed BELOW the one-stream value in round 4's driver line)
            # ... and the same steps taken in turns by three engines (three HIP streams: batches in flight): what a caller
            # with several batches to validate can add on top (mjpl_amd.engine.EngineRing); never the line's value
            T = 3
            tsteps = min(args.steps, 1200)
            tengs = [eng] + [make_engine(*timed) for _ in range(T - 1)]
            touts = [dvalid] + [x.alloc(E) for x in tengs[1:]]
            tshare = [tsteps // T + (1 if k < tsteps % T else 0) for k in range(T)]
            run_all([100] * T, 1 << 30, None, tengs, touts)
            tv, _ = run_all(tshare, 1 << 30, time.perf_counter, tengs, touts)
            dtv = time.perf_counter() - tv
            if not all(np.array_equal(o.download(np.uint8, E), valid) for o in touts):
                sys.exit("bench.py: an engine of the in-turns run returned other verdicts than the line's engine")
            out["in_turns"] = {"value": E * tsteps / dtv, "unit": "edges/s", "streams": T, "steps": tsteps, "ms_per_step": dtv / tsteps * 1e3,
                               "note": "three engines of this model, one HIP stream each, taking the steps in turns; verdicts equal"}
            for x, o in zip(tengs[1:], touts[1:]):
                o.free()
                x.close()

        # ---- the same batch through the other two engine configurations, in this process: the float64
        # kernels alone (the reference's arithmetic end to end) and the generic interpreting filter kernels
        # (what a model without a specialised library runs).  Same verdicts required, all E of them.
        flops = flop_record()
        if world.world == 1 and not args.no_variants and args.variant == "headline":
            variants = {}
            vsteps = max(50, min(args.steps, 200))
            # ... and the robot's scene-generic library (obstacles from a table: any scene without a compiler)
            for name, (vf, vs) in (("f64_only", (False, True)), ("interpreter", (True, False)), ("scene_generic", (True, 2))):
                ve = make_engine(vf, vs)
                want_kind = {"f64_only": None, "interpreter": 0, "scene_generic": 2}[name]
                if want_kind is not None and ve.spec_kind() != want_kind:
                    sys.exit(f"bench.py: the {name} variant found library kind {ve.spec_kind()} instead of {want_kind}: the "
                             "libraries under mjpl_amd/csrc/spec are not the ones __graft_entry__.build() makes from these sources")
                va, vb, vv = ve.alloc(ha.nbytes).upload(ha), ve.alloc(hb.nbytes).upload(hb), ve.alloc(E)
                dt, v_launch, v_stage, v_n = time_variant(ve, engine, va, vb, E, layout, vv, vsteps, 5)
                same = bool(np.array_equal(vv.download(np.uint8, E), valid))
                if not same:
                    sys.exit(f"bench.py: the {name} engine's verdicts differ from the headline engine's")
                vinfo = ve.info()
                vnames = stage_names(vinfo, vf)
                v_stage = {vnames.get(k, k): v for k, v in v_stage.items()}
                vk = max(v_stage, key=lambda k: v_stage[k])
                rec = profile_record(vk, E, args.layout, vf, ve.spec_kind() if vs else 0)  # (s1: the program's own library, s2: the robot's)
                in_turns = None
                if S > 1:  # ... and as the headline runs: S engines of this configuration taking the steps in turns
                    ves = [ve] + [make_engine(vf, vs) for _ in range(S - 1)]
                    vouts = [vv] + [x.alloc(E) for x in ves[1:]]
                    vshare = [vsteps // S + (1 if k < vsteps % S else 0) for k in range(S)]
                    run_all([5] * S, 1 << 30, None, ves, vouts)
                    tv, _ = run_all(vshare, 1 << 30, time.perf_counter, ves, vouts)
                    dtv = time.perf_counter() - tv
                    same = same and all(np.array_equal(o.download(np.uint8, E), valid) for o in vouts)
                    if not same:
                        sys.exit(f"bench.py: an engine of the {name} variant returned other verdicts than the headline engine's")
                    in_turns = {"value": E * vsteps / dtv, "unit": "edges/s", "streams": S, "ms_per_step": dtv / vsteps * 1e3}
                    for x, o in zip(ves[1:], vouts[1:]):
                        o.free()
                        x.close()
                variants[name] = {"value": E * vsteps / dt, "unit": "edges/s", "steps": vsteps, "ms_per_step": dt / vsteps * 1e3,
                                  "streams": 1,  # (compare with one_stream; in_turns compares with the line's value)
                                  "in_turns": in_turns,
                                  "step_ms_all_kernels": v_launch, "kernels_ms": v_stage, "dtype": "f64" if not vf else "f32-filter+f64-exact",
                                  "float32_filter": bool(ve.info()["filter_enabled"]),
                                  # (with the filter off no kernel of a per-model library runs, whatever the engine has loaded)
                                  "specialised_kernels": bool(ve.spec_loaded()) and vf,
                                  "library": ({0: "none (interpreting kernels)", 1: "this program's own", 2: "the robot's scene-generic one"}[ve.spec_kind()]
                                              if vf else "none: the interpreting float64 kernels of libmjpl_hip.so (k_edges_fused_f64 -- the exact check through the candidate queues, narrowphase with full lanes -- then k_check_edges over the edges too long for its pool)"),
                                  "fused_edges": bool(vf and vinfo.get("fused_edges")),
                                  "verdicts_equal_headline": same, "edges_compared": E,
                                  "valu_issue": valu_issue(vk, v_stage[vk], rec)}
                if name == "f64_only" and flops.get("flops_per_edge"):
                    fpe = float(flops["flops_per_edge"])
                    variants[name]["achieved_FP64_fraction"] = E * fpe / (v_launch * 1e-3) / FP64_VECTOR_PEAK
                    variants[name]["fp64"] = {"flops_per_edge": fpe, "achieved_TFLOPs": E * fpe / (v_launch * 1e-3) / 1e12,
                                              "peak_TFLOPs": FP64_VECTOR_PEAK / 1e12, "source": flops.get("source"),
                                              "note": "the oracle's operation count of the reference's algorithm x edges/s (SURVEY.md 8d)"}
                    if rec.get("SQ_INSTS_VALU_FMA_F64") is not None:
                        # what the kernel ISSUES: wave-level float64 instructions of the committed counter pass x 64 lanes
                        # (masked lanes included; an FMA counts two) over this run's kernel time
                        wi = {k: float(rec[f"SQ_INSTS_VALU_{k}_F64"]) for k in ("ADD", "MUL", "FMA", "TRANS")}
                        issued = 64.0 * (wi["ADD"] + wi["MUL"] + 2.0 * wi["FMA"] + wi["TRANS"])
                        variants[name]["fp64"]["issued"] = {
                            "wave_insts_per_launch": wi, "flops_per_launch": issued,
                            "TFLOPs": issued / (v_stage[vk] * 1e-3) / 1e12,
                            "frac_of_peak": issued / (v_stage[vk] * 1e-3) / FP64_VECTOR_PEAK, "source": rec.get("flops_source")}
                for b in (va, vb, vv):
                    b.free()
                ve.close()
            # ... and a model with MOVING boxes: Franka-P with the reference Panda's ten finger-pad boxes (panda.xml:134-241),
            # 488 pairs instead of 213 -- its own generated library (whole frames in registers, two waves per SIMD) against
            # the interpreting kernels, which also give the verdicts to compare with
            pm = scenes.franka_p(obstacles=True, pads=True)
            pe, pi = make_engine(True, True, pm), make_engine(True, False, pm)
            if pe.spec_kind() != 1:
                sys.exit("bench.py: the moving-boxes variant found no library of its own under mjpl_amd/csrc/spec "
                         "(__graft_entry__.build() makes it)")
            pa, pb, pv, pw = pe.alloc(ha.nbytes).upload(ha), pe.alloc(hb.nbytes).upload(hb), pe.alloc(E), pi.alloc(E)
            dt, p_launch, p_stage, _ = time_variant(pe, engine, pa, pb, E, layout, pv, vsteps, 5)
            dti, i_launch, _, _ = time_variant(pi, engine, pa, pb, E, layout, pw, max(10, vsteps // 5), 2)
            same = bool(np.array_equal(pv.download(np.uint8, E), pw.download(np.uint8, E)))
            if not same:
                sys.exit("bench.py: the moving-boxes library's verdicts differ from the interpreting kernels'")
            pinfo = pe.info()
            pnames = stage_names(pinfo, True)
            p_stage = {pnames.get(k, k): v for k, v in p_stage.items()}
            pk = max(p_stage, key=lambda k: p_stage[k])
            variants["moving_boxes"] = {"value": E * vsteps / dt, "unit": "edges/s", "steps": vsteps, "ms_per_step": dt / vsteps * 1e3,
                                        "streams": 1, "model": "Franka-P + 16 obstacles + the Panda's 10 finger-pad boxes (moving boxes)",
                                        "enabled_pairs": int(pinfo["npairs"]), "step_ms_all_kernels": p_launch, "kernels_ms": p_stage,
                                        "dtype": "f32-filter+f64-exact", "library": "this program's own", "fused_edges": bool(pinfo.get("fused_edges")),
                                        "waves_per_workgroup": int(pinfo.get("fused_waves", 0)),
                                        "interpreting_kernels_ms_per_step": i_launch,
                                        "verdicts_equal_interpreting_kernels": same, "edges_compared": E,
                                        "valu_issue": valu_issue(pk, p_stage[pk], profile_record(pk + "_pads", E, args.layout, True, 1))}
            for b_ in (pa, pb, pv):
                b_.free()
            pw.free()
            pe.close()
            pi.close()
            out["variants"] = variants
        if flops.get("flops_per_edge"):
            # the float32 filter does the same geometry in binary32: the float64-equivalent rate is reported for scale
            out["fp64_equivalent"] = {"flops_per_edge": flops["flops_per_edge"],
                                      "rate_TFLOPs": value * float(flops["flops_per_edge"]) / 1e12,
                                      "note": "oracle operation count x edges/s; the filter executes these in binary32",
                                      "source": flops.get("source")}
        if world.world == 1 and not args.no_cpu_baseline:
            cb, v_cpu, n = cpu_baseline(model, qidx, base, qa, qb)
            out["cpu_baseline"] = cb
            if not np.array_equal(v_cpu, valid[:n]):
                sys.exit("bench.py: GPU verdicts differ from the CPU oracle on the baseline sample")
        else:
            out["cpu_baseline"] = None
        _flush_c_stdio()
        print(json.dumps(out), flush=True)

    for e2 in engs[1:]:
        e2.close()
    world.close()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
