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
from those events.  torch is imported only for N > 1 (rendezvous, barrier, max-reduce over RCCL).
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


def profile_record(kernel, E, layout):
    """Counters of the committed rocprofv3 PMC passes of this same command (profiles/pmc.json,
    written by tools/pmc_summary.py): HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950
    correction of MI355X_MICROARCH.md "HBM") and wave-level instruction counts; {} if absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc.json")) as f:
            return json.load(f).get("%s_%d_%s" % (kernel, E, layout), {})
    except (OSError, ValueError):
        return {}


def bench_configs(args):
    """BASELINE configs[1] on one GPU: N = 65 536 Franka-P configurations, self-collision + floor,
    q ~ U[jnt_range] (default_rng(1)), verdicts checked against the oracle on a sample."""
    from mjpl_amd import engine, scenes
    model = scenes.franka_p(obstacles=False)
    qidx = scenes.planning_index(model, scenes.FRANKA_ARM_JOINTS)
    base = model.keyframe("home").qpos.copy()
    eng = engine.Engine(model)
    eng.set_planning(qidx, base)
    N = 65536
    Q = np.random.default_rng(1).uniform(model.jnt_range[qidx, 0], model.jnt_range[qidx, 1], size=(N, len(qidx)))
    h = np.ascontiguousarray(Q.T)
    dq, dv = eng.alloc(h.nbytes).upload(h), eng.alloc(N)
    if args.warmup > 0:
        eng.time_configs_dev(dq.ptr, N, engine.SOA, dv.ptr, args.warmup)
    eng.sync()
    t0 = time.perf_counter()
    ms = eng.time_configs_dev(dq.ptr, N, engine.SOA, dv.ptr, args.steps)
    elapsed = time.perf_counter() - t0
    valid = dv.download(np.uint8, N)
    out = {"metric": "validated configurations/sec, Franka-P self-collision (BASELINE configs[1])",
           "value": N * args.steps / elapsed, "unit": "configs/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None,
           "dtype": "f32-filter+f64-exact" if eng.info()["filter_enabled"] else "f64", "data": "synthetic",
           "config": {"workload": f"configs[1]: Franka-P 7-DoF, self-collision + floor, {N} configurations/launch",
                      "valid_fraction": float(valid.mean()), "step_ms_hip_events": float(np.mean(ms))},
           "roofline": {"bound": "hbm", "achieved": 57 * N / (float(np.mean(ms)) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": 57 * N / (float(np.mean(ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": None, "note": "57 algorithmic bytes per configuration; ALU/issue bound"}}
    if not args.no_cpu_baseline:
        from oracle import pyoracle
        orc = pyoracle.Oracle(model, planning_qidx=qidx, qpos_base=base)
        cores = host_cpu()[1]  # (the cores this process may use: cgroup quota, affinity)
        t0 = time.perf_counter()
        v = orc.valid_configs(Q, nthreads=cores)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": N / dt, "unit": "configs/s", "cores": cores, "kind": "port",
                               "sample": f"one pass over the {N} configurations, {cores} pthreads"}
        if not np.array_equal(v.astype(np.uint8), valid):
            sys.exit("bench.py: GPU verdicts differ from the CPU oracle")
    print(json.dumps(out), flush=True)


def bench_next_rows(args):
    """One GPU's share of BASELINE configs[3] (PoseConstraint projections of 1 M / 8 samples) or
    configs[4] (128k / 8 IK seeds -> FK -> collision filter); device-resident inputs for the
    projection, host buffers (PCIe included) for the IK seeds.  Results are checked against the
    oracle on a sample."""
    import mjpl_amd as mjpl
    from mjpl_amd import scenes
    from oracle import pyoracle
    model = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    q_home = model.keyframe("home").qpos.copy()
    cc = mjpl.CollisionConstraint(model)
    eng = cc.engine
    lo, hi = model.jnt_range[:, 0], model.jnt_range[:, 1]
    rng = np.random.default_rng(4)
    common = {"n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
              "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
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
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pc._proj.apply_dev(dqo.ptr, dq.ptr, n, dout.ptr, dok.ptr, dit.ptr)
        eng.sync()
        elapsed = time.perf_counter() - t0
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
        out = dict(common, metric="PoseConstraint projections/sec, Franka-P ee roll/pitch +-0.1 (one GPU of configs[3])",
                   value=n * args.steps / elapsed, unit="rows/s", ms_per_step=elapsed / args.steps * 1e3,
                   config={"workload": f"{n} rows per launch, {np.abs(iters).mean():.1f} projection steps per row on average",
                           "accepted_fraction": float(ok.mean())},
                   roofline={"bound": "hbm", "achieved": n * (2 * 8 * model.nq + 8 * model.nq + 1) * args.steps / elapsed / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": n * (3 * 8 * model.nq + 1) * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, "traffic": None,
                             "note": "FP64 issue / latency bound (FK + Jacobian + a certified 6x6 Cholesky solve per projection step; eigen-decomposition only near singularities)"},
                   cpu_baseline={"value": k / dtc, "unit": "rows/s", "cores": cores, "kind": "port",
                                 "sample": f"first {k} rows, {cores} pthreads"})
    else:
        solver = mjpl.HipIKSolver(model, joints, [], seed=3, num_seeds=16384, iterations=200, engine=eng)
        q_t = mjpl.random_config(model, q_home, joints, 5, [mjpl.JointLimitConstraint(model), cc])
        target = mjpl.site_pose(model, q_t, "ee_site", engine=eng)
        Q0 = solver._seeds(q_home, np.random.default_rng(3))

        def once():
            Qs, ok, its, err = eng.ik_solve("ee_site", target.translation(), target.rotation().wxyz, Q0, solver.movable,
                                            iterations=200, restarts=8, restart_seed=11)
            return Qs, ok, its, cc.valid_configs(Qs[ok])

        for _ in range(max(args.warmup, 1)):
            once()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            Qs, ok, its, free = once()
        elapsed = time.perf_counter() - t0
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
                  "cpu_quota": quota, "sample": f"first {k} seeds, {cores} threads, converged {okc.mean():.3f} "
                                                f"(GPU on the same rows {ok[:k].mean():.3f}), {len(freec)} collision checks"}
        out = dict(common, metric="IK seeds/sec: damped-least-squares seeds -> FK -> collision filter (one GPU of configs[4])",
                   value=len(Q0) * args.steps / elapsed, unit="seeds/s", ms_per_step=elapsed / args.steps * 1e3,
                   config={"workload": f"{len(Q0)} seeds per launch, <= 200 iterations, host buffers (PCIe included)",
                           "converged_fraction": float(ok.mean()), "collision_free_of_converged": float(free.mean()),
                           "mean_iterations": float(its.mean())},
                   roofline={"bound": "hbm", "achieved": len(Q0) * 2 * 8 * model.nq * args.steps / elapsed / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": len(Q0) * 2 * 8 * model.nq * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, "traffic": None,
                             "note": "latency of the iteration chain (256 waves on 1 024 SIMDs)"},
                   cpu_baseline=cpu_ik)
    print(json.dumps(out), flush=True)


def _flush_c_stdio():
    """librccl writes its version banner through C stdio, which is flushed at exit -- after Python's
    own output -- unless it is flushed here: the JSON line is to be the last line of stdout."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except (OSError, AttributeError):
        pass


def bench_rrt(args):
    """One GPU's share of BASELINE configs[3] as the planner runs it (not the headline): rounds of
    `mjpl_rrt_round` -- sample, nearest, extend with projection, validate, connect, exchange over a
    one-rank RCCL communicator -- with 131 072 lanes.  A step is one round; round 1 (empty trees)
    is the warm-up.  The path any round reports is checked against the oracle."""
    import mjpl_amd as mjpl
    from mjpl_amd import engine as eng_mod
    from mjpl_amd import scenes
    from oracle import pyoracle
    L = 131072
    rounds = args.steps if args.steps != 2000 else 5
    m = scenes.franka_p(obstacles=True)
    joints = scenes.FRANKA_ARM_JOINTS
    qidx = scenes.planning_index(m, joints)
    q_init = m.keyframe("home").qpos.copy()
    cc = mjpl.CollisionConstraint(m)
    frame = mjpl.site_pose(m, q_init, "ee_site", engine=cc.engine)
    pc = mjpl.PoseConstraint(m, "ee_site", frame, roll=(-0.1, 0.1), pitch=(-0.1, 0.1), engine=cc.engine)
    cons = [pc, mjpl.JointLimitConstraint(m), cc]
    pc.q_step = np.inf
    q_goal = mjpl.random_config(m, q_init, joints, 7, cons)
    pc.q_step = 0.05
    dev = mjpl.DeviceBiRRT(m, joints, cc, q_init, epsilon=0.05, interval_step=0.01, seed=3, goal_biasing_probability=0.05,
                           batch=L, capacity=1 << 24, pose=pc, comm=(eng_mod.comm_unique_id(), 0, 1))
    dev.rrt.reset(q_init[qidx], q_goal[qidx][None], 3)
    info = dev.rrt.round()  # warm-up: the first round grows from two single-node trees
    rows, new_nodes = [], 0
    t0 = time.perf_counter()
    for _ in range(rounds):
        t1 = time.perf_counter()
        info = dev.rrt.round()
        rows.append((time.perf_counter() - t1) * 1e3)
        new_nodes += int(info.new_nodes[0]) + int(info.new_nodes[1])
    elapsed = time.perf_counter() - t0
    path_ok = None
    if info.connected:
        path = dev.rrt.path()
        full = np.repeat(q_init[None, :], len(path), axis=0)
        full[:, qidx] = path
        orc = pyoracle.Oracle(m, planning_qidx=qidx, qpos_base=q_init)
        ok_edges = orc.valid_edges(path[:-1], path[1:], 0.01, nthreads=8)
        path_ok = bool(ok_edges.all() and np.all(pc.valid_configs(full)))
        if not path_ok:
            raise SystemExit("bench: the planner's path fails the oracle's collision check / the pose constraint")
    _flush_c_stdio()
    print(json.dumps({
        "metric": "RRT samples/sec through the frontier bi-RRT, one GPU's share of BASELINE configs[3]",
        "value": L * rounds / elapsed, "unit": "samples/s", "n_gpus": 1, "steps": rounds, "warmup": 1,
        "ms_per_step": elapsed / rounds * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64 planner + f32-filter+f64-exact validation", "data": "synthetic",
        "config": {"workload": f"{L} lanes per round, [PoseConstraint(roll, pitch +-0.1), JointLimit, Collision], "
                               "eps 0.05, interval 0.01, Franka-P + 16 obstacles, one-rank RCCL exchange per round",
                   "round_ms": rows, "new_nodes_per_s": new_nodes / elapsed,
                   "nodes": [int(info.nodes[0]), int(info.nodes[1])], "connected": bool(info.connected),
                   "path_checked_against_oracle": path_ok}}))
    dev.rrt.close()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000,
                    help="timed steps; the default keeps the timed region near one second")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--edges", type=int, default=EDGES_PER_GPU)
    ap.add_argument("--layout", choices=["soa", "aos"], default="soa")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["edges", "configs", "pose", "ik", "rrt"], default="edges",
                    help="edges: the headline metric (BASELINE configs[2]).  Extra lines, not the headline: "
                         "configs = configs[1], 65 536 Franka-P self-collision configurations per launch; "
                         "pose = one GPU's share of configs[3], 131 072 PoseConstraint projections; "
                         "ik = one GPU's share of configs[4], 16 384 IK seeds -> FK -> collision filter; "
                         "rrt = one GPU's share of configs[3] as the planner runs it: rounds of the device-resident "
                         "frontier bi-RRT, 131 072 samples each, [PoseConstraint, JointLimit, Collision], through a "
                         "one-rank RCCL communicator (--steps = timed rounds, default 5)")
    args = ap.parse_args()
    if args.workload == "configs":
        return bench_configs(args)
    if args.workload in ("pose", "ik"):
        return bench_next_rows(args)
    if args.workload == "rrt":
        return bench_rrt(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    dist = torch = None
    # MJPL_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL group, barrier, max-reduce) even
    # with one rank, to exercise it on a single-GPU box
    if world > 1 or os.environ.get("MJPL_BENCH_FORCE_DIST") == "1":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from mjpl_amd import engine, scenes

    model = scenes.franka_p(obstacles=True)
    qidx = scenes.planning_index(model, scenes.FRANKA_ARM_JOINTS)
    base = model.keyframe("home").qpos.copy()
    eng = engine.Engine(model, device=local_rank)
    eng.set_planning(qidx, base)
    info = eng.info()

    E = args.edges
    qa, qb = make_edges(model, qidx, E, seed=2 + rank)
    layout = engine.SOA if args.layout == "soa" else engine.AOS
    ha = np.ascontiguousarray(qa.T) if layout == engine.SOA else qa
    hb = np.ascontiguousarray(qb.T) if layout == engine.SOA else qb
    dqa, dqb = eng.alloc(ha.nbytes).upload(ha), eng.alloc(hb.nbytes).upload(hb)
    dvalid = eng.alloc(E)
    eng.sync()

    def barrier():
        eng.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

    # warmup (untimed)
    if args.warmup > 0:
        eng.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, STEP, layout, dvalid.ptr, args.warmup, 1 << 30)
    barrier()
    t0 = time.perf_counter()
    # EXACTLY args.steps launches, back to back on the engine's stream; every 4th one also carries
    # one HIP event after each of its kernels (the per-kernel durations roofline.achieved uses)
    launch_ms, stage_ms, nsamp = eng.time_edges_stages_dev(dqa.ptr, dqb.ptr, E, STEP, layout, dvalid.ptr,
                                                           args.steps, 4)  # syncs
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    valid = dvalid.download(np.uint8, E)

    if rank == 0:
        total_edges = E * world * args.steps
        value = total_edges / elapsed
        filt = bool(info["filter_enabled"])
        interior = eng.last_interior_edges() if filt else 0
        items = eng.last_items() if filt else 0
        # units and SURVEY.md 8(d) algorithmic bytes of every kernel of a step:
        #   endpoint pass: one edge = 2 x 56 B of columns read + 1 B verdict (113 B)
        #   item pass:     one waypoint configuration = 56 B + 1 B (57 B per check) + its 8 B (edge, index)
        #   walking pass:  one edge, 113 B + the 4 B list entry
        per_stage = {"k_filter_endpoints": (E, BYTES_PER_EDGE, "edges"),
                     "k_filter_items": (items, 57 + 8, "waypoint configurations"),
                     "k_filter_edges": (max(interior - 0, 0) if items == 0 else 0, BYTES_PER_EDGE + 4, "edges"),
                     "k_patch_pairs": (eng.last_undecided(), 56 + 16, "undecided geom pairs"),
                     "k_check_edges": (E if not filt else 0, BYTES_PER_EDGE, "edges")}
        # the dominant kernel of THIS run = the longest stage
        kernel = max(stage_ms, key=lambda k: stage_ms[k])
        kernel_ms = stage_ms[kernel]
        units, unit_bytes, unit_name = per_stage[kernel]
        achieved = unit_bytes * units / (kernel_ms * 1e-3) / 1e9
        prof = profile_record(kernel, E, args.layout)
        workload = (f"configs[2]: Franka-P 7-DoF + 16 box/sphere obstacles + floor, {E} edges/GPU, "
                    f"eps {EPS}, step {STEP} (endpoint + interior waypoints per edge)")
        out = {
            "metric": METRIC, "value": value, "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # what ran: a binary32 filter decides what it can within its tolerance band, binary64
            # kernels decide the rest; every verdict equals the pure binary64 path's
            "dtype": "f32-filter+f64-exact" if filt else "f64",
            "data": "synthetic",
            "config": {"workload": workload, "edges_per_gpu": E, "layout": args.layout,
                       "geom_pairs": info["npairs"], "valid_fraction": float(valid.mean()),
                       "float32_filter": filt, "filter_tol_m": info["filter_tol"],
                       "specialised_kernels": eng.spec_loaded(),
                       "undecided_items_last_step": eng.last_undecided(),
                       "edges_reaching_interior_pass": interior, "interior_waypoint_items": items,
                       "parallelism": f"edge-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": prof.get("hbm_bytes_per_launch"),
                         "kernel": kernel, "kernel_ms": kernel_ms, "step_ms_all_kernels": launch_ms,
                         "kernels_ms": stage_ms, "kernel_samples": nsamp,
                         "algorithmic_bytes_per_unit": unit_bytes, "units_in_this_kernel": units,
                         "unit_of_work": unit_name,
                         "whole_step": {"algorithmic_bytes": BYTES_PER_EDGE * E,
                                        "achieved_GBs": BYTES_PER_EDGE * E / (launch_ms * 1e-3) / 1e9,
                                        "frac": BYTES_PER_EDGE * E / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "note": "vector-issue / latency bound, not HBM bound (SURVEY.md 8d); see valu_issue"},
        }
        # vector-ALU issue occupancy of the dominant kernel, from the committed counter pass:
        # SQ_INSTS_VALU wave-instructions per launch / (kernel time x issue slots per second)
        if prof.get("SQ_INSTS_VALU"):
            iv = float(prof["SQ_INSTS_VALU"])
            out["valu_issue"] = {"kernel": kernel, "wave_insts_valu_per_launch": iv,
                                 "wave_insts_salu_per_launch": prof.get("SQ_INSTS_SALU"),
                                 "achieved_per_s": iv / (kernel_ms * 1e-3), "peak_per_s": VALU_ISSUE_PEAK,
                                 "frac": iv / (kernel_ms * 1e-3) / VALU_ISSUE_PEAK,
                                 "source": prof.get("source"),
                                 "note": "peak = 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU "
                                         "instruction (MI355X_MICROARCH.md, Wave scheduling)"}
        else:
            out["valu_issue"] = None
        if world == 1 and not args.no_cpu_baseline:
            cb, v_cpu, n = cpu_baseline(model, qidx, base, qa, qb)
            out["cpu_baseline"] = cb
            if not np.array_equal(v_cpu, valid[:n]):
                sys.exit("bench.py: GPU verdicts differ from the CPU oracle on the baseline sample")
        else:
            out["cpu_baseline"] = None
        _flush_c_stdio()
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
