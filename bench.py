#!/usr/bin/env python3
"""Headline benchmark: validated RRT edges/sec on the Franka 7-DoF 16-obstacle scene.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (the edge kernel: endpoint + interior waypoints of
every edge, FK + narrowphase per waypoint) over one batch of E synthetic edges that is
already resident in HBM.  Weak scaling: every rank owns its own E edges (independent units,
no data-path collective).  Rank 0 prints ONE JSON line.

The timed K steps are launched through mjpl_time_edges_dev, which brackets every launch
with HIP events on the stream the kernel runs on; roofline.achieved comes from those.
torch is imported only for N > 1 (rendezvous, barrier, max-reduce over RCCL).
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
FLOPS_PER_EDGE = 80e3        # SURVEY.md 8(d) estimate, FP64
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6     # MI355X vector FP64 (half of the 157.3 TF FP32 vector peak)


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


def cpu_baseline(model, qidx, base, qa, qb):
    """The oracle (a port, not the reference) timed on this box's host cores on a bounded
    sample of the same edges; reported, never the target."""
    from oracle import pyoracle
    orc = pyoracle.Oracle(model, planning_qidx=qidx, qpos_base=base)
    cores = os.cpu_count() or 1
    pilot = min(2048, len(qa))
    t0 = time.perf_counter()
    orc.valid_edges(qa[:pilot], qb[:pilot], STEP, nthreads=1)
    per_edge = (time.perf_counter() - t0) / pilot
    # bounded sample: about 15 core-seconds of oracle work in total, i.e. `reps` passes over
    # the first n edges of this rank's batch with every host core busy
    n = len(qa)
    reps = max(1, int(round(15.0 / (per_edge * n))))
    t0 = time.perf_counter()
    for _ in range(reps):
        v = orc.valid_edges(qa[:n], qb[:n], STEP, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": reps * n / dt, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"{reps} passes over the {n} edges of rank 0 (~{reps * n * per_edge:.0f} core-seconds), "
                      f"oracle/libmjpl_oracle.so (gcc -O2 -ffp-contract=off), {cores} pthreads; "
                      f"1-thread pilot {1.0 / per_edge:.0f} edges/s"}, v, n


def traffic_from_profile(workload_key):
    """HBM bytes per launch from the committed rocprofv3 PMC pass of this same command
    (profiles/*_traffic.json, written by tools/pmc_summary.py); None if absent."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t.get(workload_key, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


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
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"configs[1]: Franka-P 7-DoF, self-collision + floor, {N} configurations/launch",
                      "valid_fraction": float(valid.mean()), "step_ms_hip_events": float(np.mean(ms))},
           "roofline": {"bound": "hbm", "achieved": 57 * N / (float(np.mean(ms)) * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": 57 * N / (float(np.mean(ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": None, "note": "57 algorithmic bytes per configuration; ALU/issue bound"}}
    if not args.no_cpu_baseline:
        from oracle import pyoracle
        orc = pyoracle.Oracle(model, planning_qidx=qidx, qpos_base=base)
        cores = os.cpu_count() or 1
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
        cores = os.cpu_count() or 1
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
                             "note": "FP64 issue bound (6x6 Jacobi eigen-decomposition per projection step)"},
                   cpu_baseline={"value": k / dtc, "unit": "rows/s", "cores": cores, "kind": "port",
                                 "sample": f"first {k} rows, {cores} pthreads"})
    else:
        solver = mjpl.HipIKSolver(model, joints, [], seed=3, num_seeds=16384, iterations=200, engine=eng)
        q_t = mjpl.random_config(model, q_home, joints, 5, [mjpl.JointLimitConstraint(model), cc])
        target = mjpl.site_pose(model, q_t, "ee_site", engine=eng)
        Q0 = solver._seeds(q_home, np.random.default_rng(3))

        def once():
            Qs, ok, its, err = eng.ik_solve("ee_site", target.translation(), target.rotation().wxyz, Q0, solver.movable,
                                            iterations=200)
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
        out = dict(common, metric="IK seeds/sec: damped-least-squares seeds -> FK -> collision filter (one GPU of configs[4])",
                   value=len(Q0) * args.steps / elapsed, unit="seeds/s", ms_per_step=elapsed / args.steps * 1e3,
                   config={"workload": f"{len(Q0)} seeds per launch, <= 200 iterations, host buffers (PCIe included)",
                           "converged_fraction": float(ok.mean()), "collision_free_of_converged": float(free.mean()),
                           "mean_iterations": float(its.mean())},
                   roofline={"bound": "hbm", "achieved": len(Q0) * 2 * 8 * model.nq * args.steps / elapsed / 1e9,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": len(Q0) * 2 * 8 * model.nq * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, "traffic": None,
                             "note": "latency of the iteration chain (256 waves on 1 024 SIMDs)"},
                   cpu_baseline=None)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--edges", type=int, default=EDGES_PER_GPU)
    ap.add_argument("--layout", choices=["soa", "aos"], default="soa")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["edges", "configs", "pose", "ik"], default="edges",
                    help="edges: the headline metric (BASELINE configs[2]).  Extra lines, not the headline: "
                         "configs = configs[1], 65 536 Franka-P self-collision configurations per launch; "
                         "pose = one GPU's share of configs[3], 131 072 PoseConstraint projections; "
                         "ik = one GPU's share of configs[4], 16 384 IK seeds -> FK -> collision filter")
    args = ap.parse_args()
    if args.workload == "configs":
        return bench_configs(args)
    if args.workload in ("pose", "ik"):
        return bench_next_rows(args)

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
        eng.time_edges_dev(dqa.ptr, dqb.ptr, E, STEP, layout, dvalid.ptr, args.warmup)
    barrier()
    t0 = time.perf_counter()
    ms, ms_kernel = eng.time_edges_dev(dqa.ptr, dqb.ptr, E, STEP, layout, dvalid.ptr, args.steps,
                                       first_kernel=True)  # syncs
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
        launch_ms = float(np.mean(ms))          # all kernels of one step (filter + exact re-run)
        kernel_ms = float(np.mean(ms_kernel))   # the dominant kernel alone
        # The filter runs two passes: the endpoints of all edges (which also writes the interior
        # waypoints of the edges whose endpoint passed as work items), then one lane per waypoint
        # item.  The dominant kernel is the item pass: a unit is one configuration check, 56 B of
        # columns + 8 B (edge, index) read and 1 B written (57 B per check, + the item).
        interior = eng.last_interior_edges() if info["filter_enabled"] else 0
        items = eng.last_items() if info["filter_enabled"] else 0
        if items > 0:
            kernel, units, unit_bytes, unit_name = "k_filter_items", items, 57 + 8, "waypoint configurations"
        elif interior > 0:
            kernel, units, unit_bytes, unit_name = "k_filter_edges", interior, BYTES_PER_EDGE + 4, "edges"
        else:
            kernel = "k_filter_edges" if info["filter_enabled"] else "k_check_edges"
            units, unit_bytes, unit_name = E, BYTES_PER_EDGE, "edges"
        achieved = unit_bytes * units / (kernel_ms * 1e-3) / 1e9
        workload = (f"configs[2]: Franka-P 7-DoF + 16 box/sphere obstacles + floor, {E} edges/GPU, "
                    f"eps {EPS}, step {STEP} (endpoint + interior waypoints per edge)")
        out = {
            "metric": METRIC, "value": value, "unit": "edges/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "edges_per_gpu": E, "layout": args.layout,
                       "geom_pairs": info["npairs"], "valid_fraction": float(valid.mean()),
                       "float32_filter": bool(info["filter_enabled"]), "filter_tol_m": info["filter_tol"],
                       "undecided_items_last_step": eng.last_undecided(),
                       "edges_reaching_interior_pass": interior, "interior_waypoint_items": items,
                       "parallelism": f"edge-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic_from_profile("%s_%d_%s" % (kernel, E, args.layout)),
                         "kernel": kernel, "kernel_ms": kernel_ms, "step_ms_all_kernels": launch_ms,
                         "algorithmic_bytes_per_unit": unit_bytes, "units_in_this_kernel": units,
                         "unit_of_work": unit_name,
                         "note": "ALU/issue bound, not HBM bound (SURVEY.md 8d); see roofline_valu"},
            "roofline_valu": {"bound": "vector_alu", "achieved": FLOPS_PER_EDGE * E / (launch_ms * 1e-3) / 1e12,
                              "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                              "frac": FLOPS_PER_EDGE * E / (launch_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
                              "flops_per_edge_estimate": FLOPS_PER_EDGE,
                              "note": "algorithmic FP64 flop estimate of SURVEY.md 8d against the FP64 vector "
                                      "peak; the filter executes most of them in float32"},
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, v_cpu, n = cpu_baseline(model, qidx, base, qa, qb)
            out["cpu_baseline"] = cb
            if not np.array_equal(v_cpu, valid[:n]):
                sys.exit("bench.py: GPU verdicts differ from the CPU oracle on the baseline sample")
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
