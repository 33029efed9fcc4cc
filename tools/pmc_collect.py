"""Fold the per-kernel PMC summaries of one profiling round (profiles/<tag>_pmc_<kernel>.json,
written by tools/pmc_summary.py from separate rocprofv3 --pmc passes) into profiles/pmc.json, the
file bench.py reads `roofline.traffic` and `valu_issue` from.
Records are keyed by kernel, batch size, layout AND the engine configuration the pass ran under
(float32 filter on / off, specialised kernels on / off: bench.py --variant).
usage: python tools/pmc_collect.py <tag> [E] [layout] [filter 0|1] [spec 0|1|2] [flops tag]"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mjpl_amd import build as _build  # noqa: E402

# the digest of the kernel sources the counters were captured under (bench.py says so when a line's build has another)
STAMP = "%016x" % _build.src_stamp()
# kernels of the next rows' lines: (bench.py --workload, rows per launch, timed steps of the profiled command)
NEXT_ROWS = {"k_pose_apply_rows": ("pose", 131072, 1), "k_ik_solve_rows": ("ik", 16384, 1),
             "k_rrt_gen_project_rows": ("rrt", 131072, 2), "k_rrt_gen_project_ahead": ("rrt", 131072, 2), "k_nearest_mfma": ("rrt", 131072, 2), "k_filter_configs": ("configs", 65536, 1)}


def main():
    tag = sys.argv[1]
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
    layout = sys.argv[3] if len(sys.argv) > 3 else "soa"
    filt = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    spec = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    flops_tag = sys.argv[6] if len(sys.argv) > 6 else None
    path = os.path.join(ROOT, "profiles", "pmc.json")
    try:
        with open(path) as f:
            out = json.load(f)
    except (OSError, ValueError):
        out = {}
    edge_kernels = ("k_edges_fused", "k_filter_", "k_tail", "k_check_edges", "k_patch_pairs")  # (the headline workload's; the next rows' are not keyed here)
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{tag}_pmc_*.json"))):
        kernel = os.path.basename(p)[len(tag) + 5:-5]
        if not kernel.startswith(edge_kernels) or kernel in NEXT_ROWS:
            continue
        with open(p) as f:
            r = json.load(f)
        rec = {k: r[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_WAVES",
                                 "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU",
                                 "GRBM_GUI_ACTIVE", "avg_ns") if k in r}
        if "FETCH_SIZE" in r or "WRITE_SIZE" in r:
            rec["FETCH_SIZE_KiB"] = r.get("FETCH_SIZE")
            rec["WRITE_SIZE_KiB"] = r.get("WRITE_SIZE")
            rec["hbm_bytes_per_launch"] = (2 * r.get("FETCH_SIZE", 0.0) + r.get("WRITE_SIZE", 0.0)) * 1024
            rec["correction"] = ("gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section); "
                                 "FETCH_SIZE and WRITE_SIZE from separate --pmc passes")
        rec["source"] = f"profiles/{os.path.basename(p)} (tools/{'profile_pads.sh' if kernel.endswith('_pads') else 'profile_gpu.sh'} {tag})"
        # the kernel's floating-point instruction mix, from its own counter pass (tools/profile_next_rows.sh <flops tag>)
        fl = os.path.join(ROOT, "profiles", f"{flops_tag}_pmc_flops_{kernel}.json") if flops_tag else None
        if fl and os.path.exists(fl):
            with open(fl) as f:
                fr = json.load(f)
            for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"):
                if k in fr:
                    rec[k] = fr[k]
            rec["flops_source"] = f"profiles/{os.path.basename(fl)} (tools/profile_next_rows.sh {flops_tag})"
        rec["src_stamp"] = STAMP
        out[f"{kernel}_{E}_{layout}_f{filt}_s{spec}"] = rec
    # the next rows' lines (bench.py --workload pose / ik / rrt / configs): keyed by kernel and workload size; the
    # duration is the kernel's average in the SAME profiling round's kernel trace (pmc_summary.py: avg_ns, calls)
    for kernel, (workload, size, timed_steps) in NEXT_ROWS.items():
        p = os.path.join(ROOT, "profiles", f"{tag}_pmc_{kernel}.json")
        if not os.path.exists(p):
            continue
        with open(p) as f:
            r = json.load(f)
        rec = {k: r[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_WAVE_CYCLES",
                                 "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE", "avg_ns", "calls") if k in r}
        if workload == "rrt" and "calls" in r:
            # (the profiled command runs one timed round after the warm-up round -- the first round of the search, from single-node
            #  trees, 131 072 lanes all the same: two rounds' launches in the trace)
            rec["calls_per_round"] = r["calls"] / float(timed_steps)
        rec["source"] = f"profiles/{os.path.basename(p)} + profiles/{tag}_{workload}_kernel_stats.csv (tools/profile_next_rows.sh {tag})"
        rec["src_stamp"] = STAMP
        out[f"{kernel}_{workload}{size}"] = rec
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v.get("source") for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
