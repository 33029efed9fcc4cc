"""bench.py's multi-rank harness.  What the ranks need from one another in the validation workloads -- a
barrier either side of the timed region, the slowest rank's time -- travels over a Unix socket rank 0 opens
(bench.World); `python bench.py --gpus N` without a launcher starts its own N ranks.  The socket protocol runs
here with three CPU processes; on the GPU box two ranks share the one GPU (MJPL_BENCH_SHARE_GPU=1: launch,
rendezvous and timing protocol under test, not a measurement), started both ways the driver may start them."""
import json
import os
import subprocess
import sys
import tempfile
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_rendezvous_three_cpu_processes():
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        w = bench.World()
        w.connect()
        g = w.gather(10.0 + w.rank)
        uid = w.exchange((b"A" * 128) if w.rank == 0 else bytes(128))[0]
        assert uid == b"A" * 128
        w.barrier()
        assert list(g) == [10.0, 11.0, 12.0], g
        assert list(w.gather(-w.rank)) == [0.0, -1.0, -2.0]
        w.close()
    """)
    rdzv = os.path.join(tempfile.gettempdir(), "mjpl_test_%d.sock" % os.getpid())
    procs = [subprocess.Popen([sys.executable, "-c", code],
                              env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="3", MJPL_BENCH_RDZV=rdzv))
             for r in range(3)]
    assert [p.wait(timeout=120) for p in procs] == [0, 0, 0]
    assert not os.path.exists(rdzv)


def _json_line(out: str) -> dict:
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_two_ranks_on_one_gpu(launcher):
    env = dict(os.environ, MJPL_BENCH_SHARE_GPU="1")
    args = ["bench.py", "--gpus", "2", "--steps", "20", "--warmup", "2", "--no-cpu-baseline", "--no-variants"]
    if launcher == "self":
        cmd = [sys.executable, *args]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29517", *args]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _json_line(res.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["ranks_share_a_gpu"] is True and d["cpu_baseline"] is None
    one = d["config"]["edges_per_gpu"]
    assert abs(d["value"] - 2 * one * 20 / (d["ms_per_step"] * 20e-3)) <= 1e-6 * d["value"]  # whole-job rate: both ranks' edges / the slowest rank's time


@pytest.mark.gpu
def test_planner_workload_through_a_one_rank_communicator():
    res = subprocess.run([sys.executable, "bench.py", "--workload", "rrt", "--steps", "1", "--lanes", "4096", "--capacity", "1048576"],
                         cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    d = _json_line(res.stdout)
    assert d["n_gpus"] == 1 and d["config"]["rccl_ranks"] == 1 and d["config"]["nodes"][0] > 1


def test_a_failing_rank_ends_the_launch():
    """Without a GPU every rank fails when it creates its engine ("no CPU fallback"); the launcher must pass that on
    at once instead of leaving ranks at the rendezvous (here: no GPU in the container; on the GPU box: skipped)."""
    from mjpl_amd import engine
    try:
        if engine.device_count() > 0:
            pytest.skip("a GPU is present")
    except Exception:
        pass
    res = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5"], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert res.returncode != 0
    assert "no CPU fallback" in res.stderr
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
