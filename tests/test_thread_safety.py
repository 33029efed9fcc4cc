"""The process-wide state of libmjpl_hip.so under the header's "one engine per thread" contract
(include/mjpl_hip.h): the table of loaded per-model libraries is shared by every engine of the process
and is filled by whichever thread asks first.  The host half (compile a model, look its library up) runs
without a GPU; the GPU half creates, uses and destroys engines from eight threads at once."""
import ctypes as C
import os
import sys
import threading

import numpy as np
import pytest

from mjpl_amd import engine, scenes, specialise

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _models():
    from spec_models import spec_models
    return spec_models()


def _run_threads(n, fn):
    errors, barrier = [], threading.Barrier(n)

    def body(k):
        try:
            barrier.wait()  # all threads enter the library together
            fn(k)
        except BaseException as exc:  # noqa: BLE001 -- reported by the caller's assert
            errors.append((k, repr(exc)))

    threads = [threading.Thread(target=body, args=(k,)) for k in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_library_table_from_eight_threads():
    """mjpl_program_dump + mjpl_spec_probe (the look-up of mjpl_create) from 8 threads: every thread sees
    the same answer for a hash as a single-threaded look-up of the file system gives, also for hashes
    nobody has looked up before the threads start (first insertions race unless the table is locked)."""
    lib = engine.load_library()
    entries = _models()
    want = {}
    for k, (_, m, allowed, qidx, base) in enumerate(entries):
        info = specialise.dump_program(m, allowed, qidx, base)[3]
        want[k] = (int(info.hash), int(info.robot_hash),
                   os.path.exists(specialise.spec_path(int(info.hash))),
                   os.path.exists(specialise.spec_path(int(info.robot_hash), True)) and bool(info.scene_ok))
    if os.environ.get("MJPL_SPEC", "1") == "0":
        pytest.skip("MJPL_SPEC=0: no library is ever looked up")

    def work(t):
        rng = np.random.default_rng(t)
        for it in range(40):
            k = int(rng.integers(len(entries)))
            _, m, allowed, qidx, base = entries[k]
            info = specialise.dump_program(m, allowed, qidx, base)[3]
            assert (int(info.hash), int(info.robot_hash)) == want[k][:2]  # the compiler itself is re-entrant
            assert bool(lib.mjpl_spec_probe(C.c_uint64(want[k][0]), 0)) == want[k][2]
            # a hash no library exists for: inserted as "absent" by whichever thread comes first
            bogus = (0x5EED0000 + 977 * t + it) | (1 << 62)
            assert lib.mjpl_spec_probe(C.c_uint64(bogus), 0) == 0
            assert lib.mjpl_spec_probe(C.c_uint64(bogus), 1) == 0
            if want[k][3]:
                assert lib.mjpl_spec_probe(C.c_uint64(want[k][1]), 1) == 1

    _run_threads(8, work)


@pytest.mark.gpu
def test_engines_from_eight_threads():
    """Eight threads, each creating its own engines (the documented usage), checking a batch against the
    answers of an engine made before the threads started, and destroying them -- while the others do."""
    entries = _models()[:4]
    rng = np.random.default_rng(7)
    batches, want = [], []
    for _, m, allowed, qidx, base in entries:
        lo, hi = m.jnt_range[qidx, 0], m.jnt_range[qidx, 1]
        qa = rng.uniform(lo, hi, size=(3000, len(qidx)))
        d = rng.normal(size=qa.shape)
        qb = np.clip(qa + 0.05 * d / np.linalg.norm(d, axis=1, keepdims=True), lo, hi)
        with engine.Engine(m, allowed) as e:
            e.set_planning(qidx, base)
            want.append((e.check_edges(qa, qb, 0.01), e.spec_kind()))
        batches.append((qa, qb))

    def work(t):
        for it in range(6):
            k = (t + it) % len(entries)
            _, m, allowed, qidx, base = entries[k]
            with engine.Engine(m, allowed) as e:
                e.set_planning(qidx, base)
                assert e.spec_kind() == want[k][1]
                got = e.check_edges(batches[k][0], batches[k][1], 0.01)
                assert np.array_equal(got, want[k][0])

    _run_threads(8, work)
