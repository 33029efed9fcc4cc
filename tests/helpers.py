"""Test-only helpers.  OracleCollisionConstraint lets the host-side planners run on a CPU
box: it answers the Constraint interface from the CPU ORACLE (allowed in tests/ only)."""
import numpy as np

from mjpl_amd.constraint import Constraint


class OracleCollisionConstraint(Constraint):
    def __init__(self, model, allowed_collision_bodies=(), pyoracle=None):
        if pyoracle is None:
            from oracle import pyoracle
        self.model = model
        self.orc = pyoracle.Oracle(model, allowed_collision_bodies)
        self.calls = 0

    def valid_config(self, q):
        self.calls += 1
        return self.orc.valid_config(q)

    def apply(self, q_old, q):
        return q if self.valid_config(q) else None

    def valid_interval(self, start, end, step_dist):
        return self.orc.valid_collision_interval(start, end, step_dist)


class BatchedOracleCollisionConstraint(OracleCollisionConstraint):
    """The same verdicts with the row-wise methods the batched extension needs."""
    projects = False

    def valid_configs(self, Q):
        self.calls += 1
        return self.orc.valid_configs(np.asarray(Q, dtype=np.float64), nthreads=2).astype(bool)

    def valid_intervals(self, starts, ends, step_dist):
        return np.array([self.orc.valid_collision_interval(a, b, step_dist) for a, b in zip(starts, ends)], dtype=bool)


def uniform_configs(model, n, seed, fingers=0.04):
    rng = np.random.default_rng(seed)
    Q = rng.uniform(model.jnt_range[:, 0], model.jnt_range[:, 1], size=(n, model.nq))
    if model.nq == 9:
        Q[:, 7:] = fingers
    return Q


def random_edges(model, qidx, n, seed, eps=0.05):
    rng = np.random.default_rng(seed)
    lo, hi = model.jnt_range[qidx, 0], model.jnt_range[qidx, 1]
    qa = rng.uniform(lo, hi, size=(n, len(qidx)))
    d = rng.normal(size=(n, len(qidx)))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return qa, np.clip(qa + eps * d, lo, hi)
