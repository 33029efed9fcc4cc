"""Per-model specialised filter kernels.

``mjpl_create`` compiles a model into a *program* -- control words ``ip`` and constant tables
``fp`` / ``dp`` -- which the generic kernels of ``mjpl_device.h`` interpret: every body, joint, geom
and four-row chunk of partners costs scalar loads of control words, address arithmetic, loop
branches and mask bookkeeping (0.78 scalar instructions per vector instruction in the item pass,
round-1 counters).  This module turns ONE program into straight-line HIP source instead:

* forward kinematics with the model's constants as literals, terms that multiply an exact 0
  dropped and exact +-1 folded at generation time (a hinge about z contributes two products per
  quaternion component instead of four);
* one bounding cull per ENABLED partner, operands as literals, each hit mask parked in lane k of a
  register pair (``v_writelane``) so that one rolled loop per geom -- the only copy of the push and
  drain code -- picks the partners some lane passed (``ballot`` of the non-zero lanes);
* the candidate queues, the drains (full-lane narrowphase), the exact re-check of undecided pairs
  and every table they read are the generic ones: verdicts are the same.

The generated code is compiled with the kernels of ``mjpl_filter.h`` into
``mjpl_amd/csrc/spec/libmjpl_spec_<hash>.so``; an engine whose program has that hash loads it
(``mjpl_hip.hip: load_spec``).  Building needs ``hipcc`` and no GPU (``mjpl_program_dump`` compiles
the model on the host), so it runs in the build container, or on a GPU box BEFORE the process
touches the GPU.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
import threading

import numpy as np

from . import build as _build
from . import engine as _engine

SPEC_DIR = os.path.join(_build.CSRC, "spec")

# program layout (mjpl_device.h)
H_NBODYOPS, H_NPLAN, H_NSAVE, H_NSLOTS, H_OFF_BODYOPS, H_OFF_PERM, H_OFF_WCULL, H_OFF_WNARROW, H_NWORLD, H_NWPAD, \
    H_OFF_FCONST, H_SIZE = range(12)
B_PARENT, B_DOFF, B_BODYID, B_NJNT, B_SAVE, B_NGEOM, B_SIZE = range(7)
J_TYPE, J_QSRC, J_FLAGS, J_DOFF, J_SIZE = range(5)
G_TYPE, G_FLAGS, G_DOFF, G_STORE, G_GEOMID, G_SMASK, G_WMASK_LO, G_WMASK_HI, G_PMASK_LO, G_PMASK_HI, G_SIZE = range(11)
MAX_SLOTS = 32
GD_SIZE, GD_WBOUND = 7, 12
GF_SAMEPOS, GF_SAMEROT = 1, 2
JF_POS_NONZERO = 1
PARENT_CUR, PARENT_STATIC = 0, -1
JT_SLIDE, JT_HINGE = 2, 3
GT_PLANE, GT_SPHERE, GT_CAPSULE, GT_BOX = 0, 2, 3, 6
EK_PLANE, EK_STATIC, EK_SLOT = 0, 1, 2
WN_ZAXIS, WN_LEN = 0, 12
P_FIRST = 1 << 17
SLOT_NONE = 63
FC_MAXCOORD, FC_MAXANGLE = 0, 1


class ProgramInfo(C.Structure):
    _fields_ = [("hash", C.c_uint64), ("maxs", C.c_int32), ("wbox", C.c_int32), ("mbox", C.c_int32),
                ("immediate", C.c_int32), ("filter_usable", C.c_int32), ("filter_tol", C.c_float),
                ("nslots", C.c_int32), ("nsave", C.c_int32), ("spec_abi", C.c_int32),
                ("robot_hash", C.c_uint64), ("scene_rows", C.c_int32), ("scene_ok", C.c_int32)]


def dump_program(model, allowed_collision_bodies=(), qidx=None, qpos_base=None, filter_tol: float = 0.0):
    """Compile `model` on the host (no GPU) -> (ip int32[], fp float32[], dp float64[], ProgramInfo)."""
    lib = _engine.load_library()
    f = lib.mjpl_program_dump
    f.restype = C.c_int
    d = _engine._ModelDesc()
    d.nq, d.njnt, d.nbody, d.ngeom = model.nq, model.njnt, model.nbody, model.ngeom
    keep = []
    for name, typ in _engine._ModelDesc._fields_[4:]:
        arr = getattr(model, name)
        arr = _engine._i32(arr) if typ is _engine._I32P else _engine._f64(arr)
        keep.append(arr)
        setattr(d, name, arr.ctypes.data_as(typ))
    pairs = _engine._i32([(model.body(a).id, model.body(b).id) for a, b in allowed_collision_bodies]).reshape(-1, 2)
    q = None if qidx is None else _engine._i32(qidx)
    base = None if qpos_base is None else _engine._f64(qpos_base)
    info = ProgramInfo()
    nip, ntab = C.c_int32(0), C.c_int32(0)

    def call(ip, fp, dp):
        rc = f(C.byref(d), pairs.ctypes.data_as(_engine._I32P), len(pairs),
               None if q is None else q.ctypes.data_as(_engine._I32P), 0 if q is None else len(q),
               None if base is None else base.ctypes.data_as(_engine._F64P), C.c_double(filter_tol),
               None if ip is None else ip.ctypes.data_as(_engine._I32P), C.byref(nip),
               None if fp is None else fp.ctypes.data_as(C.POINTER(C.c_float)),
               None if dp is None else dp.ctypes.data_as(_engine._F64P), C.byref(ntab), C.byref(info))
        if rc != 0:
            raise _engine.MjplError(rc, lib.mjpl_last_error().decode())

    call(None, None, None)
    ip, fp, dp = np.zeros(nip.value, np.int32), np.zeros(ntab.value, np.float32), np.zeros(ntab.value, np.float64)
    call(ip, fp, dp)
    return ip, fp, dp, info


# ----------------------------------------------------------------------------- expression helpers
def lit(x: float) -> str:
    """Exact C literal of a binary32 value."""
    x = float(np.float32(x))
    if math.isnan(x):
        return "__builtin_nanf(\"\")"
    if math.isinf(x):
        return "__builtin_inff()" if x > 0 else "(-__builtin_inff())"
    return f"{x.hex()}f"


def lin(terms, const: float = 0.0) -> str:
    """sum of coef * expr with exact-zero terms dropped and exact +-1 coefficients folded;
    `terms` = [(coef, expr)].  Returns a C expression (float)."""
    parts = []
    for c, e in terms:
        c = float(np.float32(c))
        if c == 0.0:
            continue
        if c == 1.0:
            parts.append(("+", e))
        elif c == -1.0:
            parts.append(("-", e))
        elif c < 0:
            parts.append(("-", f"{lit(-c)} * {e}"))
        else:
            parts.append(("+", f"{lit(c)} * {e}"))
    const = float(np.float32(const))
    if const != 0.0:
        parts.append(("+" if const > 0 else "-", lit(abs(const))))
    if not parts:
        return "0.0f"
    out = ("-" if parts[0][0] == "-" else "") + parts[0][1]
    for sgn, e in parts[1:]:
        out += f" {sgn} {e}"
    return f"({out})"


def quat_mul_const_right(a, b) -> list[str]:
    """a (expressions) x b (constants) -> 4 expressions (mju_mulQuat)."""
    return [lin([(b[0], a[0]), (-b[1], a[1]), (-b[2], a[2]), (-b[3], a[3])]),
            lin([(b[1], a[0]), (b[0], a[1]), (b[3], a[2]), (-b[2], a[3])]),
            lin([(b[2], a[0]), (-b[3], a[1]), (b[0], a[2]), (b[1], a[3])]),
            lin([(b[3], a[0]), (b[2], a[1]), (-b[1], a[2]), (b[0], a[3])])]


def quat_mul_np(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def quat_mat_np(q):
    w, x, y, z = q
    return np.array([w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z])


def expanded_threshold(X, bound, reach: float) -> float:
    """Threshold of the expanded bounding cull of a static partner at X (binary32 coordinates) whose
    difference-form bound is `bound` (already widened by the filter's tolerance): the test
        fl(|c|^2 - 2 c.X) <= THR
    must pass whenever the real-arithmetic |c - X|^2 <= bound for a centre c with |c| <= reach.
    |c|^2 costs three roundings of at most u |c|^2 (u = 2^-24), each of the three multiply-adds one of at
    most u (|c| + |X|)^2 (every partial sum is bounded by that): the computed value exceeds the real one by
    no more than 6 u (|c| + |X|)^2.  The allowance below takes 8 u, a reach larger by one percent, and the
    threshold is rounded up to binary32."""
    X = np.asarray([float(np.float32(v)) for v in X], dtype=np.float64)
    nx = float(np.linalg.norm(X))
    b = float(np.float32(bound))
    if not math.isfinite(b):
        return b
    if not math.isfinite(reach):
        return math.inf  # (nothing bounds the centre: every lane is a candidate; the narrowphase decides)
    allow = 8.0 * 2.0 ** -24 * (1.01 * reach + nx) ** 2
    thr = b - float(X @ X) + allow
    f = np.float32(thr)
    if float(f) < thr:
        f = np.nextafter(f, np.float32(np.inf))
    return float(f)


class _Gen:
    def __init__(self, ip, fp, dp, info):
        self.ip, self.fp, self.dp, self.info = ip, fp, dp, info
        self.lines: list[str] = []
        self.ind = 3

    def w(self, s=""):
        self.lines.append("  " * self.ind + s)

    # rot_vec_quat(res, const vec, quat expr): v + 2 * cross(q_xyz, q_w v + cross(q_xyz, v))
    def rot_vec_quat(self, dst, vec, qn):
        v = [float(x) for x in vec]
        self.w(f"{{ const float t0_ = {lin([(v[0], qn[0]), (v[2], qn[2]), (-v[1], qn[3])])};")
        self.w(f"  const float t1_ = {lin([(v[1], qn[0]), (v[0], qn[3]), (-v[2], qn[1])])};")
        self.w(f"  const float t2_ = {lin([(v[2], qn[0]), (v[1], qn[1]), (-v[0], qn[2])])};")
        self.w(f"  {dst}0 = {lit(v[0])} + 2.0f * ({qn[2]} * t2_ - {qn[3]} * t1_);")
        self.w(f"  {dst}1 = {lit(v[1])} + 2.0f * ({qn[3]} * t0_ - {qn[1]} * t2_);")
        self.w(f"  {dst}2 = {lit(v[2])} + 2.0f * ({qn[1]} * t1_ - {qn[2]} * t0_); }}")


SCENE_ROWS, SCENE_HEADER, SCENE_SLOT_LANE0 = 32, 32, 40  # (mjpl_filter.h: kSceneRows, kSceneHeader)
SCENE_PLANE_ROWS, SCENE_STAGE = 2, 32 * 4 + 32              # (kScenePlaneRows, kSceneStageFloats)


class _SharedAxesReused(ValueError):
    """Boxes share an axes slot that the program's slot allocation reuses while they still refer to it."""


def generate(ip, fp, dp, info, cull_form: str | None = None, generic: bool = False, report: dict | None = None) -> str:
    """Straight-line filter code of one program (see _generate).  Moving boxes of one body with one orientation share
    the slot of their x and y axes; where the program's own slot allocation gets in the way of that, every box keeps
    its own."""
    try:
        return _generate(ip, fp, dp, info, cull_form, generic, share_axes=True, report=report)
    except _SharedAxesReused:
        return _generate(ip, fp, dp, info, cull_form, generic, share_axes=False, report=report)


def _generate(ip, fp, dp, info, cull_form: str | None = None, generic: bool = False, share_axes: bool = True, report: dict | None = None) -> str:
    """HIP source of `struct Spec` for one compiled program.
    generic: the ROBOT's code only -- forward kinematics, geom poses, the culls against earlier moving geoms --
    as literals; every static partner (floor, obstacles, the robot's own world-welded base) is a row of the
    scene table the engine keeps in front of the float32 tables (`tp[-S ...]`, scalar loads): [a0 a1 a2 thr
    desc], tested in one rolled loop per moving geom.  One such library serves the robot in any scene of up
    to 32 static geoms.
    cull_form: "expanded" (default) tests a static partner as  |c|^2 - 2 c.X <= bound - |X|^2  -- three
    fused multiply-adds and a compare per partner on top of one |c|^2 per geom -- with the threshold
    raised by a bound of the form's own rounding (see `expanded_threshold`); "difference" is the
    interpreter's |c - X|^2 <= bound (six operations and a compare, two partners per packed instruction)."""
    cull_form = cull_form or os.environ.get("MJPL_SPEC_CULL", "expanded")
    if info.immediate or not info.filter_usable:
        raise ValueError("this model runs the immediate interpreter / has no usable filter: nothing to specialise")
    mbox = bool(info.mbox)  # moving boxes: full frames in the box queue, two register slots per stored box
    g = _Gen(ip, fp, dp, info)
    w = g.w
    nbody = int(ip[H_NBODYOPS])
    nwpad = int(ip[H_NWPAD])
    off_wcull, off_wnarrow = int(ip[H_OFF_WCULL]), int(ip[H_OFF_WNARROW])
    fconst = int(ip[H_OFF_FCONST])
    maxs = int(info.maxs)

    def wc_at(wrow, f):
        return off_wcull + ((wrow >> 2) << 4) + (f << 2) + (wrow & 3)

    nstage_total = 0
    _pc = int(ip[H_OFF_BODYOPS])
    for _b in range(nbody):
        _nj, _ng = int(ip[_pc + B_NJNT]), int(ip[_pc + B_NGEOM])
        _pc += B_SIZE + _nj * J_SIZE + _ng * (G_SIZE + MAX_SLOTS)
        nstage_total += _ng
    SC = -(SCENE_HEADER + nstage_total * SCENE_STAGE + 16)  # where the scene table starts, relative to tp
    stages = []       # (case body lines, gtype, gdoff, store)
    desc = []         # per stage: list of packed partner descriptors
    pending_fk: list[str] = []   # FK of bodies since the last stage (bodies without geoms)
    pc = int(ip[H_OFF_BODYOPS])
    state_known = None  # (p, q) as numpy constants while the chain so far is constant (static parent, fixed joints)

    # How far from the world origin can a moving geom's centre be while its lane is still alive?  The body
    # origins by the triangle inequality along the chain (hinges keep lengths; an off-centre hinge adds
    # twice its offset), or -- below a slide joint, whose travel the program does not know -- by the
    # per-lane range check: a lane whose body origin leaves [-maxcoord, maxcoord]^3 is dead from there on.
    maxcoord = float(fp[fconst + FC_MAXCOORD])
    box_reach = math.sqrt(3.0) * maxcoord
    body_reach_cur = 0.0      # of the body the walk is at (PARENT_CUR refers to it)
    saved_reach: dict[int, float] = {}
    # Moving boxes: boxes of one body with one orientation (the Panda's five pads per finger) have ONE frame --
    # the x and y axes are kept once, in the second slot of the first of them; `axes_of` maps each box's own second
    # slot to the slot that holds the values, `axes_key` says whose axes a slot holds right now.
    axes_of: dict[int, int] = {}
    axes_key: dict[int, tuple] = {}
    # The edge certificate (mjpl_fused.h): how far can any point of a moving geom travel while the
    # planning joints go from one end of an edge to the other?  At most  sum_j |dq_j| rho_j  over the planning hinges j
    # above it, rho_j = the longest the chain can stretch from joint j's anchor to the geom's centre + the geom's
    # bounding radius (hinges keep lengths: a triangle inequality along the chain, as `geom_reach` above).  `anc` of a
    # body: [(planning column, that length up to the body's origin)] root first.  A planning SLIDE joint, or a moving
    # pair whose earlier geom does not hang on a prefix of the later one's joints, and the library is built without.
    cert_ok = not generic
    anc_cur: list = []
    saved_anc: dict[int, list] = {}
    slot_anc: dict[int, list] = {}   # slot -> planning columns above the geom stored there
    cert_stage: list = []            # per stage: the lines of its certificate block
    cert_levers: list = []           # per stage: (geom id, [(planning column, rho)], bounding radius) -- report["cert_levers"]

    for b in range(nbody):
        parent, bdoff, njnt, save_slot, ngeom = (int(ip[pc + k]) for k in (B_PARENT, B_DOFF, B_NJNT, B_SAVE, B_NGEOM))
        pc += B_SIZE
        bd = dp[bdoff:]
        g.lines, body_lines = [], None
        if parent == PARENT_STATIC:
            parent_reach = float(np.linalg.norm(bd[7:10]))
        elif parent == PARENT_CUR:
            parent_reach = body_reach_cur
        else:
            parent_reach = saved_reach[parent - 1]
        geom_reach = parent_reach + float(np.linalg.norm(bd[0:3]))
        for j in range(njnt):
            jtype_, jdoff_ = int(ip[pc + j * J_SIZE + J_TYPE]), int(ip[pc + j * J_SIZE + J_DOFF])
            if jtype_ == JT_SLIDE:
                geom_reach = math.inf
            else:
                geom_reach += 2.0 * float(np.linalg.norm(dp[jdoff_ + 3: jdoff_ + 6]))
        geom_reach = min(geom_reach, box_reach)
        body_reach_cur = geom_reach
        # ... and the planning joints above this body with their lever arms
        anc_parent = [] if parent == PARENT_STATIC else (anc_cur if parent == PARENT_CUR else saved_anc[parent - 1])
        blen = float(np.linalg.norm(bd[0:3]))
        anc_body = [(qs, d + blen) for qs, d in anc_parent]
        for j in range(njnt):
            jt_, qs_, jd_ = (int(ip[pc + j * J_SIZE + k]) for k in (J_TYPE, J_QSRC, J_DOFF))
            jp_ = float(np.linalg.norm(dp[jd_ + 3: jd_ + 6]))
            if jt_ == JT_SLIDE:
                if qs_ >= 0:
                    cert_ok = False
                else:  # a slide joint held at its constant: a fixed offset along its axis
                    off_ = abs(float(dp[jd_ + 7]) - float(dp[jd_ + 6]))
                    anc_body = [(qs, d + off_) for qs, d in anc_body]
            else:
                anc_body = [(qs, d + 2.0 * jp_) for qs, d in anc_body]
                if qs_ >= 0:
                    anc_body.append((qs_, jp_))
        anc_cur = anc_body
        if save_slot >= 0:
            saved_anc[save_slot] = anc_body
        if save_slot >= 0:
            saved_reach[save_slot] = geom_reach
        # ---- parent pose
        if parent == PARENT_STATIC:
            pp, pq, pR = bd[7:10].copy(), bd[10:14].copy(), bd[14:23].copy()
            npc = pp + pR.reshape(3, 3) @ bd[0:3]
            nqc = quat_mul_np(pq, bd[3:7])
            w(f"float np0 = {lit(npc[0])}, np1 = {lit(npc[1])}, np2 = {lit(npc[2])};")
            w(f"float nq0 = {lit(nqc[0])}, nq1 = {lit(nqc[1])}, nq2 = {lit(nqc[2])}, nq3 = {lit(nqc[3])};")
        else:
            if parent != PARENT_CUR:  # restore a saved pose
                k = parent - 1
                w(f"{{ const float *sv = save + (size_t){k} * 7 * sstride;")
                w("  p0 = sv[0]; p1 = sv[sstride]; p2 = sv[2 * sstride];")
                w("  q0 = sv[3 * sstride]; q1 = sv[4 * sstride]; q2 = sv[5 * sstride]; q3 = sv[6 * sstride];")
                w("  MJPL_SPEC_QUAT2MAT(); }")
            bp, bq = bd[0:3], bd[3:7]
            for r in range(3):
                w(f"float np{r} = p{r} + {lin([(bp[0], f'R{3 * r}'), (bp[1], f'R{3 * r + 1}'), (bp[2], f'R{3 * r + 2}')])};")
            e = quat_mul_const_right(["q0", "q1", "q2", "q3"], bq)
            w(f"float nq0 = {e[0]}, nq1 = {e[1]}, nq2 = {e[2]}, nq3 = {e[3]};")
        # ---- joints
        for j in range(njnt):
            jtype, qsrc, jflags, jdoff = (int(ip[pc + k]) for k in (J_TYPE, J_QSRC, J_FLAGS, J_DOFF))
            pc += J_SIZE
            jd = dp[jdoff:]
            axis, jpos, qpos0, qconst = jd[0:3], jd[3:6], float(jd[6]), float(jd[7])
            if qsrc >= 0:
                w(f"{{ const float dq = (float)q[{qsrc} * qstride] - {lit(qpos0)};")
            else:
                w(f"{{ const float dq = {lit(np.float32(qconst) - np.float32(qpos0))};")
            nqn = ["nq0", "nq1", "nq2", "nq3"]
            if jtype == JT_SLIDE:
                w("  float xa0, xa1, xa2;")
                g.rot_vec_quat("xa", axis, nqn)
                w("  np0 += xa0 * dq; np1 += xa1 * dq; np2 += xa2 * dq; }")
            else:
                if jflags & JF_POS_NONZERO:
                    w("  float an0, an1, an2;")
                    g.rot_vec_quat("an", jpos, nqn)
                    w("  an0 += np0; an1 += np1; an2 += np2;")
                w("  far = far || !(fabsf(dq) <= maxangle);  // binary32(q) is off by eps |q|")
                w("  float sn, cs; sincosf(dq * 0.5f, &sn, &cs);")
                # nq = nq (x) (cs, ax sn, ay sn, az sn)
                a = [float(np.float32(x)) for x in axis]
                sx = [f"{lit(a[k])} * sn" if a[k] not in (0.0, 1.0, -1.0) else ("sn" if a[k] == 1.0 else ("(-sn)" if a[k] == -1.0 else None))
                      for k in range(3)]

                def term(sign, left, right):
                    return None if right is None else (sign, f"{left} * {right}")

                comps = [
                    [("+", "nq0 * cs"), term("-", "nq1", sx[0]), term("-", "nq2", sx[1]), term("-", "nq3", sx[2])],
                    [term("+", "nq0", sx[0]), ("+", "nq1 * cs"), term("+", "nq2", sx[2]), term("-", "nq3", sx[1])],
                    [term("+", "nq0", sx[1]), term("-", "nq1", sx[2]), ("+", "nq2 * cs"), term("+", "nq3", sx[0])],
                    [term("+", "nq0", sx[2]), term("+", "nq1", sx[1]), term("-", "nq2", sx[0]), ("+", "nq3 * cs")],
                ]
                exprs = []
                for comp in comps:
                    ts = [t for t in comp if t is not None]
                    out = ("-" if ts[0][0] == "-" else "") + ts[0][1]
                    for sgn, ex in ts[1:]:
                        out += f" {sgn} {ex}"
                    exprs.append(out)
                w(f"  const float t0 = {exprs[0]}, t1 = {exprs[1]}, t2 = {exprs[2]}, t3 = {exprs[3]};")
                w("  nq0 = t0; nq1 = t1; nq2 = t2; nq3 = t3;")
                if jflags & JF_POS_NONZERO:
                    w("  float vv0, vv1, vv2;")
                    g.rot_vec_quat("vv", jpos, nqn)
                    w("  np0 = an0 - vv0; np1 = an1 - vv1; np2 = an2 - vv2;")
                w("}")
        # ---- normalise, rotation matrix, range check, save
        w("{ const float inv = __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(nq0 * nq0 + nq1 * nq1 + nq2 * nq2 + nq3 * nq3));")
        w("  q0 = nq0 * inv; q1 = nq1 * inv; q2 = nq2 * inv; q3 = nq3 * inv; }")
        w("p0 = np0; p1 = np1; p2 = np2;")
        w("MJPL_SPEC_QUAT2MAT();")
        w("far = far || !(fmaxf(fabsf(p0), fmaxf(fabsf(p1), fabsf(p2))) <= maxcoord);")
        w("if (far) dead = kInf;")
        if save_slot >= 0:
            w(f"{{ float *sv = save + (size_t){save_slot} * 7 * sstride;")
            w("  sv[0] = p0; sv[sstride] = p1; sv[2 * sstride] = p2;")
            w("  sv[3 * sstride] = q0; sv[4 * sstride] = q1; sv[5 * sstride] = q2; sv[6 * sstride] = q3; }")
        body_lines = g.lines
        pending_fk.extend(["{"] + body_lines + ["}"])
        # ---- geoms
        for gi in range(ngeom):
            gtype, gflags, gdoff, store, geom_id, smask = (int(ip[pc + k]) for k in (G_TYPE, G_FLAGS, G_DOFF, G_STORE, G_GEOMID, G_SMASK))
            wmask = (int(ip[pc + G_WMASK_LO]) & 0xFFFFFFFF) | ((int(ip[pc + G_WMASK_HI]) & 0xFFFFFFFF) << 32)
            pmask = (int(ip[pc + G_PMASK_LO]) & 0xFFFFFFFF) | ((int(ip[pc + G_PMASK_HI]) & 0xFFFFFFFF) << 32)
            swords = [int(x) for x in ip[pc + G_SIZE: pc + G_SIZE + MAX_SLOTS]]
            pc += G_SIZE + MAX_SLOTS
            gd = dp[gdoff:]
            fgd = fp[gdoff:]
            g.lines = []
            lpos, lquat = gd[0:3], gd[3:7]
            if gflags & GF_SAMEPOS:
                w("cx = p0; cy = p1; cz = p2;")
            else:
                for r, nm in enumerate(("cx", "cy", "cz")):
                    w(f"{nm} = p{r} + {lin([(lpos[0], f'R{3 * r}'), (lpos[1], f'R{3 * r + 1}'), (lpos[2], f'R{3 * r + 2}')])};")
            curbox = mbox and gtype == GT_BOX
            if gflags & GF_SAMEROT:
                w("zx = R2; zy = R5; zz = R8;")
                if curbox:
                    w("xx = R0; xy = R3; xz = R6; yx = R1; yy = R4; yz = R7;")
            else:
                e = quat_mul_const_right(["q0", "q1", "q2", "q3"], lquat)
                w(f"{{ const float g0 = {e[0]}, g1 = {e[1]}, g2 = {e[2]}, g3 = {e[3]};")
                if curbox:  # the whole frame, by the interpreter's own routine (same binary32 values)
                    w("  const float gq_[4] = {g0, g1, g2, g3}; float mm_[9]; mjpl::quat2mat(mm_, gq_);")
                    w("  zx = mm_[2]; zy = mm_[5]; zz = mm_[8]; xx = mm_[0]; xy = mm_[3]; xz = mm_[6]; yx = mm_[1]; yy = mm_[4]; yz = mm_[7]; }")
                else:
                    w("  zx = 2.0f * (g1 * g3 + g0 * g2); zy = 2.0f * (g2 * g3 - g0 * g1); zz = g0 * g0 - g1 * g1 - g2 * g2 + g3 * g3; }")
            w("/*CERT*/")  # (the stage's certificate block goes here once its partners are known)
            partners = []
            slot_levels: set = set()
            slot_k2 = 0.0
            if generic:
                pmask = wmask = 0  # (static partners: rows of the scene table, tested after the switch)
            wbound = fgd[GD_WBOUND: GD_WBOUND + nwpad]
            sbound = fgd[GD_WBOUND + 2 * nwpad: GD_WBOUND + 2 * nwpad + MAX_SLOTS]
            # static planes
            for wrow in range(64):
                if not (pmask >> wrow) & 1:
                    continue
                ppos = [float(fp[wc_at(wrow, f)]) for f in range(3)]
                pz = [float(x) for x in fp[off_wnarrow + wrow * WN_LEN + WN_ZAXIS: off_wnarrow + wrow * WN_LEN + WN_ZAXIS + 3]]
                k = len(partners)
                dot = lin([(pz[0], "cx"), (pz[1], "cy"), (pz[2], "cz")], -(np.float32(pz[0]) * np.float32(ppos[0]) + np.float32(pz[1]) * np.float32(ppos[1]) + np.float32(pz[2]) * np.float32(ppos[2])))
                w(f"MJPL_SPEC_HIT({k}, !({dot} + deadp > {lit(wbound[wrow])}));")
                partners.append((EK_PLANE, wrow, GT_PLANE, 1, 1 if curbox else 0, 63, 0))
            # other static geoms
            w("const float ux = cx + dead;")
            if cull_form == "expanded" and not generic:
                w("const float cc = __builtin_fmaf(cz, cz, __builtin_fmaf(cy, cy, ux * ux)) - wc0;  // (wc0: the certificate's widening, 0 without)")
            statics = []
            for wrow in range(64):
                if not (wmask >> wrow) & 1:
                    continue
                info_word = int(np.frombuffer(np.float32(fp[wc_at(wrow, 3)]).tobytes(), dtype=np.int32)[0])
                ptype, pgid = info_word & 255, info_word >> 8
                pfirst = 1 if (ptype < gtype or (ptype == gtype and pgid < geom_id)) else 0
                X, Y, Z = (float(fp[wc_at(wrow, f)]) for f in range(3))
                statics.append((len(partners), X, Y, Z, wbound[wrow]))
                partners.append((EK_STATIC, wrow, ptype, pfirst, 1 if (ptype == GT_BOX or curbox) else 0, 63, 0))
            if cull_form == "expanded" and statics:
                reach = geom_reach + float(np.linalg.norm(np.asarray(lpos, dtype=np.float64)))
                for a, b in zip(statics[0::2], statics[1::2]):
                    ta, tb = expanded_threshold(a[1:4], a[4], reach), expanded_threshold(b[1:4], b[4], reach)
                    w(f"MJPL_SPEC_CULLX2({a[0]}, {b[0]}, {lit(-2 * a[1])}, {lit(-2 * a[2])}, {lit(-2 * a[3])}, {lit(ta)}, "
                      f"{lit(-2 * b[1])}, {lit(-2 * b[2])}, {lit(-2 * b[3])}, {lit(tb)});")
                if len(statics) % 2:
                    k, X, Y, Z, bound = statics[-1]
                    w(f"MJPL_SPEC_CULLX({k}, {lit(-2 * X)}, {lit(-2 * Y)}, {lit(-2 * Z)}, {lit(expanded_threshold((X, Y, Z), bound, reach))});")
            else:
                for a, b in zip(statics[0::2], statics[1::2]):
                    w(f"MJPL_SPEC_CULL2({a[0]}, {b[0]}, {lit(a[1])}, {lit(b[1])}, {lit(a[2])}, {lit(b[2])}, {lit(a[3])}, {lit(b[3])}, "
                      f"{lit(a[4])}, {lit(b[4])});")
                if len(statics) % 2:
                    k, X, Y, Z, bound = statics[-1]
                    w(f"MJPL_SPEC_CULL({k}, {lit(X)}, {lit(Y)}, {lit(Z)}, {lit(bound)});")
            # earlier moving geoms in the slot file
            for n in range(maxs):
                if not (smask >> n) & 1:
                    continue
                pw = swords[n]
                if generic:  # (lanes 0 .. 31 belong to the scene rows)
                    while len(partners) < SCENE_SLOT_LANE0:
                        partners.append((0, 0, 0, 0, 0, 0, 0))
                k = len(partners)
                # the planning joints above the stored geom must be a prefix of those above this one (else: no certificate)
                mine_, theirs_ = [qs for qs, _ in anc_cur], slot_anc.get(n, None)
                if theirs_ is None or mine_[:len(theirs_)] != theirs_ or len(theirs_) > 7:
                    cert_ok = False
                    lvl = 0
                else:
                    lvl = len(theirs_)
                slot_levels.add(lvl)
                slot_k2 = max(slot_k2, 2.0 * math.sqrt(max(float(sbound[n]), 0.0)))
                w(f"MJPL_SPEC_SLOTCULL({k}, {n}, {lit(sbound[n])} + ws{lvl});")
                sptype = (pw >> 12) & 15
                # (a stored box keeps its x and y axes in a second slot, pw & 63; with a box on either side the pair
                # takes the box queue, whose records carry full frames)
                n2 = (pw & 63) if mbox else 63
                if n2 != 63:
                    n2 = axes_of[n2]
                partners.append((EK_SLOT, n, sptype, 1 if (pw & P_FIRST) else 0, 1 if (mbox and (curbox or sptype == GT_BOX)) else 0, n2, lvl))
            if len(partners) > 64:
                raise ValueError("a geom with more than 64 enabled partners cannot be specialised")
            store2 = ((store >> 6) & 63) if (store >= 0 and mbox) else 63
            if store >= 0 and (store & 63) in axes_key:  # (a slot that held shared axes is written again: nobody may still read them)
                if any(v == (store & 63) and k != v for k, v in axes_of.items()):
                    raise _SharedAxesReused("a shared axes slot is reused while boxes still refer to it")
                del axes_key[store & 63]
            if store2 != 63:
                key = (b, bool(gflags & GF_SAMEROT), tuple(np.float32(lquat).tolist()))
                held = [sl for sl, k_ in axes_key.items() if k_ == key] if share_axes else []
                if held:           # the frame is in the slot file already
                    axes_of[store2] = held[0]
                    store2 = 63
                else:
                    if any(v == store2 and k != v for k, v in axes_of.items()):
                        raise _SharedAxesReused("a shared axes slot is reused while boxes still refer to it")
                    axes_of[store2] = store2
                    axes_key[store2] = key
            # the certificate block of the stage: suffix sums of |dq_j| rho_j from the deepest planning joint up; level c =
            # the motion relative to a geom that hangs on the first c of them (0: to the world)
            sizes_ = [float(x) for x in gd[GD_SIZE: GD_SIZE + 3]]
            rbound_ = {GT_SPHERE: sizes_[0], GT_CAPSULE: sizes_[0] + sizes_[1], GT_BOX: math.sqrt(sum(x * x for x in sizes_))}.get(gtype, math.inf)
            lp_ = float(np.linalg.norm(np.asarray(lpos, dtype=np.float64)))
            k2w = max([2.0 * math.sqrt(max(float(st_[4]), 0.0)) for st_ in statics] + [0.0])
            if not math.isfinite(rbound_) or len(anc_cur) > 7:
                cert_ok = False
            cb = ["float s_ = 0.0f;"]
            need = sorted(slot_levels | {0})
            for lv in range(len(anc_cur), -1, -1):
                if lv < len(anc_cur):
                    qs_, d_ = anc_cur[lv]
                    rho_ = (d_ + lp_ + (rbound_ if math.isfinite(rbound_) else 0.0)) * (1.0 + 1e-6)
                    cb.append(f"s_ = __builtin_fmaf((float)adq[{qs_} * 64], {lit(rho_)}, s_);")
                if lv in need:
                    cb.append(f"dm{lv} = __builtin_fmaf(s_, 1.001f, 2.0f * tol);")
            cb.append(f"wc0 = __builtin_fmaf({lit(k2w * (1.0 + 1e-6))}, dm0, dm0 * dm0); deadp = dead - dm0;")
            for lv in sorted(slot_levels):
                cb.append(f"ws{lv} = __builtin_fmaf({lit(slot_k2 * (1.0 + 1e-6))}, dm{lv}, dm{lv} * dm{lv});")
            cert_stage.append(cb)
            cert_levers.append((geom_id, [(qs_, d_ + lp_ + (rbound_ if math.isfinite(rbound_) else 0.0)) for qs_, d_ in anc_cur], rbound_))
            if store >= 0:
                slot_anc[store & 63] = [qs for qs, _ in anc_cur]
            stages.append((pending_fk + g.lines, gtype, gdoff, store & 63 if store >= 0 else -1, store2))
            pending_fk = []
            desc.append([(kind << 0) | (index << 2) | (ptype << 10) | (pfirst << 14) | (boxq << 15) | (index2 << 16) | (lvl << 22)
                         for kind, index, ptype, pfirst, boxq, index2, lvl in partners])
    if pending_fk:  # trailing bodies without geoms influence nothing: drop them
        pending_fk = []
    # The certificate is an opt-in build (MJPL_SPEC_CERT=1): measured on the headline batch it halves the waypoint checks and
    # gains 3 % -- a workgroup's two rounds of endpoint tiles bound the launch -- while the lines it adds to the code every
    # tile runs cost 8 % without it (profiles/README.md, round 5).  Without it the generated code carries none of it.
    cert_ok = cert_ok and cull_form == "expanded" and os.environ.get("MJPL_SPEC_CERT", "0") == "1"
    if report is not None:
        report["cert_ok"], report["cert_levers"] = bool(cert_ok), cert_levers
    if not cert_ok:
        import re
        stages = [([re.sub(r" \+ ws\d\)", ")", ln).replace(") - wc0;", ");").replace("deadp", "dead") for ln in lines], gt_, gd_, st_, st2_)
                  for lines, gt_, gd_, st_, st2_ in stages]
    nstage = len(stages)
    out = []
    o = out.append
    o("// GENERATED by mjpl_amd/specialise.py -- straight-line per-configuration check of ONE compiled program.")
    o(f"// program hash {info.hash:016x}, {nbody} moving bodies, {nstage} moving geoms, "
      f"{sum(1 for d in desc for x in d if x)} literal pairs{' (static partners: scene table)' if generic else ''}, slot file width {maxs}")
    o("#define MJPL_SPEC_QUAT2MAT() \\")
    o("  do { R0 = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3; R4 = q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3; \\")
    o("       R8 = q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3; R1 = 2.0f * (q1 * q2 - q0 * q3); R2 = 2.0f * (q1 * q3 + q0 * q2); \\")
    o("       R3 = 2.0f * (q1 * q2 + q0 * q3); R5 = 2.0f * (q2 * q3 - q0 * q1); R6 = 2.0f * (q1 * q3 - q0 * q2); \\")
    o("       R7 = 2.0f * (q2 * q3 + q0 * q1); } while (0)")
    o("typedef float spec_v2f __attribute__((ext_vector_type(2)));")
    o("// (a scalar branch around the two v_writelane -- most culls pass for no lane of the wave -- measured")
    o("//  3 % SLOWER than parking unconditionally: 213 more branches per configuration)")
    o("#define MJPL_SPEC_HIT(k, pass) \\")
    o("  do { const unsigned long long m_ = __builtin_amdgcn_ballot_w64(pass); \\")
    o("       mjpl::park_mask<k>(mlo, mhi, m_); } while (0)")
    o("// ux = cx, or +inf on a lane that is not to report anything (inactive / decided / out of range)")
    o("#define MJPL_SPEC_CULL(k, X, Y, Z, BOUND) \\")
    o("  do { const float dx_ = ux - (X), dy_ = cy - (Y), dz_ = cz - (Z); \\")
    o("       MJPL_SPEC_HIT(k, !(mjpl::sqnorm3(dx_, dy_, dz_) > (BOUND))); } while (0)")
    o("// two static partners at once: packed float32 arithmetic")
    o("#define MJPL_SPEC_CULL2(ka, kb, XA, XB, YA, YB, ZA, ZB, BOUNDA, BOUNDB) \\")
    o("  do { const spec_v2f dx_ = (spec_v2f){ux, ux} - (spec_v2f){XA, XB}, dy_ = (spec_v2f){cy, cy} - (spec_v2f){YA, YB}, \\")
    o("                      dz_ = (spec_v2f){cz, cz} - (spec_v2f){ZA, ZB}; \\")
    o("       const spec_v2f s_ = __builtin_elementwise_fma(dx_, dx_, __builtin_elementwise_fma(dy_, dy_, dz_ * dz_)); \\")
    o("       const unsigned long long ma_ = __builtin_amdgcn_ballot_w64(!(s_.x > (BOUNDA))); \\")
    o("       const unsigned long long mb_ = __builtin_amdgcn_ballot_w64(!(s_.y > (BOUNDB))); \\")
    o("       mjpl::park_mask2<ka, kb>(mlo, mhi, ma_, mb_); } while (0)")
    o("// expanded form: t = |c|^2 - 2 c.X against THR = bound - |X|^2 (+ the form's rounding allowance); cc = |c|^2 with")
    o("// ux in it, so a lane that is not to report anything carries +inf or NaN here and fails the ordered compare")
    o("#define MJPL_SPEC_CULLX(k, M2X, M2Y, M2Z, THR) \\")
    o("  do { const float t_ = __builtin_fmaf(cz, (M2Z), __builtin_fmaf(cy, (M2Y), __builtin_fmaf(ux, (M2X), cc))); \\")
    o("       MJPL_SPEC_HIT(k, t_ <= (THR)); } while (0)")
    o("#define MJPL_SPEC_CULLX2(ka, kb, AX, AY, AZ, ATHR, BX, BY, BZ, BTHR) \\")
    o("  do { const float ta_ = __builtin_fmaf(cz, (AZ), __builtin_fmaf(cy, (AY), __builtin_fmaf(ux, (AX), cc))); \\")
    o("       const float tb_ = __builtin_fmaf(cz, (BZ), __builtin_fmaf(cy, (BY), __builtin_fmaf(ux, (BX), cc))); \\")
    o("       const unsigned long long ma_ = __builtin_amdgcn_ballot_w64(ta_ <= (ATHR)); \\")
    o("       const unsigned long long mb_ = __builtin_amdgcn_ballot_w64(tb_ <= (BTHR)); \\")
    o("       mjpl::park_mask2<ka, kb>(mlo, mhi, ma_, mb_); } while (0)")
    o("#define MJPL_SPEC_SLOTCULL(k, n, BOUND) \\")
    o("  do { const float dx_ = ux - sf.f[0][n], dy_ = cy - sf.f[1][n], dz_ = cz - sf.f[2][n]; \\")
    o("       MJPL_SPEC_HIT(k, !(mjpl::sqnorm3(dx_, dy_, dz_) > (BOUND))); } while (0)")
    o("")
    o(f"__constant__ int kSpecDesc[{nstage} * 64] = {{")
    for d in desc:
        o("  " + ", ".join(str(x) for x in (d + [0] * (64 - len(d)))) + ",")
    o("};")
    o("")
    if mbox:
        # Whole frames, up to 24 slots: as SlotFile's vectors (<24 x float>, widened to 32 lanes of registers each, six of
        # them) the file alone asks for 192 registers and every access for a contiguous tuple -- 2 KB of scratch per lane
        # at any occupancy.  Every slot number in this code is a literal, so plain scalars serve: the compiler keeps one
        # register per value that is in use (scalar replacement), live from the stage that stores it.
        o(f"struct SpecSlots {{ float f[6][{maxs}]; }};")
        o("static __device__ __forceinline__ void spec_put6(SpecSlots &sf, int slot, const float *t6) {")
        o("#pragma unroll")
        o("  for (int k = 0; k < 6; k++) sf.f[k][slot] = t6[k];")
        o("}")
        o("static __device__ __forceinline__ void spec_get6(const SpecSlots &sf, int slot, float *t6) {")
        o("#pragma unroll")
        o("  for (int k = 0; k < 6; k++) t6[k] = sf.f[k][slot];")
        o("}")
        o("")
    o("struct Spec {")
    o(f"  static constexpr int kNplan = {int(ip[H_NPLAN])};")
    o("  // the edge certificate: with ps.adq = this lane's |dq| per planning column, run() widens every bounding cull by what")
    o("  // the pair can move along the edge, every candidate carries that margin to its narrowphase routine, and a lane")
    o("  // none of whose candidates came closer returns V_CLEAR instead of V_NONE (mjpl_fused.h)")
    o(f"  static constexpr bool kCert = {'true' if cert_ok else 'false'};")
    o("  template <class QT>")
    o("  static __device__ __forceinline__ int run(mjpl::FP tp, const float *ltab, const QT *q, int qstride, float *save,")
    o(f"                                            int sstride, bool active, float tol, const mjpl::WaveQueue<float, {'true' if mbox else 'false'}> &wq,")
    o("                                            int item, const mjpl::PatchSink &ps) {")
    o("    using namespace mjpl;")
    o("    SpecSlots sf;" if mbox else f"    SlotFile<float, {maxs}> sf;")
    o("    float p0 = 0, p1 = 0, p2 = 0, q0 = 1, q1 = 0, q2 = 0, q3 = 0;")
    o("    float R0 = 1, R1 = 0, R2 = 0, R3 = 0, R4 = 1, R5 = 0, R6 = 0, R7 = 0, R8 = 1;")
    o("    const int lane = threadIdx.x & 63;")
    o("    const float kInf = __builtin_inff();")
    o("    float dead = active ? 0.0f : kInf;")
    o("    bool far = false;")
    o("    int fl = 0, qn = 0, qb = 0;")
    o("    wq.flags[lane] = (int)((unsigned)item << 3);")
    o("    const _Float16 *adq = kCert ? ps.adq : nullptr;")
    o("    const bool cert = adq != nullptr;  // (wave-uniform)")
    if generic:
        o(f"    // the scene table in front of the float32 tables: header, then per moving geom {SCENE_ROWS} cull rows and {SCENE_ROWS} descriptors")
        o(f"    const FP sc = tp + ({SC});")
        o("    const int sc_nplane = info_bits(sc[0]);  // the last 0 .. 2 rows of a geom are planes")
        o("    const int sc_first = info_bits(sc[1]);   // first pair of rows in use (the rows of a geom fill its table from the end)")
        o("    const float maxcoord = sc[4], maxangle = sc[5];")
        o("    const int nwpad = info_bits(sc[2]);")
        o("    const float *lwcull = ltab, *lwnarrow = ltab + info_bits(sc[3]);")
    else:
        o(f"    const float maxcoord = {lit(fp[fconst + FC_MAXCOORD])}, maxangle = {lit(fp[fconst + FC_MAXANGLE])};")
        o(f"    const int nwpad = {nwpad};")
        o(f"    const float *lwcull = ltab + {off_wcull}, *lwnarrow = ltab + {off_wnarrow};")
    o("#pragma nounroll")
    o(f"    for (int g = 0; g < {nstage}; g++) {{")
    o("      if (__builtin_amdgcn_ballot_w64(dead == 0.0f) == 0ull && qn == 0 && qb == 0) break;  // every lane decided")
    o("      float cx = 0, cy = 0, cz = 0, zx = 0, zy = 0, zz = 0;")
    if mbox:
        o("      float xx = 0, xy = 0, xz = 0, yx = 0, yy = 0, yz = 0;  // a moving box: its x and y axes")
    o("      int mlo = 0, mhi = 0;  // lane k holds the hit mask of this geom's partner k")
    if cert_ok:
        o("      // the certificate's margins of this geom by level (metres; all zero without) and what they widen the culls by")
        o("      float dm0 = 0, dm1 = 0, dm2 = 0, dm3 = 0, dm4 = 0, dm5 = 0, dm6 = 0, dm7 = 0;")
        o("      float wc0 = 0, ws0 = 0, ws1 = 0, ws2 = 0, ws3 = 0, ws4 = 0, ws5 = 0, ws6 = 0, ws7 = 0, deadp = dead;")
    o("      // ... and the descriptor of partner k: one vector load per stage, issued ahead of the stage's")
    o("      // arithmetic (a scalar load per hit stalls the wave for its whole latency)")
    if generic:
        o(f"      const int dv = lane < {SCENE_ROWS} ? info_bits(sc[{SCENE_HEADER} + g * {SCENE_STAGE} + {SCENE_ROWS * 4} + lane]) : kSpecDesc[64 * g + lane];")
    else:
        o("      const int dv = kSpecDesc[64 * g + lane];")
    o("      int gtype = 0, gdoff = 0;")
    o("      switch (g) {")
    for si, (lines, gtype, gdoff, store, store2) in enumerate(stages):
        o(f"        case {si}: {{")
        for ln in lines:
            if ln.strip() == "/*CERT*/":
                if cert_ok:
                    o("          deadp = dead;")
                    o("          if (cert) {")
                    for cl in cert_stage[si]:
                        o("            " + cl)
                    o("          }")
                continue
            o("      " + ln)
        if generic:
            o(f"          gtype = {gtype}; gdoff = info_bits(sc[8 + {si}]);")
        else:
            o(f"          gtype = {gtype}; gdoff = {gdoff};")
        o("        } break;")
    o("        default: break;")
    o("      }")
    if generic:
        o("      {  // static partners: one row [a0 a1 a2 thr] of the scene table each, two rows per scalar load, the next")
        o("         // pair fetched while this one is tested (two buffers taking turns: no copies, 16 scalar registers).")
        o("         // The planes (the last rows): a . c <= thr; the others: |c|^2 + a . c <= thr; a row that is no partner: thr = -inf.")
        o("         // One straight line over all rows, entered at the first pair in use: row offsets and the lanes")
        o("         // the masks are parked in are literals, and what a stage pays beside the tests is the one dispatch (a")
        o("         // rolled loop spent thirty scalar instructions per four rows: addresses, the lane select in M0, the counter).")
        o("        const float ux = cx + dead;")
        o("        const float cc = __builtin_fmaf(cz, cz, __builtin_fmaf(cy, cy, ux * ux));")
        o("        const float zd = 0.0f * ux;  // 0, or NaN on a lane that is not to report anything")
        o("        const float acc_a = sc_nplane >= 2 ? zd : cc, acc_b = sc_nplane >= 1 ? zd : cc;  // (the last two rows)")
        o(f"        FP rows = sc + ({SCENE_HEADER} + g * {SCENE_STAGE});")
        # rows per scalar load: two (two buffers of 8 registers); MJPL_GEN_SCENE_WIDE=1: four (2 x 16 registers) -- measured
        # slower, item pass 0.142 vs 0.137 ms: the kernels sit at the scalar-register limit and spill into vector lanes
        wide = int(os.environ.get("MJPL_GEN_SCENE_WIDE", "0"))
        o(f"        float ra[{16 if wide else 8}], rb[{16 if wide else 8}];")
        o("#define MJPL_SCENE_LOAD(dst, at, n) _Pragma(\"unroll\") for (int k_ = 0; k_ < (n); k_++) dst[k_] = rows[4 * (at) + k_]")
        o("        // (a fetch is waited for behind the arithmetic of the rows before it: scalar loads return out of order, so")
        o("        //  a wait at its first use -- after the NEXT fetch has gone out -- would drain that one, too; `rows` passes")
        o("        //  through the wait so that the next fetch cannot be moved above it, the scheduling barriers keep fetch,")
        o("        //  arithmetic and wait in this order)")
        o("#define MJPL_SCENE_WAIT(buf, n) asm volatile(\"\" : \"+s\"(rows) : \"s\"(buf[0]), \"s\"(buf[(n) - 1]))")
        o("#define MJPL_SCENE_PAIR(buf, acca, accb, ka, kb) do { \\")
        o("          const float t0_ = __builtin_fmaf(cz, (buf)[2], __builtin_fmaf(cy, (buf)[1], __builtin_fmaf(ux, (buf)[0], acca))); \\")
        o("          const float t1_ = __builtin_fmaf(cz, (buf)[6], __builtin_fmaf(cy, (buf)[5], __builtin_fmaf(ux, (buf)[4], accb))); \\")
        o("          const unsigned long long m0_ = __builtin_amdgcn_ballot_w64(t0_ <= (buf)[3]); \\")
        o("          const unsigned long long m1_ = __builtin_amdgcn_ballot_w64(t1_ <= (buf)[7]); \\")
        o("          mjpl::park_mask2<ka, kb>(mlo, mhi, m0_, m1_); } while (0)")
        o("        // (one copy of the line per entry point: falling through case labels, the buffers would arrive at every")
        o("        //  label from two places and the compiler would copy the registers there)")
        o("        switch (sc_first) {")
        for first in range(0, SCENE_ROWS, 2):
            o(f"          {'default' if first == SCENE_ROWS - 2 else f'case {first}'}: {{")
            # the rows from `first` on, in fetches of four (a leading pair where `first` is not a multiple of four) or of two
            chunks = []
            r = first
            while r < SCENE_ROWS:
                nr = 4 if (wide and r % 4 == 0) else 2
                chunks.append((r, nr))
                r += nr
            o(f"        MJPL_SCENE_LOAD(ra, {chunks[0][0]}, {4 * chunks[0][1]});")
            for k, (r, nr) in enumerate(chunks):
                cur, nxt = ("ra", "rb") if k % 2 == 0 else ("rb", "ra")
                o(f"        MJPL_SCENE_WAIT({cur}, {4 * nr});")
                o("        __builtin_amdgcn_sched_barrier(0);")
                if k + 1 < len(chunks):
                    o(f"        MJPL_SCENE_LOAD({nxt}, {chunks[k + 1][0]}, {4 * chunks[k + 1][1]});")
                o("        __builtin_amdgcn_sched_barrier(0);")
                for j in range(0, nr, 2):
                    last = r + j == SCENE_ROWS - 2
                    o(f"        MJPL_SCENE_PAIR({cur} + {4 * j}, {'acc_a' if last else 'cc'}, {'acc_b' if last else 'cc'}, {r + j}, {r + j + 1});")
                o("        __builtin_amdgcn_sched_barrier(0);")
            o("          } break;")
        o("        }")
        o("#undef MJPL_SCENE_LOAD")
        o("#undef MJPL_SCENE_WAIT")
        o("#undef MJPL_SCENE_PAIR")
        o("      }")
    o("      const float cur6[6] = {cx, cy, cz, zx, zy, zz};")
    if mbox:
        o("      const float cur6b[6] = {xx, xy, xz, yx, yy, yz};")
    o("      // partners some lane passed: bit k of the ballot <=> lane k's stored mask is non-zero")
    o("      for (unsigned long long ab = __builtin_amdgcn_ballot_w64((mlo | mhi) != 0); ab; ab &= ab - 1) {")
    o("        const int k = (int)__builtin_ctzll(ab);")
    o("        const unsigned long long pm = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(mlo, k) |")
    o("                                      ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(mhi, k) << 32);")
    o("        const int d = __builtin_amdgcn_readlane(dv, k);")
    o("        const int kind = d & 3, index = (d >> 2) & 255, ptype = (d >> 10) & 15;")
    o("        const bool pfirst = (d >> 14) & 1;")
    o("        // the candidates' certificate margin: this geom's motion relative to the partner (level 0: to the world; 0 without)")
    if cert_ok:
        o("        const int lv_ = (d >> 22) & 7;")
        o("        const float mc_ = lv_ == 0 ? dm0 : (lv_ == 1 ? dm1 : (lv_ == 2 ? dm2 : (lv_ == 3 ? dm3 : (lv_ == 4 ? dm4 : (lv_ == 5 ? dm5 : (lv_ == 6 ? dm6 : dm7))))));")
    else:
        o("        const float mc_ = 0.0f;")
    o("        float t6[6] = {0, 0, 0, 0, 0, 0};")
    if mbox:
        # (a switch over the slots in use keeps every access to the slot file a literal one)
        used1 = sorted({(x >> 2) & 255 for d_ in desc for x in d_ if (x & 3) == EK_SLOT})
        used2 = sorted({(x >> 16) & 63 for d_ in desc for x in d_ if (x & 3) == EK_SLOT} - {63})
        o("        float t6b[6] = {0, 0, 0, 0, 0, 0};")
        o("        const int index2 = (d >> 16) & 63;  // a stored box: the slot of its x and y axes")
        o("        if (kind == EK_SLOT) {")
        o("          switch (index) {")
        for n in used1:
            o(f"            case {n}: spec_get6(sf, {n}, t6); break;")
        o("            default: break;")
        o("          }")
        o("          switch (index2) {")
        for n in used2:
            o(f"            case {n}: spec_get6(sf, {n}, t6b); break;")
        o("            default: break;")
        o("          }")
        o("        }")
    else:
        o("        if (kind == EK_SLOT) slot_get6(sf, index, t6);")
    if mbox:
        o("        if ((d >> 15) & 1)")
        o("          queue_push<float, true, true>(wq, qb, dead, fl, active, far, ltab, lwcull, lwnarrow, nwpad, tol, ps, pm, kind, index,")
        o("                                        gtype, ptype, pfirst, gdoff, cur6, t6, cur6b, t6b, mc_);")
        o("        else")
        o("          queue_push<float, false, true>(wq, qn, dead, fl, active, far, ltab, lwcull, lwnarrow, nwpad, tol, ps, pm, kind, index,")
        o("                                         gtype, ptype, pfirst, gdoff, cur6, t6, nullptr, nullptr, mc_);")
    else:
        o("        if ((d >> 15) & 1)")
        o("          queue_push<float, true>(wq, qb, dead, fl, active, far, ltab, lwcull, lwnarrow, nwpad, tol, ps, pm, kind, index,")
        o("                                  gtype, ptype, pfirst, gdoff, cur6, t6, nullptr, nullptr, mc_);")
        o("        else")
        o("          queue_push<float, false>(wq, qn, dead, fl, active, far, ltab, lwcull, lwnarrow, nwpad, tol, ps, pm, kind, index,")
        o("                                   gtype, ptype, pfirst, gdoff, cur6, t6, nullptr, nullptr, mc_);")
    o("      }")
    o("      switch (g) {  // (a literal slot index keeps the slot file in registers)")
    put = "spec_put6" if mbox else "slot_put6"
    for si, (_, _, _, store, store2) in enumerate(stages):
        if store >= 0:
            if mbox and store2 != 63:
                o(f"        case {si}: {put}(sf, {store}, cur6); {put}(sf, {store2}, cur6b); break;")
            else:
                o(f"        case {si}: {put}(sf, {store}, cur6); break;")
    o("        default: break;")
    o("      }")
    o("    }")
    if mbox:
        o("    if (qn > 0) queue_drain<float, false, true, true>(wq, qn, ltab, lwcull, lwnarrow, nwpad, tol, ps);")
        o("    if (qb > 0) queue_drain<float, true, true, true>(wq, qb, ltab, lwcull, lwnarrow, nwpad, tol, ps);")
    else:
        o("    if (qn > 0) queue_drain<float, false, true>(wq, qn, ltab, lwcull, lwnarrow, nwpad, tol, ps);")
    if (info.wbox or generic) and not mbox:  # (a scene-generic library serves scenes with static boxes whatever scene it was generated from)
        o("    if (qb > 0) queue_drain<float, true, true>(wq, qb, ltab, lwcull, lwnarrow, nwpad, tol, ps);")
    o("    fl = wq.flags[lane] & 7;")
    o("    if (active && far) return V_UNSURE;  // nothing this lane's candidates said can be trusted")
    o("    return !active ? V_NONE : ((fl & 1) ? V_CONTACT : ((fl & 2) ? V_UNSURE : ((cert && !(fl & 4)) ? V_CLEAR : V_NONE)));")
    o("  }")
    o("};")
    return "\n".join(out) + "\n"



def dlit(x) -> str:
    """An exact C++17 hexadecimal literal of a float64."""
    return float(x).hex()


def generate_exact(ip, dp, info, generic: bool = False, fold: bool = True) -> str:
    """HIP source of `struct ExactSpec`: the float64 FK of k_patch_pairs (mjpl_filter.h) for one compiled program -- the
    interpreter's own statements, body by body, with the program's constants folded in (mjpl_amd/fold.py: products with
    an exact 0 or 1 and sums with an exact 0 are not executed, the unit-length test of mju_normalize4 is an interval
    test; every other operation in the interpreter's order).  Same values (up to the sign of exact zeros): the kernel's
    verdicts are the interpreter's, bit for bit; what goes away is ~700 scalar loads per wave and more than half of the
    chain's float64 instructions -- the latency of this chain is what a launch's tail kernel costs.
    fold=False: every operation of the statement (tests)."""
    from .fold import Fold
    out = []
    o = out.append
    nbody = int(ip[H_NBODYOPS])
    pc = int(ip[H_OFF_BODYOPS])
    f = Fold(indent="    ", fold=fold)
    c, v = f.c, f.v
    o("struct ExactSpec {")
    o("  // 1: geoms are numbered from the first moving one (a scene-generic library: model ids shift with the scene)")
    o(f"  static constexpr int kRelative = {1 if generic else 0};")
    o("  static __device__ __forceinline__ void fk_pair(const double *q, int qstride, double *save, int sstride, bool active,")
    o("                                                 int ga, int gb, mjpl::GeomT<double> &A, mjpl::GeomT<double> &Bg) {")
    o("    using namespace mjpl;")
    p, qt = [c(0), c(0), c(0)], [c(1), c(0), c(0), c(0)]
    R = [c(x) for x in (1, 0, 0, 0, 1, 0, 0, 0, 1)]
    stage = 0
    nsin = 0
    for b in range(nbody):
        parent, bdoff, njnt, save_slot, ngeom = (int(ip[pc + k]) for k in (B_PARENT, B_DOFF, B_NJNT, B_SAVE, B_NGEOM))
        pc += B_SIZE
        bd = dp[bdoff:]
        f.emit(f"// body op {b}")
        if parent == PARENT_CUR:
            pp, pq, pR = p, qt, R
        elif parent == PARENT_STATIC:
            pp, pq, pR = [c(x) for x in bd[7:10]], [c(x) for x in bd[10:14]], [c(x) for x in bd[14:23]]
        else:
            k0 = f.n
            f.n += 1
            f.emit(f"const double *sv{k0} = save + (size_t){parent - 1} * 7 * sstride;")
            names = [f"sv{k0}_{k}" for k in range(7)]
            f.emit("const double " + ", ".join(f"{names[k]} = sv{k0}[{k} * sstride]" for k in range(7)) + ";")
            pp, pq = [v(n) for n in names[:3]], [v(n) for n in names[3:]]
            pR = f.quat2mat(pq)
        np_ = [f.add(x, y) for x, y in zip(f.mul_mat_vec3(pR, [c(x) for x in bd[0:3]]), pp)]
        nq_ = f.mul_quat(pq, [c(x) for x in bd[3:7]])
        for j in range(njnt):
            jtype, qsrc, jflags, jdoff = (int(ip[pc + k]) for k in (J_TYPE, J_QSRC, J_FLAGS, J_DOFF))
            pc += J_SIZE
            jd = dp[jdoff:]
            qv = v(f"q[{qsrc} * qstride]") if qsrc >= 0 else c(jd[7])
            dq = f.sub(qv, c(jd[6]))
            jaxis, jpos = [c(x) for x in jd[0:3]], [c(x) for x in jd[3:6]]
            if jtype == JT_SLIDE:
                xaxis = f.rot_vec_quat(jaxis, nq_)
                np_ = [f.add(np_[r], f.mul(xaxis[r], dq)) for r in range(3)]
            else:
                xanchor = np_
                if jflags & JF_POS_NONZERO:
                    xanchor = [f.add(x, y) for x, y in zip(f.rot_vec_quat(jpos, nq_), np_)]
                half = f.mul(dq, c(0.5))
                f.emit(f"double sn{nsin}, cs{nsin};")
                f.emit(f"sincos_half({f.text(half)}, &sn{nsin}, &cs{nsin});")
                sn, cs = v(f"sn{nsin}"), v(f"cs{nsin}")
                nsin += 1
                nq_ = f.mul_quat(nq_, [cs, f.mul(jaxis[0], sn), f.mul(jaxis[1], sn), f.mul(jaxis[2], sn)])
                if jflags & JF_POS_NONZERO:
                    vec = f.rot_vec_quat(jpos, nq_)
                    np_ = [f.sub(xanchor[r], vec[r]) for r in range(3)]
        nq_ = f.normalize4(nq_)
        p, qt = np_, nq_
        R = f.quat2mat(qt)
        if save_slot >= 0:
            f.emit(f"{{ double *sv = save + (size_t){save_slot} * 7 * sstride;")
            f.emit("  " + " ".join(f"sv[{k} * sstride] = {f.text(p[k])};" for k in range(3)))
            f.emit("  " + " ".join(f"sv[{3 + k} * sstride] = {f.text(qt[k])};" for k in range(4)) + " }")
        for gi in range(ngeom):
            gflags, gdoff, geom_id = (int(ip[pc + k]) for k in (G_FLAGS, G_DOFF, G_GEOMID))
            pc += G_SIZE + MAX_SLOTS
            gd = dp[gdoff:]
            if generic:
                geom_id = stage
            stage += 1
            f.emit(f"if (__ballot(active && (ga == {geom_id} || gb == {geom_id})) != 0ull) {{")
            with f.scope():
                ind, f.ind = f.ind, f.ind + "  "
                if gflags & GF_SAMEPOS:
                    cpos = p
                else:
                    cpos = [f.add(x, y) for x, y in zip(f.mul_mat_vec3(R, [c(x) for x in gd[0:3]]), p)]
                cm = R if gflags & GF_SAMEROT else f.quat2mat(f.mul_quat(qt, [c(x) for x in gd[3:7]]))
                f.emit(f"const bool isa = ga == {geom_id}, isb = gb == {geom_id};")
                for k in range(3):
                    t = f.text(cpos[k])
                    f.emit(f"A.pos[{k}] = isa ? {t} : A.pos[{k}]; Bg.pos[{k}] = isb ? {t} : Bg.pos[{k}];")
                for k in range(9):
                    t = f.text(cm[k])
                    f.emit(f"A.m[{k}] = isa ? {t} : A.m[{k}]; Bg.m[{k}] = isb ? {t} : Bg.m[{k}];")
                f.ind = ind
            f.emit("}")
    out.extend(f.lines)
    o("  }")
    o("};")
    return "\n".join(out) + "\n"

def _exact_body_fk(f, ip, dp, pc, p, qt, R, sin_tag: str):
    """mj_kinematics of ONE body op as folded straight-line float64 code (the statements of run_config_queued<double> /
    patch_pairs_body, mjpl_device.h): `pc` at the body op's header.  (p, qt, R): the walk's current state as Fold values.
    -> (pc behind the joints, new p, qt, R, save slot, number of geoms)."""
    c, v = f.c, f.v
    parent, bdoff, njnt, save_slot, ngeom = (int(ip[pc + k]) for k in (B_PARENT, B_DOFF, B_NJNT, B_SAVE, B_NGEOM))
    pc += B_SIZE
    bd = dp[bdoff:]
    if parent == PARENT_CUR:
        pp, pq, pR = p, qt, R
    elif parent == PARENT_STATIC:
        pp, pq, pR = [c(x) for x in bd[7:10]], [c(x) for x in bd[10:14]], [c(x) for x in bd[14:23]]
    else:
        k0 = f.n
        f.n += 1
        f.emit(f"const double *sv{k0} = save + (size_t){parent - 1} * 7 * sstride;")
        names = [f"{f.prefix}sv{k0}_{k}" for k in range(7)]
        f.emit("const double " + ", ".join(f"{names[k]} = sv{k0}[{k} * sstride]" for k in range(7)) + ";")
        pp, pq = [v(n) for n in names[:3]], [v(n) for n in names[3:]]
        pR = f.quat2mat(pq)
    np_ = [f.add(x, y) for x, y in zip(f.mul_mat_vec3(pR, [c(x) for x in bd[0:3]]), pp)]
    nq_ = f.mul_quat(pq, [c(x) for x in bd[3:7]])
    for j in range(njnt):
        jtype, qsrc, jflags, jdoff = (int(ip[pc + k]) for k in (J_TYPE, J_QSRC, J_FLAGS, J_DOFF))
        pc += J_SIZE
        jd = dp[jdoff:]
        qv = v(f"q[{qsrc} * qstride]") if qsrc >= 0 else c(jd[7])
        dq = f.sub(qv, c(jd[6]))
        jaxis, jpos = [c(x) for x in jd[0:3]], [c(x) for x in jd[3:6]]
        if jtype == JT_SLIDE:
            xaxis = f.rot_vec_quat(jaxis, nq_)
            np_ = [f.add(np_[r], f.mul(xaxis[r], dq)) for r in range(3)]
        else:
            xanchor = np_
            if jflags & JF_POS_NONZERO:
                xanchor = [f.add(x, y) for x, y in zip(f.rot_vec_quat(jpos, nq_), np_)]
            half = f.mul(dq, c(0.5))
            tag = f"{sin_tag}{j}"
            f.emit(f"double sn{tag}, cs{tag};")
            f.emit(f"sincos_half({f.text(half)}, &sn{tag}, &cs{tag});")
            sn, cs = v(f"sn{tag}"), v(f"cs{tag}")
            nq_ = f.mul_quat(nq_, [cs, f.mul(jaxis[0], sn), f.mul(jaxis[1], sn), f.mul(jaxis[2], sn)])
            if jflags & JF_POS_NONZERO:
                vec = f.rot_vec_quat(jpos, nq_)
                np_ = [f.sub(xanchor[r], vec[r]) for r in range(3)]
    nq_ = f.normalize4(nq_)
    Rn = f.quat2mat(nq_)
    if save_slot >= 0:
        f.emit(f"{{ double *sv = save + (size_t){save_slot} * 7 * sstride;")
        f.emit("  " + " ".join(f"sv[{k} * sstride] = {f.text(np_[k])};" for k in range(3)))
        f.emit("  " + " ".join(f"sv[{3 + k} * sstride] = {f.text(nq_[k])};" for k in range(4)) + " }")
    return pc, np_, nq_, Rn, save_slot, ngeom


def generate_full_exact(ip, fp, dp, info) -> str | None:
    """HIP source of `struct ExactFull`: the float64 check of one configuration through the candidate queues
    (run_config_queued<double>, mjpl_device.h -- what k_edges_fused_f64 calls for every endpoint and waypoint when the
    filter is off or refused) as straight-line code for ONE compiled program: forward kinematics and geom poses with the
    constants folded in (mjpl_amd/fold.py), one bounding cull per ENABLED partner with its row of the table as literals,
    the interpreter's own pushes, drains and narrowphase routines behind them.  The culls are the interpreter's
    expressions (|c - X|^2 as x x + y y + z z, + dead, against the same bound), so the same candidates reach the same
    routines: the verdicts are the interpreter's bit for bit.  None: the model keeps the interpreting kernel (moving
    boxes, a slot file wider than 16, the immediate interpreter).
    MEASURED (round 5, tools/f64_probe.py): no faster than the interpreting kernel -- 0.4497 against 0.4473 ms per 262 144
    edges.  The interpreter fetches four partners' rows with one sixteen-value scalar load; every float64 literal here
    is two scalar moves, and the kernel spills 222 scalar registers.  So libraries carry it only when built with
    MJPL_SPEC_F64=1; what a generated float64 path would need is the partners' rows from the table (scalar loads) around
    the folded FK -- the FK is a tenth of the check."""
    from .fold import Fold
    if info.immediate or info.mbox or int(info.maxs) > 16:
        return None
    nbody, nwpad, maxs = int(ip[H_NBODYOPS]), int(ip[H_NWPAD]), int(info.maxs)
    off_wcull, off_wnarrow = int(ip[H_OFF_WCULL]), int(ip[H_OFF_WNARROW])

    def wc_at(wrow, f_):
        return off_wcull + ((wrow >> 2) << 4) + (f_ << 2) + (wrow & 3)
    state = ["p0", "p1", "p2", "q0", "q1", "q2", "q3"] + [f"R{k}" for k in range(9)]
    stages, desc, pending = [], [], []
    pc = int(ip[H_OFF_BODYOPS])
    known = True  # the walk's state is still a generation-time constant (nothing assigned to the variables yet)
    cp, cq, cR = None, None, None
    for b in range(nbody):
        f = Fold(indent="          ", prefix=f"b{b}_")
        if known and b == 0:
            p, qt, R = [f.c(0)] * 3, [f.c(1), f.c(0), f.c(0), f.c(0)], [f.c(x) for x in (1, 0, 0, 0, 1, 0, 0, 0, 1)]
        else:
            p, qt, R = [f.v(n) for n in state[:3]], [f.v(n) for n in state[3:7]], [f.v(n) for n in state[7:]]
        pc, p, qt, R, save_slot, ngeom = _exact_body_fk(f, ip, dp, pc, p, qt, R, f"b{b}j")
        for name, val in zip(state, p + qt + R):
            f.emit(f"{name} = {f.text(val)};")
        known = False
        pending.extend(["        {"] + f.lines + ["        }"])
        for gi in range(ngeom):
            gtype, gflags, gdoff, store, geom_id, smask = (int(ip[pc + k]) for k in (G_TYPE, G_FLAGS, G_DOFF, G_STORE, G_GEOMID, G_SMASK))
            wmask = (int(ip[pc + G_WMASK_LO]) & 0xFFFFFFFF) | ((int(ip[pc + G_WMASK_HI]) & 0xFFFFFFFF) << 32)
            pmask = (int(ip[pc + G_PMASK_LO]) & 0xFFFFFFFF) | ((int(ip[pc + G_PMASK_HI]) & 0xFFFFFFFF) << 32)
            swords = [int(x) for x in ip[pc + G_SIZE: pc + G_SIZE + MAX_SLOTS]]
            pc += G_SIZE + MAX_SLOTS
            gd = dp[gdoff:]
            g = Fold(indent="          ", prefix=f"g{len(stages)}_")
            sp_, sq_, sR_ = [g.v(n) for n in state[:3]], [g.v(n) for n in state[3:7]], [g.v(n) for n in state[7:]]
            cpos = sp_ if gflags & GF_SAMEPOS else [g.add(x, y) for x, y in zip(g.mul_mat_vec3(sR_, [g.c(x) for x in gd[0:3]]), sp_)]
            if gflags & GF_SAMEROT:
                zax = [sR_[2], sR_[5], sR_[8]]
            else:  # quat2zaxis(qt (x) lq): the third column of quat2mat, its expressions
                gq = g.mul_quat(sq_, [g.c(x) for x in gd[3:7]])
                m_ = g.mul
                q00, q01, q02 = m_(gq[0], gq[0]), m_(gq[0], gq[1]), m_(gq[0], gq[2])
                q11, q13, q22, q23, q33 = m_(gq[1], gq[1]), m_(gq[1], gq[3]), m_(gq[2], gq[2]), m_(gq[2], gq[3]), m_(gq[3], gq[3])
                two = g.c(2.0)
                zax = [m_(two, g.add(q13, q02)), m_(two, g.sub(q23, q01)), g.add(g.sub(g.sub(q00, q11), q22), q33)]
            for name, val in zip(("cx", "cy", "cz", "zx", "zy", "zz"), cpos + zax):
                g.emit(f"{name} = {g.text(val)};")
            cv = [g.v("cx"), g.v("cy"), g.v("cz")]
            partners = []
            wbound = gd[GD_WBOUND: GD_WBOUND + nwpad]
            sbound = gd[GD_WBOUND + 2 * nwpad: GD_WBOUND + 2 * nwpad + MAX_SLOTS]
            for wrow in range(64):  # static planes
                if not (pmask >> wrow) & 1:
                    continue
                ppos = [g.c(dp[wc_at(wrow, k)]) for k in range(3)]
                pz = [g.c(dp[off_wnarrow + wrow * WN_LEN + WN_ZAXIS + k]) for k in range(3)]
                dif = [g.sub(cv[k], ppos[k]) for k in range(3)]
                dot = g.sum([g.mul(dif[k], pz[k]) for k in range(3)])
                g.emit(f"MJPL_X64_HIT({len(partners)}, !({g.text(dot)} + dead > {dlit(wbound[wrow])}));")
                partners.append((EK_PLANE, wrow, GT_PLANE, 1, 0))
            for wrow in range(64):  # other static geoms
                if not (wmask >> wrow) & 1:
                    continue
                info_word = int(np.frombuffer(np.float64(dp[wc_at(wrow, 3)]).tobytes(), dtype=np.int32)[0])
                ptype, pgid = info_word & 255, info_word >> 8
                pfirst = 1 if (ptype < gtype or (ptype == gtype and pgid < geom_id)) else 0
                dd = [g.sub(cv[k], g.c(dp[wc_at(wrow, k)])) for k in range(3)]
                sq = g.sum([g.mul(x, x) for x in dd])
                g.emit(f"MJPL_X64_HIT({len(partners)}, !({g.text(sq)} + dead > {dlit(wbound[wrow])}));")
                partners.append((EK_STATIC, wrow, ptype, pfirst, 1 if ptype == GT_BOX else 0))
            for n in range(maxs):  # earlier moving geoms in the slot file
                if not (smask >> n) & 1:
                    continue
                pw = swords[n]
                g.emit(f"MJPL_X64_SLOTCULL({len(partners)}, {n}, {dlit(sbound[n])});")
                partners.append((EK_SLOT, n, (pw >> 12) & 15, 1 if (pw & P_FIRST) else 0, 0))
            if len(partners) > 64:
                return None
            stages.append((pending + g.lines, gtype, gdoff, store & 63 if store >= 0 else -1))
            pending = []
            desc.append([(kind << 0) | (index << 2) | (ptype << 10) | (pfirst << 14) | (boxq << 15) for kind, index, ptype, pfirst, boxq in partners])
    nstage = len(stages)
    wbox = any((x >> 15) & 1 for d in desc for x in d)
    out = []
    o = out.append
    o(f"// GENERATED: the float64 check of program {info.hash:016x} through the candidate queues -- {nstage} moving geoms, "
      f"{sum(len(d) for d in desc)} literal pairs")
    o("#define MJPL_X64_HIT(k, pass) \\")
    o("  do { const unsigned long long m_ = __builtin_amdgcn_ballot_w64(pass); \\")
    o("       mjpl::park_mask<k>(mlo, mhi, m_); } while (0)")
    o("#define MJPL_X64_SLOTCULL(k, n, BOUND) \\")
    o("  do { const double dx_ = cx - sf.f[0][n], dy_ = cy - sf.f[1][n], dz_ = cz - sf.f[2][n]; \\")
    o("       MJPL_X64_HIT(k, !(dx_ * dx_ + dy_ * dy_ + dz_ * dz_ + dead > (BOUND))); } while (0)")
    o(f"__constant__ int kExactDesc[{nstage} * 64] = {{")
    for d in desc:
        o("  " + ", ".join(str(x) for x in (d + [0] * (64 - len(d)))) + ",")
    o("};")
    o("struct ExactFull {")
    o(f"  static constexpr int kMaxs = {maxs};")
    # (outlined: inlined at the pool kernel's one call site the build returned no contact for any waypoint tile while endpoint
    #  tiles were right -- every variant with a second call site, and this outlined form, return the interpreter's verdicts on
    #  all edges; observed with ROCm 7.2's compiler at 222 spilled scalar registers, not explained)
    # (MJPL_SPEC_F64_INLINE=1: the inlined form, for tools/f64_inline_probe.py -- the experiment that looks for what breaks it)
    attr = "always_inline" if os.environ.get("MJPL_SPEC_F64_INLINE") == "1" else "noinline"
    o(f"  static __device__ __attribute__(({attr})) int run(const double *ltab, const double *q, int qstride, double *save, int sstride, bool active,")
    o("                                            const mjpl::WaveQueue<double, false> &wq, int item, const mjpl::PatchSink &ps) {")
    o("    using namespace mjpl;")
    o(f"    SlotFile<double, {maxs}> sf;")
    o("    double p0 = 0, p1 = 0, p2 = 0, q0 = 1, q1 = 0, q2 = 0, q3 = 0;")
    o("    double R0 = 1, R1 = 0, R2 = 0, R3 = 0, R4 = 1, R5 = 0, R6 = 0, R7 = 0, R8 = 1;")
    o("    const int lane = threadIdx.x & 63;")
    o("    double dead = active ? 0.0 : (double)__builtin_inff();")
    o("    int fl = 0, qn = 0, qb = 0;")
    o("    wq.flags[lane] = (int)((unsigned)item << 3);")
    o(f"    const int nwpad = {nwpad};")
    o(f"    const double *lwcull = ltab + {off_wcull}, *lwnarrow = ltab + {off_wnarrow};")
    o("#pragma nounroll")
    o(f"    for (int g = 0; g < {nstage}; g++) {{")
    o("      if (__builtin_amdgcn_ballot_w64(dead == 0.0) == 0ull && qn == 0 && qb == 0) break;  // every lane decided")
    o("      double cx = 0, cy = 0, cz = 0, zx = 0, zy = 0, zz = 0;")
    o("      int mlo = 0, mhi = 0;  // lane k holds the hit mask of this geom's partner k")
    o("      const int dv = kExactDesc[64 * g + lane];")
    o("      int gtype = 0, gdoff = 0;")
    o("      switch (g) {")
    for si, (lines, gtype, gdoff, store) in enumerate(stages):
        o(f"        case {si}: {{")
        out.extend(lines)
        o(f"          gtype = {gtype}; gdoff = {gdoff};")
        o("        } break;")
    o("        default: break;")
    o("      }")
    o("      const double cur6[6] = {cx, cy, cz, zx, zy, zz};")
    o("      for (unsigned long long ab = __builtin_amdgcn_ballot_w64((mlo | mhi) != 0); ab; ab &= ab - 1) {")
    o("        const int k = (int)__builtin_ctzll(ab);")
    o("        const unsigned long long pm = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(mlo, k) |")
    o("                                      ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(mhi, k) << 32);")
    o("        const int d = __builtin_amdgcn_readlane(dv, k);")
    o("        const int kind = d & 3, index = (d >> 2) & 255, ptype = (d >> 10) & 15;")
    o("        const bool pfirst = (d >> 14) & 1;")
    o("        double t6[6] = {0, 0, 0, 0, 0, 0};")
    o("        if (kind == EK_SLOT) slot_get6(sf, index, t6);")
    o("        if ((d >> 15) & 1)")
    o("          queue_push<double, true>(wq, qb, dead, fl, active, false, ltab, lwcull, lwnarrow, nwpad, 0.0, ps, pm, kind, index, gtype, ptype,")
    o("                                   pfirst, gdoff, cur6, t6);")
    o("        else")
    o("          queue_push<double, false>(wq, qn, dead, fl, active, false, ltab, lwcull, lwnarrow, nwpad, 0.0, ps, pm, kind, index, gtype, ptype,")
    o("                                    pfirst, gdoff, cur6, t6);")
    o("      }")
    o("      switch (g) {  // (a literal slot index keeps the slot file in registers)")
    for si, (_, _, _, store) in enumerate(stages):
        if store >= 0:
            o(f"        case {si}: slot_put6(sf, {store}, cur6); break;")
    o("        default: break;")
    o("      }")
    o("    }")
    o("    if (qn > 0) queue_drain<double, false, true>(wq, qn, ltab, lwcull, lwnarrow, nwpad, 0.0, ps);")
    if wbox:
        o("    if (qb > 0) queue_drain<double, true, true>(wq, qb, ltab, lwcull, lwnarrow, nwpad, 0.0, ps);")
    o("    fl = wq.flags[lane] & 3;")
    o("    return !active ? V_NONE : ((fl & 1) ? V_CONTACT : V_NONE);")
    o("  }")
    o("};")
    return "\n".join(out) + "\n"


_TU = """// GENERATED translation unit: the float32 filter kernels of mjpl_filter.h around one model's Spec.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
%(waves_define)s#include "mjpl_filter.h"
#include "mjpl_project.h"

namespace {
%(spec)s
%(exact)s
%(exact_full)s
}  // namespace

using namespace mjpl;
#define SPEC_GRANT(kern)                                                                            \\
  static std::atomic<size_t> granted{0};  /* (engines of several threads share the library) */      \\
  if (lds > granted.load(std::memory_order_relaxed)) {                                              \\
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1; \\
    size_t have = granted.load(std::memory_order_relaxed);                                          \\
    while (have < lds && !granted.compare_exchange_weak(have, lds, std::memory_order_relaxed)) {}   \\
  }
#define SPEC_LAUNCH(kern, ...)                                                                      \\
  do {                                                                                              \\
    SPEC_GRANT(kern)                                                                                \\
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, st, __VA_ARGS__);                        \\
    return hipGetLastError() == hipSuccess ? 0 : -1;                                                \\
  } while (0)

extern "C" {
int mjpl_spec_abi(void) { return MJPL_SPEC_ABI; }
unsigned long long mjpl_spec_src_stamp(void) { return MJPL_SRC_STAMP; }
// 0: the whole program in literals; else rows per moving geom << 8 | moving geoms of a scene-generic library
int mjpl_spec_generic(void) { return %(generic)d; }
unsigned long long mjpl_spec_hash(void) { return 0x%(hash)016xull; }
int mjpl_spec_launch_configs(hipStream_t st, unsigned grid, unsigned block, size_t lds, const int *ip, int nip, const float *fp,
                             int nfp, const double *Q, int64_t N, int layout, float tol, uint8_t *valid, int *ulist, int *ucount,
                             UndecidedConfigs uc, int *zero_next) {
  SPEC_LAUNCH((k_filter_configs<Spec, %(maxs)d, %(wbox)s, %(mbox)s>), ip, nip, fp, nfp, Q, N, layout, tol, valid, ulist, ucount, uc,
              zero_next);
}
int mjpl_spec_launch_endpoints(hipStream_t st, unsigned grid, unsigned block, size_t lds, const int *ip, int nip, const float *fp,
                               int nfp, const double *QA, const double *QB, int64_t E, int layout, float tol, uint8_t *valid,
                               int32_t *first_bad, int *status, int *ulist, int *ucount, UndecidedConfigs uc, int *slist,
                               int *scount, ItemBuffers ib, double step, int *zero_next) {
  SPEC_LAUNCH((k_filter_endpoints<Spec, %(maxs)d, %(wbox)s, %(mbox)s>), ip, nip, fp, nfp, QA, QB, E, layout, tol, valid, first_bad, status,
              ulist, ucount, uc, slist, scount, ib, step, zero_next);
}
int mjpl_spec_launch_items(hipStream_t st, unsigned grid, unsigned block, size_t lds, const int *ip, int nip, const float *fp,
                           int nfp, ItemBuffers ib, EdgeSource src, float tol, uint8_t *valid, int32_t *first_bad, int *ulist, int *ucount,
                           UndecidedConfigs uc) {
  SPEC_LAUNCH((k_filter_items<Spec, %(maxs)d, %(wbox)s, %(mbox)s>), ip, nip, fp, nfp, ib, src, tol, valid, first_bad, ulist, ucount, uc);
}
// persistent kernels (one wave per tile of 64, tiles from a device counter): the launcher sizes the grid
int mjpl_spec_launch_endpoints_pw(hipStream_t st, size_t lds, const int *ip, int nip, const float *fp, int nfp, const double *QA,
                                  const double *QB, int64_t E, int layout, float tol, uint8_t *valid, int32_t *first_bad,
                                  int *status, int *ulist, int *ucount, UndecidedConfigs uc, ItemBuffers ib, double step,
                                  int *zero_next, int *tiles) {
  auto kern = k_filter_endpoints_pw<Spec, %(maxs)d, %(wbox)s, %(mbox)s>;
  SPEC_GRANT(kern);
  const unsigned grid = persistent_grid(kern, lds, (E + 63) / 64);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, ip, nip, fp, nfp, QA, QB, E, layout, tol, valid, first_bad, status, ulist,
                     ucount, uc, ib, step, zero_next, tiles);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int mjpl_spec_launch_items_pw(hipStream_t st, size_t lds, const int *ip, int nip, const float *fp, int nfp, ItemBuffers ib,
                              EdgeSource src, float tol, uint8_t *valid, int32_t *first_bad, int *ulist, int *ucount,
                              UndecidedConfigs uc, int *tiles) {
  auto kern = k_filter_items_pw<Spec, %(maxs)d, %(wbox)s, %(mbox)s>;
  SPEC_GRANT(kern);
  const unsigned grid = persistent_grid(kern, lds, (long long)ib.cap / 64);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kBlock), lds, st, ip, nip, fp, nfp, ib, src, tol, valid, first_bad, ulist, ucount, uc, tiles);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
int mjpl_spec_launch_tail(hipStream_t st, size_t lds, TailArgs a) {
  auto kern = k_tail<ExactSpec, %(maxs)d, %(maxsd)d, %(wbox)s, %(mbox)s>;
  SPEC_GRANT(kern);
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.nw + a.np + a.nx)), dim3(kBlock), lds, st, a);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
// the whole float32 filter of an edge launch as one kernel (mjpl_fused.h); nwaves: wavefronts per workgroup, which
// must be what mjpl_spec_fused_waves reports (twelve at three waves per SIMD; eight for a model with moving boxes,
// whose code is built for two)
int mjpl_spec_fused_waves(void) { return %(fwaves)d; }
// 1: the library's check was generated with the edge certificate (MJPL_SPEC_CERT=1): its fused kernel keeps |QB - QA| rows in LDS
int mjpl_spec_fused_cert(void) { return Spec::kCert ? 1 : 0; }
int mjpl_spec_launch_fused(hipStream_t st, int nwaves, size_t lds, FusedArgs a) {
  if (nwaves != %(fwaves)d) return -1;
  auto kern = k_edges_fused<Spec, %(maxs)d, %(wbox)s, %(mbox)s, %(fwaves)d>;
  SPEC_GRANT(kern);
  return fused_launch(kern, nwaves, lds, a, st) == hipSuccess ? 0 : -1;
}
%(fused_f64)s
int mjpl_spec_launch_patch(hipStream_t st, unsigned grid, unsigned block, size_t lds, const int *ip, int nip, const double *dp,
                           int ndp, GeomTable gt, UndecidedConfigs uc, uint8_t *valid, int32_t *first_bad) {
  SPEC_LAUNCH((k_patch_pairs<ExactSpec>), ip, nip, dp, ndp, gt, uc, valid, first_bad);
}
}
// ---- PoseConstraint projections of this model's site bodies (mjpl_project.h)
%(pose)s
"""


# ---- the PoseConstraint projection as straight-line code (mjpl_project.h: PoseStatic<PS>) ----------------------
PH_NBODY, PH_NJOINT, PH_NQ, PH_SIZE = 0, 1, 2, 6  # (mjpl_pose.h)


def _model_desc(model):
    d = _engine._ModelDesc()
    d.nq, d.njnt, d.nbody, d.ngeom = model.nq, model.njnt, model.nbody, model.ngeom
    keep = []
    for name, typ in _engine._ModelDesc._fields_[4:]:
        arr = getattr(model, name)
        arr = _engine._i32(arr) if typ is _engine._I32P else _engine._f64(arr)
        keep.append(arr)
        setattr(d, name, arr.ctypes.data_as(typ))
    return d, keep


def dump_pose_chain(model, site_body: int):
    """The chain program mjpl_pose_create compiles for (model, site body), on the host (no GPU)
    -> (pi int32[], pd float64[], hash)."""
    lib = _engine.load_library()
    d, keep = _model_desc(model)
    npi, npd, h = C.c_int32(0), C.c_int32(0), C.c_uint64(0)

    def call(pi, pd):
        rc = lib.mjpl_pose_chain_dump(C.byref(d), int(site_body), None if pi is None else pi.ctypes.data_as(_engine._I32P), C.byref(npi),
                                      None if pd is None else pd.ctypes.data_as(_engine._F64P), C.byref(npd), C.byref(h))
        if rc != 0:
            raise _engine.MjplError(rc, lib.mjpl_last_error().decode())

    call(None, None)
    pi, pd = np.zeros(npi.value, np.int32), np.zeros(npd.value, np.float64)
    call(pi, pd)
    return pi, pd, int(h.value)


def pose_site_bodies(model) -> list[int]:
    """The bodies a generated projection is made for: every body that carries a site (a PoseConstraint names a site)."""
    return sorted({int(b) for b in np.asarray(model.site_bodyid).reshape(-1) if int(b) > 0})


def generate_pose(pi, pd, hash_: int, index: int, qualifier: str = "static __device__ __forceinline__", fold: bool = True) -> str:
    """`struct PoseSpec<index>`: mj_kinematics along one chain with the model's constants folded in (mjpl_amd/fold.py):
    the operations of pose_chain (mjpl_pose.h) in the same order, minus those whose result the constants decide
    (products with an exact 0 or 1, sums with an exact 0) -- the same float64 values up to the sign of exact zeros.
    The half-angle sines and cosines of the hinges are arguments: the caller computes them, one joint after the other
    or one joint per lane of a row's group (mjpl_project.h).  What goes away against the interpreting kernel: the
    reading of the chain program, the loop control, the trip through LDS of the joints' axes and anchors, and about
    half of the chain's arithmetic.  qualifier: how the member functions are declared (tests compile a chain for the host)."""
    from .fold import Fold
    nb, nj, nq = int(pi[PH_NBODY]), int(pi[PH_NJOINT]), int(pi[PH_NQ])
    f = Fold(indent="    ", fold=fold)  # (fold=False: every operation of the statement, for tests)
    c, v = f.c, f.v
    jtypes, qadrs, jids, q0s = [], [], [], []
    p, qt = [c(0), c(0), c(0)], [c(1), c(0), c(0), c(0)]
    R = [c(x) for x in (1, 0, 0, 0, 1, 0, 0, 0, 1)]
    ic, dc, jk = PH_SIZE, 0, 0
    for b in range(nb):
        njnt = int(pi[ic]); ic += 1
        f.emit(f"// chain body {b}")
        bpos, bquat = [c(x) for x in pd[dc:dc + 3]], [c(x) for x in pd[dc + 3:dc + 7]]
        dc += 7
        np_ = [f.add(x, y) for x, y in zip(f.mul_mat_vec3(R, bpos), p)]
        nq_ = f.mul_quat(qt, bquat)
        for j in range(njnt):
            jtype, qadr, jid = int(pi[ic]), int(pi[ic + 1]), int(pi[ic + 2]); ic += 3
            jtypes.append(jtype); qadrs.append(qadr); jids.append(jid); q0s.append(float(pd[dc + 6]))
            jaxis, jpos = [c(x) for x in pd[dc:dc + 3]], [c(x) for x in pd[dc + 3:dc + 6]]
            dq = f.sub(v(f"q[{qadr}]"), c(pd[dc + 6]))
            dc += 7
            xaxis = f.rot_vec_quat(jaxis, nq_)
            xanchor = [f.add(x, y) for x, y in zip(f.rot_vec_quat(jpos, nq_), np_)]
            for r in range(3):
                f.emit(f"jx[{jk}][{r}] = {f.text(xaxis[r])}; jx[{jk}][{3 + r}] = {f.text(xanchor[r])};")
            if jtype == JT_SLIDE:
                np_ = [f.add(np_[r], f.mul(xaxis[r], dq)) for r in range(3)]
            else:
                sn, cs = v(f"sn[{jk}]"), v(f"cs[{jk}]")
                qloc = [cs, f.mul(jaxis[0], sn), f.mul(jaxis[1], sn), f.mul(jaxis[2], sn)]
                nq_ = f.mul_quat(nq_, qloc)
                vec = f.rot_vec_quat(jpos, nq_)
                np_ = [f.sub(xanchor[r], vec[r]) for r in range(3)]
            jk += 1
        nq_ = f.normalize4(nq_)
        p, qt = np_, nq_
        R = f.quat2mat(qt)
    assert jk == nj
    f.emit("// the site (mj_local2Global)")
    spos = [v(f"tail[mjpl::PT_SITE_POS + {k}]") for k in range(3)]
    squat = [v(f"tail[mjpl::PT_SITE_QUAT + {k}]") for k in range(4)]
    sp = f.mul_mat_vec3(R, spos)
    for k in range(3):
        f.emit(f"out.site_xpos[{k}] = {f.text(f.add(sp[k], p[k]))};")
    sm = f.quat2mat(f.mul_quat(qt, squat))
    for k in range(9):
        f.emit(f"out.site_xmat[{k}] = {f.text(sm[k])};")
    out = []
    o = out.append
    njx = max(nj, 1)
    tab = lambda vals: ", ".join(str(x) for x in vals) or "0"  # noqa: E731
    o(f"// GENERATED: chain program {hash_:016x} -- {nb} bodies, {nj} joints, nq {nq}; "
      f"{f.ops['mul']} products, {f.ops['add']} sums, {f.ops['sqrt']} square roots after folding")
    o(f"struct PoseSpec{index} {{")
    o(f"  static constexpr int kNQ = {nq}, kNJ = {nj};")
    o(f"  static constexpr unsigned long long kHash = 0x{hash_:016x}ull;")
    o(f"  {qualifier} constexpr int jtype(int k) {{ constexpr int t[{njx}] = {{{tab(jtypes)}}}; return t[k]; }}")
    o(f"  {qualifier} constexpr int qadr(int k) {{ constexpr int t[{njx}] = {{{tab(qadrs)}}}; return t[k]; }}")
    o(f"  {qualifier} constexpr int jid(int k) {{ constexpr int t[{njx}] = {{{tab(jids)}}}; return t[k]; }}")
    o("  // half the angle of chain joint k (a hinge's: what its sine and cosine are taken of; 0 for a slide joint)")
    o(f"  {qualifier} double half_angle(int k, const double (&q)[{nq}]) {{")
    o("    switch (k) {")
    for k in range(nj):
        if jtypes[k] == JT_HINGE:
            h = Fold(indent="")
            o(f"      case {k}: return {h.text(h.sub(h.v(f'q[{qadrs[k]}]'), h.c(q0s[k])))} * 0.5;" if q0s[k] == 0.0 else
              f"      case {k}: return (q[{qadrs[k]}] - {dlit(q0s[k])}) * 0.5;")
    o("      default: return 0.0;")
    o("    }")
    o("  }")
    o(f"  {qualifier} void chain(const double (&q)[{nq}], const double (&sn)[{njx}], const double (&cs)[{njx}], double (&jx)[{njx}][6],")
    o("                                               mjpl::PoseChainOut &out, const double *tail) {")
    out.extend(f.lines)
    o("  }")
    o("};")
    return "\n".join(out) + "\n"


def generate_pose_section(model, nplan: int = 0, qidx=None) -> str:
    """All generated projections of a model and the entry points a library exports for them.  nplan: the number of
    planning joints of the program the library is built for (the planner's chunk kernel keeps a lane's rows in
    registers when it is a constant; 0: unknown); qidx: their qpos addresses (None: unknown -- the kernels read the
    planner's table)."""
    specs = []
    for k, b in enumerate(pose_site_bodies(model)):
        pi, pd, h = dump_pose_chain(model, b)
        if int(pi[PH_NJOINT]) == 0 or int(pi[PH_NJOINT]) > 12 or int(pi[PH_NQ]) > 16:
            continue  # (nothing to project / registers: the interpreting kernels serve such a chain)
        specs.append((len(specs), h, generate_pose(pi, pd, h, len(specs))))
    n = len(specs)
    np_ = int(nplan) if 0 < int(nplan) <= 16 else 0
    planq = "mjpl::PlanQRuntime"
    plan_src = []
    if np_ and qidx is not None and len(qidx) == np_:
        planq = "SpecPlanQ"
        plan_src = ["// where the program's planning columns sit in qpos (the library is named by the program's hash, which covers them)",
                    "struct SpecPlanQ {",
                    "  static __device__ __forceinline__ constexpr int at(int k, const int *) {",
                    f"    constexpr int t[{np_}] = {{{', '.join(str(int(x)) for x in qidx)}}};",
                    "    return t[k];", "  }", "};"]
    src = ["namespace {"] + [s for _, _, s in specs] + plan_src + ["}  // namespace", "", 'extern "C" {', f"int mjpl_spec_pose_count(void) {{ return {n}; }}",
           "unsigned long long mjpl_spec_pose_hash(int k) {", "  switch (k) {"]
    src += [f"    case {k}: return PoseSpec{k}::kHash;" for k, _, _ in specs]
    src += ["    default: return 0ull;", "  }", "}"]
    # Launchers of the row kernels (mjpl_rows.h).  G: lanes per row (1, 4 or 8); return 0 = launched, -1 = refused with
    # nothing launched (no such projection / lane count / number of planning joints: the caller takes the interpreting
    # kernel), -2 = the launch failed (a HIP error the caller reports; hipGetLastError has cleared it).
    done = ["    default: return -1;", "  }", "  return hipGetLastError() == hipSuccess ? 0 : -2;", "}"]

    def cases(fmt):
        out = []
        for k, _, _ in specs:
            out += [f"    case {3 * k + i}: " + fmt.format(k=k, g=g) + " break;" for i, g in enumerate((1, 4, 8))]
        return out
    pick = ["  if (G != 1 && G != 4 && G != 8) return -1;", "  switch (3 * k + (G == 8 ? 2 : (G == 4 ? 1 : 0))) {"]
    src += ["int mjpl_spec_launch_pose_apply(int k, int G, hipStream_t st, unsigned grid, const int *pi, const double *pd, const double *Qold,",
            "                                const double *Q, int64_t N, int64_t per, double *Qout, uint8_t *ok, int32_t *iters, mjpl::PosePhase ph) {"] + pick
    src += cases("hipLaunchKernelGGL((mjpl::k_pose_apply_rows<PoseSpec{k}, {g}>), dim3(grid), dim3(mjpl::kPoseBlock), 0, st, pi, pd, Qold, Q, N, per, Qout, ok, iters, ph);")
    src += done
    src += ["int mjpl_spec_launch_gen_project(int k, int G, hipStream_t st, unsigned grid, int L, int nplan, int S, double eps, int par, const int *pi,",
            "                                 const double *pd, const int *qidx, const double *qbase, const uint8_t *isplan, const double *lo,",
            "                                 const double *hi, const double *Tgt, mjpl::RrtLanes ln, mjpl::RrtCand cd, int *ctr) {"]
    if np_:
        src += [f"  if (nplan != {np_}) return -1;  // (the library's program plans {np_} joints)",
                "  if (G == 16 || G == 64) {  // sixteen lanes per row, four rows per wave or one: the next step's first Newton pass beside this step's closing evaluation (mjpl_rows.h)",
                "    switch (2 * k + (G == 64 ? 1 : 0)) {"]
        src += [f"      case {2 * k + i}: hipLaunchKernelGGL((mjpl::k_rrt_gen_project_ahead<PoseSpec{k}, {np_}, {rows}, {planq}>), dim3(grid), dim3(mjpl::kPoseBlock), 0, st, L, S, eps, par, pi, pd, qidx, qbase, isplan, lo, hi, Tgt, ln, cd, ctr); break;"
                for k, _, _ in specs for i, rows in enumerate((4, 1))]
        src += ["      default: return -1;", "    }", "    return hipGetLastError() == hipSuccess ? 0 : -2;", "  }"] + pick
        src += cases("hipLaunchKernelGGL((mjpl::k_rrt_gen_project_rows<PoseSpec{k}, " + str(np_) + ", {g}, " + planq + ">), dim3(grid), dim3(mjpl::kPoseBlock), 0, st, L, S, eps, par, pi, pd, qidx, qbase, isplan, lo, hi, Tgt, ln, cd, ctr);")
        src += done
    else:
        src += ["  return -1;  // (the number of planning joints is not a constant of this library)", "}"]
    src += ["int mjpl_spec_launch_ik_solve(int k, int G, hipStream_t st, unsigned grid, const int *pi, const double *pd, const double *Q, int64_t N, int64_t per,",
            "                               double *Qout, uint8_t *ok, int32_t *iters, double *err, int max_restarts, unsigned long long restart_seed) {"] + pick
    src += cases("hipLaunchKernelGGL((mjpl::k_ik_solve_rows<PoseSpec{k}, {g}>), dim3(grid), dim3(mjpl::kPoseBlock), 0, st, pi, pd, Q, N, per, Qout, ok, iters, err, max_restarts, (uint64_t)restart_seed);")
    src += done + ["}"]
    return "\n".join(src) + "\n"


def _mbox_waves() -> int:
    """Waves per SIMD the kernels of a model with moving boxes are built for (MJPL_SPEC_MBOX_WAVES: A/B builds)."""
    return max(1, min(3, int(os.environ.get("MJPL_SPEC_MBOX_WAVES", "2"))))


_FUSED_F64 = """// the float64 checks of an edge launch through the pool (filter off or refused) around this model's generated check
int mjpl_spec_launch_fused_f64(hipStream_t st, int nwaves, size_t lds, FusedArgs a) {
  if (nwaves != kFusedF64Waves) return -1;
  auto kern = k_edges_fused_f64<ExactFull::kMaxs, %(wbox)s, false, kFusedF64Waves, true, ExactFull>;
  SPEC_GRANT(kern);
  return fused_launch(kern, nwaves, lds, a, st) == hipSuccess ? 0 : -2;
}"""


def translation_unit(spec: str, exact: str, key: int, info, generic_word: int = 0, pose: str | None = None, exact_full: str | None = None) -> str:
    """The source of a library: the kernels of mjpl_filter.h / mjpl_fused.h instantiated around `spec` / `exact`,
    and the projections of `pose` (generate_pose_section; None: a library without any)."""
    mbox = bool(info.mbox)
    if pose is None:
        pose = 'extern "C" int mjpl_spec_pose_count(void) { return 0; }\n'
    wbox_s = "true" if (info.wbox or generic_word or mbox) else "false"
    return _TU % dict(spec=spec, exact=exact, hash=key, maxs=info.maxs, pose=pose, exact_full=exact_full or "",
                      fused_f64=(_FUSED_F64 % dict(wbox=wbox_s)) if exact_full else "",
                      wbox="true" if (info.wbox or generic_word or mbox) else "false", mbox="true" if mbox else "false",
                      maxsd=32 if mbox else info.maxs,  # (the exact kernels of a model with moving boxes: the general build)
                      # a model with moving boxes keeps whole frames in its slot file and in the box queue's records: built for
                      # two waves per SIMD (256 VGPRs), eight waves per workgroup of the fused kernel -- at three the kernels
                      # spill 60 .. 90 registers and the fused kernel's LDS no longer fits: 0.56 against 0.42 ms on Franka-P
                      # with the ten pad boxes (profiles/r04_pads.json)
                      fwaves=(4 * _mbox_waves()) if mbox else 12, waves_define=f"#define MJPL_SPEC_WAVES {_mbox_waves()}\n" if mbox else "",
                      generic=generic_word)


def spec_path(hash_: int, generic: bool = False) -> str:
    return os.path.join(SPEC_DIR, f"libmjpl_spec{'g' if generic else ''}_{hash_:016x}.so")


def build(model, allowed_collision_bodies=(), qidx=None, qpos_base=None, filter_tol: float = 0.0, force: bool = False,
          keep_source: bool = True, extra_flags=(), output: str | None = None, generic: bool = False) -> str | None:
    """Generate and compile the specialised library of (model, planning set, tolerance).  Returns the
    path of the library, or None if the model cannot be specialised (immediate interpreter).
    generic: the ROBOT's scene-generic library instead (named by the robot hash): built from any scene that
    holds the robot, it serves every scene with it -- the static geoms come from the engine's scene table."""
    ip, fp, dp, info = dump_program(model, allowed_collision_bodies, qidx, qpos_base, filter_tol)
    if info.immediate or not info.filter_usable or (generic and not info.scene_ok):
        return None
    os.makedirs(SPEC_DIR, exist_ok=True)
    key = info.robot_hash if generic else info.hash
    target = output or spec_path(key, generic)  # (output, extra_flags: timing-only variants, tools/time_variants.sh)
    deps = [os.path.join(_build.CSRC, f) for f in _build.STAMPED_HEADERS] + [__file__, os.path.join(os.path.dirname(__file__), "fold.py")]
    if not force and os.path.exists(target) and all(os.path.getmtime(d) <= os.path.getmtime(target) for d in deps):
        return target
    nstage = 0
    if generic:
        pc = int(ip[H_OFF_BODYOPS])
        for _ in range(int(ip[H_NBODYOPS])):
            nj, ng = int(ip[pc + B_NJNT]), int(ip[pc + B_NGEOM])
            pc += B_SIZE + nj * J_SIZE + ng * (G_SIZE + MAX_SLOTS)
            nstage += ng
    src = translation_unit(generate(ip, fp, dp, info, generic=generic), generate_exact(ip, dp, info, generic=generic), key, info,
                           (SCENE_ROWS << 8 | nstage) if generic else 0,  # (kSceneRows, moving geoms)
                           pose=generate_pose_section(model, int(ip[H_NPLAN]), qidx=(None if qidx is None else [int(x) for x in qidx])),
                           # (an experiment, off by default: see generate_full_exact / mjpl_fused.h)
                           exact_full=generate_full_exact(ip, fp, dp, info) if (not generic and os.environ.get("MJPL_SPEC_F64") == "1") else None)
    # Source and library appear under their final names complete or not at all (os.replace): an engine created
    # while a rebuild is running finds the old library or the new one, never half a file -- a failed dlopen would
    # be remembered as "no library" for the life of that process -- and two builds of one hash cannot interleave.
    src_path = os.path.join(SPEC_DIR, f"spec{'g' if generic else ''}_{key:016x}.hip")
    tag = f".tmp{os.getpid()}_{threading.get_ident():x}"
    src_tmp = src_path[:-4] + tag + ".hip"
    lib_tmp = target + tag
    with open(src_tmp, "w") as f:
        f.write(src)
    # -fno-slp-vectorize: left alone, the SLP vectoriser pairs the scalar binary32 arithmetic of the generated
    # code into v_pk_* instructions -- which issue no faster than the two instructions they replace -- at the
    # price of ~700 v_mov to assemble the pairs and of spilled registers: 0.279 -> 0.259 ms per step
    cmd = [_build.hipcc(), *_build.hipcc_flags(), "-fno-slp-vectorize", "-Wno-unused-variable", "-Wno-unused-but-set-variable", f"-I{_build.CSRC}",
           *extra_flags, "-o", lib_tmp, src_tmp]
    try:
        subprocess.run(cmd, check=True)
        os.replace(lib_tmp, target)
        if keep_source:
            os.replace(src_tmp, src_path)
    finally:
        for leftover in (lib_tmp, src_tmp):
            if os.path.exists(leftover):
                os.remove(leftover)
    return target


def _main(argv=None) -> int:
    """python -m mjpl_amd.specialise MODEL.xml [--joints a,b,...] [--keyframe NAME] [--allowed bodyA:bodyB,...]
    Builds the specialised library for (model, planning joints, base configuration) -- a deployment step:
    needs hipcc, no GPU; engines created for that combination afterwards load it by program hash."""
    import argparse

    from .model import load_mjcf
    ap = argparse.ArgumentParser(prog="python -m mjpl_amd.specialise", description=_main.__doc__)
    ap.add_argument("mjcf", help="primitive-only MJCF file")
    ap.add_argument("--joints", default="", help="comma-separated planning joints (default: all)")
    ap.add_argument("--keyframe", default="", help="keyframe holding the non-planning joints (default: qpos0)")
    ap.add_argument("--allowed", default="", help="allowed collision body pairs, bodyA:bodyB,...")
    ap.add_argument("--tol", type=float, default=0.0, help="filter tolerance in metres (0: the model's default)")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--generic", action="store_true",
                    help="the robot's scene-generic library: static geoms (<= 32) come from a table, so obstacles may change "
                         "without a compiler (about 10 % slower than the scene's own library)")
    args = ap.parse_args(argv)
    model = load_mjcf(args.mjcf)
    names = [n for n in args.joints.split(",") if n]
    qidx = None
    if names:
        qidx = np.asarray([int(model.jnt_qposadr[model.joint(n).id]) for n in names], dtype=np.int32)
    base = model.keyframe(args.keyframe).qpos.copy() if args.keyframe else None
    allowed = tuple(tuple(p.split(":")) for p in args.allowed.split(",") if p)
    path = build(model, allowed, qidx, base, args.tol, force=args.force, generic=args.generic)
    print(path if path else "this model runs the general builds: nothing to specialise")
    return 0


if __name__ == "__main__":
    raise SystemExit(_main())
