"""ctypes binding of ``libmjpl_hip.so`` (include/mjpl_hip.h) -- the only compute backend.

There is no CPU path in this package: if the library is missing, cannot be loaded, or finds
no gfx950 device, construction raises.  (The CPU restatement under ``oracle/`` is test
infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np

from . import build as _build
from .model import Model

SOA, AOS = 0, 1
EDGE_INTERIOR_ONLY = 1

_I32P = C.POINTER(C.c_int32)
_F64P = C.POINTER(C.c_double)
_U8P = C.POINTER(C.c_uint8)


class MjplError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libmjpl_hip status {code}: {msg}")
        self.code = code


class _ModelDesc(C.Structure):
    _fields_ = [
        ("nq", C.c_int32), ("njnt", C.c_int32), ("nbody", C.c_int32), ("ngeom", C.c_int32),
        ("body_parentid", _I32P), ("body_weldid", _I32P), ("body_jntadr", _I32P),
        ("body_jntnum", _I32P), ("body_pos", _F64P), ("body_quat", _F64P),
        ("jnt_type", _I32P), ("jnt_qposadr", _I32P), ("jnt_axis", _F64P), ("jnt_pos", _F64P),
        ("qpos0", _F64P),
        ("geom_type", _I32P), ("geom_bodyid", _I32P), ("geom_contype", _I32P),
        ("geom_conaffinity", _I32P), ("geom_size", _F64P), ("geom_pos", _F64P),
        ("geom_quat", _F64P), ("geom_rbound", _F64P), ("geom_margin", _F64P),
    ]


class Info(C.Structure):
    _fields_ = [
        ("device", C.c_int32), ("nplan", C.c_int32), ("nmoving_geoms", C.c_int32),
        ("nstatic_geoms", C.c_int32), ("npairs", C.c_int32), ("npairs_world", C.c_int32),
        ("nslots", C.c_int32), ("nsaves", C.c_int32), ("lds_bytes_configs", C.c_int32),
        ("lds_bytes_edges", C.c_int32), ("block_threads", C.c_int32), ("compute_units", C.c_int32),
        ("filter_enabled", C.c_int32), ("filter_tol", C.c_float),
        ("lds_bytes_filter", C.c_int32), ("filter_block_threads", C.c_int32),
        ("arch", C.c_char * 32),
        ("filter_max_coord", C.c_float), ("filter_err_a", C.c_float), ("filter_err_b", C.c_float),
        ("filter_poisoned_geoms", C.c_int32), ("filter_interpreter", C.c_int32),
        ("persistent_kernels", C.c_int32), ("fused_tail", C.c_int32), ("fused_edges", C.c_int32), ("fused_waves", C.c_int32),
    ]

    def as_dict(self) -> dict:
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["arch"] = self.arch.decode()
        return d


class PoseDesc(C.Structure):
    _fields_ = [
        ("site_body", C.c_int32), ("site_pos", C.c_double * 3), ("site_quat", C.c_double * 4),
        ("c_quat", C.c_double * 4), ("c_pos", C.c_double * 3), ("lo", C.c_double * 6),
        ("hi", C.c_double * 6), ("tolerance", C.c_double), ("q_step", C.c_double),
        ("jnt_range", _F64P), ("max_iters", C.c_int32),
    ]


class IKDesc(C.Structure):
    _fields_ = [
        ("site_body", C.c_int32), ("site_pos", C.c_double * 3), ("site_quat", C.c_double * 4),
        ("target_pos", C.c_double * 3), ("target_quat", C.c_double * 4), ("pos_tolerance", C.c_double),
        ("ori_tolerance", C.c_double), ("iterations", C.c_int32), ("damping", C.c_double),
        ("lm_damping", C.c_double), ("max_step", C.c_double), ("jnt_range", _F64P), ("movable", _U8P),
        ("restarts", C.c_int32), ("restart_seed", C.c_uint64),
    ]


class RrtDesc(C.Structure):
    _fields_ = [
        ("lanes", C.c_int32), ("capacity", C.c_int64), ("epsilon", C.c_double), ("interval_step", C.c_double),
        ("goal_bias", C.c_double), ("seed", C.c_uint64), ("lo", _F64P), ("hi", _F64P), ("pose", C.c_void_p),
        ("max_new_per_round", C.c_int64), ("max_steps_per_round", C.c_int32),
    ]


class RrtRoundInfo(C.Structure):
    _fields_ = [
        ("round", C.c_int32), ("new_nodes", C.c_int32 * 2), ("nodes", C.c_int32 * 2), ("connected", C.c_int32),
        ("conn_start", C.c_int32), ("conn_goal", C.c_int32), ("conn_rank", C.c_int32), ("stop_requested", C.c_int32),
    ]


# every symbol include/mjpl_hip.h declares: (restype, argtypes)
_VP = C.c_void_p
ABI = {
    "mjpl_create": (C.c_int, [C.POINTER(_ModelDesc), _I32P, C.c_int32, C.c_int32, C.POINTER(_VP)]),
    "mjpl_destroy": (None, [_VP]),
    "mjpl_set_planning": (C.c_int, [_VP, _I32P, C.c_int32, _F64P]),
    "mjpl_get_info": (C.c_int, [_VP, C.POINTER(Info)]),
    "mjpl_set_filter": (C.c_int, [_VP, C.c_int32, C.c_double]),
    "mjpl_filter_last_undecided": (C.c_int64, [_VP]),
    "mjpl_filter_undecided_pairs": (C.c_int64, [_VP, _I32P, _I32P, _I32P, _I32P, C.c_int64]),
    "mjpl_filter_last_interior_edges": (C.c_int64, [_VP]),
    "mjpl_filter_last_items": (C.c_int64, [_VP]),
    "mjpl_filter_last_certified": (C.c_int64, [_VP]),
    "mjpl_check_configs": (C.c_int, [_VP, _F64P, C.c_int64, C.c_int32, _U8P]),
    "mjpl_check_edges": (C.c_int, [_VP, _F64P, _F64P, C.c_int64, C.c_double, C.c_int32, C.c_int32, _U8P,
                                   _I32P]),
    "mjpl_fk": (C.c_int, [_VP, _F64P, C.c_int64, C.c_int32, _F64P, _F64P, _F64P, _F64P]),
    "mjpl_check_configs_dev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, _VP]),
    "mjpl_check_edges_dev": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_double, C.c_int32, C.c_int32, _VP,
                                       _VP]),
    "mjpl_take_status": (C.c_int, [_VP, _I32P]),
    "mjpl_check_configs_bits_dev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, _VP]),
    "mjpl_nearest_dev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, _VP, C.c_int64, _VP, _VP]),
    "mjpl_nearest_range_dev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, C.c_int64, _VP, C.c_int64, _VP, _VP, _VP, _VP]),
    "mjpl_nearest_last_screen": (C.c_int32, [_VP]),
    "mjpl_dev_alloc": (C.c_int, [_VP, C.c_size_t, C.POINTER(_VP)]),
    "mjpl_dev_free": (C.c_int, [_VP, _VP]),
    "mjpl_h2d": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "mjpl_d2h": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "mjpl_sync": (C.c_int, [_VP]),
    "mjpl_stream": (_VP, [_VP]),
    "mjpl_time_edges_dev": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_double, C.c_int32, _VP, C.c_int32,
                                      C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mjpl_time_edges_stages_dev": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_double, C.c_int32, _VP, C.c_int32,
                                             C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), _I32P]),
    "mjpl_time_configs_dev": (C.c_int, [_VP, _VP, C.c_int64, C.c_int32, _VP, C.c_int32,
                                        C.POINTER(C.c_float)]),
    "mjpl_pose_create": (C.c_int, [_VP, C.POINTER(PoseDesc), C.POINTER(_VP)]),
    "mjpl_pose_destroy": (None, [_VP]),
    "mjpl_pose_spec_loaded": (C.c_int, [_VP]),
    "mjpl_pose_chain_dump": (C.c_int, [C.POINTER(_ModelDesc), C.c_int32, _I32P, _I32P, _F64P, _I32P, C.POINTER(C.c_uint64)]),
    "mjpl_pose_set_q_step": (C.c_int, [_VP, C.c_double]),
    "mjpl_pose_apply": (C.c_int, [_VP, _F64P, _F64P, C.c_int64, _F64P, _U8P, _I32P]),
    "mjpl_pose_valid": (C.c_int, [_VP, _F64P, C.c_int64, _U8P, _F64P, _F64P]),
    "mjpl_pose_apply_dev": (C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, _VP, _VP]),
    "mjpl_pose_valid_dev": (C.c_int, [_VP, _VP, C.c_int64, _VP, _VP, _VP]),
    "mjpl_ik_solve": (C.c_int, [_VP, C.POINTER(IKDesc), _F64P, C.c_int64, _F64P, _U8P, _I32P, _F64P]),
    "mjpl_ik_solve_dev": (C.c_int, [_VP, C.POINTER(IKDesc), _VP, C.c_int64, _VP, _VP, _VP, _VP]),
    "mjpl_set_option": (C.c_int, [_VP, C.c_char_p, C.c_double]),
    "mjpl_get_option": (C.c_int, [_VP, C.c_char_p, C.POINTER(C.c_double)]),
    "mjpl_option_count": (C.c_int32, []),
    "mjpl_option_name": (C.c_char_p, [C.c_int32, C.POINTER(C.c_int32)]),
    "mjpl_set_spec_dir": (C.c_int, [C.c_char_p]),
    "mjpl_rrt_create": (C.c_int, [_VP, C.POINTER(RrtDesc), C.POINTER(_VP)]),
    "mjpl_rrt_destroy": (None, [_VP]),
    "mjpl_rrt_reset": (C.c_int, [_VP, _F64P, _F64P, C.c_int32, C.c_uint64]),
    "mjpl_rrt_round": (C.c_int, [_VP, C.c_int32, C.POINTER(RrtRoundInfo)]),
    "mjpl_rrt_set_world": (C.c_int, [_VP, C.c_int32, C.c_int32]),
    "mjpl_rrt_round_begin": (C.c_int, [_VP, C.c_int32, _I32P]),
    "mjpl_rrt_round_slabs": (C.c_int, [_VP, C.c_int32, C.POINTER(_VP), C.POINTER(_VP)]),
    "mjpl_rrt_round_finish": (C.c_int, [_VP, _I32P, C.POINTER(_VP), C.POINTER(_VP), _I32P, C.POINTER(RrtRoundInfo)]),
    "mjpl_rrt_path": (C.c_int, [_VP, _F64P, C.c_int32, _I32P]),
    "mjpl_rrt_get_tree": (C.c_int, [_VP, C.c_int32, _F64P, _I32P, C.c_int64, C.POINTER(C.c_int64)]),
    "mjpl_rrt_get_lanes": (C.c_int, [_VP, _F64P, _U8P]),
    "mjpl_comm_unique_id": (C.c_int, [_VP]),
    "mjpl_comm_init": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32]),
    "mjpl_comm_destroy": (C.c_int, [_VP]),
    "mjpl_allgather_dev": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "mjpl_program_dump": (C.c_int, None),   # bound in mjpl_amd/specialise.py
    "mjpl_spec_probe": (C.c_int, [C.c_uint64, C.c_int32]),
    "mjpl_spec_loaded": (C.c_int, [_VP]),
    "mjpl_set_spec": (C.c_int, [_VP, C.c_int32]),
    "mjpl_device_count": (C.c_int, []),
    "mjpl_last_error": (C.c_char_p, []),
    "mjpl_version": (C.c_char_p, []),
}

_libs: dict[str, C.CDLL] = {}


def load_library(path: str | None = None) -> C.CDLL:
    """dlopen libmjpl_hip.so and bind every declared entry point (fails loudly)."""
    path = path or os.environ.get("MJPL_HIP_LIB") or _build.LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(mjpl_amd has no CPU fallback)")
    lib = C.CDLL(path)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        if args is not None:
            fn.argtypes = args
    _libs[path] = lib
    return lib


# Options every Engine made from now on starts with (name -> value, mjpl_set_option): what the tests and tools switch
# kernels with.  `options(...)` is the scoped form.  The LIBRARY reads no switch from the environment; this module does,
# for the convenience of shells and of the test-suite's older cases: a variable MJPL_<NAME> naming a writable option is
# applied -- through the ABI -- to every engine as it is made, MJPL_SPEC=0 as mjpl_set_spec(0), MJPL_SPEC_DIR as
# mjpl_set_spec_dir.
DEFAULT_OPTIONS: dict = {}


class options:
    """with engine.options(filter=0, fused_single=0): ...  -- engines made inside start with these options."""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = DEFAULT_OPTIONS.get(k, None)
            DEFAULT_OPTIONS[k] = v
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                DEFAULT_OPTIONS.pop(k, None)
            else:
                DEFAULT_OPTIONS[k] = v


def option_names(lib=None, writable_only=False) -> list[str]:
    lib = lib or load_library()
    out = []
    for i in range(lib.mjpl_option_count()):
        w = C.c_int32(0)
        name = lib.mjpl_option_name(i, C.byref(w))
        if name and (w.value or not writable_only):
            out.append(name.decode())
    return out


def set_spec_dir(path: str | None):
    """Per-model libraries are looked for there (process-wide; None: beside the library again)."""
    lib = load_library()
    lib.mjpl_set_spec_dir(path.encode() if path else None)


def device_count() -> int:
    return int(load_library().mjpl_device_count())


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class DeviceBuffer:
    """A hipMalloc'ed block owned by an :class:`Engine`."""

    def __init__(self, eng: "Engine", nbytes: int):
        self.eng, self.nbytes = eng, int(nbytes)
        p = _VP()
        eng._ok(eng.lib.mjpl_dev_alloc(eng.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self._keep = arr  # async copy: keep the source alive until the next sync
        self.eng._ok(self.eng.lib.mjpl_h2d(self.eng.h, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype, count: int) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.eng._ok(self.eng.lib.mjpl_d2h(self.eng.h, out.ctypes.data, self.ptr, out.nbytes))
        self.eng.sync()
        return out

    def free(self):
        if self.ptr:
            self.eng.lib.mjpl_dev_free(self.eng.h, self.ptr)
            self.ptr = None


class Engine:
    """One compiled model on one MI355X (one HIP stream).  Not thread-safe, like the
    reference's per-constraint MjData (collision_constraint.py:23)."""

    def __init__(self, model: Model, allowed_collision_bodies=(), device: int = 0,
                 lib_path: str | None = None, options: dict | None = None):
        self.lib = load_library(lib_path)
        self.model = model
        self.h = None
        d = _ModelDesc()
        d.nq, d.njnt, d.nbody, d.ngeom = model.nq, model.njnt, model.nbody, model.ngeom
        keep = []
        for name, typ in _ModelDesc._fields_[4:]:
            arr = getattr(model, name)
            arr = _i32(arr) if typ is _I32P else _f64(arr)
            keep.append(arr)
            setattr(d, name, arr.ctypes.data_as(typ))
        # body names -> ids (unknown names raise KeyError from model.body, as mujoco does)
        pairs = _i32([(model.body(a).id, model.body(b).id) for a, b in allowed_collision_bodies]
                     ).reshape(-1, 2)
        h = _VP()
        rc = self.lib.mjpl_create(C.byref(d), pairs.ctypes.data_as(_I32P), len(pairs), device, C.byref(h))
        self._ok(rc)
        self.h = h
        self.nplan = model.nq
        self._projectors: "weakref.WeakSet[PoseProjector]" = weakref.WeakSet()
        self._apply_start_options(options)

    def _apply_start_options(self, explicit):
        """DEFAULT_OPTIONS, the MJPL_<NAME> variables of this process, then `explicit` -- all through mjpl_set_option; one
        compile at the end if anything was set (options that shape the compiled model take effect then)."""
        todo = dict(DEFAULT_OPTIONS)
        spec_off = None
        for name in option_names(self.lib, writable_only=True):
            v = os.environ.get("MJPL_" + name.upper())
            if v is not None and name not in todo:
                try:
                    todo[name] = float(v)
                except ValueError:
                    pass
        if os.environ.get("MJPL_SPEC") is not None:
            spec_off = os.environ["MJPL_SPEC"].strip() in ("0", "")
        if "MJPL_SPEC_DIR" in os.environ or getattr(Engine, "_spec_dir_from_env", None):
            d = os.environ.get("MJPL_SPEC_DIR")
            self.lib.mjpl_set_spec_dir(d.encode() if d else None)
            Engine._spec_dir_from_env = d
        todo.update(explicit or {})
        if "spec" in todo:
            spec_off = not todo.pop("spec")
        for k, v in todo.items():
            self._ok(self.lib.mjpl_set_option(self.h, k.encode(), float(v)))
        if todo or spec_off is not None or getattr(Engine, "_spec_dir_from_env", None) is not None:
            self._ok(self.lib.mjpl_set_spec(self.h, 0 if spec_off else 1))  # (compiles again, with the options as they stand)

    # -- plumbing
    def _ok(self, rc: int):
        if rc != 0:
            raise MjplError(rc, self.lib.mjpl_last_error().decode())

    def close(self):
        if self.h:
            # mjpl_pose handles hold device memory of their own and a pointer to this engine:
            # destroy them first so that closing the engine before its projectors leaks nothing
            for p in list(getattr(self, "_projectors", ())):
                p.close()
            self.lib.mjpl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        self._ok(self.lib.mjpl_sync(self.h))

    def info(self) -> dict:
        i = Info()
        self._ok(self.lib.mjpl_get_info(self.h, C.byref(i)))
        return i.as_dict()

    def spec_loaded(self) -> bool:
        """True if this engine's filter kernels are the model's own specialised ones
        (mjpl_amd/specialise.py), not the interpreter."""
        return bool(self.lib.mjpl_spec_loaded(self.h))

    def set_spec(self, enable):
        """False / 0: run the interpreting kernels even if this program has a specialised library; True / 1:
        look the libraries up (the program's own, else the robot's scene-generic one); 2: the generic one only."""
        self._ok(self.lib.mjpl_set_spec(self.h, int(enable)))

    def spec_kind(self) -> int:
        """0 interpreter, 1 the program's own specialised library, 2 the robot's scene-generic one."""
        return int(self.lib.mjpl_spec_loaded(self.h))

    def set_filter(self, enable: bool, tol: float = 1e-4):
        """Float32 filter in front of the exact kernels (verdicts are always the exact ones)."""
        self._ok(self.lib.mjpl_set_filter(self.h, 1 if enable else 0, float(tol)))

    def last_items(self) -> int:
        return int(self.lib.mjpl_filter_last_items(self.h))

    def last_certified(self) -> int:
        """Surviving edges of the last launch whose waypoint checks the fused kernel's edge certificate spared."""
        return int(self.lib.mjpl_filter_last_certified(self.h))

    def last_interior_edges(self) -> int:
        return int(self.lib.mjpl_filter_last_interior_edges(self.h))

    def last_undecided(self) -> int:
        return int(self.lib.mjpl_filter_last_undecided(self.h))

    def undecided_pairs(self, cap: int = 1 << 20):
        """Diagnostic: the pairs the last filter launch handed to the exact pair kernel
        -> (total, edge [m], check index [m], geom a [m], geom b [m]), m = min(total, cap)."""
        arrs = [np.zeros(cap, np.int32) for _ in range(4)]
        n = int(self.lib.mjpl_filter_undecided_pairs(self.h, *[a.ctypes.data_as(_I32P) for a in arrs], cap))
        if n < 0:
            raise RuntimeError("mjpl_filter_undecided_pairs failed")
        m = min(n, cap)
        return (n, *[a[:m] for a in arrs])

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def set_planning(self, qidx, qpos_base):
        qidx = _i32(qidx)
        base = _f64(qpos_base)
        if base.shape != (self.model.nq,):
            raise ValueError(f"qpos_base must have {self.model.nq} entries")
        self._ok(self.lib.mjpl_set_planning(self.h, qidx.ctypes.data_as(_I32P), len(qidx),
                                            base.ctypes.data_as(_F64P)))
        self.nplan = len(qidx)

    def _batch(self, Q, layout):
        Q = _f64(Q)
        want_cols = self.nplan
        if Q.ndim != 2 or (Q.shape[0] if layout == SOA else Q.shape[1]) != want_cols:
            raise ValueError(f"batch must be [{'nplan, N' if layout == SOA else 'N, nplan'}] "
                             f"with nplan={want_cols}, got {Q.shape}")
        return Q, (Q.shape[1] if layout == SOA else Q.shape[0])

    # -- host-buffer entry points
    def check_configs(self, Q, layout=AOS) -> np.ndarray:
        Q, n = self._batch(Q, layout)
        out = np.zeros(n, np.uint8)
        self._ok(self.lib.mjpl_check_configs(self.h, Q.ctypes.data_as(_F64P), n, layout,
                                             out.ctypes.data_as(_U8P)))
        return out

    def check_edges(self, QA, QB, step_dist, layout=AOS, first_bad=False, interior_only=False):
        QA, n = self._batch(QA, layout)
        QB, n2 = self._batch(QB, layout)
        if n != n2:
            raise ValueError("QA and QB must hold the same number of edges")
        out = np.zeros(n, np.uint8)
        fb = np.zeros(n, np.int32) if first_bad else None
        self._ok(self.lib.mjpl_check_edges(
            self.h, QA.ctypes.data_as(_F64P), QB.ctypes.data_as(_F64P), n, float(step_dist), layout,
            EDGE_INTERIOR_ONLY if interior_only else 0,
            out.ctypes.data_as(_U8P), fb.ctypes.data_as(_I32P) if first_bad else None))
        return (out, fb) if first_bad else out

    def fk(self, Q, layout=AOS) -> dict:
        Q, n = self._batch(Q, layout)
        m = self.model
        xpos, xquat = np.zeros((n, m.nbody, 3)), np.zeros((n, m.nbody, 4))
        gx, gm = np.zeros((n, m.ngeom, 3)), np.zeros((n, m.ngeom, 9))
        self._ok(self.lib.mjpl_fk(self.h, Q.ctypes.data_as(_F64P), n, layout,
                                  xpos.ctypes.data_as(_F64P), xquat.ctypes.data_as(_F64P),
                                  gx.ctypes.data_as(_F64P), gm.ctypes.data_as(_F64P)))
        return dict(xpos=xpos, xquat=xquat, geom_xpos=gx, geom_xmat=gm)

    # -- device-resident entry points (pointers are DeviceBuffer.ptr or foreign device pointers)
    def check_configs_dev(self, dQ, n, layout, dvalid):
        self._ok(self.lib.mjpl_check_configs_dev(self.h, dQ, n, layout, dvalid))

    def check_configs_bits_dev(self, dQ, n, layout, dbits):
        self._ok(self.lib.mjpl_check_configs_bits_dev(self.h, dQ, n, layout, dbits))

    def check_edges_dev(self, dQA, dQB, n, step_dist, layout, dvalid, dfirst_bad=None, flags=0):
        self._ok(self.lib.mjpl_check_edges_dev(self.h, dQA, dQB, n, float(step_dist), layout, flags,
                                               dvalid, dfirst_bad))

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        """RCCL communicator for the frontier planner's exchange (one per engine)."""
        if len(unique_id) != 128:
            raise ValueError("a ncclUniqueId is 128 bytes")
        self._ok(self.lib.mjpl_comm_init(self.h, C.create_string_buffer(unique_id, 128), rank, world))

    def comm_destroy(self):
        self._ok(self.lib.mjpl_comm_destroy(self.h))

    def allgather_dev(self, dsend, drecv, nbytes_per_rank: int):
        self._ok(self.lib.mjpl_allgather_dev(self.h, dsend, drecv, nbytes_per_rank))

    def take_status(self) -> int:
        """Sticky status of the device-pointer edge launches since the last take (0 or
        MJPL_E_NONFINITE = -7); synchronises."""
        st = C.c_int32(0)
        self._ok(self.lib.mjpl_take_status(self.h, C.byref(st)))
        return int(st.value)

    def nearest_dev(self, dnodes, n, cap, dqueries, m, dout_idx, dout_d2=None):
        self._ok(self.lib.mjpl_nearest_dev(self.h, dnodes, n, cap, dqueries, m, dout_idx, dout_d2))

    def nearest_range_dev(self, dnodes, n0, n, cap, dqueries, m, dout_idx, dout_d2=None, dprev_idx=None, dprev_d2=None):
        """nearest_dev over nodes [n0, n) behind the answer (dprev_idx, dprev_d2) for the nodes below n0."""
        self._ok(self.lib.mjpl_nearest_range_dev(self.h, dnodes, n0, n, cap, dqueries, m, dout_idx, dout_d2, dprev_idx, dprev_d2))

    def set_option(self, name: str, value) -> None:
        """A switch of the library by name (include/mjpl_hip.h: mjpl_set_option; the table is in tools/README.md)."""
        self._ok(self.lib.mjpl_set_option(self.h, name.encode(), float(value)))

    def get_option(self, name: str) -> float:
        v = C.c_double(0.0)
        self._ok(self.lib.mjpl_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def nearest_last_screen(self) -> int:
        """0: plain float64 scan, 1: binary32 screen, 2: matrix-core (binary16) screen -- of the last nearest_dev."""
        return int(self.lib.mjpl_nearest_last_screen(self.h))

    def time_edges_dev(self, dQA, dQB, n, step_dist, layout, dvalid, iters, first_kernel=False):
        """Per-call durations (ms) of `iters` edge launches; with first_kernel=True also the
        durations of the first (dominant) kernel of each call."""
        ms = np.zeros(iters, np.float32)
        ms1 = np.zeros(iters, np.float32)
        fp = C.POINTER(C.c_float)
        self._ok(self.lib.mjpl_time_edges_dev(self.h, dQA, dQB, n, float(step_dist), layout, dvalid,
                                              iters, ms.ctypes.data_as(fp), ms1.ctypes.data_as(fp)))
        return (ms, ms1) if first_kernel else ms

    STAGES = ("k_filter_endpoints", "k_filter_items", "k_filter_edges", "k_patch_pairs", "k_check_edges")
    # with info()["fused_tail"] the fourth stage is k_tail (walking role + pair re-check + exact edge role in
    # one launch) and stages three and five are empty; with info()["persistent_kernels"] the first two
    # kernels are k_filter_endpoints_pw / k_filter_items_pw

    def time_edges_stages_dev(self, dQA, dQB, n, step_dist, layout, dvalid, iters, sample_every=4):
        """`iters` back-to-back edge launches -> (mean ms per launch, {stage kernel: mean ms},
        sampled launches); the stages are bracketed on every `sample_every`-th launch only."""
        mean = C.c_float(0)
        st = (C.c_float * len(self.STAGES))()
        ns = C.c_int32(0)
        self._ok(self.lib.mjpl_time_edges_stages_dev(self.h, dQA, dQB, n, float(step_dist), layout, dvalid, iters,
                                                     sample_every, C.byref(mean), st, C.byref(ns)))
        return float(mean.value), {k: float(st[i]) for i, k in enumerate(self.STAGES)}, int(ns.value)

    def ik_solve(self, site: str, target_pos, target_quat, Q, movable, pos_tolerance=1e-3,
                 ori_tolerance=1e-3, iterations=500, damping=0.0, lm_damping=-1.0, max_step=0.0,
                 restarts=0, restart_seed=0):
        """Batched damped-least-squares IK: rows of Q [N, nq] are start configurations.
        -> (Q_out [N, nq], ok bool[N], iters int32[N], err [N, 2])"""
        model = self.model
        sid = model.site(site).id
        d = IKDesc()
        d.site_body = int(model.site_bodyid[sid])
        d.site_pos[:] = [float(x) for x in model.site_pos[sid]]
        d.site_quat[:] = [float(x) for x in model.site_quat[sid]]
        d.target_pos[:] = [float(x) for x in target_pos]
        d.target_quat[:] = [float(x) for x in target_quat]
        d.pos_tolerance, d.ori_tolerance, d.iterations = float(pos_tolerance), float(ori_tolerance), int(iterations)
        d.damping, d.lm_damping, d.max_step = float(damping), float(lm_damping), float(max_step)
        d.restarts, d.restart_seed = int(restarts), int(restart_seed) & (2**64 - 1)
        rng = _f64(model.jnt_range).reshape(-1)
        mv = np.ascontiguousarray(movable, dtype=np.uint8)
        if mv.shape != (model.njnt,):
            raise ValueError("`movable` must have one entry per joint")
        d.jnt_range = rng.ctypes.data_as(_F64P)
        d.movable = mv.ctypes.data_as(_U8P)
        Q = _f64(Q)
        if Q.ndim != 2 or Q.shape[1] != model.nq:
            raise ValueError(f"expected rows of {model.nq} qpos values, got shape {Q.shape}")
        n = len(Q)
        out = np.empty_like(Q)
        ok = np.zeros(n, np.uint8)
        iters = np.zeros(n, np.int32)
        err = np.zeros((n, 2))
        self._ok(self.lib.mjpl_ik_solve(self.h, C.byref(d), Q.ctypes.data_as(_F64P), n, out.ctypes.data_as(_F64P),
                                        ok.ctypes.data_as(_U8P), iters.ctypes.data_as(_I32P),
                                        err.ctypes.data_as(_F64P)))
        return out, ok.astype(bool), iters, err

    def time_configs_dev(self, dQ, n, layout, dvalid, iters) -> np.ndarray:
        ms = np.zeros(iters, np.float32)
        self._ok(self.lib.mjpl_time_configs_dev(self.h, dQ, n, layout, dvalid, iters,
                                                ms.ctypes.data_as(C.POINTER(C.c_float))))
        return ms


class PoseProjector:
    """``mjpl_pose`` handle: batched PoseConstraint.valid_config / apply / site pose for one
    (engine, site, constraint frame).  Rows are FULL qpos vectors [N, nq]."""

    def __init__(self, eng: Engine, site: str, c_quat, c_pos, bounds, tolerance: float, q_step: float,
                 max_iters: int = 1000):
        self.eng = eng
        model = eng.model
        sid = model.site(site).id
        d = PoseDesc()
        d.site_body = int(model.site_bodyid[sid])
        d.site_pos[:] = [float(x) for x in model.site_pos[sid]]
        d.site_quat[:] = [float(x) for x in model.site_quat[sid]]
        d.c_quat[:] = [float(x) for x in c_quat]
        d.c_pos[:] = [float(x) for x in c_pos]
        b = np.asarray(bounds, dtype=np.float64).reshape(6, 2)
        d.lo[:] = [float(x) for x in b[:, 0]]
        d.hi[:] = [float(x) for x in b[:, 1]]
        d.tolerance, d.q_step, d.max_iters = float(tolerance), float(q_step), int(max_iters)
        rng = _f64(model.jnt_range).reshape(-1)
        d.jnt_range = rng.ctypes.data_as(_F64P)
        self.h = None
        h = _VP()
        eng._ok(eng.lib.mjpl_pose_create(eng.h, C.byref(d), C.byref(h)))
        self.h = h
        self.nq = model.nq
        eng._projectors.add(self)

    def spec_loaded(self) -> bool:
        """True: a generated projection of the engine's per-model library serves this handle (same results)."""
        return bool(self.eng.lib.mjpl_pose_spec_loaded(self.h))

    def close(self):
        if self.h and self.eng.h:
            self.eng.lib.mjpl_pose_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_q_step(self, q_step: float):
        self.eng._ok(self.eng.lib.mjpl_pose_set_q_step(self.h, float(q_step)))

    def _rows(self, Q):
        Q = _f64(Q)
        if Q.ndim != 2 or Q.shape[1] != self.nq:
            raise ValueError(f"expected rows of {self.nq} qpos values, got shape {Q.shape}")
        return Q

    def apply(self, Q_old, Q):
        """-> (Q_projected [N, nq], ok bool[N], iters int32[N])"""
        Q_old, Q = self._rows(Q_old), self._rows(Q)
        if Q_old.shape != Q.shape:
            raise ValueError("Q_old and Q must have the same shape")
        n = len(Q)
        out = np.empty_like(Q)
        ok = np.zeros(n, np.uint8)
        iters = np.zeros(n, np.int32)
        self.eng._ok(self.eng.lib.mjpl_pose_apply(
            self.h, Q_old.ctypes.data_as(_F64P), Q.ctypes.data_as(_F64P), n, out.ctypes.data_as(_F64P),
            ok.ctypes.data_as(_U8P), iters.ctypes.data_as(_I32P)))
        return out, ok.astype(bool), iters

    def valid(self, Q, poses: bool = False):
        """-> valid bool[N]  (and, with poses=True, site xpos [N, 3], xmat [N, 3, 3])"""
        Q = self._rows(Q)
        n = len(Q)
        valid = np.zeros(n, np.uint8)
        xpos = np.empty((n, 3)) if poses else None
        xmat = np.empty((n, 9)) if poses else None
        self.eng._ok(self.eng.lib.mjpl_pose_valid(
            self.h, Q.ctypes.data_as(_F64P), n, valid.ctypes.data_as(_U8P),
            xpos.ctypes.data_as(_F64P) if poses else None, xmat.ctypes.data_as(_F64P) if poses else None))
        if poses:
            return valid.astype(bool), xpos, xmat.reshape(n, 3, 3)
        return valid.astype(bool)

    def apply_dev(self, dQ_old, dQ, n, dQ_out, dok, diters=None):
        self.eng._ok(self.eng.lib.mjpl_pose_apply_dev(self.h, dQ_old, dQ, n, dQ_out, dok, diters))


class DeviceRRT:
    """``mjpl_rrt`` handle: both trees of a frontier bi-RRT resident in HBM (include/mjpl_hip.h).
    Batches are over the engine's planning columns, so call ``Engine.set_planning`` first."""

    def __init__(self, eng: Engine, lanes: int, capacity: int, lo, hi, epsilon=0.05, interval_step=None,
                 goal_bias=0.05, seed=0, pose: PoseProjector | None = None, max_new_per_round=0, max_steps_per_round=0):
        self.eng, self.nplan, self.lanes = eng, eng.nplan, int(lanes)
        lo, hi = _f64(lo), _f64(hi)
        if lo.shape != (self.nplan,) or hi.shape != (self.nplan,):
            raise ValueError("lo / hi must have one entry per planning column")
        d = RrtDesc()
        d.lanes, d.capacity, d.epsilon = int(lanes), int(capacity), float(epsilon)
        d.interval_step = float(interval_step) if interval_step else 0.0
        d.goal_bias, d.seed = float(goal_bias), int(seed) & (2**64 - 1)
        d.lo, d.hi = lo.ctypes.data_as(_F64P), hi.ctypes.data_as(_F64P)
        d.pose = pose.h if pose is not None else None
        d.max_new_per_round = int(max_new_per_round)
        d.max_steps_per_round = int(max_steps_per_round)
        self._pose = pose  # keep the handle alive
        self.h = None
        h = _VP()
        eng._ok(eng.lib.mjpl_rrt_create(eng.h, C.byref(d), C.byref(h)))
        self.h = h
        eng._projectors.add(self)  # closed with (before) its engine

    def close(self):
        if self.h and self.eng.h:
            self.eng.lib.mjpl_rrt_destroy(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, q_init, q_goals, seed: int):
        q_init, q_goals = _f64(q_init), _f64(np.atleast_2d(q_goals))
        if q_init.shape != (self.nplan,) or q_goals.shape[1] != self.nplan:
            raise ValueError("configurations must hold the planning columns")
        self.eng._ok(self.eng.lib.mjpl_rrt_reset(self.h, q_init.ctypes.data_as(_F64P), q_goals.ctypes.data_as(_F64P),
                                                 len(q_goals), int(seed) & (2**64 - 1)))

    def round(self, request_stop: bool = False) -> RrtRoundInfo:
        info = RrtRoundInfo()
        self.eng._ok(self.eng.lib.mjpl_rrt_round(self.h, 1 if request_stop else 0, C.byref(info)))
        return info

    # -- the round split at its exchange step (a launcher that moves the slabs itself)
    def set_world(self, rank: int, world: int):
        self.eng._ok(self.eng.lib.mjpl_rrt_set_world(self.h, int(rank), int(world)))

    def round_begin(self, request_stop: bool = False) -> np.ndarray:
        """-> this rank's exchange header, int32[8] (include/mjpl_hip.h)"""
        head = np.zeros(8, np.int32)
        self.eng._ok(self.eng.lib.mjpl_rrt_round_begin(self.h, 1 if request_stop else 0, head.ctypes.data_as(_I32P)))
        return head

    def round_slabs(self, which: int):
        """-> (device pointer of the rows, of the parents) of this rank's new nodes of pass `which`"""
        rows, par = _VP(), _VP()
        self.eng._ok(self.eng.lib.mjpl_rrt_round_slabs(self.h, int(which), C.byref(rows), C.byref(par)))
        return rows.value, par.value

    def round_finish(self, heads, drows_all, dparents_all, stride_rows) -> RrtRoundInfo:
        heads = np.ascontiguousarray(heads, dtype=np.int32)
        rows = (_VP * 2)(*[p or None for p in drows_all])
        pars = (_VP * 2)(*[p or None for p in dparents_all])
        stride = np.ascontiguousarray(stride_rows, dtype=np.int32)
        info = RrtRoundInfo()
        self.eng._ok(self.eng.lib.mjpl_rrt_round_finish(self.h, heads.ctypes.data_as(_I32P), rows, pars,
                                                        stride.ctypes.data_as(_I32P), C.byref(info)))
        return info

    def path(self, maxlen: int = 65536) -> np.ndarray:
        out = np.empty((maxlen, self.nplan))
        n = C.c_int32(0)
        self.eng._ok(self.eng.lib.mjpl_rrt_path(self.h, out.ctypes.data_as(_F64P), maxlen, C.byref(n)))
        return out[: n.value].copy()

    def tree(self, t: int):
        """-> (Q [n, nplan], parent int32[n]) of tree t (0 = start, 1 = goal)"""
        n = C.c_int64(0)
        self.eng._ok(self.eng.lib.mjpl_rrt_get_tree(self.h, t, None, None, 0, C.byref(n)))
        Q = np.empty((n.value, self.nplan))
        par = np.empty(n.value, np.int32)
        self.eng._ok(self.eng.lib.mjpl_rrt_get_tree(self.h, t, Q.ctypes.data_as(_F64P), par.ctypes.data_as(_I32P), n.value,
                                                    C.byref(n)))
        return Q, par

    def lanes_state(self):
        T = np.empty((self.lanes, self.nplan))
        on = np.empty(self.lanes, np.uint8)
        self.eng._ok(self.eng.lib.mjpl_rrt_get_lanes(self.h, T.ctypes.data_as(_F64P), on.ctypes.data_as(_U8P)))
        return T, on.astype(bool)


class EngineRing:
    """Several engines of one model on one GPU taking batches in turns: every engine has its own HIP stream and
    scratch, so the kernels of consecutive `check_edges_dev` / `check_configs_dev` calls overlap -- the next batch
    fills the chip while this one's kernels drain (three engines validate 262 144-edge batches
    at 1.55e9 edges/s, one at 1.08e9).  Calls return when they are enqueued; `sync()` waits for all engines.
    Results are those of a single engine, batch by batch."""

    def __init__(self, model: Model, allowed_collision_bodies=(), device: int = 0, engines: int = 3,
                 planning_qidx=None, qpos_base=None):
        if engines < 1:
            raise ValueError("`engines` must be >= 1")
        self.engines = [Engine(model, allowed_collision_bodies, device) for _ in range(engines)]
        if planning_qidx is not None:
            for e in self.engines:
                e.set_planning(planning_qidx, qpos_base)
        self._turn = 0

    def next(self) -> "Engine":
        """The engine whose turn it is (device buffers of any engine of the ring may be passed to any other)."""
        e = self.engines[self._turn % len(self.engines)]
        self._turn += 1
        return e

    def check_edges_dev(self, dQA, dQB, n, step_dist, layout, dvalid, dfirst_bad=None, flags=0) -> "Engine":
        e = self.next()
        e.check_edges_dev(dQA, dQB, n, step_dist, layout, dvalid, dfirst_bad, flags)
        return e

    def sync(self):
        for e in self.engines:
            e.sync()

    def close(self):
        for e in self.engines:
            e.close()
        self.engines = []


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library: 128 bytes for rank 0 to hand to the other ranks."""
    lib = load_library()
    buf = C.create_string_buffer(128)
    rc = lib.mjpl_comm_unique_id(buf)
    if rc != 0:
        raise MjplError(rc, lib.mjpl_last_error().decode())
    return buf.raw
