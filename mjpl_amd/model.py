"""Kinematic/collision model tables for the batched collision-validation path.

The reference hands a ``mujoco.MjModel`` to ``CollisionConstraint``
(src/mjpl/constraint/collision_constraint.py:10-24) and reads ``model.body(name).id``,
``model.geom_bodyid`` (:61, :93), ``model.jnt_range`` (joint_limit_constraint.py:16-17),
``model.joint(j).name`` / ``jnt_type`` / ``jnt_qposadr`` / ``nq`` / ``njnt``
(src/mjpl/utils.py:10-38).  ``mujoco`` is not available to this build, so :class:`Model`
carries the same-named fields (the subset the hot path touches) and is produced either
programmatically (:class:`ModelBuilder`) or from primitive-only MJCF (:func:`load_mjcf`).

Compile-time conventions follow MuJoCo's model compiler [MJ-recalled: user_model.cc /
user_objects.cc]: bodies are numbered depth-first in document order, geoms by body id,
quaternions are normalised by division by their norm, joint axes are normalised,
``fromto`` capsules become pos/quat/half-length, ``body_weldid`` merges joint-less
bodies with their parent, ``geom_rbound`` is the bounding-sphere radius.
"""
from __future__ import annotations

import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

# mjtJoint / mjtGeom
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE = 0, 1, 2, 3
GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 4, 5, 6, 7

_JNT_NAMES = {"free": JNT_FREE, "ball": JNT_BALL, "slide": JNT_SLIDE, "hinge": JNT_HINGE}
_GEOM_NAMES = {
    "plane": GEOM_PLANE, "hfield": GEOM_HFIELD, "sphere": GEOM_SPHERE, "capsule": GEOM_CAPSULE,
    "ellipsoid": GEOM_ELLIPSOID, "cylinder": GEOM_CYLINDER, "box": GEOM_BOX, "mesh": GEOM_MESH,
}
_EPS = 1e-14  # mjEPS


def _normalize(v: np.ndarray) -> np.ndarray:
    """mjuu_normvec: divide by the 2-norm unless it is within mjEPS of 1."""
    v = np.asarray(v, dtype=np.float64).copy()
    s = 0.0
    for x in v:
        s += float(x) * float(x)
    if s < _EPS:
        raise ValueError("cannot normalise a zero vector")
    n = math.sqrt(s)
    if abs(n - 1.0) > _EPS:
        v = v / n
    return v


def _mul_quat(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
    ])


def _z2quat(vec) -> np.ndarray:
    """Quaternion rotating +z onto ``vec`` (mjuu_z2quat)."""
    vec = _normalize(vec)
    axis = np.cross([0.0, 0.0, 1.0], vec)
    s = float(np.sqrt(axis @ axis))
    if s < 1e-10:
        axis = np.array([1.0, 0.0, 0.0])
    else:
        axis = axis / s
    ang = math.atan2(s, vec[2])
    q = np.empty(4)
    q[0] = math.cos(ang / 2)
    q[1:] = axis * math.sin(ang / 2)
    return q


def _euler2quat(e, seq="xyz") -> np.ndarray:
    q = np.array([1.0, 0.0, 0.0, 0.0])
    for ang, ax in zip(e, seq):
        r = np.array([math.cos(ang / 2), 0.0, 0.0, 0.0])
        r["xyz".index(ax.lower()) + 1] = math.sin(ang / 2)
        # lower-case: rotating (intrinsic) frame, upper-case: fixed frame
        q = _mul_quat(q, r) if ax.islower() else _mul_quat(r, q)
    return q


class _Named:
    """``model.body(name)``-style accessor (id, name and per-element views)."""

    def __init__(self, model: "Model", kind: str, idx: int):
        self._m, self._k, self.id = model, kind, int(idx)

    @property
    def name(self) -> str:
        return getattr(self._m, f"{self._k}_names")[self.id]

    def __getattr__(self, attr):
        m = object.__getattribute__(self, "_m")
        k = object.__getattribute__(self, "_k")
        prefix = {"joint": "jnt", "keyframe": "key"}.get(k, k)
        arr = getattr(m, f"{prefix}_{attr}", None)
        if arr is None:
            raise AttributeError(attr)
        return arr[self.id]


@dataclass
class Model:
    """The MjModel subset the collision path reads (same field names as mjModel)."""

    nq: int
    njnt: int
    nbody: int
    ngeom: int
    nsite: int
    body_parentid: np.ndarray
    body_weldid: np.ndarray
    body_jntadr: np.ndarray
    body_jntnum: np.ndarray
    body_geomadr: np.ndarray
    body_geomnum: np.ndarray
    body_pos: np.ndarray
    body_quat: np.ndarray
    jnt_type: np.ndarray
    jnt_qposadr: np.ndarray
    jnt_dofadr: np.ndarray
    jnt_bodyid: np.ndarray
    jnt_axis: np.ndarray
    jnt_pos: np.ndarray
    jnt_range: np.ndarray
    qpos0: np.ndarray
    geom_type: np.ndarray
    geom_bodyid: np.ndarray
    geom_contype: np.ndarray
    geom_conaffinity: np.ndarray
    geom_size: np.ndarray
    geom_pos: np.ndarray
    geom_quat: np.ndarray
    geom_rbound: np.ndarray
    geom_margin: np.ndarray
    site_bodyid: np.ndarray
    site_pos: np.ndarray
    site_quat: np.ndarray
    body_names: list = field(default_factory=list)
    joint_names: list = field(default_factory=list)
    geom_names: list = field(default_factory=list)
    site_names: list = field(default_factory=list)
    keyframe_names: list = field(default_factory=list)
    key_qpos: np.ndarray = field(default_factory=lambda: np.zeros((0, 0)))

    # -- name lookups (ValueError/KeyError on unknown names, as mujoco raises KeyError)
    def _lookup(self, kind: str, key) -> _Named:
        names = getattr(self, f"{kind}_names")
        if isinstance(key, (int, np.integer)):
            if not 0 <= int(key) < len(names):
                raise KeyError(f"Invalid {kind} index {key}")
            return _Named(self, kind, int(key))
        if key not in names:
            raise KeyError(f"Invalid name '{key}'. Valid names: {names}")
        return _Named(self, kind, names.index(key))

    def body(self, key) -> _Named:
        return self._lookup("body", key)

    def joint(self, key) -> _Named:
        return self._lookup("joint", key)

    def geom(self, key) -> _Named:
        return self._lookup("geom", key)

    def site(self, key) -> _Named:
        return self._lookup("site", key)

    def keyframe(self, key) -> _Named:
        return self._lookup("keyframe", key)

    @property
    def key_names(self):
        return self.keyframe_names


# ----------------------------------------------------------------------------- builder


@dataclass
class _Body:
    name: str
    parent: "_Body | None"
    pos: np.ndarray
    quat: np.ndarray
    joints: list = field(default_factory=list)
    geoms: list = field(default_factory=list)
    sites: list = field(default_factory=list)
    children: list = field(default_factory=list)


class ModelBuilder:
    """Programmatic construction of a :class:`Model` (stand-in for an MJCF compile)."""

    def __init__(self):
        self.world = _Body("world", None, np.zeros(3), np.array([1.0, 0, 0, 0]))
        self._bodies = {"world": self.world}
        self._keys: list[tuple[str, np.ndarray]] = []
        self._anon = 0

    def add_body(self, name, parent="world", pos=(0, 0, 0), quat=(1, 0, 0, 0)):
        if name is None:
            self._anon += 1
            name = f"_body{self._anon}"
        if name in self._bodies:
            raise ValueError(f"repeated body name '{name}'")
        p = self._bodies[parent]
        b = _Body(name, p, np.asarray(pos, float), _normalize(quat))
        p.children.append(b)
        self._bodies[name] = b
        return name

    def add_joint(self, body, name=None, type="hinge", axis=(0, 0, 1), pos=(0, 0, 0),
                  range=(0.0, 0.0), ref=0.0):
        jt = _JNT_NAMES[type] if isinstance(type, str) else int(type)
        if jt not in (JNT_SLIDE, JNT_HINGE):
            raise ValueError("only 1-DoF slide/hinge joints are supported (reference README.md:20)")
        if body == "world":
            raise ValueError("the world body cannot have joints")
        self._bodies[body].joints.append(
            dict(name=name or "", type=jt, axis=_normalize(axis), pos=np.asarray(pos, float),
                 range=np.asarray(range, float), ref=float(ref)))

    def add_geom(self, body="world", type="sphere", size=(0.0,), pos=(0, 0, 0),
                 quat=(1, 0, 0, 0), fromto=None, contype=1, conaffinity=1, margin=0.0, name=None):
        gt = _GEOM_NAMES[type] if isinstance(type, str) else int(type)
        sz = np.zeros(3)
        size = np.atleast_1d(np.asarray(size, float))
        sz[: len(size)] = size
        pos = np.asarray(pos, float)
        quat = _normalize(quat)
        if fromto is not None:
            if gt not in (GEOM_CAPSULE, GEOM_CYLINDER, GEOM_BOX, GEOM_ELLIPSOID):
                raise ValueError("fromto requires capsule, cylinder, box or ellipsoid")
            ft = np.asarray(fromto, float)
            vec = ft[0:3] - ft[3:6]
            length = float(np.sqrt(vec @ vec))
            if length < _EPS:
                raise ValueError("fromto points too close")
            if gt in (GEOM_CAPSULE, GEOM_CYLINDER):
                sz[1] = length / 2
            else:
                sz[2] = length / 2
            pos = (ft[0:3] + ft[3:6]) / 2
            quat = _z2quat(vec)
        self._bodies[body].geoms.append(
            dict(name=name or "", type=gt, size=sz, pos=pos, quat=quat, contype=int(contype),
                 conaffinity=int(conaffinity), margin=float(margin)))

    def add_site(self, body, name, pos=(0, 0, 0), quat=(1, 0, 0, 0)):
        self._bodies[body].sites.append(
            dict(name=name or "", pos=np.asarray(pos, float), quat=_normalize(quat)))

    def add_keyframe(self, name, qpos):
        self._keys.append((name, np.asarray(qpos, float)))

    # -- compile
    @staticmethod
    def _rbound(gt, sz):
        if gt == GEOM_SPHERE:
            return sz[0]
        if gt == GEOM_CAPSULE:
            return sz[0] + sz[1]
        if gt == GEOM_CYLINDER:
            return math.sqrt(sz[0] * sz[0] + sz[1] * sz[1])
        if gt in (GEOM_BOX, GEOM_ELLIPSOID):
            return math.sqrt(sz[0] * sz[0] + sz[1] * sz[1] + sz[2] * sz[2]) if gt == GEOM_BOX \
                else max(sz)
        return 0.0  # plane / hfield / mesh (mesh unsupported)

    def compile(self) -> Model:
        order: list[_Body] = []

        def dfs(b):
            order.append(b)
            for c in b.children:
                dfs(c)

        dfs(self.world)
        bid = {id(b): i for i, b in enumerate(order)}
        nb = len(order)
        parent = np.zeros(nb, np.int32)
        weld = np.zeros(nb, np.int32)
        jadr = np.full(nb, -1, np.int32)
        jnum = np.zeros(nb, np.int32)
        gadr = np.full(nb, -1, np.int32)
        gnum = np.zeros(nb, np.int32)
        bpos = np.zeros((nb, 3))
        bquat = np.zeros((nb, 4))
        jn = dict(type=[], qadr=[], body=[], axis=[], pos=[], range=[], ref=[], name=[])
        ge = dict(type=[], body=[], ct=[], ca=[], size=[], pos=[], quat=[], rb=[], mg=[], name=[])
        st = dict(body=[], pos=[], quat=[], name=[])
        for i, b in enumerate(order):
            parent[i] = bid[id(b.parent)] if b.parent is not None else 0
            bpos[i], bquat[i] = b.pos, b.quat
            if b.joints:
                jadr[i] = len(jn["type"])
                jnum[i] = len(b.joints)
            weld[i] = i if (b.joints or i == 0) else weld[parent[i]]
            for j in b.joints:
                jn["qadr"].append(len(jn["type"]))
                jn["type"].append(j["type"]); jn["body"].append(i); jn["axis"].append(j["axis"])
                jn["pos"].append(j["pos"]); jn["range"].append(j["range"]); jn["ref"].append(j["ref"])
                jn["name"].append(j["name"])
            if b.geoms:
                gadr[i] = len(ge["type"])
                gnum[i] = len(b.geoms)
            for g in b.geoms:
                ge["type"].append(g["type"]); ge["body"].append(i); ge["ct"].append(g["contype"])
                ge["ca"].append(g["conaffinity"]); ge["size"].append(g["size"])
                ge["pos"].append(g["pos"]); ge["quat"].append(g["quat"])
                ge["rb"].append(self._rbound(g["type"], g["size"])); ge["mg"].append(g["margin"])
                ge["name"].append(g["name"])
            for s in b.sites:
                st["body"].append(i); st["pos"].append(s["pos"]); st["quat"].append(s["quat"])
                st["name"].append(s["name"])
        nj, ng, ns = len(jn["type"]), len(ge["type"]), len(st["body"])

        def arr(x, shape, dtype=np.float64):
            return np.ascontiguousarray(np.asarray(x, dtype=dtype).reshape(shape))

        nq = nj
        keys = np.zeros((len(self._keys), nq))
        for k, (_, q) in enumerate(self._keys):
            if q.shape != (nq,):
                raise ValueError(f"keyframe qpos must have {nq} values")
            keys[k] = q
        return Model(
            nq=nq, njnt=nj, nbody=nb, ngeom=ng, nsite=ns,
            body_parentid=parent, body_weldid=weld, body_jntadr=jadr, body_jntnum=jnum,
            body_geomadr=gadr, body_geomnum=gnum,
            body_pos=arr(bpos, (nb, 3)), body_quat=arr(bquat, (nb, 4)),
            jnt_type=arr(jn["type"], (nj,), np.int32), jnt_qposadr=arr(jn["qadr"], (nj,), np.int32),
            jnt_dofadr=arr(jn["qadr"], (nj,), np.int32), jnt_bodyid=arr(jn["body"], (nj,), np.int32),
            jnt_axis=arr(jn["axis"], (nj, 3)), jnt_pos=arr(jn["pos"], (nj, 3)),
            jnt_range=arr(jn["range"], (nj, 2)), qpos0=arr(jn["ref"], (nj,)),
            geom_type=arr(ge["type"], (ng,), np.int32), geom_bodyid=arr(ge["body"], (ng,), np.int32),
            geom_contype=arr(ge["ct"], (ng,), np.int32),
            geom_conaffinity=arr(ge["ca"], (ng,), np.int32),
            geom_size=arr(ge["size"], (ng, 3)), geom_pos=arr(ge["pos"], (ng, 3)),
            geom_quat=arr(ge["quat"], (ng, 4)), geom_rbound=arr(ge["rb"], (ng,)),
            geom_margin=arr(ge["mg"], (ng,)),
            site_bodyid=arr(st["body"], (ns,), np.int32), site_pos=arr(st["pos"], (ns, 3)),
            site_quat=arr(st["quat"], (ns, 4)),
            body_names=[b.name for b in order], joint_names=jn["name"], geom_names=ge["name"],
            site_names=st["name"], keyframe_names=[k for k, _ in self._keys], key_qpos=keys,
        )


# ----------------------------------------------------------------------------- MJCF writer


def to_mjcf(model: Model, name: str = "mjpl_amd_model") -> str:
    """Primitive-only MJCF of a :class:`Model`: what ``mujoco.MjModel.from_xml_string`` needs to
    rebuild the same kinematic tree and collision geoms, so that real MuJoCo (where it is
    installed) can be asked for the verdicts this package computes (tools/crosscheck_mujoco.py).
    Numbers are written with ``repr`` (round-trip exact); ``autolimits`` is off and every joint
    carries its range explicitly; planes get a finite rendering size, which MuJoCo ignores for
    collisions."""
    inv_j = {v: k for k, v in _JNT_NAMES.items()}
    inv_g = {v: k for k, v in _GEOM_NAMES.items()}

    def v(a):
        return " ".join(repr(float(x)) for x in np.asarray(a).ravel())

    children: dict[int, list[int]] = {}
    for b in range(1, model.nbody):
        children.setdefault(int(model.body_parentid[b]), []).append(b)
    out = [f'<mujoco model="{name}">', '  <compiler angle="radian" autolimits="false"/>',
           '  <option gravity="0 0 0"/>']

    def emit_body(b: int, ind: str):
        if b == 0:
            out.append(ind + "<worldbody>")
        else:
            out.append(f'{ind}<body name="{model.body_names[b]}" pos="{v(model.body_pos[b])}" '
                       f'quat="{v(model.body_quat[b])}">')
        for j in range(int(model.body_jntadr[b]), int(model.body_jntadr[b]) + int(model.body_jntnum[b])) \
                if model.body_jntnum[b] else []:
            lo, hi = model.jnt_range[j]
            limited = "true" if (lo != 0.0 or hi != 0.0) else "false"
            nm = f'name="{model.joint_names[j]}" ' if model.joint_names[j] else ""
            out.append(f'{ind}  <joint {nm}type="{inv_j[int(model.jnt_type[j])]}" axis="{v(model.jnt_axis[j])}" '
                       f'pos="{v(model.jnt_pos[j])}" limited="{limited}" range="{v(model.jnt_range[j])}" '
                       f'ref="{repr(float(model.qpos0[model.jnt_qposadr[j]]))}"/>')
        for g in range(model.ngeom):
            if int(model.geom_bodyid[g]) != b:
                continue
            t = int(model.geom_type[g])
            size = model.geom_size[g].copy()
            nsz = {GEOM_PLANE: 3, GEOM_SPHERE: 1, GEOM_CAPSULE: 2, GEOM_CYLINDER: 2}.get(t, 3)
            if t == GEOM_PLANE and size[0] == 0 and size[1] == 0:
                size[2] = size[2] if size[2] > 0 else 0.1  # infinite plane: only the grid spacing must be > 0
            nm = f'name="{model.geom_names[g]}" ' if model.geom_names[g] else ""
            out.append(f'{ind}  <geom {nm}type="{inv_g[t]}" size="{v(size[:nsz])}" pos="{v(model.geom_pos[g])}" '
                       f'quat="{v(model.geom_quat[g])}" contype="{int(model.geom_contype[g])}" '
                       f'conaffinity="{int(model.geom_conaffinity[g])}" margin="{repr(float(model.geom_margin[g]))}"/>')
        for k in range(model.nsite):
            if int(model.site_bodyid[k]) == b:
                out.append(f'{ind}  <site name="{model.site_names[k]}" pos="{v(model.site_pos[k])}" '
                           f'quat="{v(model.site_quat[k])}"/>')
        for c in children.get(b, []):
            emit_body(c, ind + "  ")
        out.append(ind + ("</worldbody>" if b == 0 else "</body>"))

    emit_body(0, "  ")
    if len(model.keyframe_names):
        out.append("  <keyframe>")
        for k, kn in enumerate(model.keyframe_names):
            out.append(f'    <key name="{kn}" qpos="{v(model.key_qpos[k])}"/>')
        out.append("  </keyframe>")
    out.append("</mujoco>")
    return "\n".join(out) + "\n"


# ----------------------------------------------------------------------------- MJCF reader


def _floats(s):
    return [float(x) for x in s.split()]


class _Defaults:
    """Nested <default class=...> tables for joint / geom / site attributes."""

    def __init__(self):
        self.tables: dict[str, dict[str, dict[str, str]]] = {"main": {"joint": {}, "geom": {}, "site": {}}}

    def read(self, elem, parent="main"):
        cls = elem.get("class") or "main"
        if cls not in self.tables:
            self.tables[cls] = {k: dict(v) for k, v in self.tables[parent].items()}
        for child in elem:
            if child.tag in ("joint", "geom", "site"):
                self.tables[cls][child.tag].update(child.attrib)
        for child in elem:
            if child.tag == "default":
                self.read(child, cls)

    def resolve(self, tag, elem, childclass):
        cls = elem.get("class") or childclass or "main"
        if cls not in self.tables:
            raise ValueError(f"unknown default class '{cls}'")
        attrs = dict(self.tables[cls][tag])
        attrs.update({k: v for k, v in elem.attrib.items() if k != "class"})
        return attrs


def _expand_includes(root, base_dir):
    for parent in list(root.iter()):
        for i, child in enumerate(list(parent)):
            if child.tag == "include":
                inc = ET.parse(os.path.join(base_dir, child.get("file"))).getroot()
                _expand_includes(inc, base_dir)
                parent.remove(child)
                for k, sub in enumerate(list(inc)):
                    parent.insert(i + k, sub)


def load_mjcf(path_or_xml: str) -> Model:
    """Compile primitive-only MJCF into a :class:`Model` (see :func:`parse_mjcf`)."""
    return parse_mjcf(path_or_xml).compile()


def parse_mjcf(path_or_xml: str) -> ModelBuilder:
    """Read primitive-only MJCF (bodies, slide/hinge joints, plane/sphere/capsule/box
    geoms, sites, default classes, includes, keyframes) into a :class:`ModelBuilder`.

    Mesh/visual geoms with ``contype=conaffinity=0`` are dropped (they can never collide);
    a colliding mesh geom raises, since the north-star path is primitive-primitive only.
    """
    if path_or_xml.lstrip().startswith("<"):
        root = ET.fromstring(path_or_xml)
        base = "."
    else:
        root = ET.parse(path_or_xml).getroot()
        base = os.path.dirname(os.path.abspath(path_or_xml))
    _expand_includes(root, base)

    radians = False
    eulerseq = "xyz"
    for comp in root.findall("compiler"):
        if comp.get("angle"):
            radians = comp.get("angle") == "radian"
        eulerseq = comp.get("eulerseq", eulerseq)
    ang = 1.0 if radians else math.pi / 180.0

    defaults = _Defaults()
    for d in root.findall("default"):
        defaults.read(d)

    def orient(a):
        if "quat" in a:
            return _floats(a["quat"])
        if "euler" in a:
            return _euler2quat([x * ang for x in _floats(a["euler"])], eulerseq)
        if "axisangle" in a:
            v = _floats(a["axisangle"])
            ax = _normalize(v[:3])
            th = v[3] * ang
            return [math.cos(th / 2), *(ax * math.sin(th / 2))]
        if "zaxis" in a:
            return _z2quat(_floats(a["zaxis"]))
        return [1.0, 0, 0, 0]

    mb = ModelBuilder()

    def walk(elem, body_name, childclass):
        for child in elem:
            if child.tag == "geom":
                a = defaults.resolve("geom", child, childclass)
                ct, ca = int(a.get("contype", 1)), int(a.get("conaffinity", 1))
                gtype = a.get("type", "sphere")
                if "mesh" in a and "type" not in a:
                    gtype = "mesh"
                if gtype in ("mesh", "hfield", "sdf"):
                    if ct == 0 and ca == 0:
                        continue
                    raise ValueError(f"colliding '{gtype}' geoms are outside the primitive path")
                mb.add_geom(body_name, gtype, _floats(a.get("size", "0")),
                            _floats(a.get("pos", "0 0 0")), orient(a),
                            _floats(a["fromto"]) if "fromto" in a else None,
                            ct, ca, float(a.get("margin", 0.0)), a.get("name"))
            elif child.tag == "joint":
                a = defaults.resolve("joint", child, childclass)
                jt = a.get("type", "hinge")
                rng = _floats(a.get("range", "0 0"))
                ref = float(a.get("ref", 0.0))
                if jt == "hinge":
                    rng = [r * ang for r in rng]
                    ref *= ang
                mb.add_joint(body_name, a.get("name"), jt, _floats(a.get("axis", "0 0 1")),
                             _floats(a.get("pos", "0 0 0")), rng, ref)
            elif child.tag == "freejoint":
                raise ValueError("free joints are outside the 1-DoF planning path")
            elif child.tag == "site":
                a = defaults.resolve("site", child, childclass)
                mb.add_site(body_name, a.get("name"), _floats(a.get("pos", "0 0 0")), orient(a))
            elif child.tag == "body":
                cc = child.get("childclass", childclass)
                nm = mb.add_body(child.get("name"), body_name,
                                 _floats(child.get("pos", "0 0 0")), orient(child.attrib))
                walk(child, nm, cc)

    for wb in root.findall("worldbody"):
        walk(wb, "world", None)
    for kf in root.findall("keyframe"):
        for key in kf.findall("key"):
            if key.get("qpos"):
                mb.add_keyframe(key.get("name", ""), _floats(key.get("qpos")))
    return mb
