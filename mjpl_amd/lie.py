"""Minimal SE3 / SO3 value types with the slice of ``mink.lie``'s interface that the reference's
PoseConstraint API takes and returns (``reference_frame: SE3``, pose_constraint.py:21;
``utils.site_pose -> SE3``, src/mjpl/utils.py:60-75).  ``mink`` is a third-party wheel that is
not installed here; these are plain NumPy holders for the host side of the boundary (the
projection itself runs in ``libmjpl_hip.so``).  Quaternions are ``wxyz`` as in MuJoCo.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


def _mul_quat(a, b):
    return np.array([
        a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
        a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
        a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
        a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0],
    ])


@dataclass(frozen=True)
class RollPitchYaw:
    roll: float
    pitch: float
    yaw: float


class SO3:
    def __init__(self, wxyz):
        self.wxyz = np.asarray(wxyz, dtype=np.float64).copy()
        assert self.wxyz.shape == (4,)

    @staticmethod
    def identity() -> "SO3":
        return SO3([1.0, 0.0, 0.0, 0.0])

    @staticmethod
    def from_matrix(matrix) -> "SO3":
        """Rotation matrix -> unit quaternion (the branch structure of mju_mat2Quat)."""
        m = np.asarray(matrix, dtype=np.float64).reshape(9)
        q = np.zeros(4)
        if m[0] + m[4] + m[8] > 0:
            q[0] = 0.5 * np.sqrt(1 + m[0] + m[4] + m[8])
            q[1] = 0.25 * (m[7] - m[5]) / q[0]
            q[2] = 0.25 * (m[2] - m[6]) / q[0]
            q[3] = 0.25 * (m[3] - m[1]) / q[0]
        elif m[0] > m[4] and m[0] > m[8]:
            q[1] = 0.5 * np.sqrt(1 + m[0] - m[4] - m[8])
            q[0] = 0.25 * (m[7] - m[5]) / q[1]
            q[2] = 0.25 * (m[1] + m[3]) / q[1]
            q[3] = 0.25 * (m[2] + m[6]) / q[1]
        elif m[4] > m[8]:
            q[2] = 0.5 * np.sqrt(1 - m[0] + m[4] - m[8])
            q[0] = 0.25 * (m[2] - m[6]) / q[2]
            q[1] = 0.25 * (m[1] + m[3]) / q[2]
            q[3] = 0.25 * (m[5] + m[7]) / q[2]
        else:
            q[3] = 0.5 * np.sqrt(1 - m[0] - m[4] + m[8])
            q[0] = 0.25 * (m[3] - m[1]) / q[3]
            q[1] = 0.25 * (m[2] + m[6]) / q[3]
            q[2] = 0.25 * (m[5] + m[7]) / q[3]
        n = np.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
        if n < 1e-15:
            q[:] = (1, 0, 0, 0)
        elif abs(n - 1) > 1e-15:
            q *= 1 / n
        return SO3(q)

    @staticmethod
    def from_rpy_radians(roll: float, pitch: float, yaw: float) -> "SO3":
        def about(axis, angle):
            q = np.zeros(4)
            q[0] = np.cos(angle / 2)
            q[1 + axis] = np.sin(angle / 2)
            return SO3(q)
        return about(2, yaw) @ about(1, pitch) @ about(0, roll)

    def as_matrix(self) -> np.ndarray:
        w, x, y, z = self.wxyz
        return np.array([
            [w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z],
        ])

    def as_rpy_radians(self) -> RollPitchYaw:
        q0, q1, q2, q3 = self.wxyz
        return RollPitchYaw(
            roll=float(np.arctan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2))),
            pitch=float(np.arcsin(2 * (q0 * q2 - q3 * q1))),
            yaw=float(np.arctan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3))),
        )

    def inverse(self) -> "SO3":
        return SO3(self.wxyz * np.array([1.0, -1.0, -1.0, -1.0]))

    def multiply(self, other: "SO3") -> "SO3":
        return SO3(_mul_quat(self.wxyz, other.wxyz))

    __matmul__ = multiply

    def apply(self, target) -> np.ndarray:
        t = np.asarray(target, dtype=np.float64)
        assert t.shape == (3,)
        padded = np.concatenate([np.zeros(1), t])
        return _mul_quat(_mul_quat(self.wxyz, padded), self.inverse().wxyz)[1:]

    @staticmethod
    def exp(tangent) -> "SO3":
        """Rotation vector -> rotation."""
        w = np.asarray(tangent, dtype=np.float64)
        theta = float(np.linalg.norm(w))
        if theta < 1e-10:
            q = np.concatenate([[1.0], 0.5 * w])
            return SO3(q / np.linalg.norm(q))
        return SO3(np.concatenate([[np.cos(theta / 2)], np.sin(theta / 2) * w / theta]))

    def log(self) -> np.ndarray:
        """Rotation vector of the shortest rotation."""
        q = self.wxyz if self.wxyz[0] >= 0 else -self.wxyz
        s = float(np.linalg.norm(q[1:]))
        if s < 1e-12:
            return 2.0 * q[1:]
        return 2.0 * np.arctan2(s, q[0]) / s * q[1:]


class SE3:
    def __init__(self, wxyz_xyz):
        self.wxyz_xyz = np.asarray(wxyz_xyz, dtype=np.float64).copy()
        assert self.wxyz_xyz.shape == (7,)

    @staticmethod
    def identity() -> "SE3":
        return SE3([1.0, 0, 0, 0, 0, 0, 0])

    @staticmethod
    def from_rotation_and_translation(rotation: SO3, translation) -> "SE3":
        return SE3(np.concatenate([rotation.wxyz, np.asarray(translation, dtype=np.float64)]))

    @staticmethod
    def from_translation(translation) -> "SE3":
        return SE3.from_rotation_and_translation(SO3.identity(), translation)

    @staticmethod
    def from_rotation(rotation: SO3) -> "SE3":
        return SE3.from_rotation_and_translation(rotation, np.zeros(3))

    def rotation(self) -> SO3:
        return SO3(self.wxyz_xyz[:4])

    def translation(self) -> np.ndarray:
        return self.wxyz_xyz[4:].copy()

    def inverse(self) -> "SE3":
        r_inv = self.rotation().inverse()
        return SE3.from_rotation_and_translation(r_inv, -(r_inv.apply(self.translation())))

    def multiply(self, other: "SE3") -> "SE3":
        return SE3.from_rotation_and_translation(
            self.rotation() @ other.rotation(),
            self.rotation().apply(other.translation()) + self.translation())

    __matmul__ = multiply

    @staticmethod
    def _V(w: np.ndarray):
        """Left Jacobian of SO(3) at rotation vector w (couples translation in exp / log)."""
        theta = float(np.linalg.norm(w))
        K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        if theta < 1e-6:
            return np.eye(3) + 0.5 * K + K @ K / 6.0
        return (np.eye(3) + (1 - np.cos(theta)) / theta ** 2 * K
                + (theta - np.sin(theta)) / theta ** 3 * K @ K)

    @staticmethod
    def exp(tangent) -> "SE3":
        """se(3) tangent (v[3], w[3]) -> pose."""
        t = np.asarray(tangent, dtype=np.float64)
        return SE3.from_rotation_and_translation(SO3.exp(t[3:]), SE3._V(t[3:]) @ t[:3])

    def log(self) -> np.ndarray:
        w = self.rotation().log()
        return np.concatenate([np.linalg.solve(SE3._V(w), self.translation()), w])

    def minus(self, other: "SE3") -> np.ndarray:
        """Right-minus: log(other^-1 @ self), the tangent that takes ``other`` to ``self``."""
        return other.inverse().multiply(self).log()

    def interpolate(self, other: "SE3", alpha: float = 0.5) -> "SE3":
        """``self @ exp(alpha * log(self^-1 @ other))``, alpha in [0, 1]."""
        if alpha < 0.0 or alpha > 1.0:
            raise ValueError(f"Expected alpha within [0.0, 1.0] but received {alpha}")
        return self.multiply(SE3.exp(alpha * other.minus(self)))

    def __repr__(self) -> str:
        return f"SE3(wxyz={self.wxyz_xyz[:4]}, xyz={self.wxyz_xyz[4:]})"
