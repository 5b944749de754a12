"""Bullet 3.2.x utilities reached through `pybullet` on the hot path — float64 restatements.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference calls these through the third-party
`pybullet` module (pinned `^3.2.7`, reference pyproject.toml:19; not vendored, not installable
here).  Call sites: `p.getMatrixFromQuaternion` (envs/BaseAviary.py:836,771;
control/DSLPIDControl.py:187,240), `p.getEulerFromQuaternion` (envs/BaseAviary.py:518;
control/DSLPIDControl.py:144,241), `p.getQuaternionFromEuler` (envs/BaseAviary.py:488).
Formulas: btMatrix3x3::setRotation, pybullet.c pybullet_getEulerFromQuaternion,
btQuaternion::setEulerZYX.  Quaternions are (x, y, z, w).
"""
import math

import numpy as np


def matrix_from_quaternion(q):
    """btMatrix3x3::setRotation — insensitive to the quaternion's norm.  Returns 3x3 ndarray."""
    x, y, z, w = (float(q[0]), float(q[1]), float(q[2]), float(q[3]))
    d = x * x + y * y + z * z + w * w
    s = 2.0 / d
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    return np.array([[1.0 - (yy + zz), xy - wz, xz + wy],
                     [xy + wz, 1.0 - (xx + zz), yz - wx],
                     [xz - wy, yz + wx, 1.0 - (xx + yy)]], dtype=np.float64)


def euler_from_quaternion(q):
    """pybullet_getEulerFromQuaternion incl. its gimbal branches.  Returns (roll, pitch, yaw)."""
    x, y, z, w = (float(q[0]), float(q[1]), float(q[2]), float(q[3]))
    sqx, sqy, sqz, squ = x * x, y * y, z * z, w * w
    sarg = -2.0 * (x * z - w * y)
    if sarg <= -0.99999:
        return (0.0, -0.5 * math.pi, 2.0 * math.atan2(x, -y))
    if sarg >= 0.99999:
        return (0.0, 0.5 * math.pi, 2.0 * math.atan2(-x, y))
    return (math.atan2(2.0 * (y * z + w * x), squ - sqx - sqy + sqz),
            math.asin(sarg),
            math.atan2(2.0 * (x * y + w * z), squ + sqx - sqy - sqz))


def quaternion_from_euler(rpy):
    """btQuaternion::setEulerZYX(yaw, pitch, roll) followed by normalize()."""
    hr, hp, hy = 0.5 * float(rpy[0]), 0.5 * float(rpy[1]), 0.5 * float(rpy[2])
    cr, sr = math.cos(hr), math.sin(hr)
    cp, sp = math.cos(hp), math.sin(hp)
    cy, sy = math.cos(hy), math.sin(hy)
    x = sr * cp * cy - cr * sp * sy
    y = cr * sp * cy + sr * cp * sy
    z = cr * cp * sy - sr * sp * cy
    w = cr * cp * cy + sr * sp * sy
    n = math.sqrt(x * x + y * y + z * z + w * w)
    return (x / n, y / n, z / n, w / n)


# ---- the same three, vectorised over a leading batch axis (used by batched_oracle.py) ----------

def matrix_from_quaternion_b(q):
    """q [...,4] -> R [...,3,3]."""
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    s = 2.0 / (x * x + y * y + z * z + w * w)
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    R = np.empty(q.shape[:-1] + (3, 3), dtype=np.float64)
    R[..., 0, 0] = 1.0 - (yy + zz); R[..., 0, 1] = xy - wz; R[..., 0, 2] = xz + wy
    R[..., 1, 0] = xy + wz; R[..., 1, 1] = 1.0 - (xx + zz); R[..., 1, 2] = yz - wx
    R[..., 2, 0] = xz - wy; R[..., 2, 1] = yz + wx; R[..., 2, 2] = 1.0 - (xx + yy)
    return R


def euler_from_quaternion_b(q):
    """q [...,4] -> rpy [...,3]."""
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    sqx, sqy, sqz, squ = x * x, y * y, z * z, w * w
    sarg = -2.0 * (x * z - w * y)
    lo = sarg <= -0.99999
    hi = sarg >= 0.99999
    roll = np.arctan2(2.0 * (y * z + w * x), squ - sqx - sqy + sqz)
    pitch = np.arcsin(np.clip(sarg, -1.0, 1.0))
    yaw = np.arctan2(2.0 * (x * y + w * z), squ + sqx - sqy - sqz)
    roll = np.where(lo | hi, 0.0, roll)
    pitch = np.where(lo, -0.5 * np.pi, np.where(hi, 0.5 * np.pi, pitch))
    yaw = np.where(lo, 2.0 * np.arctan2(x, -y), np.where(hi, 2.0 * np.arctan2(-x, y), yaw))
    return np.stack([roll, pitch, yaw], axis=-1)


def quaternion_from_euler_b(rpy):
    h = 0.5 * np.asarray(rpy, dtype=np.float64)
    cr, sr = np.cos(h[..., 0]), np.sin(h[..., 0])
    cp, sp = np.cos(h[..., 1]), np.sin(h[..., 1])
    cy, sy = np.cos(h[..., 2]), np.sin(h[..., 2])
    q = np.stack([sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
                  cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy], axis=-1)
    return q / np.linalg.norm(q, axis=-1, keepdims=True)
