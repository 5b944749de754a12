"""ctypes front-end of oracle/gpd_oracle.c with the interface of `batched_oracle.BatchedAviary`.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Float64 C loops over aviaries and drones: ~100x faster
than the numpy versions at full BASELINE sizes, so that `tests/test_gpu_fullsize.py` can check 65 536
drones over 1920 physics steps in seconds.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import bullet_math as bm
from .aviary_oracle import ACT_DIM, UrdfConstants

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libgpd_oracle.so")
ACT_CODE = {"rpm": 0, "pid": 1, "vel": 2, "one_d_rpm": 3, "one_d_pid": 4, "raw_rpm": 5, "direct_rpm": 6}
TASK_CODE = {"none": 0, "hover": 1, "multihover": 2}
MODEL_CODE = {"cf2x": 0, "cf2p": 1, "racer": 2}
_d3, _d4, _d12 = ctypes.c_double * 3, ctypes.c_double * 4, ctypes.c_double * 12


class OrcParams(ctypes.Structure):
    _fields_ = [("drone_model", ctypes.c_int32), ("pad_", ctypes.c_int32),
                ("M", ctypes.c_double), ("L", ctypes.c_double), ("KF", ctypes.c_double), ("KM", ctypes.c_double),
                ("GRAVITY", ctypes.c_double), ("J", _d3), ("J_INV", _d3), ("prop_x", _d4), ("prop_y", _d4),
                ("gnd_eff_coeff", ctypes.c_double), ("prop_radius", ctypes.c_double), ("gnd_eff_h_clip", ctypes.c_double),
                ("drag_coeff", _d3), ("dw_coeff", _d3), ("hover_rpm", ctypes.c_double), ("max_rpm", ctypes.c_double),
                ("pid_gravity", ctypes.c_double), ("pid_kf", ctypes.c_double),
                ("p_for", _d3), ("i_for", _d3), ("d_for", _d3), ("p_tor", _d3), ("i_tor", _d3), ("d_tor", _d3),
                ("mixer", _d12), ("pwm2rpm_scale", ctypes.c_double), ("pwm2rpm_const", ctypes.c_double),
                ("min_pwm", ctypes.c_double), ("max_pwm", ctypes.c_double), ("speed_limit", ctypes.c_double),
                ("ground_z", ctypes.c_double)]


class OrcCfg(ctypes.Structure):
    _fields_ = [("num_envs", ctypes.c_int32), ("drones_per_env", ctypes.c_int32), ("act_type", ctypes.c_int32),
                ("substeps", ctypes.c_int32), ("physics_flags", ctypes.c_uint32), ("task", ctypes.c_int32),
                ("pyb_freq", ctypes.c_int32), ("auto_reset", ctypes.c_int32), ("pyb_dt", ctypes.c_double),
                ("ctrl_dt", ctypes.c_double), ("xy_bound", ctypes.c_double), ("z_bound", ctypes.c_double),
                ("tilt_bound", ctypes.c_double), ("term_dist", ctypes.c_double), ("episode_len_sec", ctypes.c_double)]


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "gpd_oracle.c")):
            build()
        _lib = ctypes.CDLL(_LIB)
        sizes = (ctypes.c_int32 * 2)()
        _lib.orc_struct_sizes(sizes)
        assert tuple(sizes) == (ctypes.sizeof(OrcParams), ctypes.sizeof(OrcCfg)), "oracle struct mismatch"
    return _lib


def make_params(C: UrdfConstants, pid_consts: UrdfConstants = None, pid_model="cf2x") -> OrcParams:
    p = OrcParams()
    p.drone_model = MODEL_CODE[C.DRONE_MODEL]
    p.M, p.L, p.KF, p.KM, p.GRAVITY = C.M, C.L, C.KF, C.KM, C.GRAVITY
    for k in range(3):
        p.J[k], p.J_INV[k], p.drag_coeff[k] = C.J[k, k], C.J_INV[k, k], C.DRAG_COEFF[k]
    for k in range(4):
        p.prop_x[k], p.prop_y[k] = C.PROP_OFFSETS[k, 0], C.PROP_OFFSETS[k, 1]
    p.gnd_eff_coeff, p.prop_radius, p.gnd_eff_h_clip = C.GND_EFF_COEFF, C.PROP_RADIUS, C.GND_EFF_H_CLIP
    p.dw_coeff[0], p.dw_coeff[1], p.dw_coeff[2] = C.DW_COEFF_1, C.DW_COEFF_2, C.DW_COEFF_3
    p.hover_rpm, p.max_rpm, p.speed_limit = C.HOVER_RPM, C.MAX_RPM, C.SPEED_LIMIT
    p.ground_z = C.COLLISION_H / 2 - C.COLLISION_Z_OFFSET
    pc = pid_consts or C
    p.pid_gravity, p.pid_kf = 9.8 * pc.M, pc.KF
    gains = dict(p_for=[.4, .4, 1.25], i_for=[.05, .05, .05], d_for=[.2, .2, .5], p_tor=[70000., 70000., 60000.],
                 i_tor=[.0, .0, 500.], d_tor=[20000., 20000., 12000.])
    for name, vals in gains.items():
        for k in range(3):
            getattr(p, name)[k] = vals[k]
    mixer = [[-.5, -.5, -1], [-.5, .5, 1], [.5, .5, -1], [.5, -.5, 1]] if pid_model == "cf2x" else \
        [[0, -1, -1], [1, 0, 1], [0, 1, -1], [-1, 0, 1]]
    for k, v in enumerate(np.array(mixer, dtype=np.float64).reshape(-1)):
        p.mixer[k] = v
    p.pwm2rpm_scale, p.pwm2rpm_const, p.min_pwm, p.max_pwm = 0.2685, 4070.3, 20000, 65535
    return p


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


class CAviary:
    """Same constructor / attributes / `step()` as `BatchedAviary`, arithmetic in C."""

    def __init__(self, urdf_path, drone_model="cf2x", num_envs=1, num_drones=1, initial_xyzs=None, initial_rpys=None,
                 physics_flags=0, pyb_freq=240, ctrl_freq=240, act="rpm", task="none", pid_urdf_path=None,
                 episode_len_sec=8, auto_reset=False, target_pos=None):
        self.L = lib()
        self.C = C = UrdfConstants(urdf_path, drone_model)
        self.E, self.D, self.S = num_envs, num_drones, int(pyb_freq / ctrl_freq)
        self.ACT, self.TASK, self.PHYS = act, task, physics_flags
        E, D, N = self.E, self.D, num_envs * num_drones
        pc = UrdfConstants(pid_urdf_path or urdf_path, "cf2x") if act in ("pid", "vel", "one_d_pid") else None
        self.params = make_params(C, pc)
        xy = 1.5 if task == "hover" else 2.0
        self.cfg = OrcCfg(num_envs=E, drones_per_env=D, act_type=ACT_CODE[act], substeps=self.S, physics_flags=physics_flags,
                          task=TASK_CODE[task], pyb_freq=pyb_freq, auto_reset=int(auto_reset), pyb_dt=1. / pyb_freq,
                          ctrl_dt=1. / ctrl_freq, xy_bound=xy, z_bound=2.0, tilt_bound=.4, term_dist=.0001,
                          episode_len_sec=episode_len_sec)
        if initial_xyzs is None:
            i = np.arange(D, dtype=np.float64)
            initial_xyzs = np.stack([i * 4 * C.L, i * 4 * C.L, np.ones(D) * (C.COLLISION_H / 2 - C.COLLISION_Z_OFFSET + .1)], axis=-1)
        self.INIT_XYZS = np.ascontiguousarray(np.broadcast_to(np.asarray(initial_xyzs, dtype=np.float64), (E, D, 3)))
        self.INIT_RPYS = np.ascontiguousarray(np.broadcast_to(np.zeros(3) if initial_rpys is None else
                                                              np.asarray(initial_rpys, dtype=np.float64), (E, D, 3)))
        self.INIT_QUAT = np.ascontiguousarray(bm.quaternion_from_euler_b(self.INIT_RPYS))
        if target_pos is not None:
            self.TARGET_POS = np.ascontiguousarray(np.broadcast_to(np.asarray(target_pos, dtype=np.float64), (E, D, 3)))
        elif task == "hover":
            self.TARGET_POS = np.ascontiguousarray(np.broadcast_to(np.array([0, 0, 1.]), (E, D, 3)))
        elif task == "multihover":
            self.TARGET_POS = self.INIT_XYZS + np.array([[0, 0, 1 / (i + 1)] for i in range(D)])[None]
        else:
            self.TARGET_POS = np.zeros((E, D, 3))
        z = lambda k: np.zeros((E, D, k))
        self.pos, self.quat, self.vel, self.rpy_rates, self.ang_v, self.rpy = z(3), z(4), z(3), z(3), z(3), z(3)
        self.last_rpm, self.pid_state = z(4), z(9)
        self.step_counter = np.zeros(E, dtype=np.int64)
        self.obs = z(12)
        self.reward = np.zeros(E)
        self.terminated, self.truncated = np.zeros(E, dtype=np.uint8), np.zeros(E, dtype=np.uint8)
        self.term_obs = z(12)
        self.reset()

    def reset(self, mask=None):
        m = np.ones(self.E, dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        self.pos[m], self.quat[m] = self.INIT_XYZS[m], self.INIT_QUAT[m]
        self.vel[m] = 0; self.rpy_rates[m] = 0; self.ang_v[m] = 0; self.last_rpm[m] = 0
        self.rpy[m] = bm.euler_from_quaternion_b(self.quat[m])
        self.step_counter[m] = 0
        return self.obs12()

    def obs12(self):
        return np.concatenate([self.pos, self.rpy, self.vel, self.ang_v], axis=-1)

    def state20(self):
        return np.concatenate([self.pos, self.quat, self.rpy, self.vel, self.ang_v, self.last_rpm], axis=-1)

    def step(self, action):
        self.step_in_place(action)
        return (self.obs.copy(), self.reward.copy(), self.terminated.astype(bool), self.truncated.astype(bool),
                self.term_obs.copy() if self.cfg.auto_reset else None)

    def step_in_place(self, action):
        """One step; results stay in self.obs / reward / terminated / truncated / term_obs (no copies: for timing)."""
        a = np.ascontiguousarray(np.asarray(action, dtype=np.float64).reshape(self.E, self.D, ACT_DIM.get(self.ACT, 4)))
        rc = self.L.orc_step(ctypes.byref(self.params), ctypes.byref(self.cfg), _ptr(self.pos), _ptr(self.quat),
                             _ptr(self.vel), _ptr(self.rpy_rates), _ptr(self.ang_v), _ptr(self.rpy), _ptr(self.last_rpm),
                             _ptr(self.pid_state), _ptr(self.step_counter), _ptr(a), _ptr(self.TARGET_POS),
                             _ptr(self.INIT_XYZS), _ptr(self.INIT_QUAT), _ptr(self.obs), _ptr(self.reward),
                             _ptr(self.terminated), _ptr(self.truncated), _ptr(self.term_obs))
        assert rc == 0


def downwash_all_pairs(urdf_path, xyz, drone_model="cf2x", threads=1):
    """`BaseAviary._downwash` for ONE aviary of any size (the reference's O(n^2) loop, float64 C): body-z force per drone."""
    L = lib()
    p = make_params(UrdfConstants(urdf_path, drone_model))
    pos = np.ascontiguousarray(np.asarray(xyz, dtype=np.float64).reshape(-1, 3))
    out = np.zeros(len(pos))
    L.orc_set_threads(threads)
    try:
        rc = L.orc_downwash_all_pairs(ctypes.byref(p), len(pos), _ptr(pos), _ptr(out))
    finally:
        L.orc_set_threads(1)
    assert rc == 0
    return out


def downwash_some(urdf_path, xyz, receivers, drone_model="cf2x", threads=1):
    """`downwash_all_pairs` for the drones `receivers` (indices) only: every drone is a source, a sample of them receivers."""
    L = lib()
    p = make_params(UrdfConstants(urdf_path, drone_model))
    pos = np.ascontiguousarray(np.asarray(xyz, dtype=np.float64).reshape(-1, 3))
    recv = np.ascontiguousarray(np.asarray(receivers, dtype=np.int32))
    out = np.zeros(len(recv))
    L.orc_set_threads(threads)
    try:
        rc = L.orc_downwash_some(ctypes.byref(p), len(pos), _ptr(pos), len(recv), _ptr(recv), _ptr(out))
    finally:
        L.orc_set_threads(1)
    assert rc == 0
    return out


def swarm_substep_seconds(xyz, threads=1, budget_s=10.0, urdf_path=None):
    """Seconds per all-pairs downwash pass over the drones at `xyz` (what dominates a sub-step of one large world on the CPU),
    averaged over the passes that fit `budget_s` -> (seconds, passes)."""
    import time
    urdf_path = urdf_path or os.path.join(os.path.dirname(_HERE), "gym_pybullet_drones_amd", "assets", "cf2x.urdf")
    downwash_all_pairs(urdf_path, xyz[:256], threads=threads)          # (page in)
    reps, t0 = 0, time.perf_counter()
    while reps == 0 or time.perf_counter() - t0 < budget_s:
        downwash_all_pairs(urdf_path, xyz, threads=threads)
        reps += 1
    return (time.perf_counter() - t0) / reps, reps
