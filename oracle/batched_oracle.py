"""The oracle's arithmetic vectorised over [E envs, D drones] with numpy float64.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Same formulas, same ordering as
oracle/aviary_oracle.py (which is pinned against the reference's own code); checked against it in
tests/test_oracle_batched.py.  Exists so that parity runs at thousands of drones finish in
seconds.  Adds the one thing the single-aviary reference leaves to its caller: the same-step
auto-reset of SB3's DummyVecEnv (examples/learn.py:54-58 is the calling pattern; reset semantics
= envs/BaseAviary.py:451-477, i.e. the PID state and the action history survive a reset).
"""
import numpy as np

from . import bullet_math as bm
from .aviary_oracle import ACT_DIM, BULLET_DAMPING, PHYS_DAMP, PHYS_DRAG, PHYS_DW, PHYS_GND, PHYS_GROUND, UrdfConstants


def _cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def _norm(a):
    return np.sqrt(np.sum(a * a, axis=-1))


class BatchedPID:
    """control/DSLPIDControl.py:82-259 over a leading batch shape."""

    def __init__(self, consts: UrdfConstants, shape, g=9.8):
        assert consts.DRONE_MODEL in ("cf2x", "cf2p")
        self.GRAVITY, self.KF = g * consts.M, consts.KF
        self.P_FOR, self.I_FOR, self.D_FOR = np.array([.4, .4, 1.25]), np.array([.05, .05, .05]), np.array([.2, .2, .5])
        self.P_TOR, self.I_TOR = np.array([70000., 70000., 60000.]), np.array([.0, .0, 500.])
        self.D_TOR = np.array([20000., 20000., 12000.])
        self.SCALE, self.CONST, self.MIN_PWM, self.MAX_PWM = 0.2685, 4070.3, 20000, 65535
        if consts.DRONE_MODEL == "cf2x":
            self.MIXER = np.array([[-.5, -.5, -1], [-.5, .5, 1], [.5, .5, -1], [.5, -.5, 1]])
        else:
            self.MIXER = np.array([[0, -1, -1], [+1, 0, 1], [0, 1, -1], [-1, 0, 1]])
        self.shape = tuple(shape)
        self.reset()

    def reset(self):
        self.integral_pos_e = np.zeros(self.shape + (3,))
        self.last_rpy = np.zeros(self.shape + (3,))
        self.integral_rpy_e = np.zeros(self.shape + (3,))

    def compute(self, dt, pos, quat, vel, tpos, trpy=None, tvel=None, trates=None):
        z3 = np.zeros(self.shape + (3,))
        trpy = z3 if trpy is None else trpy
        tvel = z3 if tvel is None else tvel
        trates = z3 if trates is None else trates
        R = bm.matrix_from_quaternion_b(quat)
        rpy = bm.euler_from_quaternion_b(quat)
        e_p, e_v = tpos - pos, tvel - vel
        acc = np.clip(self.integral_pos_e + e_p * dt, -2., 2.)
        acc[..., 2] = np.clip(acc[..., 2], -0.15, .15)
        self.integral_pos_e = acc
        f_des = self.P_FOR * e_p + self.I_FOR * acc + self.D_FOR * e_v
        f_des[..., 2] = f_des[..., 2] + self.GRAVITY
        along = np.maximum(0., np.sum(f_des * R[..., :, 2], axis=-1))
        base_pwm = (np.sqrt(along / (4 * self.KF)) - self.CONST) / self.SCALE
        zb = f_des / _norm(f_des)[..., None]
        heading = np.stack([np.cos(trpy[..., 2]), np.sin(trpy[..., 2]), np.zeros(self.shape)], axis=-1)
        yb = _cross(zb, heading)
        yb = yb / _norm(yb)[..., None]
        xb = _cross(yb, zb)
        Rd = np.stack([xb, yb, zb], axis=-1)                     # columns
        # intrinsic XYZ Euler angles of Rd (scipy 'XYZ'); only the yaw is ever used by callers
        des_yaw = np.arctan2(-Rd[..., 0, 1], Rd[..., 0, 0])
        RdT_R = np.einsum('...ki,...kj->...ij', Rd, R)
        skew = RdT_R - np.swapaxes(RdT_R, -1, -2)
        e_R = np.stack([skew[..., 2, 1], skew[..., 0, 2], skew[..., 1, 0]], axis=-1)
        e_w = trates - (rpy - self.last_rpy) / dt
        self.last_rpy = rpy
        acc_r = np.clip(self.integral_rpy_e - e_R * dt, -1500., 1500.)
        acc_r[..., 0:2] = np.clip(acc_r[..., 0:2], -1., 1.)
        self.integral_rpy_e = acc_r
        tau = np.clip(-self.P_TOR * e_R + self.D_TOR * e_w + self.I_TOR * acc_r, -3200, 3200)
        pwm = np.clip(base_pwm[..., None] + np.einsum('mk,...k->...m', self.MIXER, tau), self.MIN_PWM, self.MAX_PWM)
        return self.SCALE * pwm + self.CONST, e_p, des_yaw - rpy[..., 2]


class BatchedAviary:
    """E independent aviaries of D drones each; arrays are [E, D, k]."""

    def __init__(self, urdf_path, drone_model="cf2x", num_envs=1, num_drones=1, initial_xyzs=None, initial_rpys=None,
                 physics_flags=0, pyb_freq=240, ctrl_freq=240, act="rpm", task="none", pid_urdf_path=None,
                 episode_len_sec=8, auto_reset=False, target_pos=None):
        self.C = C = UrdfConstants(urdf_path, drone_model)
        self.E, self.D = num_envs, num_drones
        self.PHYS, self.ACT, self.TASK = physics_flags, act, task
        self.PYB_FREQ, self.CTRL_FREQ = pyb_freq, ctrl_freq
        self.S = int(pyb_freq / ctrl_freq)
        self.PYB_TIMESTEP, self.CTRL_TIMESTEP = 1. / pyb_freq, 1. / ctrl_freq
        self.EPISODE_LEN_SEC, self.auto_reset = episode_len_sec, auto_reset
        E, D = self.E, self.D
        if initial_xyzs is None:
            i = np.arange(D, dtype=np.float64)
            initial_xyzs = np.stack([i * 4 * C.L, i * 4 * C.L,
                                     np.ones(D) * (C.COLLISION_H / 2 - C.COLLISION_Z_OFFSET + .1)], axis=-1)
        self.INIT_XYZS = np.broadcast_to(np.asarray(initial_xyzs, dtype=np.float64), (E, D, 3)).copy()
        self.INIT_RPYS = np.broadcast_to(np.zeros(3) if initial_rpys is None else
                                         np.asarray(initial_rpys, dtype=np.float64), (E, D, 3)).copy()
        self.INIT_QUAT = bm.quaternion_from_euler_b(self.INIT_RPYS)
        if target_pos is not None:
            self.TARGET_POS = np.broadcast_to(np.asarray(target_pos, dtype=np.float64), (E, D, 3)).copy()
        elif task == "hover":
            self.TARGET_POS = np.broadcast_to(np.array([0, 0, 1.]), (E, D, 3)).copy()
        elif task == "multihover":
            self.TARGET_POS = self.INIT_XYZS + np.array([[0, 0, 1 / (i + 1)] for i in range(D)])[None]
        else:
            self.TARGET_POS = np.zeros((E, D, 3))
        if act in ("pid", "vel", "one_d_pid"):
            self.pid = BatchedPID(UrdfConstants(pid_urdf_path or urdf_path, "cf2x"), (E, D))
        self.pos = np.zeros((E, D, 3)); self.quat = np.zeros((E, D, 4)); self.vel = np.zeros((E, D, 3))
        self.rpy_rates = np.zeros((E, D, 3)); self.ang_v = np.zeros((E, D, 3)); self.rpy = np.zeros((E, D, 3))
        self.last_rpm = np.zeros((E, D, 4)); self.step_counter = np.zeros(E, dtype=np.int64)
        self.reset()

    def reset(self, mask=None):
        m = np.ones(self.E, dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        self.pos[m] = self.INIT_XYZS[m]
        self.quat[m] = self.INIT_QUAT[m]
        self.vel[m] = 0; self.rpy_rates[m] = 0; self.ang_v[m] = 0; self.last_rpm[m] = 0
        self.rpy[m] = bm.euler_from_quaternion_b(self.quat[m])
        self.step_counter[m] = 0
        return self.obs12()

    def obs12(self):
        return np.concatenate([self.pos, self.rpy, self.vel, self.ang_v], axis=-1)

    def downwash_force_all(self):
        """[E, D] body-z downwash force on every drone from the drones above it in its aviary
        (envs/BaseAviary.py:798-804, all pairs with dz > 0 and dxy < 10)."""
        C = self.C
        dz = self.pos[:, None, :, 2] - self.pos[:, :, None, 2]            # [E, i, j] = z_j - z_i
        dxy = _norm(self.pos[:, None, :, 0:2] - self.pos[:, :, None, 0:2])
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            alpha = C.DW_COEFF_1 * (C.PROP_RADIUS / (4 * dz)) ** 2
            beta = C.DW_COEFF_2 * dz + C.DW_COEFF_3
            term = -alpha * np.exp(-.5 * (dxy / beta) ** 2)
        return np.sum(np.where((dz > 0) & (dxy < 10), term, 0.0), axis=-1)

    def state20(self):
        return np.concatenate([self.pos, self.quat, self.rpy, self.vel, self.ang_v, self.last_rpm], axis=-1)

    def _rpm_from_action(self, action):
        C = self.C
        a = np.asarray(action, dtype=np.float64).reshape(self.E, self.D, ACT_DIM[self.ACT])
        if self.ACT == "raw_rpm":
            return np.clip(a, 0, C.MAX_RPM)
        if self.ACT == "rpm":
            return C.HOVER_RPM * (1 + 0.05 * a)
        if self.ACT == "one_d_rpm":
            return np.repeat(C.HOVER_RPM * (1 + 0.05 * a), 4, axis=-1)
        dt = self.CTRL_TIMESTEP
        if self.ACT == "pid":
            d = a - self.pos
            n = _norm(d)
            with np.errstate(invalid="ignore", divide="ignore"):
                stepped = self.pos + d / n[..., None]
            tpos = np.where((n <= 1)[..., None], a, stepped)
            rpm, _, _ = self.pid.compute(dt, self.pos, self.quat, self.vel, tpos)
        elif self.ACT == "vel":
            n = _norm(a[..., 0:3])
            with np.errstate(invalid="ignore", divide="ignore"):
                unit = np.where((n != 0)[..., None], a[..., 0:3] / n[..., None], 0.0)
            trpy = np.concatenate([np.zeros((self.E, self.D, 2)), self.rpy[..., 2:3]], axis=-1)
            tvel = C.SPEED_LIMIT * np.abs(a[..., 3:4]) * unit
            rpm, _, _ = self.pid.compute(dt, self.pos, self.quat, self.vel, self.pos.copy(), trpy, tvel)
        elif self.ACT == "one_d_pid":
            tpos = self.pos.copy()
            tpos[..., 2] += 0.1 * a[..., 0]
            rpm, _, _ = self.pid.compute(dt, self.pos, self.quat, self.vel, tpos)
        else:
            raise ValueError(self.ACT)
        return rpm

    def _substep(self, rpm, last_rpm):
        C, h = self.C, self.PYB_TIMESTEP
        R = bm.matrix_from_quaternion_b(self.quat)
        sq = rpm ** 2
        f = sq * C.KF
        if self.PHYS & PHYS_GND:
            hts = self.pos[..., 2:3] + R[..., 2, 0:1] * C.PROP_OFFSETS[:, 0] + R[..., 2, 1:2] * C.PROP_OFFSETS[:, 1] \
                + R[..., 2, 2:3] * C.PROP_OFFSETS[:, 2]
            hts = np.clip(hts, C.GND_EFF_H_CLIP, np.inf)
            gnd = sq * C.KF * C.GND_EFF_COEFF * (C.PROP_RADIUS / (4 * hts)) ** 2
            rpy = bm.euler_from_quaternion_b(self.quat)
            on = (np.abs(rpy[..., 0]) < np.pi / 2) & (np.abs(rpy[..., 1]) < np.pi / 2)
            f = f + np.where(on[..., None], gnd, 0.0)
        fz = np.sum(f, axis=-1)
        if self.PHYS & PHYS_DW:
            fz = fz + self.downwash_force_all()
        F = R[..., :, 2] * fz[..., None]
        F[..., 2] -= C.GRAVITY
        if self.PHYS & PHYS_DRAG:
            F = F - C.DRAG_COEFF * self.vel * np.sum(2 * np.pi * last_rpm / 60, axis=-1)[..., None]
        yaw_t = sq * C.KM * (-1.0 if C.DRONE_MODEL == "racer" else 1.0)
        tz = -yaw_t[..., 0] + yaw_t[..., 1] - yaw_t[..., 2] + yaw_t[..., 3]
        if C.DRONE_MODEL == "cf2p":
            tx, ty = (f[..., 1] - f[..., 3]) * C.L, (-f[..., 0] + f[..., 2]) * C.L
        else:
            arm = C.L / np.sqrt(2)
            tx = (f[..., 0] + f[..., 1] - f[..., 2] - f[..., 3]) * arm
            ty = (-f[..., 0] + f[..., 1] + f[..., 2] - f[..., 3]) * arm
            if C.DRONE_MODEL == "cf2x":
                tx = -tx
        w = self.rpy_rates
        Jd = np.diag(C.J)
        tau = np.stack([tx, ty, tz], axis=-1) - _cross(w, Jd * w)
        w_dot, a = tau * np.diag(C.J_INV), F / C.M
        if self.PHYS & PHYS_DAMP:                         # extension (aviary_oracle.PHYS_DAMP): Bullet's default multibody damping
            a = a - BULLET_DAMPING * (1.0 + _norm(self.vel))[..., None] * self.vel
            w_dot = w_dot - BULLET_DAMPING * (1.0 + _norm(w))[..., None] * w
        w = w + h * w_dot
        v = self.vel + h * a
        x = self.pos + h * v
        if self.PHYS & PHYS_GROUND:                       # extension (aviary_oracle.PHYS_GROUND): the plane at z = 0
            z_rest = C.COLLISION_H / 2 - C.COLLISION_Z_OFFSET
            hit = (x[..., 2] < z_rest) | ((x[..., 2] <= z_rest) & (v[..., 2] < 0))      # (second clause: the tie x_z == z_rest)
            x = np.where(hit[..., None], np.stack([x[..., 0], x[..., 1], np.full_like(x[..., 2], z_rest)], axis=-1), x)
            v = np.where(hit[..., None], np.stack([np.zeros_like(v[..., 0]), np.zeros_like(v[..., 1]), np.maximum(v[..., 2], 0.0)], axis=-1), v)
        n = _norm(w)
        th = n * h / 2
        qx, qy, qz, qw = (self.quat[..., k] for k in range(4))
        p, q_, r = w[..., 0], w[..., 1], w[..., 2]
        lam = np.stack([r * qy - q_ * qz + p * qw, -r * qx + p * qz + q_ * qw,
                        q_ * qx - p * qy + r * qw, -p * qx - q_ * qy - r * qz], axis=-1)
        with np.errstate(divide="ignore", invalid="ignore"):
            qn = np.cos(th)[..., None] * self.quat + (np.sin(th) / n)[..., None] * lam
        still = np.isclose(n, 0)
        self.quat = np.where(still[..., None], self.quat, qn)
        self.pos, self.vel, self.rpy_rates = x, v, w
        self.ang_v = np.einsum('...ij,...j->...i', R, w)

    def step(self, action):
        """-> obs12 [E,D,12], reward [E], terminated [E], truncated [E], terminal_obs12 (or None)."""
        rpm = self._rpm_from_action(action)
        for s in range(self.S):
            self._substep(rpm, self.last_rpm if s == 0 else rpm)
        self.last_rpm = rpm.copy()
        self.rpy = bm.euler_from_quaternion_b(self.quat)
        obs = self.obs12()
        if self.TASK == "none":
            reward = -np.ones(self.E)
            term = np.zeros(self.E, dtype=bool)
            trunc = np.zeros(self.E, dtype=bool)
        else:
            dist = _norm(self.TARGET_POS - self.pos)
            reward = np.sum(np.maximum(0, 2 - dist ** 4), axis=-1)
            term = np.sum(dist, axis=-1) < .0001
            xy = 1.5 if self.TASK == "hover" else 2.0
            out = (np.abs(self.pos[..., 0]) > xy) | (np.abs(self.pos[..., 1]) > xy) | (self.pos[..., 2] > 2.0) \
                | (np.abs(self.rpy[..., 0]) > .4) | (np.abs(self.rpy[..., 1]) > .4)
            trunc = out.any(axis=-1) | (self.step_counter / self.PYB_FREQ > self.EPISODE_LEN_SEC)
        self.step_counter = self.step_counter + self.S
        term_obs = None
        if self.auto_reset:
            done = term | trunc
            term_obs = obs.copy()
            if done.any():
                self.reset(done)
                obs = self.obs12()
        return obs, reward, term, trunc, term_obs
