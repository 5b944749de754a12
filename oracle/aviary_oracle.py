"""Loop-structured float64 restatement of the reference's hot path — the CPU "port".

TEST INFRASTRUCTURE (see oracle/__init__.py): imported only by tests/, by
`__graft_entry__.smoke()` and by `bench.py`'s `cpu_baseline` leg.

It keeps the reference's structure on purpose (one Python iteration per drone, numpy on 3- and
4-vectors, float64) so that timing it says something about the reference's own cost profile:
  * constants          envs/BaseAviary.py:74-128, 985-1017; control/BaseControl.py:35-39
  * step ordering      envs/BaseAviary.py:341-383
  * integrator         envs/BaseAviary.py:815-892  (_dynamics, _integrateQ)
  * add-on forces      envs/BaseAviary.py:715-811  re-expressed inside the explicit integrator
                       (a DERIVATION — the reference applies them only through
                       PyBullet's integrator.  The force *formulas* are pinned by
                       tests/golden/force_models_*.npz.)
  * cache/state vector envs/BaseAviary.py:509-519, 541-561
  * PID                control/DSLPIDControl.py:82-259
  * action mapping/obs envs/BaseRLAviary.py:160-239, 307-320; envs/CtrlAviary.py:106-140
  * tasks              envs/HoverAviary.py:51-132; envs/MultiHoverAviary.py:57-145
The PyBullet state store is replaced by plain arrays (values are kept exactly as written).
"""
import math
import xml.etree.ElementTree as etxml
from collections import deque

import numpy as np
from scipy.spatial.transform import Rotation

from . import bullet_math as bm

PHYS_GND, PHYS_DRAG, PHYS_DW = 1, 2, 4
#: EXTENSION, not in the reference's Physics.DYN (include/gpd.h, GPD_PHYS_GROUND): the plane at z = 0.  A drone whose
#: collision cylinder would sink below it is put back on it (base-link height COLLISION_H/2 - COLLISION_Z_OFFSET), loses
#: its downward velocity and sticks laterally.  UNPINNED by construction: the reference resolves ground contact only
#: through Bullet's solver (Physics.PYB*), which is not available; this restates the kernel's definition in float64.
PHYS_GROUND = 8
#: EXTENSION, not in the reference's Physics.DYN either (include/gpd.h, GPD_PHYS_DAMP): Bullet's default damping of a body that
#: `p.loadURDF` created (envs/BaseAviary.py:488-494 loads the drone without `useMaximalCoordinates`, i.e. as a btMultiBody), which
#: every `Physics.PYB*` run of the reference carries and no line of the reference mentions.  THIRD-PARTY ORIGIN: Bullet 3.2.x
#: (`pybullet ^3.2.7`, pyproject.toml:19; sources not under /root/reference), src/BulletDynamics/Featherstone/btMultiBody.cpp:
#: the constructor sets m_linearDamping = m_angularDamping = 0.04 (PyBullet's documented `changeDynamics` defaults, which the
#: reference never changes), and computeAccelerationsArticulatedBodyAlgorithmMultiDof adds to the base's bias force
#:     m_baseMass * v * (K1 + K2 |v|)   and   m_baseInertia (.) w * (K1 + K2 |w|),   K1 = K2 = the damping coefficient
#: ("adding damping terms (only)"; DAMPING_K1_LINEAR = DAMPING_K2_LINEAR = m_linearDamping, same for ANGULAR), i.e.
#:     dv/dt -= d (1 + |v|) v,      dw/dt -= d (1 + |w|) w,      d = 0.04,
#: evaluated on the velocities at the start of the sub-step, like every other force of the explicit integrator.  (A body in
#: maximal coordinates would instead get btRigidBody::applyDamping, v *= (1 - d)^dt: the same to first order at |v| << 1.)
#: PARITY UNPINNED: restated from knowledge of the Bullet sources; there is no Bullet here to run it against.
PHYS_DAMP = 16
BULLET_DAMPING = 0.04
ACT_DIM = {"rpm": 4, "pid": 3, "vel": 4, "one_d_rpm": 1, "one_d_pid": 1, "raw_rpm": 4}


class UrdfConstants:
    """Positional URDF reader + derived constants (envs/BaseAviary.py:992-1015, :117-128)."""

    def __init__(self, urdf_path, drone_model, g=9.8):
        tree = etxml.parse(urdf_path).getroot()
        self.DRONE_MODEL = drone_model          # "cf2x" | "cf2p" | "racer"
        self.G = g
        self.M = float(tree[1][0][1].attrib['value'])
        self.L = float(tree[0].attrib['arm'])
        self.THRUST2WEIGHT_RATIO = float(tree[0].attrib['thrust2weight'])
        ixx, iyy, izz = (float(tree[1][0][2].attrib[k]) for k in ('ixx', 'iyy', 'izz'))
        self.J = np.diag([ixx, iyy, izz])
        self.J_INV = np.linalg.inv(self.J)
        self.KF = float(tree[0].attrib['kf'])
        self.KM = float(tree[0].attrib['km'])
        self.COLLISION_H = float(tree[1][2][1][0].attrib['length'])
        self.COLLISION_R = float(tree[1][2][1][0].attrib['radius'])
        self.COLLISION_Z_OFFSET = [float(s) for s in tree[1][2][0].attrib['xyz'].split(' ')][2]
        self.MAX_SPEED_KMH = float(tree[0].attrib['max_speed_kmh'])
        self.GND_EFF_COEFF = float(tree[0].attrib['gnd_eff_coeff'])
        self.PROP_RADIUS = float(tree[0].attrib['prop_radius'])
        dxy, dz = float(tree[0].attrib['drag_coeff_xy']), float(tree[0].attrib['drag_coeff_z'])
        self.DRAG_COEFF = np.array([dxy, dxy, dz])
        self.DW_COEFF_1 = float(tree[0].attrib['dw_coeff_1'])
        self.DW_COEFF_2 = float(tree[0].attrib['dw_coeff_2'])
        self.DW_COEFF_3 = float(tree[0].attrib['dw_coeff_3'])
        # rotor link origins (prop0..3 are robot children 2,4,6,8: link, joint, link, joint, ...)
        self.PROP_OFFSETS = np.array([[float(s) for s in tree[2 + 2 * k][0][0].attrib['xyz'].split()]
                                      for k in range(4)])
        self.GRAVITY = self.G * self.M
        self.HOVER_RPM = np.sqrt(self.GRAVITY / (4 * self.KF))
        self.MAX_RPM = np.sqrt((self.THRUST2WEIGHT_RATIO * self.GRAVITY) / (4 * self.KF))
        self.MAX_THRUST = (4 * self.KF * self.MAX_RPM ** 2)
        if drone_model == "cf2p":
            self.MAX_XY_TORQUE = (self.L * self.KF * self.MAX_RPM ** 2)
        else:
            self.MAX_XY_TORQUE = (2 * self.L * self.KF * self.MAX_RPM ** 2) / np.sqrt(2)
        self.MAX_Z_TORQUE = (2 * self.KM * self.MAX_RPM ** 2)
        self.GND_EFF_H_CLIP = 0.25 * self.PROP_RADIUS * np.sqrt(
            (15 * self.MAX_RPM ** 2 * self.KF * self.GND_EFF_COEFF) / self.MAX_THRUST)
        self.SPEED_LIMIT = 0.03 * self.MAX_SPEED_KMH * (1000 / 3600)


class OracleDSLPID:
    """control/DSLPIDControl.py:19-259 for one drone."""

    def __init__(self, consts: UrdfConstants, g=9.8):
        assert consts.DRONE_MODEL in ("cf2x", "cf2p")
        self.GRAVITY = g * consts.M
        self.KF = consts.KF
        self.P_COEFF_FOR = np.array([.4, .4, 1.25])
        self.I_COEFF_FOR = np.array([.05, .05, .05])
        self.D_COEFF_FOR = np.array([.2, .2, .5])
        self.P_COEFF_TOR = np.array([70000., 70000., 60000.])
        self.I_COEFF_TOR = np.array([.0, .0, 500.])
        self.D_COEFF_TOR = np.array([20000., 20000., 12000.])
        self.PWM2RPM_SCALE = 0.2685
        self.PWM2RPM_CONST = 4070.3
        self.MIN_PWM = 20000
        self.MAX_PWM = 65535
        if consts.DRONE_MODEL == "cf2x":
            self.MIXER_MATRIX = np.array([[-.5, -.5, -1], [-.5, .5, 1], [.5, .5, -1], [.5, -.5, 1]])
        else:
            self.MIXER_MATRIX = np.array([[0, -1, -1], [+1, 0, 1], [0, 1, -1], [-1, 0, 1]])
        self.reset()

    def reset(self):
        self.control_counter = 0
        self.last_rpy = np.zeros(3)
        self.integral_pos_e = np.zeros(3)
        self.integral_rpy_e = np.zeros(3)

    def computeControl(self, control_timestep, cur_pos, cur_quat, cur_vel, cur_ang_vel, target_pos,
                       target_rpy=np.zeros(3), target_vel=np.zeros(3), target_rpy_rates=np.zeros(3)):
        """One controller tick -> (rpm[4], pos_e[3], yaw_e).  `cur_ang_vel` is unused, as upstream."""
        dt = control_timestep
        self.control_counter += 1
        R = bm.matrix_from_quaternion(cur_quat)
        cur_rpy = np.array(bm.euler_from_quaternion(cur_quat))

        # -- position PID -> desired force vector, collective PWM, desired attitude (:187-205)
        e_p = target_pos - cur_pos
        e_v = target_vel - cur_vel
        acc = np.clip(self.integral_pos_e + e_p * dt, -2., 2.)
        acc[2] = np.clip(acc[2], -0.15, .15)
        self.integral_pos_e = acc
        f_des = self.P_COEFF_FOR * e_p + self.I_COEFF_FOR * acc + self.D_COEFF_FOR * e_v
        f_des = f_des + np.array([0, 0, self.GRAVITY])
        along_body_z = max(0., np.dot(f_des, R[:, 2]))
        base_pwm = (math.sqrt(along_body_z / (4 * self.KF)) - self.PWM2RPM_CONST) / self.PWM2RPM_SCALE
        zb = f_des / np.linalg.norm(f_des)
        heading = np.array([math.cos(target_rpy[2]), math.sin(target_rpy[2]), 0])
        yb = np.cross(zb, heading)
        yb = yb / np.linalg.norm(yb)
        xb = np.cross(yb, zb)
        R_des = np.vstack([xb, yb, zb]).transpose()
        des_euler = Rotation.from_matrix(R_des).as_euler('XYZ', degrees=False)

        # -- attitude PID on SO(3) -> torques -> mixer -> PWM -> RPM (:240-259).  The reference
        #    rebuilds the target rotation from des_euler through a quaternion whose components it
        #    unpacks as (w,x,y,z) but passes back in the same order, i.e. the identical 4-vector.
        R_des = Rotation.from_quat(Rotation.from_euler('XYZ', des_euler, degrees=False).as_quat()).as_matrix()
        skew = R_des.T @ R - R.T @ R_des
        e_R = np.array([skew[2, 1], skew[0, 2], skew[1, 0]])
        e_w = target_rpy_rates - (cur_rpy - self.last_rpy) / dt          # finite difference, no unwrap
        self.last_rpy = cur_rpy
        acc_r = np.clip(self.integral_rpy_e - e_R * dt, -1500., 1500.)
        acc_r[0:2] = np.clip(acc_r[0:2], -1., 1.)
        self.integral_rpy_e = acc_r
        tau = -self.P_COEFF_TOR * e_R + self.D_COEFF_TOR * e_w + self.I_COEFF_TOR * acc_r
        tau = np.clip(tau, -3200, 3200)
        pwm = np.clip(base_pwm + self.MIXER_MATRIX @ tau, self.MIN_PWM, self.MAX_PWM)
        return self.PWM2RPM_SCALE * pwm + self.PWM2RPM_CONST, e_p, des_euler[2] - cur_rpy[2]


class OracleAviary:
    """One aviary of `num_drones` drones: BaseAviary(DYN [+GND|DRAG|DW]) + BaseRLAviary + task.

    act  : "rpm" | "pid" | "vel" | "one_d_rpm" | "one_d_pid" (BaseRLAviary) or "raw_rpm" (CtrlAviary)
    task : "hover" | "multihover" | "none"
    """

    def __init__(self, urdf_path, drone_model="cf2x", num_drones=1, initial_xyzs=None, initial_rpys=None,
                 physics_flags=0, pyb_freq=240, ctrl_freq=240, act="rpm", task="none",
                 pid_urdf_path=None, episode_len_sec=8):
        self.C = UrdfConstants(urdf_path, drone_model)
        self.NUM_DRONES = num_drones
        self.PHYS = physics_flags
        self.PYB_FREQ, self.CTRL_FREQ = pyb_freq, ctrl_freq
        if pyb_freq % ctrl_freq != 0:
            raise ValueError("pyb_freq is not divisible by ctrl_freq")
        self.PYB_STEPS_PER_CTRL = int(pyb_freq / ctrl_freq)
        self.CTRL_TIMESTEP = 1. / ctrl_freq
        self.PYB_TIMESTEP = 1. / pyb_freq
        self.ACT, self.TASK = act, task
        self.EPISODE_LEN_SEC = episode_len_sec
        C = self.C
        if initial_xyzs is None:   # envs/BaseAviary.py:194-197
            self.INIT_XYZS = np.vstack([np.array([x * 4 * C.L for x in range(num_drones)]),
                                        np.array([y * 4 * C.L for y in range(num_drones)]),
                                        np.ones(num_drones) * (C.COLLISION_H / 2 - C.COLLISION_Z_OFFSET + .1)]
                                       ).transpose().reshape(num_drones, 3)
        else:
            self.INIT_XYZS = np.array(initial_xyzs, dtype=np.float64).reshape(num_drones, 3)
        self.INIT_RPYS = np.zeros((num_drones, 3)) if initial_rpys is None else \
            np.array(initial_rpys, dtype=np.float64).reshape(num_drones, 3)
        if task == "hover":
            self.TARGET_POS = np.array([[0, 0, 1.]])
        elif task == "multihover":
            self.TARGET_POS = self.INIT_XYZS + np.array([[0, 0, 1 / (i + 1)] for i in range(num_drones)])
        else:
            self.TARGET_POS = None
        # action buffer, envs/BaseRLAviary.py:66-67,153-154 (filled once, never cleared)
        self.ACTION_BUFFER_SIZE = int(ctrl_freq // 2)
        self.action_buffer = deque(maxlen=self.ACTION_BUFFER_SIZE)
        if act != "raw_rpm":
            for _ in range(self.ACTION_BUFFER_SIZE):
                self.action_buffer.append(np.zeros((num_drones, ACT_DIM[act])))
        if act in ("pid", "vel", "one_d_pid"):   # always CF2X controllers, envs/BaseRLAviary.py:75-76
            pc = UrdfConstants(pid_urdf_path or urdf_path, "cf2x")
            self.ctrl = [OracleDSLPID(pc) for _ in range(num_drones)]
        self._housekeeping()

    # ---- reset --------------------------------------------------------------------------------
    def _housekeeping(self):
        n = self.NUM_DRONES
        self.step_counter = 0
        self.last_clipped_action = np.zeros((n, 4))
        self.pos = self.INIT_XYZS.astype(np.float64).copy()
        self.quat = np.array([bm.quaternion_from_euler(self.INIT_RPYS[i]) for i in range(n)])
        self.vel = np.zeros((n, 3))
        self.ang_v = np.zeros((n, 3))
        self.rpy_rates = np.zeros((n, 3))
        self.rpy = np.array([bm.euler_from_quaternion(self.quat[i]) for i in range(n)])
        # the "store": what the PyBullet body holds between cache refreshes
        self._s_pos, self._s_quat = self.pos.copy(), self.quat.copy()
        self._s_vel, self._s_ang_v = self.vel.copy(), self.ang_v.copy()

    def reset(self):
        self._housekeeping()
        return self._computeObs()

    def _refresh_cache(self):   # envs/BaseAviary.py:509-519
        for i in range(self.NUM_DRONES):
            self.pos[i], self.quat[i] = self._s_pos[i], self._s_quat[i]
            self.rpy[i] = bm.euler_from_quaternion(self.quat[i])
            self.vel[i], self.ang_v[i] = self._s_vel[i], self._s_ang_v[i]

    def _getDroneStateVector(self, i):
        return np.hstack([self.pos[i], self.quat[i], self.rpy[i], self.vel[i], self.ang_v[i],
                          self.last_clipped_action[i]]).reshape(20,)

    # ---- add-on force formulas (what the reference hands to p.applyExternalForce) ---------------
    def ground_effect_forces(self, rpm, i):
        """envs/BaseAviary.py:739-743 -> 4 body-z forces at the rotors (zeros when tilted > pi/2)."""
        C = self.C
        R = bm.matrix_from_quaternion(self.quat[i])
        prop_heights = np.array([self.pos[i, 2] + R[2, 0] * C.PROP_OFFSETS[k, 0] + R[2, 1] * C.PROP_OFFSETS[k, 1]
                                 + R[2, 2] * C.PROP_OFFSETS[k, 2] for k in range(4)])
        prop_heights = np.clip(prop_heights, C.GND_EFF_H_CLIP, np.inf)
        gnd = np.array(rpm ** 2) * C.KF * C.GND_EFF_COEFF * (C.PROP_RADIUS / (4 * prop_heights)) ** 2
        if np.abs(self.rpy[i, 0]) < np.pi / 2 and np.abs(self.rpy[i, 1]) < np.pi / 2:
            return gnd
        return np.zeros(4)

    def drag_force_body(self, rpm, i):
        """envs/BaseAviary.py:771-774 -> force in the BODY frame (applied LINK_FRAME at the COM)."""
        R = bm.matrix_from_quaternion(self.quat[i])
        drag_factors = -1 * self.C.DRAG_COEFF * np.sum(np.array(2 * np.pi * rpm / 60))
        return np.dot(R.T, drag_factors * np.array(self.vel[i]))

    def downwash_force(self, i):
        """envs/BaseAviary.py:798-804 -> scalar body-z force on drone i from every drone above it."""
        C = self.C
        total = 0.0
        for j in range(self.NUM_DRONES):
            delta_z = self.pos[j, 2] - self.pos[i, 2]
            delta_xy = np.linalg.norm(np.array(self.pos[j, 0:2]) - np.array(self.pos[i, 0:2]))
            if delta_z > 0 and delta_xy < 10:
                alpha = C.DW_COEFF_1 * (C.PROP_RADIUS / (4 * delta_z)) ** 2
                beta = C.DW_COEFF_2 * delta_z + C.DW_COEFF_3
                total += -alpha * np.exp(-.5 * (delta_xy / beta) ** 2)
        return total

    # ---- integrator -----------------------------------------------------------------------------
    def _dynamics(self, rpm, i):
        """envs/BaseAviary.py:831-877 with the add-on terms folded in per SURVEY.md App. A.4."""
        C, h = self.C, self.PYB_TIMESTEP
        x, q, v, w = self.pos[i], self.quat[i], self.vel[i], self.rpy_rates[i]   # cached state
        R = bm.matrix_from_quaternion(q)
        sq = np.array(rpm ** 2)
        f = sq * C.KF                                     # rotor thrusts
        if self.PHYS & PHYS_GND:                          # extra per-rotor thrust, CURRENT rpm (:356,365)
            f = f + self.ground_effect_forces(rpm, i)
        fz_body = np.sum(f)
        if self.PHYS & PHYS_DW:                           # body-z force at the COM (:805-811)
            fz_body = fz_body + self.downwash_force(i)
        F_world = R @ np.array([0, 0, fz_body]) - np.array([0, 0, C.GRAVITY])
        if self.PHYS & PHYS_DRAG:                         # body-frame force at the COM, PREVIOUS action (:359,366)
            F_world = F_world + R @ self.drag_force_body(self.last_clipped_action[i, :], i)
        yaw_t = sq * C.KM
        if C.DRONE_MODEL == "racer":
            yaw_t = -yaw_t
        tz = -yaw_t[0] + yaw_t[1] - yaw_t[2] + yaw_t[3]
        if C.DRONE_MODEL == "cf2p":
            tx, ty = (f[1] - f[3]) * C.L, (-f[0] + f[2]) * C.L
        else:
            arm = C.L / np.sqrt(2)
            tx = (f[0] + f[1] - f[2] - f[3]) * arm
            ty = (-f[0] + f[1] + f[2] - f[3]) * arm
            if C.DRONE_MODEL == "cf2x":
                tx = -tx
        tau = np.array([tx, ty, tz]) - np.cross(w, C.J @ w)
        w_dot = C.J_INV @ tau
        a = F_world / C.M
        if self.PHYS & PHYS_DAMP:                         # extension (see PHYS_DAMP): Bullet's default multibody damping
            a = a - BULLET_DAMPING * (1.0 + np.linalg.norm(v)) * v
            w_dot = w_dot - BULLET_DAMPING * (1.0 + np.linalg.norm(w)) * w
        v = v + h * a                                     # semi-implicit Euler: x uses the NEW v
        w = w + h * w_dot
        x = x + h * v
        if self.PHYS & PHYS_GROUND:                       # extension (see PHYS_GROUND): the plane at z = 0
            z_rest = C.COLLISION_H / 2 - C.COLLISION_Z_OFFSET
            if x[2] < z_rest or (x[2] <= z_rest and v[2] < 0):      # (second clause: the tie x_z == z_rest, see include/gpd.h)
                x = np.array([x[0], x[1], z_rest])
                v = np.array([0.0, 0.0, max(v[2], 0.0)])
        q = self._integrateQ(q, w, h)
        self._s_pos[i], self._s_quat[i] = x, q            # not renormalised
        self._s_vel[i], self._s_ang_v[i] = v, R @ w       # world rates use the PRE-update rotation
        self.rpy_rates[i, :] = w

    @staticmethod
    def _integrateQ(quat, omega, dt):
        """envs/BaseAviary.py:879-892: exact exponential of a constant body rate over dt."""
        n = np.linalg.norm(omega)
        if np.isclose(n, 0):
            return quat
        p, q, r = omega
        half_skew = 0.5 * np.array([[0, r, -q, p], [-r, 0, p, q], [q, -p, 0, r], [-p, -q, -r, 0]])
        th = n * dt / 2
        return (np.eye(4) * np.cos(th) + 2 / n * half_skew * np.sin(th)) @ quat

    # ---- action mapping, envs/BaseRLAviary.py:187-239 and envs/CtrlAviary.py:140 -------------
    @staticmethod
    def _calculateNextStep(current_position, destination, step_size=1):   # envs/BaseAviary.py:1132-1150
        direction = destination - current_position
        distance = np.linalg.norm(direction)
        if distance <= step_size:
            return destination
        return current_position + direction / distance * step_size

    def _preprocessAction(self, action):
        C = self.C
        if self.ACT == "raw_rpm":
            return np.array([np.clip(action[i, :], 0, C.MAX_RPM) for i in range(self.NUM_DRONES)])
        self.action_buffer.append(action)
        rpm = np.zeros((self.NUM_DRONES, 4))
        for k in range(action.shape[0]):
            target = action[k, :]
            if self.ACT == "rpm":
                rpm[k, :] = np.array(C.HOVER_RPM * (1 + 0.05 * target))
            elif self.ACT == "one_d_rpm":
                rpm[k, :] = np.repeat(C.HOVER_RPM * (1 + 0.05 * target), 4)
            else:
                state = self._getDroneStateVector(k)
                if self.ACT == "pid":
                    next_pos = self._calculateNextStep(state[0:3], target, 1)
                    rpm[k, :], _, _ = self.ctrl[k].computeControl(self.CTRL_TIMESTEP, state[0:3], state[3:7],
                                                                  state[10:13], state[13:16], next_pos)
                elif self.ACT == "vel":
                    if np.linalg.norm(target[0:3]) != 0:
                        v_unit_vector = target[0:3] / np.linalg.norm(target[0:3])
                    else:
                        v_unit_vector = np.zeros(3)
                    rpm[k, :], _, _ = self.ctrl[k].computeControl(
                        self.CTRL_TIMESTEP, state[0:3], state[3:7], state[10:13], state[13:16],
                        target_pos=state[0:3], target_rpy=np.array([0, 0, state[9]]),
                        target_vel=C.SPEED_LIMIT * np.abs(target[3]) * v_unit_vector)
                elif self.ACT == "one_d_pid":
                    rpm[k, :], _, _ = self.ctrl[k].computeControl(
                        self.CTRL_TIMESTEP, state[0:3], state[3:7], state[10:13], state[13:16],
                        target_pos=state[0:3] + 0.1 * np.array([0, 0, target[0]]))
                else:
                    raise ValueError(self.ACT)
        return rpm

    # ---- obs / task, envs/BaseRLAviary.py:307-320, HoverAviary.py, MultiHoverAviary.py ----------
    def obs12(self):
        out = np.zeros((self.NUM_DRONES, 12))
        for i in range(self.NUM_DRONES):
            s = self._getDroneStateVector(i)
            out[i, :] = np.hstack([s[0:3], s[7:10], s[10:13], s[13:16]]).reshape(12,)
        return out

    def _computeObs(self):
        if self.ACT == "raw_rpm":   # CtrlAviary: (N,20) state
            return np.array([self._getDroneStateVector(i) for i in range(self.NUM_DRONES)])
        ret = self.obs12()
        for i in range(self.ACTION_BUFFER_SIZE):
            ret = np.hstack([ret, np.array([self.action_buffer[i][j, :] for j in range(self.NUM_DRONES)])])
        return ret

    def _computeReward(self):
        if self.TASK == "none":
            return -1
        ret = 0
        for i in range(self.NUM_DRONES):
            ret += max(0, 2 - np.linalg.norm(self.TARGET_POS[i, :] - self.pos[i]) ** 4)
        return ret

    def _computeTerminated(self):
        if self.TASK == "none":
            return False
        dist = 0
        for i in range(self.NUM_DRONES):
            dist += np.linalg.norm(self.TARGET_POS[i, :] - self.pos[i])
        return bool(dist < .0001)

    def _computeTruncated(self):
        if self.TASK == "none":
            return False
        xy = 1.5 if self.TASK == "hover" else 2.0
        for i in range(self.NUM_DRONES):
            s = self._getDroneStateVector(i)
            if (abs(s[0]) > xy or abs(s[1]) > xy or s[2] > 2.0 or abs(s[7]) > .4 or abs(s[8]) > .4):
                return True
        return bool(self.step_counter / self.PYB_FREQ > self.EPISODE_LEN_SEC)

    # ---- step, envs/BaseAviary.py:341-383 -------------------------------------------------------
    def step(self, action):
        clipped_action = np.reshape(self._preprocessAction(np.asarray(action, dtype=np.float64)), (self.NUM_DRONES, 4))
        for _ in range(self.PYB_STEPS_PER_CTRL):
            if self.PYB_STEPS_PER_CTRL > 1:
                self._refresh_cache()
            for i in range(self.NUM_DRONES):
                self._dynamics(clipped_action[i, :], i)
            self.last_clipped_action = clipped_action
        self._refresh_cache()
        obs = self._computeObs()
        reward = self._computeReward()
        terminated = self._computeTerminated()
        truncated = self._computeTruncated()
        self.step_counter = self.step_counter + (1 * self.PYB_STEPS_PER_CTRL)
        return obs, reward, terminated, truncated
