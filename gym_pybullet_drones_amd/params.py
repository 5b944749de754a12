"""Airframe constants: URDF -> python floats -> the by-value `GpdParams` kernel argument.

Restates what `BaseAviary.__init__` derives (`envs/BaseAviary.py:74,97-128`), what
`_parseURDFParameters` reads (`:985-1017`) and the DSLPID gains
(`control/DSLPIDControl.py:37-60`, `control/BaseControl.py:35-39`).  The parser looks
elements up by tag name (the reference indexes by position); both agree on the
reference's files and on the compact files shipped in `assets/`.
"""
import ctypes
import math
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

from .utils.enums import DroneModel

ASSETS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")
G = 9.8  # envs/BaseAviary.py:74, control/BaseControl.py:20 -- not 9.81


def urdf_path(drone_model: DroneModel) -> str:
    return os.path.join(ASSETS_DIR, drone_model.value + ".urdf")


def parse_urdf(path: str) -> dict:
    """Read the physical constants of one airframe file."""
    root = ET.parse(path).getroot()
    props = root.find("properties").attrib
    base = root.find("link")          # first link = base_link
    inertial = base.find("inertial")
    inertia = inertial.find("inertia").attrib
    coll = base.find("collision")
    cyl = coll.find("geometry").find("cylinder").attrib
    offs = []
    for k in range(4):
        link = root.find(f"link[@name='prop{k}_link']")
        xyz = [float(s) for s in link.find("inertial").find("origin").attrib["xyz"].split()]
        offs.append(xyz)
    out = {
        "M": float(inertial.find("mass").attrib["value"]),
        "L": float(props["arm"]),
        "THRUST2WEIGHT_RATIO": float(props["thrust2weight"]),
        "IXX": float(inertia["ixx"]), "IYY": float(inertia["iyy"]), "IZZ": float(inertia["izz"]),
        "KF": float(props["kf"]), "KM": float(props["km"]),
        "COLLISION_H": float(cyl["length"]), "COLLISION_R": float(cyl["radius"]),
        "COLLISION_Z_OFFSET": [float(s) for s in coll.find("origin").attrib["xyz"].split()][2],
        "MAX_SPEED_KMH": float(props["max_speed_kmh"]),
        "GND_EFF_COEFF": float(props["gnd_eff_coeff"]),
        "PROP_RADIUS": float(props["prop_radius"]),
        "DRAG_COEFF_XY": float(props["drag_coeff_xy"]), "DRAG_COEFF_Z": float(props["drag_coeff_z"]),
        "DW_COEFF_1": float(props["dw_coeff_1"]), "DW_COEFF_2": float(props["dw_coeff_2"]),
        "DW_COEFF_3": float(props["dw_coeff_3"]),
        "PROP_OFFSETS": np.array(offs, dtype=np.float64),
    }
    return out


class GpdParams(ctypes.Structure):
    """ctypes mirror of `struct GpdParams` (include/gpd.h)."""
    _fields_ = [
        ("drone_model", ctypes.c_int32),
        ("M", ctypes.c_float), ("inv_M", ctypes.c_float), ("L", ctypes.c_float), ("KF", ctypes.c_float),
        ("KM", ctypes.c_float),
        ("GRAVITY", ctypes.c_float),
        ("J", ctypes.c_float * 3), ("J_INV", ctypes.c_float * 3),
        ("prop_x", ctypes.c_float * 4), ("prop_y", ctypes.c_float * 4),
        ("gnd_eff_coeff", ctypes.c_float), ("prop_radius", ctypes.c_float), ("gnd_eff_h_clip", ctypes.c_float),
        ("drag_coeff", ctypes.c_float * 3), ("dw_coeff", ctypes.c_float * 3),
        ("hover_rpm", ctypes.c_float), ("max_rpm", ctypes.c_float),
        ("hover_thrust", ctypes.c_float), ("hover_resid", ctypes.c_float), ("km_over_kf", ctypes.c_float),
        ("pid_gravity", ctypes.c_float), ("pid_kf", ctypes.c_float), ("pid_inv_4kf", ctypes.c_float),
        ("p_for", ctypes.c_float * 3), ("i_for", ctypes.c_float * 3), ("d_for", ctypes.c_float * 3),
        ("p_tor", ctypes.c_float * 3), ("i_tor", ctypes.c_float * 3), ("d_tor", ctypes.c_float * 3),
        ("mixer", ctypes.c_float * 12),
        ("pwm2rpm_scale", ctypes.c_float), ("inv_pwm2rpm_scale", ctypes.c_float), ("pwm2rpm_const", ctypes.c_float),
        ("min_pwm", ctypes.c_float), ("max_pwm", ctypes.c_float),
        ("speed_limit", ctypes.c_float), ("ground_z", ctypes.c_float),
    ]


#: DSLPID mixers, control/DSLPIDControl.py:47-60
MIXER = {
    DroneModel.CF2X: np.array([[-.5, -.5, -1], [-.5, .5, 1], [.5, .5, -1], [.5, -.5, 1]], dtype=np.float64),
    DroneModel.CF2P: np.array([[0, -1, -1], [1, 0, 1], [0, 1, -1], [-1, 0, 1]], dtype=np.float64),
}


@dataclass
class PIDGains:
    """DSLPIDControl coefficients (control/DSLPIDControl.py:37-46)."""
    P_COEFF_FOR: np.ndarray = field(default_factory=lambda: np.array([.4, .4, 1.25]))
    I_COEFF_FOR: np.ndarray = field(default_factory=lambda: np.array([.05, .05, .05]))
    D_COEFF_FOR: np.ndarray = field(default_factory=lambda: np.array([.2, .2, .5]))
    P_COEFF_TOR: np.ndarray = field(default_factory=lambda: np.array([70000., 70000., 60000.]))
    I_COEFF_TOR: np.ndarray = field(default_factory=lambda: np.array([.0, .0, 500.]))
    D_COEFF_TOR: np.ndarray = field(default_factory=lambda: np.array([20000., 20000., 12000.]))
    PWM2RPM_SCALE: float = 0.2685
    PWM2RPM_CONST: float = 4070.3
    MIN_PWM: float = 20000
    MAX_PWM: float = 65535


class DroneParams:
    """All constants of one airframe, in float64, with the reference's attribute names."""

    def __init__(self, drone_model: DroneModel = DroneModel.CF2X, urdf: str = None, g: float = G):
        self.DRONE_MODEL = drone_model
        self.URDF_PATH = urdf or urdf_path(drone_model)
        u = parse_urdf(self.URDF_PATH)
        self.G = g
        self.M = u["M"]
        self.L = u["L"]
        self.THRUST2WEIGHT_RATIO = u["THRUST2WEIGHT_RATIO"]
        self.J = np.diag([u["IXX"], u["IYY"], u["IZZ"]])
        self.J_INV = np.linalg.inv(self.J)
        self.KF = u["KF"]
        self.KM = u["KM"]
        self.COLLISION_H = u["COLLISION_H"]
        self.COLLISION_R = u["COLLISION_R"]
        self.COLLISION_Z_OFFSET = u["COLLISION_Z_OFFSET"]
        self.MAX_SPEED_KMH = u["MAX_SPEED_KMH"]
        self.GND_EFF_COEFF = u["GND_EFF_COEFF"]
        self.PROP_RADIUS = u["PROP_RADIUS"]
        self.DRAG_COEFF = np.array([u["DRAG_COEFF_XY"], u["DRAG_COEFF_XY"], u["DRAG_COEFF_Z"]])
        self.DW_COEFF_1 = u["DW_COEFF_1"]
        self.DW_COEFF_2 = u["DW_COEFF_2"]
        self.DW_COEFF_3 = u["DW_COEFF_3"]
        self.PROP_OFFSETS = u["PROP_OFFSETS"]
        # derived, envs/BaseAviary.py:117-128
        self.GRAVITY = self.G * self.M
        self.HOVER_RPM = np.sqrt(self.GRAVITY / (4 * self.KF))
        self.MAX_RPM = np.sqrt((self.THRUST2WEIGHT_RATIO * self.GRAVITY) / (4 * self.KF))
        self.MAX_THRUST = 4 * self.KF * self.MAX_RPM ** 2
        if drone_model == DroneModel.CF2P:
            self.MAX_XY_TORQUE = self.L * self.KF * self.MAX_RPM ** 2
        else:
            self.MAX_XY_TORQUE = (2 * self.L * self.KF * self.MAX_RPM ** 2) / np.sqrt(2)
        self.MAX_Z_TORQUE = 2 * self.KM * self.MAX_RPM ** 2
        self.GND_EFF_H_CLIP = 0.25 * self.PROP_RADIUS * np.sqrt(
            (15 * self.MAX_RPM ** 2 * self.KF * self.GND_EFF_COEFF) / self.MAX_THRUST)
        # ActionType.VEL, envs/BaseRLAviary.py:94-95
        self.SPEED_LIMIT = 0.03 * self.MAX_SPEED_KMH * (1000 / 3600)

    def default_init_xyzs(self, num_drones: int) -> np.ndarray:
        """envs/BaseAviary.py:194-197."""
        i = np.arange(num_drones, dtype=np.float64)
        z = np.ones(num_drones) * (self.COLLISION_H / 2 - self.COLLISION_Z_OFFSET + .1)
        return np.vstack([i * 4 * self.L, i * 4 * self.L, z]).transpose().reshape(num_drones, 3)

    def to_struct(self, pid_model: DroneModel = DroneModel.CF2X, pid_g: float = G,
                  gains: PIDGains = None, pid_params: "DroneParams" = None) -> GpdParams:
        """Pack into the C struct.  `pid_model` selects the controller airframe: the RL aviaries
        always build CF2X controllers (envs/BaseRLAviary.py:75-76), standalone DSLPIDControl uses
        its own `drone_model`."""
        gains = gains or PIDGains()
        s = GpdParams()
        s.drone_model = self.DRONE_MODEL.code
        s.M, s.L, s.KF, s.KM, s.GRAVITY = self.M, self.L, self.KF, self.KM, self.GRAVITY
        s.inv_M = 1.0 / self.M
        for k in range(3):
            s.J[k] = self.J[k, k]
            s.J_INV[k] = self.J_INV[k, k]
            s.drag_coeff[k] = self.DRAG_COEFF[k]
        for k in range(4):
            s.prop_x[k] = self.PROP_OFFSETS[k, 0]
            s.prop_y[k] = self.PROP_OFFSETS[k, 1]
        s.gnd_eff_coeff, s.prop_radius, s.gnd_eff_h_clip = self.GND_EFF_COEFF, self.PROP_RADIUS, self.GND_EFF_H_CLIP
        s.dw_coeff[0], s.dw_coeff[1], s.dw_coeff[2] = self.DW_COEFF_1, self.DW_COEFF_2, self.DW_COEFF_3
        s.hover_rpm, s.max_rpm = self.HOVER_RPM, self.MAX_RPM
        h32 = float(np.float32(self.HOVER_RPM))
        s.hover_thrust = self.GRAVITY / 4
        s.hover_resid = self.KF * h32 * h32 - self.GRAVITY / 4
        s.km_over_kf = self.KM / self.KF
        if pid_model in MIXER:
            cp = pid_params or (self if pid_model == self.DRONE_MODEL else DroneParams(pid_model))
            s.pid_gravity = pid_g * cp.M
            s.pid_kf = cp.KF
            s.pid_inv_4kf = 1.0 / (4.0 * cp.KF)
            for k in range(3):
                s.p_for[k], s.i_for[k], s.d_for[k] = gains.P_COEFF_FOR[k], gains.I_COEFF_FOR[k], gains.D_COEFF_FOR[k]
                s.p_tor[k], s.i_tor[k], s.d_tor[k] = gains.P_COEFF_TOR[k], gains.I_COEFF_TOR[k], gains.D_COEFF_TOR[k]
            for k in range(12):
                s.mixer[k] = MIXER[pid_model].reshape(-1)[k]
            s.pwm2rpm_scale, s.pwm2rpm_const = gains.PWM2RPM_SCALE, gains.PWM2RPM_CONST
            s.inv_pwm2rpm_scale = 1.0 / gains.PWM2RPM_SCALE
            s.min_pwm, s.max_pwm = gains.MIN_PWM, gains.MAX_PWM
        s.speed_limit = self.SPEED_LIMIT
        s.ground_z = self.COLLISION_H / 2 - self.COLLISION_Z_OFFSET      # base-link height with the cylinder on the plane
        return s


def trunc_counter(episode_len_sec: float, pyb_freq: int) -> int:
    """Largest integer c with `c / pyb_freq > episode_len_sec` False (float64 semantics of
    `self.step_counter/self.PYB_FREQ > self.EPISODE_LEN_SEC`, envs/HoverAviary.py:114), so the
    kernel can test `step_counter > c` in integers."""
    c = int(math.floor(episode_len_sec * pyb_freq))
    while (c + 1) / pyb_freq <= episode_len_sec:
        c += 1
    while c / pyb_freq > episode_len_sec:
        c -= 1
    return c


def euler_to_quat(rpy) -> np.ndarray:
    """Bullet's getQuaternionFromEuler, float64, batched over leading dims."""
    rpy = np.asarray(rpy, dtype=np.float64)
    h = rpy * 0.5
    cr, sr = np.cos(h[..., 0]), np.sin(h[..., 0])
    cp, sp = np.cos(h[..., 1]), np.sin(h[..., 1])
    cy, sy = np.cos(h[..., 2]), np.sin(h[..., 2])
    q = np.stack([sr * cp * cy - cr * sp * sy,
                  cr * sp * cy + sr * cp * sy,
                  cr * cp * sy - sr * sp * cy,
                  cr * cp * cy + sr * sp * sy], axis=-1)
    return q / np.linalg.norm(q, axis=-1, keepdims=True)
