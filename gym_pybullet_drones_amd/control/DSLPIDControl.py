"""`DSLPIDControl`: the Crazyflie cascaded PID (reference `control/DSLPIDControl.py:9-287`) on the GPU.

`DSLPIDControl` is the drop-in single-drone class (numpy in / numpy out, same signature and
return triple); `DSLPIDControlBatch` runs n independent controllers per call on torch tensors.
Both call `gpd_pid` (include/gpd.h) — the same device function the fused step kernel uses for
`ActionType.PID/VEL/ONE_D_PID`.  Controller state (integral_pos_e, last_rpy, integral_rpy_e)
lives in a [9][ld] float32 block: in HBM for the batch class, in page-locked host memory the
device addresses directly for the single-drone class, whose call is then ONE library call (`gpd_pid_sync`:
the launch and the wait for it; inputs written and outputs read by the host in place; no copies).
"""
import ctypes

import numpy as np
import torch

from .. import _native
from ..params import MIXER, PIDGains
from ..utils.enums import ActionType, DroneModel
from .BaseControl import BaseControl


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _f32(x, n, k, device):
    if x is None:
        return None
    t = torch.as_tensor(np.asarray(x, dtype=np.float32) if not torch.is_tensor(x) else x, dtype=torch.float32, device=device)
    return t.reshape(n, k).contiguous()


class DSLPIDControlBatch(BaseControl):
    """n independent DSLPID controllers, one lane each."""

    def __init__(self, num_controllers: int, drone_model: DroneModel = DroneModel.CF2X, g: float = 9.8, device=None,
                 host_visible: bool = False):
        if drone_model not in (DroneModel.CF2X, DroneModel.CF2P):
            raise ValueError("[ERROR] in DSLPIDControl.__init__(), DSLPIDControl requires DroneModel.CF2X or DroneModel.CF2P")
        self.lib = _native.lib()
        if device is None:
            if not torch.cuda.is_available():
                raise _native.GpdError("DSLPIDControl runs on an MI355X only: no CUDA/HIP device available")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.n = int(num_controllers)
        self.ld = (self.n + 63) // 64 * 64
        gains = PIDGains()
        self.P_COEFF_FOR, self.I_COEFF_FOR, self.D_COEFF_FOR = gains.P_COEFF_FOR, gains.I_COEFF_FOR, gains.D_COEFF_FOR
        self.P_COEFF_TOR, self.I_COEFF_TOR, self.D_COEFF_TOR = gains.P_COEFF_TOR, gains.I_COEFF_TOR, gains.D_COEFF_TOR
        self.PWM2RPM_SCALE, self.PWM2RPM_CONST = gains.PWM2RPM_SCALE, gains.PWM2RPM_CONST
        self.MIN_PWM, self.MAX_PWM = gains.MIN_PWM, gains.MAX_PWM
        self.MIXER_MATRIX = MIXER[drone_model].copy()
        if host_visible:
            with torch.cuda.device(self.device):
                self._state = torch.zeros((9, self.ld), dtype=torch.float32).pin_memory()
        else:
            self._state = torch.zeros((9, self.ld), dtype=torch.float32, device=self.device)
        super().__init__(drone_model=drone_model, g=g)
        self._coefficients_changed()

    def _coefficients_changed(self):
        gains = PIDGains(self.P_COEFF_FOR, self.I_COEFF_FOR, self.D_COEFF_FOR, self.P_COEFF_TOR, self.I_COEFF_TOR,
                         self.D_COEFF_TOR, self.PWM2RPM_SCALE, self.PWM2RPM_CONST, self.MIN_PWM, self.MAX_PWM)
        self._params = self._drone_params.to_struct(pid_model=self.DRONE_MODEL, pid_g=self._g, gains=gains)

    def reset(self):
        """Previous-step and integral errors of both loops are set to zero."""
        super().reset()
        self._state.zero_()

    # controller members with the reference's names, read back from the device
    @property
    def integral_pos_e(self):
        return self._state[0:3, :self.n].t().cpu().numpy().astype(np.float64)

    @property
    def last_rpy(self):
        return self._state[3:6, :self.n].t().cpu().numpy().astype(np.float64)

    @property
    def integral_rpy_e(self):
        return self._state[6:9, :self.n].t().cpu().numpy().astype(np.float64)

    def computeControl(self, control_timestep, cur_pos, cur_quat, cur_vel, cur_ang_vel, target_pos,
                       target_rpy=None, target_vel=None, target_rpy_rates=None):
        """Batched control step -> (rpm [n,4], pos_e [n,3], yaw_e [n]) float32 device tensors."""
        n, dev = self.n, self.device
        self.control_counter += 1
        pos, quat, vel = _f32(cur_pos, n, 3, dev), _f32(cur_quat, n, 4, dev), _f32(cur_vel, n, 3, dev)
        tpos, trpy = _f32(target_pos, n, 3, dev), _f32(target_rpy, n, 3, dev)
        tvel, trates = _f32(target_vel, n, 3, dev), _f32(target_rpy_rates, n, 3, dev)
        rpm = torch.empty((n, 4), dtype=torch.float32, device=dev)
        pos_e = torch.empty((n, 3), dtype=torch.float32, device=dev)
        yaw_e = torch.empty((n,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = self.lib.gpd_pid(ctypes.byref(self._params), _ptr(self._state), self.ld, float(control_timestep),
                                  _ptr(pos), _ptr(quat), _ptr(vel), _ptr(tpos), _ptr(trpy), _ptr(tvel), _ptr(trates),
                                  _ptr(rpm), _ptr(pos_e), _ptr(yaw_e), n,
                                  ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _native.check(rc, "gpd_pid")
        return rpm, pos_e, yaw_e


class DSLPIDControl(DSLPIDControlBatch):
    """PID control class for Crazyflies — single drone, numpy interface of the reference."""

    # one page-locked block for the call's operands, float offsets (the two 16-byte accesses of gpd_pid_kernel first)
    _QUAT, _RPM, _POS, _VEL, _TPOS, _TRPY, _TVEL, _TRATES, _POS_E, _YAW_E, _IO_FLOATS = 0, 4, 8, 11, 14, 17, 20, 23, 26, 29, 32

    def __init__(self, drone_model: DroneModel, g: float = 9.8, device=None):
        super().__init__(1, drone_model=drone_model, g=g, device=device, host_visible=True)
        with torch.cuda.device(self.device):
            self._io_t = torch.zeros((self._IO_FLOATS,), dtype=torch.float32).pin_memory()
        self._io = self._io_t.numpy()
        base = self._io_t.data_ptr()
        at = lambda off: ctypes.c_void_p(base + 4 * off)      # noqa: E731
        self._pid_args = (_ptr(self._state), self.ld, at(self._POS), at(self._QUAT), at(self._VEL), at(self._TPOS), at(self._TRPY),
                          at(self._TVEL), at(self._TRATES), at(self._RPM), at(self._POS_E), at(self._YAW_E))

    def _member(self, row):
        torch.cuda.current_stream(self.device).synchronize()
        return self._state[row:row + 3, 0].numpy().astype(np.float64)

    integral_pos_e = property(lambda self: self._member(0))
    last_rpy = property(lambda self: self._member(3))
    integral_rpy_e = property(lambda self: self._member(6))

    def computeControl(self, control_timestep, cur_pos, cur_quat, cur_vel, cur_ang_vel, target_pos,
                       target_rpy=np.zeros(3), target_vel=np.zeros(3), target_rpy_rates=np.zeros(3)):
        """-> (rpm (4,), pos_e (3,), yaw_e float), float64 numpy like the reference."""
        self.control_counter += 1
        io, a = self._io, self._pid_args
        io[self._QUAT:self._QUAT + 4] = cur_quat
        io[self._POS:self._POS + 3] = cur_pos
        io[self._VEL:self._VEL + 3] = cur_vel
        io[self._TPOS:self._TPOS + 3] = target_pos
        io[self._TRPY:self._TRPY + 3] = target_rpy
        io[self._TVEL:self._TVEL + 3] = target_vel
        io[self._TRATES:self._TRATES + 3] = target_rpy_rates
        with torch.cuda.device(self.device):        # ONE library call: the launch and the wait for it (include/gpd.h: gpd_pid_sync)
            rc = self.lib.gpd_pid_sync(ctypes.byref(self._params), a[0], a[1], float(control_timestep), a[2], a[3], a[4], a[5], a[6], a[7], a[8],
                                       a[9], a[10], a[11], 1, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        _native.check(rc, "gpd_pid_sync")
        out = io.astype(np.float64)
        return out[self._RPM:self._RPM + 4], out[self._POS_E:self._POS_E + 3], float(out[self._YAW_E])

    def _one23DInterface(self, thrust):
        """1, 2 or 4 desired thrusts -> 4 PWMs (reference `:263-287`)."""
        DIM = len(np.array(thrust))
        pwm = np.clip((np.sqrt(np.array(thrust) / (self.KF * (4 // DIM))) - self.PWM2RPM_CONST) / self.PWM2RPM_SCALE,
                      self.MIN_PWM, self.MAX_PWM)
        if DIM in [1, 4]:
            return np.repeat(pwm, 4 // DIM)
        if DIM == 2:
            return np.hstack([pwm, np.flip(pwm)])
        raise ValueError("[ERROR] in DSLPIDControl._one23DInterface()")


def pid_rpm_for_action(env, action, act_type=None):
    """RPMs the embedded controllers of a `BaseRLAviary` produce for `action` (advances their state).

    Host-side twin of the kernel's PID action decoding (`envs/BaseRLAviary.py:193-235`), used only when
    someone calls `BaseRLAviary._preprocessAction` directly; `step()` does all of this in the kernel.
    """
    core = env._core
    act_type = act_type or env.ACT_TYPE
    n, dev = env.NUM_DRONES, core.device
    pos, quat, vel, rpy = env.pos, env.quat, env.vel, env.rpy
    tpos, trpy, tvel = pos.copy(), np.zeros((n, 3)), np.zeros((n, 3))
    for k in range(n):
        a = action[k]
        if act_type == ActionType.PID:
            tpos[k] = env._calculateNextStep(pos[k], a, 1)
        elif act_type == ActionType.VEL:
            nn = np.linalg.norm(a[0:3])
            unit = a[0:3] / nn if nn != 0 else np.zeros(3)
            trpy[k, 2] = rpy[k, 2]
            tvel[k] = env.SPEED_LIMIT * np.abs(a[3]) * unit
        elif act_type == ActionType.ONE_D_PID:
            tpos[k] = pos[k] + 0.1 * np.array([0, 0, a[0]])
    rpm = torch.empty((n, 4), dtype=torch.float32, device=dev)
    args = [_f32(x, n, k, dev) for x, k in ((pos, 3), (quat, 4), (vel, 3), (tpos, 3), (trpy, 3), (tvel, 3))]
    with torch.cuda.device(dev):
        rc = core.lib.gpd_pid(ctypes.byref(core._params), _ptr(core.pid), core.ld, float(env.CTRL_TIMESTEP),
                              *[_ptr(a) for a in args], _ptr(None), _ptr(rpm), _ptr(None), _ptr(None), n,
                              ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _native.check(rc, "gpd_pid")
    return rpm.cpu().numpy().astype(np.float64)
