"""`BaseAviary`: the reference's single-aviary `gymnasium.Env` surface on top of the HIP engine.

Mirrors the public interface of the reference class (`envs/BaseAviary.py:25-216` constructor and
constants, `reset` `:220-255`, `step` `:259-383`, `render`/`close` `:387-421`, the kinematic cache
`:509-519`, `_getDroneStateVector` `:541-561`, the eight subclass hooks `:1021-1104`,
`_calculateNextStep` `:1108-1150`) so subclasses written against the reference keep working.  What
is different underneath: there is no PyBullet — the whole `step()` body (action mapping, PID,
physics sub-steps, cache refresh, observation, reward, termination, truncation) is one launch of
the fused gfx950 kernel (`engine.SimCore`, E=1 aviary x NUM_DRONES), and this class only exposes the
handful of resulting floats as numpy attributes with the reference's names and shapes.  The aviary's
state lives in page-locked host memory the device addresses directly (`SimCore(host_visible=True)`):
a `step()` is the action row written by the host, ONE launch, one stream synchronisation, and numpy
reads of what the kernel wrote -- no copy engine, no second launch (`bench.py`: `dropin_single_env`).

Deviations from the reference, on purpose:
  * `physics`: only the explicit integrator exists.  `Physics.DYN` is the reference's `Physics.DYN`.  `Physics.PYB*` (the
    default!) runs the same integrator PLUS a ground plane at z = 0 (`GPD_PHYS_GROUND`, a model the reference's DYN does not have,
    standing in for Bullet's contact solver: the advertised observation space, z >= 0, needs it) -- above the plane a `PYB` run is
    bit for bit the reference's DYN; `PYB_GND/DRAG/DW/GND_DRAG_DW` also enable those force models *inside* the explicit integrator.
    Bullet's default multibody damping 0.04 (`GPD_PHYS_DAMP`; restated from the Bullet sources, parity unpinned -- include/gpd.h)
    is OPT-IN: `utils.enums.set_pyb_like("damped")` or `GPD_PYB_LIKE=damped`.  `set_pyb_like(False)` / `GPD_PYB_LIKE=0` removes
    the plane too: `Physics.PYB` is then exactly `Physics.DYN`.  A one-time `UserWarning` says all this when a PYB member is used.
  * arithmetic is float32 on the device (the reference is float64 numpy); attributes are float64
    numpy copies of the float32 results.
  * GUI, video recording, cameras, obstacles are not available (`gui=True`/`record=True` raise).
"""
import os
import time

import numpy as np
import torch

from .. import engine
from .._gym_shim import Env
from ..params import DroneParams
from ..utils.enums import ACT_DIRECT_RPM, DroneModel, Physics, warn_if_pyb


class BaseAviary(Env):
    """Base class for "drone aviary" Gym environments (MI355X-native)."""

    #: set by subclasses whose action mapping / task is evaluated inside the kernel
    _TASK = engine.TASK_NONE

    def __init__(self,
                 drone_model: DroneModel = DroneModel.CF2X,
                 num_drones: int = 1,
                 neighbourhood_radius: float = np.inf,
                 initial_xyzs=None,
                 initial_rpys=None,
                 physics: Physics = Physics.PYB,
                 pyb_freq: int = 240,
                 ctrl_freq: int = 240,
                 gui=False,
                 record=False,
                 obstacles=False,
                 user_debug_gui=True,
                 vision_attributes=False,
                 output_folder='results',
                 device=None):
        if gui or record or vision_attributes:
            raise NotImplementedError("GUI, video recording and camera observations are outside the MI355X hot path")
        # constants, under the reference's names
        if pyb_freq % ctrl_freq:
            raise ValueError('[ERROR] in BaseAviary.__init__(), pyb_freq is not divisible by env_freq.')
        self.G, self.RAD2DEG, self.DEG2RAD = 9.8, 180 / np.pi, np.pi / 180
        self.CTRL_FREQ, self.PYB_FREQ, self.PYB_STEPS_PER_CTRL = ctrl_freq, pyb_freq, pyb_freq // ctrl_freq
        self.CTRL_TIMESTEP, self.PYB_TIMESTEP = 1.0 / ctrl_freq, 1.0 / pyb_freq
        self.NUM_DRONES, self.NEIGHBOURHOOD_RADIUS, self.DRONE_MODEL = num_drones, neighbourhood_radius, drone_model
        self.GUI, self.RECORD, self.OBSTACLES, self.USER_DEBUG = False, False, obstacles, user_debug_gui
        self.PHYSICS = physics
        warn_if_pyb(physics)                  # Physics.PYB* runs the explicit integrator here: say so, once
        self.URDF, self.OUTPUT_FOLDER = drone_model.value + ".urdf", output_folder
        self.CLIENT = -1                      # there is no PyBullet client
        P = DroneParams(drone_model)
        for name in ("M", "L", "THRUST2WEIGHT_RATIO", "J", "J_INV", "KF", "KM", "COLLISION_H", "COLLISION_R",
                     "COLLISION_Z_OFFSET", "MAX_SPEED_KMH", "GND_EFF_COEFF", "PROP_RADIUS", "DRAG_COEFF",
                     "DW_COEFF_1", "DW_COEFF_2", "DW_COEFF_3", "GRAVITY", "HOVER_RPM", "MAX_RPM", "MAX_THRUST",
                     "MAX_XY_TORQUE", "MAX_Z_TORQUE", "GND_EFF_H_CLIP"):
            setattr(self, name, getattr(P, name))
        self._drone_params = P
        if initial_xyzs is None:
            self.INIT_XYZS = P.default_init_xyzs(self.NUM_DRONES)
        elif np.array(initial_xyzs).shape == (self.NUM_DRONES, 3):
            self.INIT_XYZS = np.array(initial_xyzs, dtype=np.float64)
        else:
            raise ValueError("[ERROR] invalid initial_xyzs in BaseAviary.__init__(), try initial_xyzs.reshape(NUM_DRONES,3)")
        if initial_rpys is None:
            self.INIT_RPYS = np.zeros_like(self.INIT_XYZS)
        elif np.shape(initial_rpys) == (num_drones, 3):
            self.INIT_RPYS = np.array(initial_rpys, dtype=np.float64)
        else:
            raise ValueError("[ERROR] invalid initial_rpys in BaseAviary.__init__(), try initial_rpys.reshape(NUM_DRONES,3)")
        self.action_space, self.observation_space = self._actionSpace(), self._observationSpace()
        # the engine: one aviary of NUM_DRONES drones; GPD_HOST_VISIBLE=0 keeps its state in HBM instead (A/B)
        fused = self._fusedActionCode()
        self._fused_action = fused is not None
        task_kw = self._taskConfig()
        self._core = engine.SimCore(drone_model=drone_model, num_envs=1, drones_per_env=num_drones,
                                    physics=physics, pyb_freq=pyb_freq, ctrl_freq=ctrl_freq,
                                    act_code=fused if self._fused_action else ACT_DIRECT_RPM,
                                    task=self._TASK, initial_xyzs=self.INIT_XYZS, initial_rpys=self.INIT_RPYS,
                                    auto_reset=False, track_rpm=True, device=device,
                                    host_visible=os.environ.get("GPD_HOST_VISIBLE", "1") != "0",
                                    # a subclass that overrides _preprocessAction and calls super()'s for a PID action type
                                    # still needs the embedded controllers' state (BaseRLAviary.py:75-76)
                                    force_pid=bool(getattr(getattr(self, "ACT_TYPE", None), "uses_pid", False)), **task_kw)
        self.DRONE_IDS = np.arange(1, self.NUM_DRONES + 1)
        if self._core.host_visible:           # numpy windows onto the buffers the kernel writes (valid after SimCore.drain())
            c, n = self._core, self.NUM_DRONES
            self._host_views = {"obs": c.obs12.numpy(), "P": c.kin_P.numpy()[:n], "Q": c.kin_Q.numpy()[:n], "V": c.kin_V.numpy()[:n],
                                "W": c.kin_W.numpy()[:n], "rpm": c.last_rpm.numpy()[:, :n], "reward": c.reward.numpy(),
                                "terminated": c.terminated.numpy(), "truncated": c.truncated.numpy()}
            self._action_row = c.action_host.numpy()
            # the body rates sit in three planes of the state block (P[:, 3] | V[:, 3] | W): one gather instead of a stack of three slices
            i = np.arange(n)
            self._host_views["kin_flat"] = c.kin_store.numpy().reshape(-1)
            self._rates_at = np.stack([4 * i + 3, 8 * c.ld + 4 * i + 3, 12 * c.ld + i], axis=1).reshape(-1)
        self._housekeeping()
        self._updateAndStoreKinematicInformation()


    def reset(self, seed: int = None, options: dict = None):
        """Resets the environment -> (obs, info)."""
        super().reset(seed=seed, options=options)
        self._housekeeping()
        self._updateAndStoreKinematicInformation()
        return self._computeObs(), self._computeInfo()


    def step(self, action):
        """Advances the environment by one control step -> (obs, reward, terminated, truncated, info)."""
        action = np.asarray(action)
        if self._fused_action:
            self._recordAction(action)
            row = action
        else:
            row = self._preprocessAction(action)        # the subclass's own mapping: RPMs, fed to the kernel as they are
        core = self._core
        if core.host_visible:
            self._action_row[...] = np.reshape(row, self._action_row.shape)       # (casts to float32 where the kernel reads it)
            core.step_host()
        else:
            core.step(torch.as_tensor(np.ascontiguousarray(row, dtype=np.float32).reshape(self.NUM_DRONES, -1), device=core.device))
        self._updateAndStoreKinematicInformation()
        # the hooks, in the reference's order (a subclass may rely on it): obs, reward, terminated, truncated, info
        result = (self._computeObs(), self._computeReward(), self._computeTerminated(), self._computeTruncated(), self._computeInfo())
        self.step_counter += self.PYB_STEPS_PER_CTRL
        return result


    def render(self, mode='human', close=False):
        """Text only (there is no GUI): one table row per drone -- position, velocity, attitude in degrees, world body rates --
        under a header with the step count and simulated vs wall-clock time.  Prints it and returns the string."""
        wall = max(time.time() - self.RESET_TIME, 1e-9)
        sim = self.step_counter * self.PYB_TIMESTEP
        lines = [f"step {self.step_counter}  sim {sim:.2f} s @ {self.PYB_FREQ} Hz  wall {wall:.2f} s  ({sim / wall:.1f}x real time)",
                 "drone        x        y        z       vx       vy       vz     roll    pitch      yaw       wx       wy       wz"]
        for i in range(self.NUM_DRONES):
            cells = (*self.pos[i], *self.vel[i], *np.degrees(self.rpy[i]), *self.ang_v[i])
            lines.append(f"{i:5d} " + " ".join(f"{c:8.3f}" for c in cells))
        text = "\n".join(lines)
        print(text)
        return text

    def close(self):
        """Terminates the environment (nothing to disconnect from)."""

    def getPyBulletClient(self):
        return self.CLIENT

    def getDroneIds(self):
        return self.DRONE_IDS


    def _housekeeping(self):
        """Zero the counters and put every drone back at its initial pose (on the device)."""
        self.RESET_TIME = time.time()
        self.step_counter = 0
        self._core.reset()

    def _updateAndStoreKinematicInformation(self):
        """Refresh the numpy kinematic cache (float64 copies of the float32 results) from what the latest launch wrote.  With the
        state in host-visible memory these are plain numpy reads; otherwise one packed device-to-host copy."""
        n = self.NUM_DRONES
        core = self._core
        if core.host_visible:
            v = self._host_views
            obs = v["obs"].astype(np.float64)                      # pos | rpy | vel | ang_v, as the kernel's observation row has them
            self.pos, self.rpy, self.vel, self.ang_v = obs[:, 0:3], obs[:, 3:6], obs[:, 6:9], obs[:, 9:12]
            self.quat = v["Q"].astype(np.float64)
            self.rpy_rates = v["kin_flat"][self._rates_at].astype(np.float64).reshape(n, 3)
            self.last_clipped_action = v["rpm"].T.astype(np.float64)
            self._k_reward = float(v["reward"][0])
            self._k_terminated = bool(v["terminated"][0])
            self._k_truncated = bool(v["truncated"][0])
            return
        packed = torch.cat([core.state_vectors(), core.body_rates(n),
                            core.reward.expand(n, 1), core.terminated.to(torch.float32).expand(n, 1),
                            core.truncated.to(torch.float32).expand(n, 1)], dim=1).cpu().numpy().astype(np.float64)
        self.pos = packed[:, 0:3].copy()
        self.quat = packed[:, 3:7].copy()
        self.rpy = packed[:, 7:10].copy()
        self.vel = packed[:, 10:13].copy()
        self.ang_v = packed[:, 13:16].copy()
        self.last_clipped_action = packed[:, 16:20].copy()
        self.rpy_rates = packed[:, 20:23].copy()
        self._k_reward = float(packed[0, 23])
        self._k_terminated = bool(packed[0, 24] != 0)
        self._k_truncated = bool(packed[0, 25] != 0)

    def _parseURDFParameters(self):
        """The tuple the reference's URDF parser returns, in its order (envs/BaseAviary.py:985-1017): M, L,
        THRUST2WEIGHT_RATIO, J, J_INV, KF, KM, COLLISION_H, COLLISION_R, COLLISION_Z_OFFSET, MAX_SPEED_KMH, GND_EFF_COEFF,
        PROP_RADIUS, DRAG_COEFF, DW_COEFF_1, DW_COEFF_2, DW_COEFF_3 -- from the shipped copy of the same `assets/<model>.urdf`
        (`params.DroneParams`), for callers and subclasses that unpack it themselves."""
        P = DroneParams(self.DRONE_MODEL)
        return tuple(getattr(P, n) for n in ("M", "L", "THRUST2WEIGHT_RATIO", "J", "J_INV", "KF", "KM", "COLLISION_H", "COLLISION_R",
                                             "COLLISION_Z_OFFSET", "MAX_SPEED_KMH", "GND_EFF_COEFF", "PROP_RADIUS", "DRAG_COEFF",
                                             "DW_COEFF_1", "DW_COEFF_2", "DW_COEFF_3"))

    def _getDroneStateVector(self, nth_drone):
        """(20,) state vector: pos3 | quat4 | rpy3 | vel3 | ang_v3 | last_clipped_action4."""
        i = nth_drone
        return np.concatenate((self.pos[i], self.quat[i], self.rpy[i], self.vel[i], self.ang_v[i], self.last_clipped_action[i]))

    def _getAdjacencyMatrix(self):
        """(NUM_DRONES, NUM_DRONES) adjacency matrix for NEIGHBOURHOOD_RADIUS (BaseAviary.py:658-675)."""
        d = np.linalg.norm(self.pos[:, None, :] - self.pos[None, :, :], axis=-1)
        return (d < self.NEIGHBOURHOOD_RADIUS).astype(np.float64)

    def _normalizedActionToRPM(self, action):
        """[-1, 1] -> [0, MAX_RPM], non-linear (BaseAviary.py:896-914; unused by the RL aviaries)."""
        if np.abs(action).max() > 1:
            print(f"[ERROR] BaseAviary._normalizedActionToRPM() at step {self.step_counter}: an action outside [-1, 1]")
        return np.where(action <= 0, (action + 1) * self.HOVER_RPM, self.HOVER_RPM + (self.MAX_RPM - self.HOVER_RPM) * action)

    # engine configuration hooks (new)

    def _fusedActionCode(self):
        """Kernel action code when the action->RPM mapping runs inside the kernel, else None (the
        subclass's `_preprocessAction` is then called in Python and its RPMs are fed directly)."""
        return None

    def _recordAction(self, action):
        """Called with the raw action when the mapping is fused (BaseRLAviary keeps its action buffer)."""

    def _taskConfig(self) -> dict:
        """Extra `SimCore` keyword arguments describing the task evaluated in the kernel."""
        return {}

    # the reference's subclass hooks

    def _actionSpace(self):
        raise NotImplementedError

    def _observationSpace(self):
        raise NotImplementedError

    def _computeObs(self):
        raise NotImplementedError

    def _preprocessAction(self, action):
        raise NotImplementedError

    def _computeReward(self):
        raise NotImplementedError

    def _computeTerminated(self):
        raise NotImplementedError

    def _computeTruncated(self):
        raise NotImplementedError

    def _computeInfo(self):
        raise NotImplementedError


    def _calculateNextStep(self, current_position, destination, step_size=1):
        """Waypoint at most `step_size` away from `current_position` towards `destination`."""
        direction = destination - current_position
        distance = np.linalg.norm(direction)
        if distance <= step_size:
            return destination
        return current_position + (direction / distance) * step_size
