"""Batched aviaries: E independent Hover / MultiHover / control aviaries advanced by ONE kernel launch.

New relative to the reference, which vectorises only through SB3's `make_vec_env(..., n_envs=1)`
(`examples/learn.py:54-58`).  The per-aviary semantics are exactly those of `HoverAviary` /
`MultiHoverAviary` / `CtrlAviary`; on top of that the batch follows the SB3 `DummyVecEnv`
convention: an aviary that terminates or is truncated is reset inside the same `step()` and the
observation returned for it is the first one of the new episode (the last one of the finished
episode is available as `info["terminal_observation"]` when `keep_terminal_obs=True`).  As in the
reference, a reset neither clears the action history nor the embedded PID state
(SURVEY.md App. B.2/B.3).

Everything stays on the GPU: actions come in and observations / rewards / flags go out as torch
tensors on the aviary's device, nothing synchronises with the host.
"""
import numpy as np
import torch

from .. import engine
from ..params import DroneParams
from ..utils.enums import ACT_RAW_RPM, ActionType, DroneModel, ObservationType, Physics

_TASKS = {"none": engine.TASK_NONE, "hover": engine.TASK_HOVER, "multihover": engine.TASK_MULTIHOVER}


class VectorAviary:
    """E aviaries x D drones.  Observations `(E, D, 12)` (or `(E, D, 12 + H*A)` with `full_obs`)."""

    def __init__(self,
                 num_envs: int,
                 num_drones: int = 1,
                 drone_model: DroneModel = DroneModel.CF2X,
                 initial_xyzs=None,
                 initial_rpys=None,
                 physics: Physics = Physics.DYN,
                 pyb_freq: int = 240,
                 ctrl_freq: int = 30,
                 obs: ObservationType = ObservationType.KIN,
                 act=ActionType.RPM,
                 task: str = "hover",
                 target_pos=None,
                 episode_len_sec: float = 8,
                 auto_reset: bool = True,
                 full_obs: bool = False,
                 keep_terminal_obs: bool = False,
                 track_rpm: bool = False,
                 device=None):
        if obs != ObservationType.KIN:
            raise NotImplementedError("only ObservationType.KIN is on the MI355X hot path")
        self.NUM_ENVS, self.NUM_DRONES = int(num_envs), int(num_drones)
        self.DRONE_MODEL, self.PHYSICS = drone_model, physics
        self.PYB_FREQ, self.CTRL_FREQ = pyb_freq, ctrl_freq
        self.PYB_STEPS_PER_CTRL = pyb_freq // ctrl_freq
        self.CTRL_TIMESTEP, self.PYB_TIMESTEP = 1. / ctrl_freq, 1. / pyb_freq
        self.EPISODE_LEN_SEC = episode_len_sec
        self.ACT_TYPE = act
        act_code = ACT_RAW_RPM if act == "raw_rpm" else act.code
        P = DroneParams(drone_model)
        self.HOVER_RPM, self.MAX_RPM = P.HOVER_RPM, P.MAX_RPM
        if initial_xyzs is None:
            initial_xyzs = P.default_init_xyzs(self.NUM_DRONES)
        init = np.asarray(initial_xyzs, dtype=np.float64)
        if target_pos is None:
            if task == "hover":
                target_pos = np.broadcast_to(np.array([0., 0., 1.]), (self.NUM_DRONES, 3)).copy()
            elif task == "multihover":
                target_pos = init + np.array([[0, 0, 1 / (i + 1)] for i in range(self.NUM_DRONES)])
        xy = 1.5 if task == "hover" else 2.0
        self.core = engine.SimCore(drone_model=drone_model, num_envs=num_envs, drones_per_env=num_drones,
                                   physics=physics, pyb_freq=pyb_freq, ctrl_freq=ctrl_freq, act_code=act_code,
                                   task=_TASKS[task], initial_xyzs=init, initial_rpys=initial_rpys,
                                   target_pos=target_pos, episode_len_sec=episode_len_sec, xy_bound=xy,
                                   auto_reset=auto_reset, track_rpm=track_rpm, keep_terminal_obs=keep_terminal_obs,
                                   device=device)
        self.device = self.core.device
        self.ACT_DIM = self.core.A
        self.INIT_XYZS, self.INIT_RPYS, self.TARGET_POS = self.core.INIT_XYZS, self.core.INIT_RPYS, self.core.TARGET_POS
        self.full_obs = bool(full_obs)
        self.ACTION_BUFFER_SIZE = int(ctrl_freq // 2)
        self.OBS_DIM = 12 + (self.ACTION_BUFFER_SIZE * self.ACT_DIM if full_obs else 0)
        if full_obs:
            # doubled ring: the action of step t is written to slots p and p+H (p = t mod H), so the
            # window [p+1, p+1+H) always holds the last H actions oldest-first, contiguously
            H = self.ACTION_BUFFER_SIZE
            self._hist = torch.zeros((self.core.N, 2 * H, self.ACT_DIM), dtype=torch.float32, device=self.device)
            self._hist_pos = H - 1

    # ---- gymnasium-VectorEnv-like surface ----------------------------------------------------
    @property
    def num_envs(self):
        return self.NUM_ENVS

    def _obs(self):
        E, D = self.NUM_ENVS, self.NUM_DRONES
        o = self.core.obs12.view(E, D, 12)
        if not self.full_obs:
            return o
        return torch.cat([o, self.action_history().reshape(E, D, -1)], dim=-1)

    def action_history(self) -> torch.Tensor:
        """(E*D, H, A) view of the last H actions, oldest first (zero-copy)."""
        H, p = self.ACTION_BUFFER_SIZE, self._hist_pos
        return self._hist[:, p + 1:p + 1 + H, :]

    def reset(self, seed=None, options=None, mask=None):
        """Reset all aviaries (or those selected by the boolean/uint8 tensor `mask` [E])."""
        self.core.reset(mask=mask)
        return self._obs(), {}

    def step(self, action: torch.Tensor):
        """action: float32 tensor (E, D, A) on `self.device` -> (obs, reward[E], terminated[E], truncated[E], info)."""
        if self.full_obs:
            H = self.ACTION_BUFFER_SIZE
            p = (self._hist_pos + 1) % H
            a = action.reshape(self.core.N, self.ACT_DIM).to(torch.float32)
            self._hist[:, p, :] = a
            self._hist[:, p + H, :] = a
            self._hist_pos = p
        _, reward, terminated, truncated = self.core.step(action)
        info = {}
        if self.core.term_obs12 is not None:
            info["terminal_observation"] = self.core.term_obs12.view(self.NUM_ENVS, self.NUM_DRONES, 12)
        return self._obs(), reward, terminated, truncated, info

    def state_vectors(self) -> torch.Tensor:
        """(E, D, 20) `_getDroneStateVector`-ordered states (needs `track_rpm=True` for the RPM columns)."""
        return self.core.state_vectors().view(self.NUM_ENVS, self.NUM_DRONES, 20)

    def close(self):
        pass


class VectorHoverAviary(VectorAviary):
    """E x `HoverAviary` (one drone each)."""

    def __init__(self, num_envs: int, drone_model: DroneModel = DroneModel.CF2X, initial_xyzs=None, initial_rpys=None,
                 physics: Physics = Physics.DYN, pyb_freq: int = 240, ctrl_freq: int = 30,
                 obs: ObservationType = ObservationType.KIN, act: ActionType = ActionType.RPM, **kw):
        super().__init__(num_envs=num_envs, num_drones=1, drone_model=drone_model, initial_xyzs=initial_xyzs,
                         initial_rpys=initial_rpys, physics=physics, pyb_freq=pyb_freq, ctrl_freq=ctrl_freq, obs=obs,
                         act=act, task="hover", **kw)


class VectorMultiHoverAviary(VectorAviary):
    """E x `MultiHoverAviary` (`num_drones` drones each, default 2)."""

    def __init__(self, num_envs: int, num_drones: int = 2, drone_model: DroneModel = DroneModel.CF2X,
                 initial_xyzs=None, initial_rpys=None, physics: Physics = Physics.DYN, pyb_freq: int = 240,
                 ctrl_freq: int = 30, obs: ObservationType = ObservationType.KIN, act: ActionType = ActionType.RPM, **kw):
        super().__init__(num_envs=num_envs, num_drones=num_drones, drone_model=drone_model, initial_xyzs=initial_xyzs,
                         initial_rpys=initial_rpys, physics=physics, pyb_freq=pyb_freq, ctrl_freq=ctrl_freq, obs=obs,
                         act=act, task="multihover", **kw)


class VectorCtrlAviary(VectorAviary):
    """E x `CtrlAviary`: raw RPM actions clipped to [0, MAX_RPM], no task."""

    def __init__(self, num_envs: int, num_drones: int = 1, drone_model: DroneModel = DroneModel.CF2X,
                 initial_xyzs=None, initial_rpys=None, physics: Physics = Physics.DYN, pyb_freq: int = 240,
                 ctrl_freq: int = 240, **kw):
        kw.setdefault("auto_reset", False)
        kw.setdefault("track_rpm", True)
        super().__init__(num_envs=num_envs, num_drones=num_drones, drone_model=drone_model, initial_xyzs=initial_xyzs,
                         initial_rpys=initial_rpys, physics=physics, pyb_freq=pyb_freq, ctrl_freq=ctrl_freq,
                         act="raw_rpm", task="none", **kw)
