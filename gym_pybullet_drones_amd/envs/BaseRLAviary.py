"""`BaseRLAviary`: action types, kinematic observation + action history (single aviary).

Reference: `envs/BaseRLAviary.py` — constructor `:16-95`, `_actionSpace` `:132-156`,
`_preprocessAction` `:160-239`, `_observationSpace` `:243-280`, `_computeObs` `:284-322`.
The five `ActionType` mappings (incl. the embedded DSLPID controllers, always built for CF2X,
`:75-76`) run inside the fused kernel; this class keeps the host-side pieces: spaces, the
`ctrl_freq//2`-deep action buffer (never cleared by `reset()`, App. B.2) and obs assembly.

Deviation: the observation is always float32 (the reference's dtype drifts).
"""
from collections import deque

import numpy as np
import torch

from .._gym_shim import spaces
from ..utils.enums import ActionType, DroneModel, ObservationType, Physics
from .BaseAviary import BaseAviary


class BaseRLAviary(BaseAviary):
    """Base single and multi-agent environment class for reinforcement learning."""

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
                 obs: ObservationType = ObservationType.KIN,
                 act: ActionType = ActionType.RPM,
                 device=None):
        if obs != ObservationType.KIN:
            raise NotImplementedError("ObservationType.RGB (camera rendering) is outside the MI355X hot path")
        # the last half second of actions is part of every observation row
        self.ACTION_BUFFER_SIZE = int(ctrl_freq // 2)
        self.action_buffer = deque(maxlen=self.ACTION_BUFFER_SIZE)
        self.OBS_TYPE = obs
        self.ACT_TYPE = act
        if act.uses_pid and drone_model not in (DroneModel.CF2X, DroneModel.CF2P):
            raise ValueError("[ERROR] in BaseRLAviary.__init()__, no controller is available for the specified drone_model")
        super().__init__(drone_model=drone_model, num_drones=num_drones, neighbourhood_radius=neighbourhood_radius,
                         initial_xyzs=initial_xyzs, initial_rpys=initial_rpys, physics=physics, pyb_freq=pyb_freq,
                         ctrl_freq=ctrl_freq, gui=gui, record=record, obstacles=True, user_debug_gui=False,
                         vision_attributes=False, device=device)
        if act == ActionType.VEL:
            self.SPEED_LIMIT = 0.03 * self.MAX_SPEED_KMH * (1000 / 3600)


    def _fusedActionCode(self):
        if type(self)._preprocessAction is BaseRLAviary._preprocessAction:
            return self.ACT_TYPE.code
        return None

    def _recordAction(self, action):
        a = np.array(action, dtype=np.float32).reshape(self.NUM_DRONES, -1)
        self.action_buffer.append(a)
        # the same history as one (NUM_DRONES, H * A) row block, oldest action first: what every observation row ends with
        w = a.shape[1]
        self._history_rows[:, :-w] = self._history_rows[:, w:]
        self._history_rows[:, -w:] = a
        self._history_tail = a

    def _actionSpace(self):
        """Box of shape (NUM_DRONES, 4 | 3 | 1) in [-1, 1]."""
        size = self.ACT_TYPE.dim
        act_lower_bound = np.array([-1 * np.ones(size) for _ in range(self.NUM_DRONES)])
        act_upper_bound = np.array([+1 * np.ones(size) for _ in range(self.NUM_DRONES)])
        for _ in range(self.ACTION_BUFFER_SIZE):
            self.action_buffer.append(np.zeros((self.NUM_DRONES, size), dtype=np.float32))
        self._history_rows = np.zeros((self.NUM_DRONES, self.ACTION_BUFFER_SIZE * size), dtype=np.float32)
        self._history_tail = self.action_buffer[-1]
        return spaces.Box(low=act_lower_bound, high=act_upper_bound, dtype=np.float32)


    def _preprocessAction(self, action):
        """action (NUM_DRONES, A) -> RPMs (NUM_DRONES, 4).

        `step()` does NOT call this (the mapping is fused into the kernel); it is kept for callers
        and subclasses that use it directly.  For the PID action types it advances the embedded
        controllers exactly like the reference does.
        """
        action = np.asarray(action, dtype=np.float64).reshape(self.NUM_DRONES, -1)
        self._recordAction(action)
        if self.ACT_TYPE == ActionType.RPM:
            return np.array(self.HOVER_RPM * (1 + 0.05 * action))
        if self.ACT_TYPE == ActionType.ONE_D_RPM:
            return np.repeat(self.HOVER_RPM * (1 + 0.05 * action), 4, axis=1)
        from ..control.DSLPIDControl import pid_rpm_for_action
        return pid_rpm_for_action(self, action)


    def _observationSpace(self):
        """Box of shape (NUM_DRONES, 12 + ACTION_BUFFER_SIZE * A)."""
        lo, hi = -np.inf, np.inf
        obs_lower_bound = np.array([[lo, lo, 0, lo, lo, lo, lo, lo, lo, lo, lo, lo] for _ in range(self.NUM_DRONES)])
        obs_upper_bound = np.array([[hi] * 12 for _ in range(self.NUM_DRONES)])
        tail = self.ACTION_BUFFER_SIZE * self.ACT_TYPE.dim
        obs_lower_bound = np.hstack([obs_lower_bound, -np.ones((self.NUM_DRONES, tail))])
        obs_upper_bound = np.hstack([obs_upper_bound, +np.ones((self.NUM_DRONES, tail))])
        return spaces.Box(low=obs_lower_bound, high=obs_upper_bound, dtype=np.float32)


    def _computeObs(self):
        """(NUM_DRONES, 12 + H*A) float32: pos | rpy | vel | ang_v, then the H most recent actions, oldest first."""
        if self._core.host_visible:
            kin = self._host_views["obs"]           # the kernel's own float32 row: what the float64 attributes are copies of
        else:
            kin = np.hstack([self.pos, self.rpy, self.vel, self.ang_v]).astype('float32')
        if self.action_buffer[-1] is not self._history_tail:       # somebody appended to the deque directly: follow it
            self._history_rows = np.concatenate([np.asarray(a, dtype=np.float32).reshape(self.NUM_DRONES, -1) for a in self.action_buffer], axis=1)
            self._history_tail = self.action_buffer[-1]
        return np.concatenate([kin, self._history_rows], axis=1)
