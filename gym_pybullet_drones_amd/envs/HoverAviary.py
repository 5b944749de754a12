"""`HoverAviary`: single-drone hover task (reference `envs/HoverAviary.py:11-132`).

Reward `max(0, 2 - ||target - pos||^4)`, termination within 1e-4 m of the target, truncation when
the drone leaves the |x|,|y| <= 1.5, z <= 2 box, tilts more than 0.4 rad or after 8 s — all
evaluated by the fused kernel (`GPD_TASK_HOVER`); the hooks below return the kernel's results.
"""
import numpy as np

from .. import engine
from ..utils.enums import ActionType, DroneModel, ObservationType, Physics
from .BaseRLAviary import BaseRLAviary


class HoverAviary(BaseRLAviary):
    """Single agent RL problem: hover at position."""

    _TASK = engine.TASK_HOVER

    def __init__(self,
                 drone_model: DroneModel = DroneModel.CF2X,
                 initial_xyzs=None,
                 initial_rpys=None,
                 physics: Physics = Physics.PYB,
                 pyb_freq: int = 240,
                 ctrl_freq: int = 30,
                 gui=False,
                 record=False,
                 obs: ObservationType = ObservationType.KIN,
                 act: ActionType = ActionType.RPM,
                 device=None):
        self.TARGET_POS = np.array([0, 0, 1])
        self.EPISODE_LEN_SEC = 8
        super().__init__(drone_model=drone_model, num_drones=1, initial_xyzs=initial_xyzs, initial_rpys=initial_rpys,
                         physics=physics, pyb_freq=pyb_freq, ctrl_freq=ctrl_freq, gui=gui, record=record, obs=obs,
                         act=act, device=device)

    def _taskConfig(self):
        return dict(target_pos=np.asarray(self.TARGET_POS, dtype=np.float64).reshape(1, 3),
                    episode_len_sec=self.EPISODE_LEN_SEC, xy_bound=1.5, z_bound=2.0, tilt_bound=.4, term_dist=.0001)

    def _computeReward(self):
        return self._k_reward

    def _computeTerminated(self):
        return self._k_terminated

    def _computeTruncated(self):
        return self._k_truncated

    def _computeInfo(self):
        return {"answer": 42}
