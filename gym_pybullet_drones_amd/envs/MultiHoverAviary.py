"""`MultiHoverAviary`: N drones (default 2) hovering at staggered heights (reference
`envs/MultiHoverAviary.py:11-145`).

Targets `INIT_XYZS + [0, 0, 1/(i+1)]`; reward = sum of the per-drone hover rewards; terminated when
the SUM of distances is below 1e-4; truncated when ANY drone leaves the |x|,|y| <= 2, z <= 2 box or
tilts more than 0.4 rad, or after 8 s.  Evaluated by the fused kernel (`GPD_TASK_MULTIHOVER`): the
drones of the aviary sit in adjacent lanes and reduce through LDS.
"""
import numpy as np

from .. import engine
from ..utils.enums import ActionType, DroneModel, ObservationType, Physics
from .BaseRLAviary import BaseRLAviary


class MultiHoverAviary(BaseRLAviary):
    """Multi-agent RL problem: leader-follower."""

    _TASK = engine.TASK_MULTIHOVER

    def __init__(self,
                 drone_model: DroneModel = DroneModel.CF2X,
                 num_drones: int = 2,
                 neighbourhood_radius: float = np.inf,
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
        self.EPISODE_LEN_SEC = 8
        super().__init__(drone_model=drone_model, num_drones=num_drones, neighbourhood_radius=neighbourhood_radius,
                         initial_xyzs=initial_xyzs, initial_rpys=initial_rpys, physics=physics, pyb_freq=pyb_freq,
                         ctrl_freq=ctrl_freq, gui=gui, record=record, obs=obs, act=act, device=device)

    def _taskConfig(self):
        self.TARGET_POS = self.INIT_XYZS + np.array([[0, 0, 1 / (i + 1)] for i in range(self.NUM_DRONES)])
        return dict(target_pos=self.TARGET_POS, episode_len_sec=self.EPISODE_LEN_SEC, xy_bound=2.0, z_bound=2.0,
                    tilt_bound=.4, term_dist=.0001)

    def _computeReward(self):
        return self._k_reward

    def _computeTerminated(self):
        return self._k_terminated

    def _computeTruncated(self):
        return self._k_truncated

    def _computeInfo(self):
        return {"answer": 42}
