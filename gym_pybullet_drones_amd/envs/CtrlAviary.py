"""`CtrlAviary`: control-research aviary — raw RPM actions, 20-float state observations
(reference `envs/CtrlAviary.py:11-200`).  The RPM clip to [0, MAX_RPM] (`:140`) runs in the kernel
(`GPD_ACT_RAW_RPM`).
"""
import numpy as np

from .._gym_shim import spaces
from ..utils.enums import ACT_RAW_RPM, DroneModel, Physics
from ._state_vector_aviary import StateVectorAviary


class CtrlAviary(StateVectorAviary):
    """Multi-drone environment class for control applications."""

    _STOCK_ACTION_CODE = ACT_RAW_RPM

    def __init__(self, drone_model: DroneModel = DroneModel.CF2X, num_drones: int = 1, neighbourhood_radius: float = np.inf,
                 initial_xyzs=None, initial_rpys=None, physics: Physics = Physics.PYB, pyb_freq: int = 240, ctrl_freq: int = 240,
                 gui=False, record=False, obstacles=False, user_debug_gui=True, output_folder='results', device=None):
        # (the reference's argument list, `envs/CtrlAviary.py:16-30`, + `device`)
        super().__init__(drone_model, num_drones, neighbourhood_radius, initial_xyzs, initial_rpys, physics, pyb_freq, ctrl_freq,
                         gui, record, obstacles, user_debug_gui, output_folder=output_folder, device=device)

    def _actionSpace(self):
        shape = (self.NUM_DRONES, 4)
        return spaces.Box(low=np.zeros(shape), high=np.full(shape, self.MAX_RPM), dtype=np.float32)

    def _preprocessAction(self, action):
        return np.clip(np.asarray(action)[:self.NUM_DRONES], 0, self.MAX_RPM)
