"""`CtrlAviary`: control-research aviary — raw RPM actions, 20-float state observations
(reference `envs/CtrlAviary.py:11-200`).  The RPM clip to [0, MAX_RPM] (`:140`) runs in the kernel
(`GPD_ACT_RAW_RPM`).
"""
import numpy as np

from .._gym_shim import spaces
from ..utils.enums import ACT_RAW_RPM, DroneModel, Physics
from .BaseAviary import BaseAviary


class CtrlAviary(BaseAviary):
    """Multi-drone environment class for control applications."""

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
                 output_folder='results',
                 device=None):
        super().__init__(drone_model=drone_model, num_drones=num_drones, neighbourhood_radius=neighbourhood_radius,
                         initial_xyzs=initial_xyzs, initial_rpys=initial_rpys, physics=physics, pyb_freq=pyb_freq,
                         ctrl_freq=ctrl_freq, gui=gui, record=record, obstacles=obstacles,
                         user_debug_gui=user_debug_gui, output_folder=output_folder, device=device)

    def _fusedActionCode(self):
        if type(self)._preprocessAction is CtrlAviary._preprocessAction:
            return ACT_RAW_RPM
        return None

    def _actionSpace(self):
        lo = np.zeros((self.NUM_DRONES, 4))
        hi = np.full((self.NUM_DRONES, 4), self.MAX_RPM)
        return spaces.Box(low=lo, high=hi, dtype=np.float32)

    def _observationSpace(self):
        inf, pi = np.inf, np.pi
        lo = np.array([[-inf, -inf, 0., -1., -1., -1., -1., -pi, -pi, -pi, -inf, -inf, -inf, -inf, -inf, -inf, 0., 0., 0., 0.]
                       for _ in range(self.NUM_DRONES)])
        hi = np.array([[inf, inf, inf, 1., 1., 1., 1., pi, pi, pi, inf, inf, inf, inf, inf, inf,
                        self.MAX_RPM, self.MAX_RPM, self.MAX_RPM, self.MAX_RPM] for _ in range(self.NUM_DRONES)])
        return spaces.Box(low=lo, high=hi, dtype=np.float32)

    def _computeObs(self):
        return np.array([self._getDroneStateVector(i) for i in range(self.NUM_DRONES)])

    def _preprocessAction(self, action):
        return np.array([np.clip(action[i, :], 0, self.MAX_RPM) for i in range(self.NUM_DRONES)])

    def _computeReward(self):
        return -1

    def _computeTerminated(self):
        return False

    def _computeTruncated(self):
        return False

    def _computeInfo(self):
        return {"answer": 42}
