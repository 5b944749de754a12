"""What `CtrlAviary` and `VelocityAviary` share: neither has a task, both observe the 20-float state vector of every drone
(reference `envs/CtrlAviary.py:98-121, 146-200`, `envs/VelocityAviary.py:105-127, 172-228`)."""
import numpy as np

from .._gym_shim import spaces
from .BaseAviary import BaseAviary

_INF, _PI = np.inf, np.pi
#: bounds of one state-vector row: x y z | quaternion | r p y | velocity | body rates | the four RPMs (upper RPM bound: MAX_RPM)
_ROW_LOW = (-_INF, -_INF, 0.) + (-1.,) * 4 + (-_PI,) * 3 + (-_INF,) * 6 + (0.,) * 4
_ROW_HIGH = (_INF,) * 3 + (1.,) * 4 + (_PI,) * 3 + (_INF,) * 6


class StateVectorAviary(BaseAviary):
    """An aviary without a task whose observation is `_getDroneStateVector` of every drone, shape (NUM_DRONES, 20)."""

    #: the kernel's action code of the class whose `_preprocessAction` is still the stock one (set by the subclass)
    _STOCK_ACTION_CODE = None

    def _fusedActionCode(self):
        stock = next(c for c in type(self).__mro__ if "_STOCK_ACTION_CODE" in vars(c))
        return self._STOCK_ACTION_CODE if type(self)._preprocessAction is stock._preprocessAction else None

    def _observationSpace(self):
        low = np.tile(np.array(_ROW_LOW), (self.NUM_DRONES, 1))
        high = np.tile(np.array(_ROW_HIGH + (self.MAX_RPM,) * 4), (self.NUM_DRONES, 1))
        return spaces.Box(low=low, high=high, dtype=np.float32)

    def _computeObs(self):
        return np.stack([self._getDroneStateVector(d) for d in range(self.NUM_DRONES)])

    # no task: the reference's placeholders (`envs/CtrlAviary.py:146-200`)
    def _computeReward(self):
        return -1

    def _computeTerminated(self):
        return False

    def _computeTruncated(self):
        return False

    def _computeInfo(self):
        return {"answer": 42}
