"""`VelocityAviary`: high-level planning aviary — velocity commands tracked by the embedded DSLPID
controllers, 20-float state observations (reference `envs/VelocityAviary.py:11-228`).

The action of one drone is `[vx, vy, vz, fraction of SPEED_LIMIT]`; the reference turns it into
`target_vel = SPEED_LIMIT * |a[3]| * unit(a[0:3])`, `target_pos = current position`,
`target_rpy = [0, 0, current yaw]` and calls `DSLPIDControl.computeControl` per drone
(`_preprocessAction`, `:129-168`).  That is exactly `GPD_ACT_VEL` of the fused kernel
(`include/gpd.h`), so `step()` is one launch; the controllers are always built for CF2X (`:59-60`).
"""
import numpy as np

from .._gym_shim import spaces
from ..utils.enums import ActionType, DroneModel, Physics
from ._state_vector_aviary import StateVectorAviary


class VelocityAviary(StateVectorAviary):
    """Multi-drone environment class for high-level planning."""

    _STOCK_ACTION_CODE = ActionType.VEL.code

    def __init__(self, drone_model: DroneModel = DroneModel.CF2X, num_drones: int = 1, neighbourhood_radius: float = np.inf,
                 initial_xyzs=None, initial_rpys=None, physics: Physics = Physics.PYB, pyb_freq: int = 240, ctrl_freq: int = 240,
                 gui=False, record=False, obstacles=False, user_debug_gui=True, output_folder='results', device=None):
        # (the reference's argument list, `envs/VelocityAviary.py:16-30`, + `device`)
        if drone_model not in (DroneModel.CF2X, DroneModel.CF2P):
            # the reference builds no controller for other models and fails at the first step (:59-60, :152)
            raise ValueError("[ERROR] in VelocityAviary.__init__(), no controller is available for the specified drone_model")
        super().__init__(drone_model, num_drones, neighbourhood_radius, initial_xyzs, initial_rpys, physics, pyb_freq, ctrl_freq,
                         gui, record, obstacles, user_debug_gui, output_folder=output_folder, device=device)
        self.SPEED_LIMIT = 0.03 * self.MAX_SPEED_KMH * (1000 / 3600)

    def _actionSpace(self):
        low = np.tile(np.array([-1., -1., -1., 0.]), (self.NUM_DRONES, 1))
        return spaces.Box(low=low, high=np.ones((self.NUM_DRONES, 4)), dtype=np.float32)

    def _preprocessAction(self, action):
        """(NUM_DRONES, 4) velocity commands -> (NUM_DRONES, 4) RPMs.  `step()` does not call this (the mapping
        is fused into the kernel); kept for callers that use it directly -- it advances the embedded controllers."""
        from ..control.DSLPIDControl import pid_rpm_for_action
        return pid_rpm_for_action(self, np.asarray(action, dtype=np.float64).reshape(self.NUM_DRONES, 4),
                                  act_type=ActionType.VEL)
