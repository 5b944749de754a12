"""`BaseControl`: controller base class (reference `control/BaseControl.py:8-216`).

Keeps the reference's surface — `reset()`, `computeControlFromState()`, `setPIDCoefficients()`,
`_getURDFParameter()` — for controllers whose arithmetic runs in the HIP library.
"""
import numpy as np

from ..params import DroneParams, G
from ..utils.enums import DroneModel


class BaseControl(object):
    """Base class for control."""

    def __init__(self, drone_model: DroneModel, g: float = G):
        self.DRONE_MODEL = drone_model
        self._drone_params = DroneParams(drone_model)
        self._g = g
        self.GRAVITY = g * self._getURDFParameter('m')
        self.KF = self._getURDFParameter('kf')
        self.KM = self._getURDFParameter('km')
        self.reset()

    def reset(self):
        """Reset the control classes: the general use counter is set to zero."""
        self.control_counter = 0

    def computeControlFromState(self, control_timestep, state, target_pos, target_rpy=np.zeros(3),
                                target_vel=np.zeros(3), target_rpy_rates=np.zeros(3)):
        """`computeControl` fed from a (20,) state vector as returned in `obs` by `CtrlAviary.step()`."""
        return self.computeControl(control_timestep=control_timestep, cur_pos=state[0:3], cur_quat=state[3:7],
                                   cur_vel=state[10:13], cur_ang_vel=state[13:16], target_pos=target_pos,
                                   target_rpy=target_rpy, target_vel=target_vel, target_rpy_rates=target_rpy_rates)

    def computeControl(self, control_timestep, cur_pos, cur_quat, cur_vel, cur_ang_vel, target_pos,
                       target_rpy=np.zeros(3), target_vel=np.zeros(3), target_rpy_rates=np.zeros(3)):
        raise NotImplementedError

    def setPIDCoefficients(self, p_coeff_pos=None, i_coeff_pos=None, d_coeff_pos=None, p_coeff_att=None,
                           i_coeff_att=None, d_coeff_att=None):
        """Sets the coefficients of a PID controller (raises if the controller has none)."""
        names = ['P_COEFF_FOR', 'I_COEFF_FOR', 'D_COEFF_FOR', 'P_COEFF_TOR', 'I_COEFF_TOR', 'D_COEFF_TOR']
        if not all(hasattr(self, a) for a in names):
            raise AttributeError("[ERROR] in BaseControl.setPIDCoefficients(), not all PID coefficients exist as "
                                 "attributes in the instantiated control class.")
        for name, val in zip(names, (p_coeff_pos, i_coeff_pos, d_coeff_pos, p_coeff_att, i_coeff_att, d_coeff_att)):
            if val is not None:
                setattr(self, name, np.asarray(val, dtype=np.float64))
        self._coefficients_changed()

    def _coefficients_changed(self):
        pass

    def _getURDFParameter(self, parameter_name: str):
        """Reads a parameter of the controlled airframe (reference `_getURDFParameter`, `:181-216`)."""
        P = self._drone_params
        table = {'m': P.M, 'ixx': P.J[0, 0], 'iyy': P.J[1, 1], 'izz': P.J[2, 2], 'arm': P.L,
                 'thrust2weight': P.THRUST2WEIGHT_RATIO, 'kf': P.KF, 'km': P.KM, 'max_speed_kmh': P.MAX_SPEED_KMH,
                 'gnd_eff_coeff': P.GND_EFF_COEFF, 'prop_radius': P.PROP_RADIUS, 'drag_coeff_xy': P.DRAG_COEFF[0],
                 'drag_coeff_z': P.DRAG_COEFF[2], 'dw_coeff_1': P.DW_COEFF_1, 'dw_coeff_2': P.DW_COEFF_2,
                 'dw_coeff_3': P.DW_COEFF_3, 'length': P.COLLISION_H, 'radius': P.COLLISION_R,
                 'collision_z_offset': P.COLLISION_Z_OFFSET}
        return table[parameter_name]
