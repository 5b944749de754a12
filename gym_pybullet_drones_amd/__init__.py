"""MI355X-native vectorised quadrotor simulator — drop-in for the hot path of gym-pybullet-drones.

Installable (`pip install -e .`, pyproject.toml: the build step compiles `csrc/libgpd.so` for gfx950);
`gym-pybullet-drones_amd` at the repository root is a symbolic link to this directory.

    from gym_pybullet_drones_amd.envs import HoverAviary, MultiHoverAviary      # reference-shaped, 1 aviary
    from gym_pybullet_drones_amd.envs import VectorHoverAviary                  # E aviaries per launch
    from gym_pybullet_drones_amd.control import DSLPIDControl
    from gym_pybullet_drones_amd.utils.enums import DroneModel, Physics, ActionType, ObservationType

`install_as("gym_pybullet_drones")` makes the reference's own import paths resolve to this
package (`from gym_pybullet_drones.envs.HoverAviary import HoverAviary`).
"""
import importlib
import sys

from ._gym_shim import register as _register

from ._version import __version__  # noqa: F401

_register(id='ctrl-aviary-v0', entry_point='gym_pybullet_drones_amd.envs:CtrlAviary')
_register(id='velocity-aviary-v0', entry_point='gym_pybullet_drones_amd.envs:VelocityAviary')
_register(id='hover-aviary-v0', entry_point='gym_pybullet_drones_amd.envs:HoverAviary')
_register(id='multihover-aviary-v0', entry_point='gym_pybullet_drones_amd.envs:MultiHoverAviary')


def install_as(name: str = "gym_pybullet_drones"):
    """Alias this package (and its envs/control/utils modules) under another top-level name."""
    me = sys.modules[__name__]
    sys.modules[name] = me
    for sub in ("envs", "envs.BaseAviary", "envs.BaseRLAviary", "envs.HoverAviary", "envs.MultiHoverAviary",
                "envs.CtrlAviary", "envs.VelocityAviary", "envs.VectorAviary", "control", "control.BaseControl",
                "control.DSLPIDControl", "utils", "utils.enums", "utils.Logger", "utils.utils"):
        sys.modules[f"{name}.{sub}"] = importlib.import_module(f"{__name__}.{sub}")
    return me
