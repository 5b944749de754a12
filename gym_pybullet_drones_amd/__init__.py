"""Import alias: the package lives in the sibling directory `gym-pybullet-drones_amd/` (a hyphen is
not importable), this stub points `gym_pybullet_drones_amd` at it."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gym-pybullet-drones_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _f.name, "exec"))
del _f
