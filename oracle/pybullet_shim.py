"""A stand-in for the `pybullet` module, just large enough to run the reference's Physics.DYN path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  `pybullet` (pinned ^3.2.7, reference
pyproject.toml:19) is a third-party C++ engine that is neither vendored in the reference nor
installable here.  In Physics.DYN the reference uses it only as (a) a state store
(`resetBasePositionAndOrientation`/`resetBaseVelocity` write, `getBasePositionAndOrientation`/
`getBaseVelocity` read back the same numbers; envs/BaseAviary.py:517-519,865-875) and (b) three
quaternion utilities (oracle/bullet_math.py).  Installing this module as `sys.modules["pybullet"]`
lets the reference's own `BaseAviary`/`BaseRLAviary`/`HoverAviary`/`MultiHoverAviary`/
`DSLPIDControl` Python execute unmodified (tests/golden/make_golden.py), which pins the oracle.

`applyExternalForce`/`applyExternalTorque` do not integrate anything: they RECORD what the
reference asked for, so the force formulas of `_groundEffect`/`_drag`/`_downwash`
(envs/BaseAviary.py:715-811) can be captured and compared with the oracle's derivation.
`stepSimulation` raises: the Featherstone integrator (Physics.PYB*) is not restated.
"""
import xml.etree.ElementTree as _ET

import numpy as _np

from . import bullet_math as _bm

DIRECT, GUI = 2, 1
LINK_FRAME, WORLD_FRAME = 1, 2
URDF_USE_INERTIA_FROM_FILE = 2
COV_ENABLE_RGB_BUFFER_PREVIEW = COV_ENABLE_DEPTH_BUFFER_PREVIEW = COV_ENABLE_SEGMENTATION_MARK_PREVIEW = 0
ER_TINY_RENDERER = 0
ER_SEGMENTATION_MASK_OBJECT_AND_LINKINDEX = 0
STATE_LOGGING_VIDEO_MP4 = 0

_bodies = {}       # id -> dict(pos, orn, lin, ang, prop_offsets)
_next_id = [0]
applied = []       # records of applyExternalForce/Torque since the last clear_applied()


def connect(mode, options=""):
    return 0


def disconnect(physicsClientId=0):
    pass


def resetSimulation(physicsClientId=0):
    _bodies.clear()
    _next_id[0] = 0
    applied.clear()


def setGravity(x, y, z, physicsClientId=0):
    pass


def setRealTimeSimulation(flag, physicsClientId=0):
    pass


def setTimeStep(dt, physicsClientId=0):
    pass


def setAdditionalSearchPath(path, physicsClientId=0):
    pass


def loadURDF(fileName, basePosition=(0, 0, 0), baseOrientation=(0, 0, 0, 1), flags=0,
             physicsClientId=0, **kw):
    bid = _next_id[0]
    _next_id[0] += 1
    offs = []
    try:
        root = _ET.parse(fileName).getroot()
        for k in range(4):
            link = root.find(f"link[@name='prop{k}_link']")
            offs.append([float(s) for s in link.find("inertial").find("origin").attrib["xyz"].split()])
    except Exception:
        offs = []
    _bodies[bid] = dict(pos=tuple(float(v) for v in basePosition),
                        orn=tuple(float(v) for v in baseOrientation),
                        lin=(0.0, 0.0, 0.0), ang=(0.0, 0.0, 0.0),
                        prop_offsets=_np.array(offs, dtype=_np.float64))
    return bid


def getQuaternionFromEuler(rpy, physicsClientId=0):
    return _bm.quaternion_from_euler(rpy)


def getMatrixFromQuaternion(q, physicsClientId=0):
    return tuple(_bm.matrix_from_quaternion(q).reshape(9))


def getEulerFromQuaternion(q, physicsClientId=0):
    return _bm.euler_from_quaternion(q)


def resetBasePositionAndOrientation(bid, pos, orn, physicsClientId=0):
    b = _bodies[int(bid)]
    b["pos"] = tuple(float(v) for v in pos)
    b["orn"] = tuple(float(v) for v in orn)


def resetBaseVelocity(bid, linearVelocity=(0, 0, 0), angularVelocity=(0, 0, 0), physicsClientId=0):
    b = _bodies[int(bid)]
    b["lin"] = tuple(float(v) for v in linearVelocity)
    b["ang"] = tuple(float(v) for v in angularVelocity)


def getBasePositionAndOrientation(bid, physicsClientId=0):
    b = _bodies[int(bid)]
    return b["pos"], b["orn"]


def getBaseVelocity(bid, physicsClientId=0):
    b = _bodies[int(bid)]
    return b["lin"], b["ang"]


def getLinkStates(bid, linkIndices, computeLinkVelocity=0, computeForwardKinematics=0, physicsClientId=0):
    """World position of each link's inertial frame: prop links 0-3 sit at base + R*offset, link 4
    (centre of mass) at the base.  Only element [i][0] (the position) is meaningful."""
    b = _bodies[int(bid)]
    R = _bm.matrix_from_quaternion(b["orn"])
    out = []
    for li in linkIndices:
        off = b["prop_offsets"][li] if li < 4 else _np.zeros(3)
        wp = _np.array(b["pos"]) + R @ off
        out.append((tuple(wp), b["orn"], (0, 0, 0), (0, 0, 0, 1), tuple(wp), b["orn"], b["lin"], b["ang"]))
    return tuple(out)


def applyExternalForce(objectUniqueId, linkIndex, forceObj, posObj, flags, physicsClientId=0):
    applied.append(("force", int(objectUniqueId), int(linkIndex),
                    tuple(float(v) for v in forceObj), tuple(float(v) for v in posObj), int(flags)))


def applyExternalTorque(objectUniqueId, linkIndex, torqueObj, flags, physicsClientId=0):
    applied.append(("torque", int(objectUniqueId), int(linkIndex),
                    tuple(float(v) for v in torqueObj), None, int(flags)))


def clear_applied():
    applied.clear()


def stepSimulation(physicsClientId=0):
    raise RuntimeError("pybullet_shim: the Bullet integrator (Physics.PYB*) is not available; "
                       "only Physics.DYN can run on the shim")


def getCameraImage(*a, **k):
    raise RuntimeError("pybullet_shim: no renderer")


def configureDebugVisualizer(*a, **k):
    pass


def addUserDebugParameter(*a, **k):
    return 0


def readUserDebugParameter(*a, **k):
    return 0
