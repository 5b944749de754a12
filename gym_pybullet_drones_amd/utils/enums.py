"""Enumerations shared by the aviaries, the controllers and the C-ABI.

Member names and string values are kept value-compatible with the reference
(`gym_pybullet_drones/utils/enums.py:3-48`) so that user code written against it
(`Physics.DYN`, `ActionType("one_d_rpm")`, ...) keeps working.  What is new here is
the integer side: every enum that reaches the HIP kernel carries the code the
C-ABI in `include/gpd.h` expects (`GPD_MODEL_*`, `GPD_ACT_*`, `GPD_PHYS_*`).
"""
from enum import Enum


class DroneModel(Enum):
    CF2X = "cf2x"    # Crazyflie 2.x, X configuration
    CF2P = "cf2p"    # Crazyflie 2.x, + configuration
    RACE = "racer"   # racing quad, X configuration

    @property
    def code(self) -> int:
        """`GPD_MODEL_*` value of include/gpd.h."""
        return {"cf2x": 0, "cf2p": 1, "racer": 2}[self.value]


class Physics(Enum):
    PYB = "pyb"
    DYN = "dyn"
    PYB_GND = "pyb_gnd"
    PYB_DRAG = "pyb_drag"
    PYB_DW = "pyb_dw"
    PYB_GND_DRAG_DW = "pyb_gnd_drag_dw"

    @property
    def flags(self) -> int:
        """Bit mask of the aerodynamic add-on terms (`GPD_PHYS_GND|DRAG|DW`).

        The explicit integrator is the only engine in this package: the `PYB_*`
        members select which of the reference's add-on force models
        (`BaseAviary.py:354-367`) are evaluated *inside* the explicit integrator
        (SURVEY.md App. A.4).  Plain `PYB` maps to no add-on, i.e. to `DYN`.
        """
        return {"pyb": 0, "dyn": 0, "pyb_gnd": 1, "pyb_drag": 2, "pyb_dw": 4,
                "pyb_gnd_drag_dw": 7}[self.value]

    @property
    def ground(self) -> bool:
        """Whether the ground plane acts (`GPD_PHYS_GROUND`).  In the reference every `PYB*` member resolves contact with
        the plane loaded at `BaseAviary.py:479` through Bullet's solver, while `DYN` overwrites the pose every step
        (`:865-875`) and a drone falls through z = 0.  Here: `PYB*` -> plane on (the minimal contact model of
        include/gpd.h), `DYN` -> off, exactly the reference's `DYN`."""
        return self != Physics.DYN

    @property
    def damping(self) -> bool:
        """Whether Bullet's default damping CAN act (`GPD_PHYS_DAMP`, with `pyb_like="damped"`): every `PYB*` run of the reference
        integrates the drone as a Bullet multibody with linear and angular damping 0.04 (`p.loadURDF`, `BaseAviary.py:488-494`; the
        reference never calls `changeDynamics`), `DYN` has none (`:831-877`)."""
        return self != Physics.DYN

    def mask(self, pyb_like=None) -> int:
        """The `GPD_PHYS_*` mask a kernel launch gets for this member: the add-on force models, plus -- for `PYB*` -- the stand-ins
        for what Bullet itself adds to such a run:
            pyb_like=False / "off"     nothing: exactly "the reference's explicit `Physics.DYN` integrator + the selected add-on models"
            pyb_like=True / "ground"   the ground plane (`GPD_PHYS_GROUND`; the advertised observation space, z >= 0, needs it)
            pyb_like="damped"          the ground plane AND Bullet's default multibody damping 0.04 (`GPD_PHYS_DAMP`)
        `pyb_like=None` reads the process default (`set_pyb_like`, env `GPD_PYB_LIKE` = 0 | 1 | damped): "ground".  The damping is
        OPT-IN since round 5: it is restated from the Bullet sources and pinned against nothing (tests/test_pybullet_optional.py
        pins it the first time a box has PyBullet); above the plane a default `Physics.PYB*` run is therefore bit for bit the
        reference's explicit integrator with the selected add-on models."""
        mode = _pyb_mode(_pyb_like if pyb_like is None else pyb_like)
        m = self.flags
        if mode != "off" and self.ground:
            m |= PHYS_GROUND
        if mode == "damped" and self.damping:
            m |= PHYS_DAMP
        return m


class ImageType(Enum):
    RGB = 0
    DEP = 1
    SEG = 2
    BW = 3


class ActionType(Enum):
    RPM = "rpm"
    PID = "pid"
    VEL = "vel"
    ONE_D_RPM = "one_d_rpm"
    ONE_D_PID = "one_d_pid"

    @property
    def code(self) -> int:
        """`GPD_ACT_*` value of include/gpd.h."""
        return {"rpm": 0, "pid": 1, "vel": 2, "one_d_rpm": 3, "one_d_pid": 4}[self.value]

    @property
    def dim(self) -> int:
        """Per-drone action width (`BaseRLAviary.py:141-146`)."""
        return {"rpm": 4, "pid": 3, "vel": 4, "one_d_rpm": 1, "one_d_pid": 1}[self.value]

    @property
    def uses_pid(self) -> bool:
        return self in (ActionType.PID, ActionType.VEL, ActionType.ONE_D_PID)


class ObservationType(Enum):
    KIN = "kin"
    RGB = "rgb"


_warned_pyb = False


def warn_if_pyb(physics) -> None:
    """One `UserWarning` per process when a `Physics.PYB*` member is requested: this package has no Bullet -- the
    explicit `Physics.DYN` integrator runs instead (with the selected add-on force models inside it), so there is
    no Featherstone integrator and no collision shapes; the ground is the contact model of `GPD_PHYS_GROUND` (a plane at
    z = 0 the airframe's collision cylinder rests on), not Bullet's solver, and the damping is `GPD_PHYS_DAMP` (Bullet's
    default multibody damping restated inside the explicit integrator)."""
    global _warned_pyb
    if _warned_pyb or not isinstance(physics, Physics) or physics == Physics.DYN:
        return
    _warned_pyb = True
    import warnings
    mode = _pyb_mode(_pyb_like)
    extra = {"off": ".  Trajectories follow the reference's Physics.DYN (pyb_like is off: no ground plane, no damping).",
             "ground": ".  On top of it this package adds a ground plane at z = 0 (GPD_PHYS_GROUND; a model the reference's Physics.DYN does "
                       "NOT have, standing in for Bullet's contact solver): above the plane trajectories follow the reference's Physics.DYN "
                       "exactly.  Bullet's default multibody damping of 0.04 (GPD_PHYS_DAMP; restated from the Bullet sources, parity "
                       "unpinned) is opt-in: set_pyb_like('damped') / GPD_PYB_LIKE=damped / pyb_like='damped'.  set_pyb_like(False) / "
                       "GPD_PYB_LIKE=0 / pyb_like=False remove the plane too.",
             "damped": ".  On top of it this package adds two models the reference's Physics.DYN does NOT have, standing in for what Bullet "
                       "does in Physics.PYB*: a ground plane at z = 0 (GPD_PHYS_GROUND) and Bullet's default multibody damping of 0.04 "
                       "(GPD_PHYS_DAMP; restated from the Bullet sources, parity unpinned).  set_pyb_like(False) / GPD_PYB_LIKE=0 / "
                       "pyb_like=False turn both off: trajectories then follow the reference's Physics.DYN exactly."}[mode]
    warnings.warn(f"Physics.{physics.name} was requested, but PyBullet's integrator does not exist in this package: the "
                  f"explicit Physics.DYN integrator (envs/BaseAviary.py:815-877 of the reference) is used instead"
                  + (", with the " + "/".join(n for b, n in ((1, "ground-effect"), (2, "drag"), (4, "downwash")) if physics.flags & b)
                     + " model(s) evaluated inside it" if physics.flags else "") + extra,
                  UserWarning, stacklevel=3)


#: physics add-on bits, mirrored in include/gpd.h
PHYS_GND, PHYS_DRAG, PHYS_DW, PHYS_GROUND, PHYS_DAMP = 1, 2, 4, 8, 16

import os as _os


def _pyb_mode(v) -> str:
    """False / 0 / "off" -> "off";  True / 1 / "ground" -> "ground";  "damped" (or 2) -> "damped" """
    if isinstance(v, str):
        t = v.strip().lower()
        if t in ("0", "false", "off", "no", ""):
            return "off"
        if t in ("damped", "damp", "2", "bullet"):
            return "damped"
        if t in ("1", "true", "on", "yes", "ground"):
            return "ground"
        raise ValueError(f"pyb_like / GPD_PYB_LIKE: {v!r} is none of 0 | off, 1 | ground, damped")
    if v is None:
        return "ground"
    if isinstance(v, bool):
        return "ground" if v else "off"
    return {0: "off", 1: "ground"}.get(int(v), "damped")


_pyb_like = _pyb_mode(_os.environ.get("GPD_PYB_LIKE", "1"))


def set_pyb_like(on) -> None:
    """Process-wide default of `Physics.mask()`: what `Physics.PYB*` adds to the explicit integrator -- False: nothing; True (the
    default): the ground plane; "damped": the plane and Bullet's default damping (both extensions with no counterpart in the
    reference's explicit integrator, include/gpd.h).  The drop-in classes keep the reference's constructor signatures, so this
    switch (or `GPD_PYB_LIKE=0|1|damped`) is their knob; the batched classes also take `pyb_like=` per instance."""
    global _pyb_like
    _pyb_like = _pyb_mode(on)
#: raw-RPM action clipped to [0, MAX_RPM] (CtrlAviary, `CtrlAviary.py:140`); kernel-only code
ACT_RAW_RPM = 5
#: RPMs taken as they are (output of a user subclass's own `_preprocessAction`); kernel-only code
ACT_DIRECT_RPM = 6
