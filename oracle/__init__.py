"""CPU oracle of the hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Float64 restatements of the reference algorithm (utiasDSL/gym-pybullet-drones, files cited per
function) used ONLY by `tests/`, by `__graft_entry__.smoke()` and by the `cpu_baseline` leg of
`bench.py`, as the checker the HIP path is compared with.  Nothing under
`gym_pybullet_drones_amd/` imports this package; the product path fails loudly when the HIP
library is missing instead of falling back to anything here.

Modules
  bullet_math.py     Bullet 3.2.x quaternion utilities the reference calls through `pybullet`
                     (third-party, pinned `pybullet ^3.2.7` in the reference's pyproject.toml:19,
                     source not vendored) restated from the published btMatrix3x3/btQuaternion/
                     pybullet.c formulas.
  pybullet_shim.py   a stand-in `pybullet` module (state store + the utilities above) that lets
                     the REFERENCE'S OWN PYTHON run in this container for golden-vector generation
                     (tests/golden/make_golden.py).
  aviary_oracle.py   per-drone, loop-structured restatement of BaseAviary(DYN)+BaseRLAviary+
                     Hover/MultiHoverAviary+DSLPIDControl — the `cpu_baseline` "port".
  batched_oracle.py  the same arithmetic vectorised over drones with numpy (float64), for parity
                     checks at sizes where the loop version is too slow.
  gpd_oracle.c       the same arithmetic in plain C (float64), for full-size parity runs.

Parity pinning: the reference ships no golden vectors and its tests never run Physics.DYN
(SURVEY.md §4).  The oracle is pinned against the reference's own code executed here over
`pybullet_shim` (fixtures in tests/golden/, generator committed).  What remains unpinned is the
shim itself, i.e. the four Bullet utility formulas, which are cross-checked against scipy.
"""
