"""The GPU test of the pair-balance experiment (scratch/exp_r05/pair_balance.patch): copy next to tests/ after applying the patch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pair_balanced_groups_change_no_bit_and_even_out_the_cus(gpu_device):
    """Round 5 (VERDICT r04 "next" #3): at every binning `gpd_swarm_bin` re-deals the groups of 64 sorted drones to the force kernel's
    workgroup slots by the batches their wake lists held, boustrophedon over rounds of 256 slots (workgroup b runs on CU b mod 256,
    profiles/r05_workgroup_placement.txt), so that the four groups sharing a CU add up to equal work.  Sums are integers: state vectors
    and forces are bit for bit those of the unbalanced world (`balance_groups=False`: group b on workgroup b, rounds 2-4) and of a world
    that bins every sub-step; the permutation stays a permutation; and the heaviest CU of the balanced placement carries less than the
    heaviest CU of the identity placement would."""
    from gym_pybullet_drones_amd.envs import SwarmAviary
    from gym_pybullet_drones_amd.utils.enums import Physics
    rng = np.random.default_rng(55)
    N = 65536
    # a scene with a dense patch: 12 layers on a 4 m lattice, every fourth site of one quadrant doubled up (more pairs per group there)
    side = int(np.ceil(np.sqrt(N / 12)))
    idx = rng.permutation(side * side * 12)[:N]
    layer, site = idx // (side * side), idx % (side * side)
    xy = np.stack([(site % side) * 4.0 + layer % 4, (site // side) * 4.0 + layer // 4], axis=1) - 2.0 * side + rng.uniform(-0.1, 0.1, size=(N, 2))
    dense = (xy[:, 0] > 0) & (xy[:, 1] > 0)
    xy[dense] *= 0.8                                      # the quadrant pulled together: 1.56 x the density
    xyz = np.concatenate([xy, (1.0 + layer)[:, None]], axis=1)
    kw = dict(initial_xyzs=xyz, initial_rpys=rng.uniform(-0.05, 0.05, size=(N, 3)), physics=Physics.PYB_GND_DRAG_DW, device=gpu_device)
    bal, plain, every = SwarmAviary(N, rebin_every=8, **kw), SwarmAviary(N, rebin_every=8, balance_groups=False, **kw), SwarmAviary(N, rebin_every=1, **kw)
    assert bal.balanced and not plain.balanced and bal._group_perm is not None and plain._group_perm is None
    va, _ = bal.reset(); vb, _ = plain.reset(); vc, _ = every.reset()
    assert torch.equal(va, vb) and torch.equal(va, vc)
    rpm = torch.as_tensor((bal.HOVER_RPM * (1 + 0.01 * rng.uniform(-1, 1, size=(N, 4)))).astype(np.float32), device=gpu_device)
    G = (bal.n_rows + 63) // 64
    for k in range(26):                                   # four binnings: the second one is the first that has counts to deal by
        va, *_ = bal.step(rpm); vb, *_ = plain.step(rpm); vc, *_ = every.step(rpm)
        assert torch.equal(va, vb) and torch.equal(va, vc), k
        assert torch.equal(bal.dw_force[:N], plain.dw_force[:N]) and torch.equal(bal.dw_force[:N], every.dw_force[:N]), k
    perm = bal._group_perm.cpu().numpy()
    cur = int(perm[2 * G])
    p = perm[cur * G:(cur + 1) * G]
    assert sorted(p.tolist()) == list(range(G)) and not np.array_equal(p, np.arange(G))
    # work per CU (batches of the four waves of the groups on it), balanced slots vs the same groups on their own index
    nb = bal._pair_nb.cpu().numpy().astype(np.int64)[:, :, 0] & 0xffff          # [slot][wave]
    w_slot = nb.sum(axis=1)
    w_group = np.zeros(G, dtype=np.int64); w_group[p] = w_slot                   # slot b holds group p[b]
    cu = lambda w: np.bincount(np.arange(len(w)) % 256, weights=w, minlength=256)   # noqa: E731
    print("batches per CU: balanced max %.0f, identity max %.0f, mean %.1f" % (cu(w_slot).max(), cu(w_group).max(), cu(w_slot).mean()))
    assert cu(w_slot).max() < cu(w_group).max() and cu(w_slot).max() <= 1.1 * cu(w_slot).mean() + 4
