import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from test_gpu_rollout import _pair, _actions
dev = torch.device("cuda:0")
for (act, flags, D, S, model) in [("rpm", 0, 1, 1, "cf2x"), ("pid", 0, 1, 1, "cf2x"), ("rpm", 4, 2, 8, "cf2x")]:
    rng = np.random.default_rng(1)
    E, K = 1536 // D, 6
    a, b = _pair(act, flags, D, S, model, dev, E, rng, auto_reset=False)
    acts = torch.as_tensor(_actions(rng, act, (K, E, D), a.P.HOVER_RPM).astype(np.float32), device=dev)
    obs_s = []
    for k in range(K):
        o, r, te, tr = a.step(acts[k]); obs_s.append(o.clone())
    obs, rew, te, tr = b.rollout(acts)
    d = (torch.stack(obs_s) - obs).abs()
    print(act, flags, D, S, "max diff per step", d.amax(dim=(1, 2)).cpu().numpy(), "per col step0", d[0].amax(dim=0).cpu().numpy())
    print("  kin diff", (a.kin - b.kin).abs().amax(dim=1).cpu().numpy())
