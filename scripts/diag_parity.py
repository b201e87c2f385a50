import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from tests.test_gpu_parity import *
model = panda_tabletop_model()
N = 1024
gpu, cpu = make_pair(model, N)
q, qd, tq, cube = random_tabletop_state(N, 2)
for px in (gpu, cpu):
    set_state(px, model, N, q, qd, tq, cube)
    px.step(1)
a, b = get_state(gpu, model, N), get_state(cpu, model, N)
same = (a["cnt"] == b["cnt"]).all(0)
print("same cnt frac", same.float().mean().item())
err = (a["qd"] - b["qd"]).abs()
print("per-dof max qd err", err.max(0)[0])
worst = err.max(1)[0].argsort(descending=True)[:8]
for e in worst.tolist():
    pairs = torch.nonzero(b["cnt"][:, e]).flatten().tolist()
    print("env", e, "err", err[e].numpy().round(5), "same", same[e].item(), "pairs", [(p, int(b["cnt"][p, e]), model.shape_owner[model.arrays['pair_shape'][p][0]], model.shape_owner[model.arrays['pair_shape'][p][1]]) for p in pairs])
    print("   qd gpu", a["qd"][e].numpy().round(4)); print("   qd cpu", b["qd"][e].numpy().round(4))
bad = torch.nonzero(~same).flatten().tolist()
for e in bad[:6]:
    d = torch.nonzero(a["cnt"][:, e] != b["cnt"][:, e]).flatten().tolist()
    print("cnt mismatch env", e, [(p, int(a["cnt"][p, e]), int(b["cnt"][p, e]), model.shape_owner[model.arrays['pair_shape'][p][0]], model.shape_owner[model.arrays['pair_shape'][p][1]]) for p in d])
