"""PegInsertionSide diagnostics on the GPU: contact-count distribution, worst pairs, step-time split"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import maniskill_amd.envs  # noqa
import gymnasium as gym

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
env = gym.make("PegInsertionSide-v1", num_envs=N, obs_mode="state", control_mode="pd_joint_delta_pos")
base = env.unwrapped
env.reset(seed=0)
px = base.scene.px
model = base.scene.model
worst = torch.zeros(N, dtype=torch.int32, device="cuda")
for t in range(120):
    env.step(2 * torch.rand(N, 8, device="cuda") - 1)
    cnt = px.read_internal("contact_count", model.n_pair)
    cnt = cnt.reshape(model.n_pair, N)
    tot = cnt.sum(0).int()
    worst = torch.maximum(worst, tot)
print("overflow envs", px.overflow_count())
print("per-env max contacts: mean %.1f  p50 %d p90 %d p99 %d max %d" % (worst.float().mean(), *[int(torch.quantile(worst.float(), q)) for q in (0.5, 0.9, 0.99, 1.0)]))
e = int(tot.argmax())
print("env", e, "contacts now", int(tot[e]))
A = model.arrays
names = model.shape_names if hasattr(model, "shape_names") else None
for p in torch.nonzero(cnt[:, e]).flatten().tolist():
    sa, sb = int(A["pair_shape"][p][0]), int(A["pair_shape"][p][1])
    print("  pair", p, "shapes", sa, sb, "types", int(A["shape_type"][sa]), int(A["shape_type"][sb]), "kind/idx", int(A["shape_body_kind"][sa]), int(A["shape_body_index"][sa]), int(A["shape_body_kind"][sb]), int(A["shape_body_index"][sb]), "n", int(cnt[p, e]))
print("row names", getattr(model, "row_names", None))

# time split
def timed(f, n=30):
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.time() - t) / n * 1e3
a = 2 * torch.rand(N, 8, device="cuda") - 1
print("env.step        %.3f ms" % timed(lambda: env.step(a)))
print("physics only    %.3f ms" % timed(lambda: (px.step(5), px.gpu_fetch_all())))
print("evaluate        %.3f ms" % timed(lambda: base.evaluate()))
info = base.get_info()
print("get_info        %.3f ms" % timed(lambda: base.get_info()))
print("get_obs         %.3f ms" % timed(lambda: base.get_obs(info)))
print("reward          %.3f ms" % timed(lambda: base.get_reward(obs=None, action=a, info=info)))
print("is_grasping     %.3f ms" % timed(lambda: base.agent.is_grasping(base.peg, max_angle=20)))
