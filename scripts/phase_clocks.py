"""cycles per phase of k_solve16 (debug build with -DMSSIM_PHASE_CLOCKS, loaded through MSSIM_LIB).
usage: python scripts/phase_clocks.py [env_id] [N] [steps]   (builds the variant library itself)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "maniskill_amd", "_native", "libmssim_clk.so")
src = os.path.join(ROOT, "maniskill_amd", "csrc", "mssim_kernels.hip")
if "--build" in sys.argv or not os.path.exists(lib):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-fno-hip-fp32-correctly-rounded-divide-sqrt",
                    "-DMSSIM_PHASE_CLOCKS", "-o", lib, src], check=True)
    if "--build" in sys.argv:
        sys.exit(0)
os.environ["MSSIM_LIB"] = lib
import torch
import maniskill_amd.envs  # noqa
import gymnasium as gym

args = [a for a in sys.argv[1:] if not a.startswith("--")]
env_id = args[0] if len(args) > 0 else "PickCube-v1"
N = int(args[1]) if len(args) > 1 else 4096
steps = int(args[2]) if len(args) > 2 else 100
env = gym.make(env_id, num_envs=N, obs_mode="state", control_mode="pd_joint_delta_pos", **({"robot_uids": os.environ["MS_ROBOT"]} if os.environ.get("MS_ROBOT") else {}))  # (MS_ROBOT=fetch with Empty-v1)
env.reset(seed=0)
dbg = ctypes.CDLL(lib)
buf = (ctypes.c_ulonglong * 32)()
names = ["state load", "RNEA+CRBA tail", "Gauss-Jordan", "free bodies", "row build", "PGS", "pair impulses", "writeback+FK",
         "tail: copy-out + task epilogue", "#launches with a patch > 4 points", "np: task setup (shapes from LDS)", "np: plane", "np: box-box (one lane per pair)", "np: stage A rounds (plane / one-lane box-box) + staging",
         "#coop MPR task slots (max over groups)", "end: FK + carry", "end: velocity sum", "pgs: integrate/loop head", "pgs: limit rows", "pgs: contacts (LDS)", "pgs: contacts (global)",
         "np: contact patches + records", "np: shape table", "np: cull", "np: coop box-box (stage C)", "np: coop MPR (stage B)",
         "#survivor tasks per wave", "#plane tasks", "#box-box tasks", "#max contacts in block", "#contacts in block (4 envs)", "np: mesh triangles (stage T)"]
for phase_name, k in (("fresh episodes", steps), ("after %d more unreset steps" % steps, steps)):
    torch.cuda.synchronize()
    dbg.mssim_debug_phase_clocks(buf, 1)
    for _ in range(k):
        env.step(2 * torch.rand(N, env.unwrapped.single_action_space.shape[0], device="cuda") - 1)
    torch.cuda.synchronize()
    dbg.mssim_debug_phase_clocks(buf, 1)
    tot = sum(buf[i] for i in range(32))
    launches = k
    blocks = (N + 3) // 4
    print(f"{env_id} N={N} {phase_name}: {tot / launches / blocks:.0f} cycles per block-launch")
    for i, nm in enumerate(names):
        if not buf[i]:
            continue
        print(f"  {nm:14s} {buf[i] / launches / blocks:9.0f} cycles  {100.0 * buf[i] / tot:5.1f} %")

# slowest blocks of the last launch
import numpy as np
nb = min(2048, 4 * 8 * (((N + 15) // 16 + 7) // 8))  # one record per wave (4 envs); blocks are 4 waves
arr = (ctypes.c_uint * (nb * 32))()
dbg.mssim_debug_phase_blocks(arr, nb)
a = np.frombuffer(arr, dtype=np.uint32).reshape(nb, 32).astype(np.float64)
cyc = a[:, :26].sum(1) - a[:, 14]
order = np.argsort(-cyc)
print(f"last launch: block cycles mean {cyc.mean():.0f}  p50 {np.median(cyc):.0f}  p99 {np.percentile(cyc, 99):.0f}  max {cyc.max():.0f}")
for b in order[:3]:
    print(f" block {b}: total {cyc[b]:.0f}; contacts(4 envs, summed over substeps) {a[b, 30]:.0f}, max-per-env sum {a[b, 29]:.0f}, coop-MPR slots {a[b, 14]:.0f}, survivor tasks {a[b, 26]:.0f}, box-box {a[b, 28]:.0f}, plane {a[b, 27]:.0f}")
    for i, nm in enumerate(names):
        if i < 26 and i != 14 and a[b, i] > 0.01 * cyc[b]:
            print(f"    {nm:40s} {a[b, i]:9.0f}  {100 * a[b, i] / cyc[b]:5.1f} %")

print(" slowest 16 blocks: total | coop MPR | PGS contacts | row build | contacts | MPR slots")
for b in order[:16]:
    print(f"   {cyc[b]:9.0f} {a[b, 25]:9.0f} {a[b, 19] + a[b, 20]:9.0f} {a[b, 4]:9.0f} {a[b, 30]:5.0f} {a[b, 14]:4.0f}")
print(" blocks with MPR slots: %d of %d; mean total with / without: %.0f / %.0f" % ((a[:, 14] > 0).sum(), nb, cyc[a[:, 14] > 0].mean() if (a[:, 14] > 0).any() else 0, cyc[a[:, 14] == 0].mean()))

h = (ctypes.c_uint * 32)()
dbg.mssim_debug_mpr_hist(h)
print("MPR portal-discovery iterations (bins of 4):", list(h)[:16])
print("MPR refinement iterations       (bins of 4):", list(h)[16:])

mc = (ctypes.c_ulonglong * 8)()
dbg.mssim_debug_mpr_clocks(mc, 0)
if mc[4]:
    print("MPR as seen by thread 0 of each wave (whole run): calls %d, cycles to portal %.0f, refinement %.0f (%.1f iterations, %.0f cycles each), contact point %.0f"
          % (mc[4], mc[0] / mc[4], mc[1] / mc[4], mc[3] / mc[4], mc[1] / max(mc[3], 1), mc[2] / mc[4]))
