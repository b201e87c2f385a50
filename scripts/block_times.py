"""Per-block start / duration / placement of the fused kernel's most recent launch, measured in the
product kernel (debug build with -DMSSIM_BLOCK_TIMES: two clock reads per block).
usage: python scripts/block_times.py [env_id] [N] [steps]   (--build: only build the variant library)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "maniskill_amd", "_native", "libmssim_blk.so")
src = os.path.join(ROOT, "maniskill_amd", "csrc", "mssim_kernels.hip")
if "--build" in sys.argv or not os.path.exists(lib):
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-fno-hip-fp32-correctly-rounded-divide-sqrt",
                    "-DMSSIM_BLOCK_TIMES", "-o", lib, src], check=True)
    if "--build" in sys.argv:
        sys.exit(0)
os.environ["MSSIM_LIB"] = lib
import numpy as np
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
nb = min(2048, 4 * 8 * (((N + 15) // 16 + 7) // 8))  # one record per wave (4 envs); blocks are 4 waves
arr = (ctypes.c_uint * (nb * 32))()
for rep in range(3):
    for _ in range(steps):
        env.step(2 * torch.rand(N, env.unwrapped.single_action_space.shape[0], device="cuda") - 1)
    torch.cuda.synchronize()
    dbg.mssim_debug_phase_blocks(arr, nb)
    a = np.frombuffer(arr, dtype=np.uint32).reshape(nb, 32).astype(np.int64)
    used = a[:, 2] > 0
    a = a[used]
    start = (a[:, 0] | (a[:, 1] << 32)).astype(np.float64)
    start = (start - start.min()) * 0.01  # us (100 MHz)
    dur = a[:, 2] * 0.01
    end = start + dur
    cyc = a[:, 3].astype(np.float64)
    print(f"launch after {steps * (rep + 1)} steps: {len(a)} blocks, span {end.max():.1f} us; duration mean {dur.mean():.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f} us; "
          f"start offset p50 {np.median(start):.1f} p99 {np.percentile(start, 99):.1f} max {start.max():.1f} us; core clock {np.median(cyc / dur):.0f} MHz")
    hw, xcc = a[:, 4], a[:, 5] & 0xF
    cu = (xcc << 16) | (hw & 0xFF00)  # xcc | se sh cu
    simd = (hw >> 4) & 3
    ncu = len(np.unique(cu))
    cnt = np.bincount(np.unique(cu, return_inverse=True)[1])
    slot = np.unique(cu * 4 + simd, return_counts=True)[1]
    print(f"  placement: {ncu} CUs used, blocks per CU min {cnt.min()} max {cnt.max()} (histogram {np.bincount(cnt).tolist()}); blocks per SIMD max {slot.max()} ({(slot > 1).sum()} SIMDs hold more than one)")
    order = np.argsort(-end)
    print("  last blocks to finish: end us | start us | dur us | contacts(4 envs, all substeps) | max-env contacts | MPR rounds | box-box | blocks on its CU / SIMD")
    inv_cu = np.unique(cu, return_inverse=True)[1]
    inv_sl = np.unique(cu * 4 + simd, return_inverse=True)[1]
    for b in order[:12]:
        print(f"    {end[b]:7.1f} {start[b]:7.1f} {dur[b]:7.1f} {a[b, 30]:5d} {a[b, 29]:5d} {a[b, 14]:4d} {a[b, 28]:4d}   {cnt[inv_cu[b]]} / {slot[inv_sl[b]]}")
    print("  their generic-convex tasks: uncached | new slot | moved | empty manifold asked again | growth query | refresh only")
    for b in order[:12]:
        print(f"    {a[b, 10]:5d} {a[b, 11]:5d} {a[b, 12]:5d} {a[b, 13]:5d} {a[b, 15]:5d} {a[b, 16]:5d}")
    print("  their stage B cycles: slot assignment | slot load | refresh + setup | MPR query | merge | store (last round: + wait)")
    for b in order[:12]:
        print("    " + " ".join(f"{a[b, k]:8d}" for k in (17, 18, 19, 20, 21, 22)))
    print("  their patch-pass cycles: keys + sleeping | anchor search | 4-point selection | records")
    for b in order[:12]:
        print("    " + " ".join(f"{a[b, k]:8d}" for k in (8, 25, 23, 24)))
    print(f"  all blocks: uncached {a[:, 10].sum()} new slot {a[:, 11].sum()} moved {a[:, 12].sum()} empty asked again {a[:, 13].sum()} growth {a[:, 15].sum()} refresh only {a[:, 16].sum()}")
    print(f"  sweeps run per wave and launch (of {5 * 17} possible): mean {a[:, 9].mean():.1f} p50 {np.median(a[:, 9]):.0f} max {a[:, 9].max()}; limit-row visits: mean {a[:, 31].mean():.1f} p50 {np.median(a[:, 31]):.0f} max {a[:, 31].max()}; "
          f"of the 12 last blocks: sweeps {[int(a[b, 9]) for b in order[:12]]} limit visits {[int(a[b, 31]) for b in order[:12]]}")
    c = np.corrcoef(dur, a[:, 29])[0, 1]
    print(f"  correlation duration ~ max-env contacts: {c:.2f}; mean duration by MPR rounds: " + ", ".join(f"{k}: {dur[a[:, 14] == k].mean():.0f} us (n={int((a[:, 14] == k).sum())})" for k in np.unique(a[:, 14])[:8]))
