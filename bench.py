"""Headline benchmark: env-steps/sec of the PickCube-v1 vectorised env loop (state obs).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Protocol restated from the reference's own harness (mani_skill/examples/benchmarking/gpu_sim.py:
91-106, profiling.py:90-113): reset(seed=2022...), warm-up, then K env.step() calls with actions
2*U[0,1)-1 drawn on device, wall time bracketed by device syncs; value = K * num_envs_total / s.
One process per GPU, `--envs-per-gpu` envs each (weak scaling). The env path has no exchange step, so there
is no collective on it (barrier + max-over-ranks timing only); `--gather` adds the optional centralised-learner
exchange: the step outputs all-gathered over RCCL, in rollout chunks, inside the timed region.
Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# host threads actually available to this process (the GPU box exposes a CPU share, not all cores)
def _host_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 CPU quota, when the box limits this job to a share of the host
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(n, 16) if n > 64 else n  # a one-GPU box is a 16-core share of a larger host


HOST_CORES = _host_cores()
os.environ.setdefault("OMP_NUM_THREADS", str(HOST_CORES))

import torch  # noqa: E402

ALG_BYTES_PER_ENV_STEP = 1332  # 192 B read + 1140 B written per env-step (f32)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
PMC_SUMMARY = os.path.join(ROOT, "profiles", "round1", "pmc_summary.json")  # separate --pmc passes, see scripts/summarize_pmc.py


def pmc_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of this
    same command (counters cannot be collected from inside the process); None if not collected"""
    try:
        with open(PMC_SUMMARY) as f:
            kernels = json.load(f)["kernels"]
        for name, v in kernels.items():  # template arguments vary (k_solve16<true, 9>): match by prefix
            if name.startswith(kernel_key):
                return v["hbm_bytes_per_launch_raw"]
        return None
    except Exception:
        return None



def timed_steps(env, steps, gather, barrier):
    base = env.unwrapped
    dev = base.device
    N = base.num_envs
    barrier()
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        a = 2 * torch.rand((N, 8), device=dev) - 1
        obs, rew, term, trunc, info = env.step(a)
        if gather is not None:
            gather.add(obs, rew, term | trunc)  # packed record into the rollout chunk; one RCCL all-gather per chunk
    if gather is not None:
        gather.flush()
        gather.result()  # the last collective has to be complete inside the timed region
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    barrier()
    return time.perf_counter() - t0


def cpu_baseline(seconds_budget=20.0):
    """the oracle (CPU restatement, kind "port") timed on this host's cores on a bounded sample of
    the same workload: PickCube-v1 env.step loop behind the same env layer"""
    import gymnasium as gym

    from tests import oracle_backend as ob

    ob.register("f32", "cpu_oracle_f32")
    n = 512
    torch.set_num_threads(HOST_CORES)
    env = gym.make("PickCube-v1", num_envs=n, sim_backend="cpu_oracle_f32")
    env.reset(seed=[2022 + i for i in range(n)])
    for _ in range(2):
        env.step(2 * torch.rand(n, 8) - 1)
    t0 = time.perf_counter()
    k = 0
    while True:
        env.step(2 * torch.rand(n, 8) - 1)
        k += 1
        if time.perf_counter() - t0 > seconds_budget or k >= 200:
            break
    dt = time.perf_counter() - t0
    env.close()
    cores = int(os.environ.get("OMP_NUM_THREADS", HOST_CORES))
    return dict(
        value=round(k * n / dt, 1),
        unit="env-steps/s",
        cores=cores,
        kind="port",
        sample=f"PickCube-v1 env.step loop, {n} envs x {k} control steps (5 substeps), in-repo f32 CPU oracle (OpenMP over envs) behind the same env layer",
    )


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--control-freq", type=int, default=20, help="sim_freq is 100: 20 -> 5 substeps (reference default), 25 -> 4")
    ap.add_argument("--env-id", default="PickCube-v1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gather", action="store_true",
                    help="centralised-learner mode: all-gather obs/reward/done over RCCL inside the timed region (--gpus > 1). Off by "
                         "default: the env path has no exchange step, every rank's learner shard consumes its own envs")
    ap.add_argument("--no-gather", action="store_true", help="(default behaviour; kept for older command lines)")
    ap.add_argument("--gather-every", type=int, default=8, help="control steps per all-gather (rollout chunk) when --gpus > 1")
    args = ap.parse_args()

    import maniskill_amd.envs  # noqa: F401
    import gymnasium as gym
    import torch.distributed as dist

    from maniskill_amd import native

    if not os.path.exists(native.NATIVE_LIB_PATH) and int(os.environ.get("LOCAL_RANK", 0)) == 0:
        import __graft_entry__

        __graft_entry__.build()  # fresh checkout: hipcc is on the GPU box too
    for _ in range(600):  # other ranks wait for rank 0's build
        if os.path.exists(native.NATIVE_LIB_PATH):
            break
        time.sleep(0.5)

    from maniskill_amd.distributed import RolloutGather, shard_seeds, world_info

    rank, local_rank, world = world_info()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus > 1 must be launched with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the simulation core is HIP-only (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    def barrier():
        if world > 1:
            dist.barrier()

    n = args.envs_per_gpu
    substeps = 100 // args.control_freq
    global_seeds = [2022 + i for i in range(n * world)]
    torch.manual_seed(2022 + rank)  # reproducible action stream (A/B comparisons between builds)
    env = gym.make(args.env_id, num_envs=n, sim_backend=f"cuda:{local_rank}", sim_config=dict(control_freq=args.control_freq))
    base = env.unwrapped
    seeds = shard_seeds(global_seeds, rank, world)
    obs, _ = env.reset(seed=seeds)
    gather = RolloutGather(n, obs.shape[1], dev, chunk=args.gather_every) if (world > 1 and args.gather and not args.no_gather) else None
    for _ in range(args.warmup):
        o, r, te, tr, _ = env.step(2 * torch.rand((n, 8), device=dev) - 1)
        if gather is not None:
            gather.add(o, r, te | tr)
    if gather is not None:
        gather.flush()
        gather.result()
    env.reset(seed=seeds)
    px = base.scene.px
    px.profile_enable(True)
    elapsed = timed_steps(env, args.steps, gather, barrier)
    prof = px.profile_read()
    px.profile_enable(False)
    overflow = px.overflow_count()
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        total_envs = n * world
        value = args.steps * total_envs / elapsed
        solve_ms, solve_n = prof["solve"]
        narrow_ms, narrow_n = prof["narrow"]
        avg_solve_s = (solve_ms / max(solve_n, 1)) * 1e-3
        fused = narrow_n == 0 and solve_n == args.steps  # k_solve16<FUSED>: one launch = n envs x one control step
        alg_bytes_per_launch = ALG_BYTES_PER_ENV_STEP * n / (1 if fused else substeps)
        if fused:
            kernel_name, kernel_key = f"k_solve16<FUSED, NDOF, TASK> (whole env.step in one launch: action map, {substeps} substeps incl. narrowphase, copy-out, task epilogue; 16 lanes/env)", "k_solve16<true"
        elif (px.model.n_dof + 6 * px.model.n_free) <= 16 and os.environ.get("MSSIM_SOLVER") != "lane":
            kernel_name, kernel_key = "k_solve16 (one substep, 16 lanes/env; narrowphase in k_narrow)", "k_solve16<false"
        else:
            kernel_name, kernel_key = "k_solve (one env per lane)", "k_solve"
        achieved = alg_bytes_per_launch / avg_solve_s / 1e9 if avg_solve_s > 0 else 0.0
        out = {
            "metric": "env-steps/sec (whole node), PickCube-v1 state-obs 4096 envs/GPU",
            "value": round(value, 1),
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.env_id}, Panda, state obs [N,42], pd_joint_delta_pos, num_envs={n}/GPU, sim 100 Hz / control {args.control_freq} Hz = {substeps} substeps, "
                "15+1 solver iterations, random actions 2*U-1, no resets inside the timed region",
                "envs_per_gpu": n,
                "substeps": substeps,
                "parallelism": f"env-sharded x{world}, no collective on the env path" + (f"; centralised-learner mode: one packed RCCL all-gather of obs/reward/done per {args.gather_every} control steps (asynchronous, overlaps the next chunk)" if gather is not None else ""),
                "solver_overflow_envs": overflow,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": round(achieved, 4),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": pmc_traffic(kernel_key),
                "avg_kernel_ms": round(avg_solve_s * 1e3, 4),
                "launches": solve_n,
                "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                "narrowphase_avg_kernel_ms": None if fused else round(narrow_ms / max(narrow_n, 1), 4),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            env.close()
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as ex:  # the GPU measurement must not be lost if the CPU checker cannot be built / loaded
                out["cpu_baseline"] = dict(value=None, unit="env-steps/s", cores=0, kind="port", sample=f"unavailable: {type(ex).__name__}: {ex}")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
