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
    return n


HOST_CORES = _host_cores()
os.environ.setdefault("OMP_NUM_THREADS", str(HOST_CORES))

import torch  # noqa: E402

ALG_BYTES_PER_ENV_STEP = 1332  # 192 B read + 1140 B written per env-step (f32)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
# MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, a wave64 VALU instruction issues over 2 cycles of a SIMD-32 at 2.4 GHz ...
VALU_PEAK_WAVE_INSTS_PER_S = 1024 * 2.4e9 / 2
# ... but ONE wave alone on its SIMD sustains one VALU instruction per 4 cycles (the guide's per-instruction constants): the
# control-step kernel takes the CU's whole register file and LDS, so it never has a second wave -- this is its attainable peak
VALU_PEAK_ONE_WAVE_PER_SIMD = 1024 * 2.4e9 / 4
PROFILE_DIR = os.path.join(ROOT, "profiles", "round3")  # separate --pmc passes of this same command, scripts/collect_profiles.py


def kernel_source_sha():
    """identity of the kernels a counter summary belongs to: sha256 over the HIP sources + the ABI header (the same on
    this box and wherever the counters were collected; a rebuilt .so is not byte-stable)"""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "maniskill_amd", "csrc")
    for p in sorted(os.path.join(csrc, f) for f in os.listdir(csrc)) + [os.path.join(ROOT, "include", "mssim.h")]:
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def committed_counters(kernel_key, protocol):
    """(HBM bytes corrected, HBM bytes raw, wave-VALU instructions) per launch of the dominant kernel from the committed
    rocprofv3 --pmc passes (counters cannot be collected from inside the process). None unless the summaries were
    collected on exactly the kernel sources that are running now AND under the protocol of this run (`--steps / --warmup /
    envs / control frequency`: a launch over fresh episodes is not a launch after 1000 unreset steps) -- a stale or a mixed
    number is not reported. Corrected = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM: gfx950 tallies a 128-byte
    read request as 64 bytes)."""
    traffic = raw = valu = None
    try:
        with open(os.path.join(PROFILE_DIR, "pmc_summary.json")) as f:
            d = json.load(f)
        if d.get("kernel_source_sha") == kernel_source_sha():
            for name, v in d["protocols"][protocol]["kernels"].items():  # template arguments vary (k_solve16<9, 1>): match by prefix
                if name.startswith(kernel_key):
                    traffic, raw = v["hbm_bytes_per_launch"], v["hbm_bytes_per_launch_raw"]
    except Exception:
        pass
    try:
        with open(os.path.join(PROFILE_DIR, "sq_counters.json")) as f:
            d = json.load(f)
        if d.get("kernel_source_sha") == kernel_source_sha():
            valu = d["protocols"][protocol]["counters"]["SQ_INSTS_VALU"]["per_launch"]
    except Exception:
        pass
    return traffic, raw, valu


def protocol_of(args):
    """what a counter pass has to share with a timed run for their per-launch figures to be divided by each other"""
    return f"{args.env_id} envs={args.envs_per_gpu} control_freq={args.control_freq} steps={args.steps} warmup={args.warmup}"



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
    out = dict(
        value=round(k * n / dt, 1),
        unit="env-steps/s",
        cores=cores,
        kind="port",
        sample=f"PickCube-v1 env.step loop, {n} envs x {k} control steps (5 substeps), in-repo f32 CPU oracle (OpenMP over envs) behind the same env layer",
    )
    out["config1"] = cpu_config1()
    return out


def cpu_config1(steps=2500):
    """BASELINE.json configs[0] -- the reference's own CPU-runnable case (gpu_sim.py:71-85): 4 worker processes x 1 env
    of PushCube-v1, state obs, step-only -- on the in-repo CPU restatement (the reference's SAPIEN CPU PhysX is absent).
    Child processes of scripts/cpu_config1.py; they never open the GPU."""
    import subprocess

    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "cpu_config1.py"), str(steps), "4"], capture_output=True, text=True, timeout=240)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return dict(value=d["env_steps_per_s"], unit="env-steps/s", cores=d["processes"], kind="port",
                    sample=f"PushCube-v1 (panda_wristcam), 4 processes x 1 env x {steps} control steps (5 substeps), in-repo f32 CPU oracle, one thread per process")
    except Exception as ex:
        return dict(value=None, unit="env-steps/s", cores=0, kind="port", sample=f"unavailable: {type(ex).__name__}: {ex}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed control steps (the reference harness times 1000, gpu_sim.py:96-106)")
    ap.add_argument("--warmup", type=int, default=5, help="untimed control steps before the timed region (the reference: reset, 1 step, reset, gpu_sim.py:91-93)")
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

    # rank 0 (re)builds a missing or stale library (hipcc is on the GPU box too; the build moves the finished file into
    # place, so nobody dlopens a partial one), the other ranks wait for its marker before they load it
    marker = os.path.join(os.path.dirname(native.NATIVE_LIB_PATH), f".built.{os.environ.get('MASTER_PORT', '0')}.{os.getppid()}")
    if int(os.environ.get("LOCAL_RANK", 0)) == 0:
        import atexit

        import __graft_entry__

        if os.path.exists(marker):  # (a marker left by an earlier launch with the same port and parent: not ours)
            os.remove(marker)
        __graft_entry__.build()
        if int(os.environ.get("WORLD_SIZE", 1)) > 1:
            with open(marker + ".tmp", "w") as fh:
                fh.write(kernel_source_sha())  # what the library was built from
            os.replace(marker + ".tmp", marker)
            atexit.register(lambda: os.path.exists(marker) and os.remove(marker))
    else:
        t_wait = time.time()
        while True:
            if os.path.exists(marker) and open(marker).read() == kernel_source_sha() and os.path.getmtime(marker) >= t_wait - 1800:
                break
            if time.time() - t_wait > 900:
                raise SystemExit(f"rank {os.environ.get('RANK')}: rank 0 did not finish building {native.NATIVE_LIB_PATH} within 15 minutes")
            time.sleep(0.5)

    from maniskill_amd.distributed import RolloutGather, set_env_index_offset, shard_seeds, world_info

    rank, local_rank, world = world_info()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus > 1 must be launched with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the simulation core is HIP-only (no CPU fallback)")
    # MSSIM_BENCH_REHEARSE=1: a dry run of the multi-rank path on a box with ONE GPU (every rank on cuda:0, gloo for the barrier and
    # the max over ranks, no gather): launch, build handshake, seed sharding, timing protocol and the JSON line are the real
    # ones, the number is not a measurement of anything (the ranks share a GPU) and the line says so
    rehearse = os.environ.get("MSSIM_BENCH_REHEARSE") == "1" and world > 1
    dev_index = 0 if rehearse else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
            args.no_gather = True
        else:
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm

    def barrier():
        if world > 1:
            dist.barrier()

    n = args.envs_per_gpu
    substeps = 100 // args.control_freq
    global_seeds = [2022 + i for i in range(n * world)]
    torch.manual_seed(2022 + rank)  # reproducible action stream (A/B comparisons between builds)
    set_env_index_offset(rank * n)  # this rank holds envs [rank * n, (rank + 1) * n) of the global run
    env = gym.make(args.env_id, num_envs=n, sim_backend=f"cuda:{dev_index}", sim_config=dict(control_freq=args.control_freq))
    base = env.unwrapped
    seeds = shard_seeds(global_seeds, rank, world)
    obs, _ = env.reset(seed=seeds)
    # N > 1: two timed regions of K steps each, back to back -- (1) the env path as it is (no collective: `value`),
    # (2) the centralised-learner exchange of BASELINE config 4 on top (RCCL all-gather of obs / reward / done over xGMI
    # in rollout chunks: `gathered`). `--gather` makes (2) the headline instead; `--no-gather` skips (2).
    gather = RolloutGather(n, obs.shape[1], dev, chunk=args.gather_every) if (world > 1 and not args.no_gather) else None

    def warmup(g):
        for _ in range(args.warmup):
            o, r, te, tr, _ = env.step(2 * torch.rand((n, 8), device=dev) - 1)
            if g is not None:
                g.add(o, r, te | tr)
        if g is not None:
            g.flush()
            g.result()
        env.reset(seed=seeds)

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    px = base.scene.px
    warmup(None)
    px.profile_enable(True)
    elapsed = max_over_ranks(timed_steps(env, args.steps, None, barrier))
    prof = px.profile_read()
    px.profile_enable(False)
    overflow = px.overflow_count()
    elapsed_gathered = None
    if gather is not None:
        warmup(gather)
        elapsed_gathered = max_over_ranks(timed_steps(env, args.steps, gather, barrier))
        if args.gather:
            elapsed, elapsed_gathered = elapsed_gathered, elapsed

    if rank == 0:
        total_envs = n * world
        value = args.steps * total_envs / elapsed
        solve_ms, solve_n = prof["solve"]
        narrow_ms, narrow_n = prof["narrow"]
        avg_solve_s = (solve_ms / max(solve_n, 1)) * 1e-3
        assert solve_n == args.steps and narrow_n == 0, (solve_n, narrow_n)  # one launch = n envs x one control step
        alg_bytes_per_launch = ALG_BYTES_PER_ENV_STEP * n
        kernel_name, kernel_key = f"k_solve16<NDOF, TASK> (whole env.step in one launch: action map, {substeps} substeps incl. narrowphase + contact patches, copy-out, task epilogue; 16 lanes/env)", "k_solve16<"
        achieved = alg_bytes_per_launch / avg_solve_s / 1e9 if avg_solve_s > 0 else 0.0
        traffic, traffic_raw, valu_insts = committed_counters(kernel_key, protocol_of(args))
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
                "parallelism": f"env-sharded x{world}" + (", headline = centralised-learner mode (all-gather inside the timed region)" if (args.gather and gather is not None) else ", no collective on the env path"),
                "solver_overflow_envs": overflow,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": round(achieved, 4),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_raw": traffic_raw,
                "traffic_source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command in profiles/round3/pmc_summary.json: 2 x FETCH_SIZE + WRITE_SIZE "
                                  f"(gfx950 correction of MI355X_MICROARCH.md; raw sum beside it), reported only for kernel sources {kernel_source_sha()} and protocol '{protocol_of(args)}'",
                "avg_kernel_ms": round(avg_solve_s * 1e3, 4),
                "launches": solve_n,
                "algorithmic_bytes_per_launch": alg_bytes_per_launch,
            },
            # the bound that applies (neither HBM nor MFMA): vector-instruction issue
            "roofline_valu": {
                "bound": "valu-issue",
                "achieved": None if valu_insts is None or avg_solve_s <= 0 else round(valu_insts / avg_solve_s / 1e12, 4),
                "peak": VALU_PEAK_WAVE_INSTS_PER_S / 1e12,
                "unit": "T wave-instructions/s",
                "frac": None if valu_insts is None or avg_solve_s <= 0 else valu_insts / avg_solve_s / VALU_PEAK_WAVE_INSTS_PER_S,
                "peak_one_wave_per_simd": VALU_PEAK_ONE_WAVE_PER_SIMD / 1e12,
                "frac_of_one_wave_peak": None if valu_insts is None or avg_solve_s <= 0 else valu_insts / avg_solve_s / VALU_PEAK_ONE_WAVE_PER_SIMD,
                "wave_valu_instructions_per_launch": valu_insts,
                "source": "SQ_INSTS_VALU from profiles/round3/sq_counters.json (same source-hash and protocol rule) / HIP-event kernel time of this run; `peak`: one wave64 "
                          "instruction per 2 cycles and SIMD (chip), `peak_one_wave_per_simd`: per 4 cycles -- what a kernel at one wave per SIMD, as this one, can reach",
            },
        }
        if elapsed_gathered is not None:
            other = "no collective" if args.gather else f"one packed RCCL all-gather of obs / reward / done per {args.gather_every} control steps (asynchronous, overlaps the next chunk), inside the timed region"
            out["gathered" if not args.gather else "ungathered"] = {
                "value": round(args.steps * total_envs / elapsed_gathered, 1),
                "unit": "env-steps/s",
                "ms_per_step": round(elapsed_gathered / args.steps * 1e3, 4),
                "what": f"second timed region of {args.steps} steps in the same run: {other}",
            }
        if rehearse:
            out["rehearsal"] = f"{world} ranks sharing one GPU (MSSIM_BENCH_REHEARSE=1): the multi-rank code path, not a measurement"
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
