"""BASELINE.json config (1) on the in-repo CPU restatement: P worker processes x 1 env of PushCube-v1
(panda_wristcam, state obs, pd_joint_delta_pos), random actions, the protocol of the reference's CPU
mode (examples/benchmarking/gpu_sim.py:71-106: one sub-env per process, step-only). The reference's own
CPU path (SAPIEN CPU PhysX) cannot run here, so this is labelled "in-repo CPU restatement", never
"SAPIEN CPU". Prints one JSON line per process count: 4 (the reference's configuration) and one per
host core.   usage: python scripts/cpu_config1.py [steps]"""
import json, multiprocessing as mp, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, steps, barrier, out):
    os.environ["OMP_NUM_THREADS"] = "1"
    sys.path.insert(0, ROOT)
    import torch

    torch.set_num_threads(1)
    import maniskill_amd.envs  # noqa  (first: installs the gymnasium stand-in where the real module is absent)
    import gymnasium as gym
    from tests import oracle_backend as ob

    ob.register("f32", "cpu_oracle_f32")
    env = gym.make("PushCube-v1", num_envs=1, obs_mode="state", control_mode="pd_joint_delta_pos", sim_backend="cpu_oracle_f32")
    adim = env.unwrapped.single_action_space.shape[0]
    torch.manual_seed(2022 + rank)
    env.reset(seed=2022 + rank)
    env.step(2 * torch.rand(1, adim) - 1)
    env.reset(seed=2022 + rank)
    barrier.wait()
    t = time.perf_counter()
    for _ in range(steps):
        env.step(2 * torch.rand(1, adim) - 1)
    out.put((rank, time.perf_counter() - t))
    barrier.wait()
    env.close()


def run(procs, steps):
    ctx = mp.get_context("spawn")
    barrier, out = ctx.Barrier(procs), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, steps, barrier, out)) for r in range(procs)]
    for p in ps:
        p.start()
    times = [out.get(timeout=900)[1] for _ in ps]
    for p in ps:
        p.join()
    wall = max(times)
    print(json.dumps(dict(config="PushCube-v1, 1 env per process, state obs, pd_joint_delta_pos, step-only", processes=procs, steps=steps,
                          env_steps_per_s=round(procs * steps / wall, 1), ms_per_env_step=round(1e3 * wall / steps, 3),
                          kind="in-repo CPU restatement (f32 oracle), one thread per process")), flush=True)


if __name__ == "__main__":
    # CPU-only measurement: the workers must not open the GPU (a GPU box admits few processes on its card)
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        os.environ[k] = ""
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("MSSIM_HOST_CORES", "16")))  # a 1-GPU box shares 16 cores
    counts = [int(a) for a in sys.argv[2:]] or sorted({4, cores})  # explicit process counts: bench.py asks for 4 only
    for procs in counts:
        run(procs, steps)
