"""Test-only glue: builds / loads the CPU oracle (oracle/) and registers it as a sim backend.

The oracle is the CHECKER. Only tests, `__graft_entry__.smoke()` and bench.py's `cpu_baseline`
leg may use this module; the package `maniskill_amd` never imports it.
"""
import os
import subprocess

import torch

from maniskill_amd import native
from maniskill_amd.physx import system as px_system

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def oracle_path(precision="f64"):
    return os.path.join(ORACLE_DIR, "_build", f"libmssim_ref_{precision}.so")


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def load_oracle(precision="f64") -> native.NativeLib:
    if not os.path.exists(oracle_path(precision)):
        build_oracle()
    return native.NativeLib.load(oracle_path(precision), prefix="mssim_ref_")


def register(precision="f64", name="physx_cpu"):
    """make `MssimSystem(device='cpu', backend=name)` run on the oracle"""
    lib = load_oracle(precision)
    px_system.register_backend(name, lambda device: (lib, -1))
    return name


def make_system(model, num_envs, precision="f64", timestep=None):
    name = register(precision, f"oracle_{precision}")
    px = px_system.MssimSystem(device="cpu", backend=name)
    if timestep is not None:
        px.timestep = timestep
    px.gpu_init(model, num_envs)
    return px
