"""The C-ABI libraries load (no GPU needed) and export every entry point include/mssim.h declares;
the ctypes struct mirrors the header's field order."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "mssim.h")).read()
DECLARED = sorted(set(re.findall(r"MSSIM_FN\((\w+)\)\(", HEADER)))


def test_header_declares_the_expected_surface():
    for name in ("create", "destroy", "bind_buffers", "apply", "fetch", "step", "update_kinematics", "create_pair_query",
                 "query_pair_impulses", "create_body_query", "query_body_impulses", "set_timestep", "get_timestep",
                 "apply_action", "task_pick_outputs", "task_push_outputs", "task_peg_outputs", "link_jacobian", "last_error", "abi_version"):
        assert name in DECLARED


def test_hip_library_exports_every_declared_symbol():
    from maniskill_amd import native

    if not os.path.exists(native.NATIVE_LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = ctypes.CDLL(native.NATIVE_LIB_PATH)  # loading needs no GPU
    for name in DECLARED:
        assert hasattr(lib, "mssim_" + name), f"libmssim.so does not export mssim_{name}"
    assert sorted(native.NativeLib.EXPORTS) == DECLARED, "native.py binds a different set than the header declares"
    nl = native.NativeLib.load()
    assert nl.abi_version() == native.ABI_VERSION


def test_oracle_library_exports_every_declared_symbol(oracle_lib):
    for name in DECLARED:
        assert hasattr(oracle_lib.lib, "mssim_ref_" + name)


def test_model_desc_struct_matches_header():
    from maniskill_amd import native

    body = HEADER[HEADER.index("typedef struct mssim_model_desc {") : HEADER.index("} mssim_model_desc;")]
    fields = re.findall(r"^\s*(?:const\s+)?(?:int32_t|float)\s*\*?\s*(\w+)(?:\[\d+\])?;", body, flags=re.M)
    assert fields == [f[0] for f in native.ModelDesc._fields_]


def test_create_rejects_bad_models(oracle_lib):
    from maniskill_amd import native
    from maniskill_amd.model.scenes import panda_tabletop_model

    model = panda_tabletop_model(with_robot=False)
    desc, keep = native.make_model_desc(model)
    desc.abi_version = 99
    h = ctypes.c_void_p()
    assert oracle_lib.create(ctypes.byref(desc), 4, -1, ctypes.byref(h)) != 0
    assert b"ABI" in oracle_lib.last_error(None)
