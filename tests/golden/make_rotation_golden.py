"""Generates tests/golden/rotation_golden.npz by importing the REFERENCE's
mani_skill/utils/geometry/rotation_conversions.py standalone (it only imports torch) and
evaluating it on seeded random inputs. Run in the build container only (the reference checkout
does not exist on the GPU box); the .npz is the committed fixture.

    python tests/golden/make_rotation_golden.py /root/reference
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ref_root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
path = os.path.join(ref_root, "mani_skill/utils/geometry/rotation_conversions.py")
spec = importlib.util.spec_from_file_location("ref_rotation_conversions", path)
rc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rc)

g = torch.Generator().manual_seed(20240607)
B = 64
qa = torch.randn(B, 4, generator=g)
qa = qa / qa.norm(dim=1, keepdim=True)
qb = torch.randn(B, 4, generator=g)
qb = qb / qb.norm(dim=1, keepdim=True)
pts = torch.randn(B, 3, generator=g)
eul = (torch.rand(B, 3, generator=g) * 2 - 1) * 3.0
aa = torch.randn(B, 3, generator=g)
mats = rc.quaternion_to_matrix(qa)
out = dict(
    qa=qa, qb=qb, pts=pts, eul=eul, aa=aa,
    quaternion_multiply=rc.quaternion_multiply(qa, qb),
    quaternion_raw_multiply=rc.quaternion_raw_multiply(qa, qb),
    quaternion_apply=rc.quaternion_apply(qa, pts),
    quaternion_invert=rc.quaternion_invert(qa),
    quaternion_to_matrix=mats,
    matrix_to_quaternion=rc.matrix_to_quaternion(mats),
    euler_XYZ=rc.euler_angles_to_matrix(eul, "XYZ"),
    euler_ZYX=rc.euler_angles_to_matrix(eul, "ZYX"),
    axis_angle_to_quaternion=rc.axis_angle_to_quaternion(aa),
    quaternion_to_axis_angle=rc.quaternion_to_axis_angle(qa),
    standardize_quaternion=rc.standardize_quaternion(qa),
)
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rotation_golden.npz"), **{k: v.numpy() for k, v in out.items()})
print("wrote", len(out), "arrays")
