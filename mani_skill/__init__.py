"""`mani_skill` import-path alias of `maniskill_amd`.

Task / learner code written against the reference API (`from mani_skill.envs.sapien_env import
BaseEnv`, `from mani_skill.utils.structs.pose import Pose`, ...) resolves to this repository's own
implementation. No reference code lives here: every `mani_skill.X` module IS `maniskill_amd.X`.
"""
import importlib
import importlib.abc
import importlib.util
import sys

import maniskill_amd

_PREFIX, _TARGET = "mani_skill", "maniskill_amd"


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        return importlib.import_module(self.target)

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if fullname == _PREFIX or not fullname.startswith(_PREFIX + "."):
            return None
        real = _TARGET + fullname[len(_PREFIX):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except (ImportError, ValueError):
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(real))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

PACKAGE_DIR = maniskill_amd.PACKAGE_DIR
PACKAGE_ASSET_DIR = maniskill_amd.PACKAGE_ASSET_DIR
ASSET_DIR = maniskill_amd.ASSET_DIR
__version__ = maniskill_amd.__version__

import maniskill_amd.envs  # noqa: E402,F401  registers PickCube-v1 / PushCube-v1 with gym
