"""Thin own-code stand-ins for third-party modules the reference imports by name but that are not
installable here: `sapien` (only `Pose` and a few inert records), `gymnasium`
(Env / Wrapper / spaces / vector / registry subset), `transforms3d.euler.euler2quat`.
`install()` registers a stand-in in `sys.modules` ONLY when the real module is absent.
"""
import importlib
import importlib.util
import sys


def _missing(name):
    if name in sys.modules:
        return False
    try:
        return importlib.util.find_spec(name) is None
    except (ImportError, ValueError):
        return True


def install():
    if _missing("gymnasium"):
        from . import gym_min

        gym_min.install_as("gymnasium")
    if _missing("sapien"):
        from . import sapien_min

        sapien_min.install_as("sapien")
    if _missing("transforms3d"):
        from . import transforms3d_min

        transforms3d_min.install_as("transforms3d")
