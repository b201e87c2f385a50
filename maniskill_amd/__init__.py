"""maniskill_amd -- MI355X-native batched rigid-body simulation step behind ManiSkill's env API.

See DESIGN.md. The hot path (articulated dynamics, narrowphase, contact solver) is hand-written
HIP for gfx950 in `maniskill_amd/csrc`, exported through the C ABI of `include/mssim.h`.
"""
import os

__version__ = "0.1.0"

PACKAGE_DIR = os.path.dirname(os.path.abspath(__file__))
PACKAGE_ASSET_DIR = os.path.join(PACKAGE_DIR, "assets")
ASSET_DIR = PACKAGE_ASSET_DIR

from .compat import install as _install_compat

_install_compat()  # stand-ins for gymnasium / sapien / transforms3d only where the real module is absent
