from .registration import REGISTERED_AGENTS, register_agent
from . import robots  # noqa: E402,F401  registers the bundled robots (panda, panda_wristcam)
