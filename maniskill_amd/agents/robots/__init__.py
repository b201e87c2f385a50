from .panda import Panda, PandaWristCam
from .fetch import Fetch
