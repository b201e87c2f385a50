from .panda import Panda, PandaWristCam
