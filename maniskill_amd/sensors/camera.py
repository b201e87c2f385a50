"""CameraConfig as an inert record (mani_skill/sensors/camera.py:33-68): task files construct
camera configs in `_default_sensor_configs`; this build has no renderer so they are never used."""
from dataclasses import dataclass
from typing import Optional

from maniskill_amd.utils.structs.pose import Pose


@dataclass
class CameraConfig:
    uid: str
    pose: Pose
    width: int
    height: int
    fov: float = None
    near: float = 0.01
    far: float = 100
    intrinsic: object = None
    entity_uid: Optional[str] = None
    mount: object = None
    shader_pack: Optional[str] = "minimal"
    shader_config: object = None

    def __post_init__(self):
        self.pose = Pose.create(self.pose)
