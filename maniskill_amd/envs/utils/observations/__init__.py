"""obs-mode string parsing (counterpart of mani_skill/envs/utils/observations/__init__.py:37).
Only state modes can be produced by this build; visual textures parse but cannot be rendered."""
from dataclasses import dataclass

ALL_VISUAL_TEXTURES = ["rgb", "depth", "segmentation", "position", "normal", "albedo"]


@dataclass
class CameraObsTextures:
    rgb: bool = False
    depth: bool = False
    segmentation: bool = False
    position: bool = False
    normal: bool = False
    albedo: bool = False


@dataclass
class ObservationModeStruct:
    state_dict: bool
    state: bool
    visual: CameraObsTextures

    @property
    def use_state(self):
        return self.state or self.state_dict


def parse_obs_mode_to_struct(obs_mode: str) -> ObservationModeStruct:
    if obs_mode == "rgbd":
        return ObservationModeStruct(False, False, CameraObsTextures(rgb=True, depth=True))
    if obs_mode == "pointcloud":
        return ObservationModeStruct(False, False, CameraObsTextures(rgb=True, position=True, segmentation=True))
    if obs_mode == "sensor_data":
        return ObservationModeStruct(False, False, CameraObsTextures(rgb=True, depth=True, position=True, segmentation=True))
    if obs_mode in ("state", "state_dict", "none"):
        return ObservationModeStruct(obs_mode == "state_dict", obs_mode == "state", CameraObsTextures())
    tex = obs_mode.split("+")
    for t in tex:
        if t not in ALL_VISUAL_TEXTURES and t not in ("state", "state_dict"):
            raise ValueError(f"Invalid texture type '{t}' requested in the obs mode '{obs_mode}'. Each individual texture must be one of {ALL_VISUAL_TEXTURES}")
    return ObservationModeStruct(
        "state_dict" in tex, "state" in tex, CameraObsTextures(**{t: True for t in tex if t in ALL_VISUAL_TEXTURES})
    )
