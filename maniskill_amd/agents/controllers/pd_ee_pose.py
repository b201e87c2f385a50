"""End-effector space controllers: configs are accepted so robot definitions import unchanged, but
constructing one raises -- batched IK on device is the next widening step (SURVEY.md 8f rank 3;
reference: mani_skill/agents/controllers/pd_ee_pose.py, controllers/utils/kinematics.py:124-186)."""
from dataclasses import dataclass
from typing import Sequence, Union

from .base_controller import BaseController, ControllerConfig


class PDEEPosController(BaseController):
    def __init__(self, *a, **kw):
        raise NotImplementedError("pd_ee_* control modes need batched IK, which this build does not implement yet (SURVEY.md 8f)")


class PDEEPoseController(PDEEPosController):
    pass


@dataclass
class PDEEPosControllerConfig(ControllerConfig):
    pos_lower: Union[float, Sequence[float]] = None
    pos_upper: Union[float, Sequence[float]] = None
    stiffness: Union[float, Sequence[float]] = None
    damping: Union[float, Sequence[float]] = None
    force_limit: Union[float, Sequence[float]] = 1e10
    friction: Union[float, Sequence[float]] = 0.0
    ee_link: str = None
    urdf_path: str = None
    frame: str = "root_translation"
    use_delta: bool = True
    use_target: bool = False
    interpolate: bool = False
    normalize_action: bool = True
    drive_mode: str = "force"
    controller_cls = PDEEPosController


@dataclass
class PDEEPoseControllerConfig(ControllerConfig):
    pos_lower: Union[float, Sequence[float]] = None
    pos_upper: Union[float, Sequence[float]] = None
    rot_lower: Union[float, Sequence[float]] = None
    rot_upper: Union[float, Sequence[float]] = None
    stiffness: Union[float, Sequence[float]] = None
    damping: Union[float, Sequence[float]] = None
    force_limit: Union[float, Sequence[float]] = 1e10
    friction: Union[float, Sequence[float]] = 0.0
    ee_link: str = None
    urdf_path: str = None
    frame: str = "root_translation:root_aligned_body_rotation"
    use_delta: bool = True
    use_target: bool = False
    interpolate: bool = False
    normalize_action: bool = True
    drive_mode: str = "force"
    controller_cls = PDEEPoseController
