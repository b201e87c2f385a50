from .base_controller import BaseController, CombinedController, ControllerConfig, DictController
from .passive_controller import PassiveController, PassiveControllerConfig
from .pd_base_vel import PDBaseForwardVelController, PDBaseForwardVelControllerConfig, PDBaseVelController, PDBaseVelControllerConfig
from .pd_ee_pose import PDEEPosController, PDEEPosControllerConfig, PDEEPoseController, PDEEPoseControllerConfig
from .pd_joint_pos import (
    PDJointPosController,
    PDJointPosControllerConfig,
    PDJointPosMimicController,
    PDJointPosMimicControllerConfig,
)
from .pd_joint_pos_vel import PDJointPosVelController, PDJointPosVelControllerConfig
from .pd_joint_vel import PDJointVelController, PDJointVelControllerConfig


def deepcopy_dict(configs: dict):
    """deep copy of a (nested) dict of controller configs (mani_skill/agents/controllers/__init__.py)"""
    import copy

    out = {}
    for k, v in configs.items():
        out[k] = deepcopy_dict(v) if isinstance(v, dict) else copy.deepcopy(v)
    return out
