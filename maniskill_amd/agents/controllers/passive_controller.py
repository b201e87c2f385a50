"""Passive joints: a damping-only drive, no action (behavioural counterpart of
mani_skill/agents/controllers/passive_controller.py). Built on joint_drive.apply_joint_gains like the PD controllers."""
from dataclasses import dataclass
from typing import Sequence, Union

import numpy as np
from gymnasium import spaces

from .base_controller import BaseController, ControllerConfig
from .joint_drive import apply_joint_gains

_NO_ACTION = np.zeros(0, dtype=np.float32)


class PassiveController(BaseController):
    config: "PassiveControllerConfig"

    def _initialize_action_space(self):
        # zero-dimensional box: the controller takes part in combined action spaces without consuming a column
        self.single_action_space = spaces.Box(_NO_ACTION, _NO_ACTION, dtype=np.float32)

    def set_drive_property(self):
        apply_joint_gains(self.joints, stiffness=0.0, damping=self.config.damping, force_limit=self.config.force_limit,
                          friction=self.config.friction, drive_mode="force")

    def set_action(self, action):
        """nothing to set"""


@dataclass
class PassiveControllerConfig(ControllerConfig):
    damping: Union[float, Sequence[float]]
    force_limit: Union[float, Sequence[float]] = 1e10
    friction: Union[float, Sequence[float]] = 0.0
    controller_cls = PassiveController
