"""Passive controller: joints with a damping-only drive and an empty action space."""
from dataclasses import dataclass
from typing import Sequence, Union

import numpy as np
from gymnasium import spaces

from .base_controller import BaseController, ControllerConfig


class PassiveController(BaseController):
    config: "PassiveControllerConfig"

    def set_drive_property(self):
        n = len(self.joints)
        d = np.broadcast_to(self.config.damping, n)
        f = np.broadcast_to(self.config.force_limit, n)
        for i, joint in enumerate(self.joints):
            joint.set_drive_properties(0, d[i], force_limit=f[i])

    def _initialize_action_space(self):
        self.single_action_space = spaces.Box(np.empty(0, np.float32), np.empty(0, np.float32), dtype=np.float32)

    def set_action(self, action):
        pass


@dataclass
class PassiveControllerConfig(ControllerConfig):
    damping: Union[float, Sequence[float]]
    force_limit: Union[float, Sequence[float]] = 1e10
    friction: Union[float, Sequence[float]] = 0.0
    controller_cls = PassiveController
