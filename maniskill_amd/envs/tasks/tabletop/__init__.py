from .pick_cube import PickCubeEnv
from .push_cube import PushCubeEnv
from .peg_insertion_side import PegInsertionSideEnv
