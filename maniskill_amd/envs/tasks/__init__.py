from .tabletop import PegInsertionSideEnv, PickCubeEnv, PushCubeEnv
from .empty_env import EmptyEnv
