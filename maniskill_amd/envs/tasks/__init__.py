from .tabletop import PegInsertionSideEnv, PickCubeEnv, PushCubeEnv
