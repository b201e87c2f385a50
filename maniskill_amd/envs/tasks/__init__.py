from .tabletop import PickCubeEnv, PushCubeEnv
