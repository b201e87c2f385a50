from .base_env import SceneManipulationEnv
