from . import tasks  # noqa: F401  registers the task environments
from . import scenes  # noqa: F401  registers SceneManipulation-v1
