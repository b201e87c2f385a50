from . import tasks  # noqa: F401  registers the task environments
