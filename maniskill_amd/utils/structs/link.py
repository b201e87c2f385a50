from .actor import Link  # noqa: F401  (module path parity with mani_skill/utils/structs/link.py)
