from .articulation import ArticulationJoint  # noqa: F401  (module path parity)
