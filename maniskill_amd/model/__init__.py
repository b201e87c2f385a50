from .compile import (
    ActorRecord,
    ArticulationRecord,
    CompiledModel,
    SceneModelBuilder,
    ShapeRecord,
    shapes_from_urdf_link,
)
from .urdf import RobotDescription, parse_urdf
