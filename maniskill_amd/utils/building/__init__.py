from .actor_builder import ActorBuilder
from .articulation_builder import ArticulationBuilder
from .urdf_loader import URDFLoader
