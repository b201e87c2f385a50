"""ground plane (mani_skill/utils/building/ground.py:18-45); the checker-board visual is dropped"""
import sapien


def build_ground(scene, floor_width: int = 100, floor_length: int = None, xy_origin: tuple = (0, 0), altitude=0, name="ground",
                 texture_file=None, texture_square_len=4, mipmap_levels=4, add_collision=True):
    ground = scene.create_actor_builder()
    if add_collision:
        ground.add_plane_collision(sapien.Pose(p=[0, 0, altitude], q=[0.7071068, 0, -0.7071068, 0]))
    ground.initial_pose = sapien.Pose(p=[0, 0, 0], q=[1, 0, 0, 0])
    return ground.build_static(name=name)
