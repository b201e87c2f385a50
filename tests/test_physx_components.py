"""Builder-time component surface of `physx`: a body assembled from `PhysxRigidDynamicComponent /
PhysxRigidStaticComponent / PhysxCollisionShape* / PhysxMaterial` the way the reference's
`ActorBuilder.build_physx_component` assembles it (mani_skill/utils/building/actor_builder.py:57-163) compiles to the
same model, byte for byte, as the same body assembled with the builder's `add_*_collision` calls."""
import numpy as np
import torch

from maniskill_amd import physx
from maniskill_amd.model import geom
from maniskill_amd.model.compile import SceneModelBuilder
from maniskill_amd.model.scenes import TABLE_HEIGHT
from tests import oracle_backend as ob


def _models_equal(a, b):
    assert a.arrays.keys() == b.arrays.keys()
    for k in a.arrays:
        x, y = np.asarray(a.arrays[k]), np.asarray(b.arrays[k])
        assert x.dtype == y.dtype and x.shape == y.shape and x.tobytes() == y.tobytes(), k
    assert a.scalars == b.scalars
    assert (a.link_names, a.free_names, a.kin_names, a.static_names, a.shape_owner) == (b.link_names, b.free_names, b.kin_names, b.static_names, b.shape_owner)


def _pickcube_bodies_by_components(default_material):
    """the PickCube scene content except the robot: table (kinematic box), ground (static plane), cube (dynamic box),
    goal site (kinematic, no shapes) -- utils/scene_builder/table/scene_builder.py:20-58, building/ground.py:36-45,
    envs/tasks/tabletop/pick_cube.py:60-81"""
    out = []
    table = physx.PhysxRigidDynamicComponent()
    table.kinematic = True
    box = physx.PhysxCollisionShapeBox(half_size=[2.418 / 2, 1.209 / 2, TABLE_HEIGHT / 2], material=default_material)
    box.local_pose = geom.pose([0, 0, TABLE_HEIGHT / 2])
    box.set_collision_groups([1, 1, 0, 0])
    box.set_density(1000)
    box.set_patch_radius(0)
    box.set_min_patch_radius(0)
    table.attach(box)
    out.append(("table-workspace", table, geom.pose([-0.12, 0, -TABLE_HEIGHT], geom.rpy_to_quat([0, 0, np.pi / 2]))))

    ground = physx.PhysxRigidStaticComponent()
    plane = physx.PhysxCollisionShapePlane(material=default_material)
    plane.local_pose = geom.pose(q=[0.7071068, 0, -0.7071068, 0])
    ground.attach(plane)
    out.append(("ground", ground, geom.pose([0, 0, -TABLE_HEIGHT])))

    cube = physx.PhysxRigidDynamicComponent()
    cbox = physx.PhysxCollisionShapeBox(half_size=[0.02] * 3, material=default_material)
    cube.attach(cbox)
    out.append(("cube", cube, geom.pose([0, 0, 0.02])))

    goal = physx.PhysxRigidDynamicComponent()
    goal.kinematic = True
    out.append(("goal_site", goal, geom.pose()))
    return out


def test_component_api_compiles_to_the_builders_model():
    import sapien

    import maniskill_amd.envs  # noqa: F401
    from maniskill_amd.envs.scene import ManiSkillScene
    from maniskill_amd.utils.building.ground import build_ground

    ob.register("f64", "oracle_f64_env")
    scene = ManiSkillScene(2, device="cpu", backend_name="oracle_f64_env")
    # the builder way (what the tasks do)
    b = scene.create_actor_builder()
    b.add_box_collision(pose=sapien.Pose(p=[0, 0, TABLE_HEIGHT / 2]), half_size=(2.418 / 2, 1.209 / 2, TABLE_HEIGHT / 2))
    b.initial_pose = sapien.Pose(p=[-0.12, 0, -TABLE_HEIGHT], q=geom.rpy_to_quat([0, 0, np.pi / 2]))
    b.build_kinematic("table-workspace")
    build_ground(scene, altitude=-TABLE_HEIGHT)
    b = scene.create_actor_builder()
    b.add_box_collision(half_size=[0.02] * 3)
    b.initial_pose = sapien.Pose(p=[0, 0, 0.02])
    cube_actor = b.build("cube")
    b = scene.create_actor_builder()
    b.initial_pose = sapien.Pose()
    b.build_kinematic("goal_site")
    via_builder = scene._builder.compile(num_envs=2)

    # the component way
    sb = SceneModelBuilder()
    for name, comp, pose in _pickcube_bodies_by_components(scene.default_material):
        sb.add_actor(comp.to_record(name, pose))
    via_components = sb.compile(num_envs=2)
    _models_equal(via_builder, via_components)

    # ActorBuilder.build_physx_component hands out the component the actor was registered from
    comp = cube_actor._px_component
    assert isinstance(comp, physx.PhysxRigidDynamicComponent) and not comp.kinematic
    assert isinstance(comp.collision_shapes[0], physx.PhysxCollisionShapeBox) and isinstance(comp.collision_shapes[0].physical_material, physx.PhysxMaterial)
    assert comp.collision_shapes[0].get_collision_groups() == [1, 1, 0, 0]

    # after gpu_init: rows, introspection lists, contacts
    scene._setup()
    px = scene.px
    names = [c.name for c in px.rigid_dynamic_components]
    assert names == ["cube", "table-workspace", "goal_site"] and [c.name for c in px.rigid_static_components] == ["ground"]
    assert comp.gpu_pose_index == scene.model.row_of("cube") * 2 and comp.entity is cube_actor
    row = px.cuda_rigid_body_data.torch()[comp.gpu_pose_index]
    assert torch.allclose(row[:3], torch.tensor([0.0, 0.0, 0.02]))
    px.step(3)
    contacts = scene.get_contacts(0)
    assert len(contacts) == 1
    pair = {contacts[0].bodies[0].name, contacts[0].bodies[1].name}
    assert pair == {"cube", "table-workspace"}
    w = 1000 * 0.04**3 * 9.81 * px.timestep  # weight of the cube times dt
    assert abs(abs(float(contacts[0].points[0].impulse[2])) - w) < 0.03 * w


def test_unsupported_shapes_say_so():
    import pytest

    c = physx.PhysxRigidDynamicComponent()
    with pytest.raises(NotImplementedError):
        c.set_locked_motion_axes([True, False, False, False, False, False])
    c.set_locked_motion_axes([False] * 6)
