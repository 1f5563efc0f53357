"""Host-side drop-in surface (upper side): env classes, body / kilobot / light views,
spaces.  Runs on CPU with the oracle-backed stand-in simulator from tests/oracle_backend.py; the GPU
twin of these checks is tests/test_env_api_gpu.py."""
import numpy as np
import pytest
import torch

from gym_kilobots_amd.envs import (KilobotsEnv, DirectControlKilobotsEnv, BatchedKilobotsEnv, UnknownLightTypeException,
                                    UnknownObjectException)
from gym_kilobots_amd.lib import (SimpleVelocityControlKilobot, SimpleAccelerationControlKilobot, PhototaxisKilobot,
                                  SimplePhototaxisKilobot, CircularGradientLight, Body, Circle, Light)
from gym_kilobots_amd import dist as kdist
from tests.oracle_backend import OracleBackend

O_DRIVE_SP = 3   # KB_DRIVE_SIMPLE_PHOTOTAXIS


class VelEnv(DirectControlKilobotsEnv):
    def _configure_environment(self):
        for i in range(6):
            self._add_kilobot(SimpleVelocityControlKilobot(self.world, position=(-0.25 + 0.1 * i, 0.05 * i),
                                                           orientation=0.3 * i, velocity=[0.005, 0.1]))

    def get_reward(self, state, action, new_state):
        return float(np.linalg.norm(new_state['kilobots'][:, :2] - state['kilobots'][:, :2]))


class PhotoEnv(KilobotsEnv):
    def _configure_environment(self):
        self._light = CircularGradientLight(position=np.array([0.1, 0.0]), radius=0.3,
                                            bounds=(np.array([-1.1, -0.825]), np.array([1.1, 0.825])))
        for i in range(5):
            self._add_kilobot(PhototaxisKilobot(self.world, position=(0.04 * i, 0.0), light=self._light))

    def get_reward(self, state, action, new_state):
        return 1.


def test_constructor_leaves_env_empty_until_reset_like_the_reference():
    env = VelEnv(sim_factory=OracleBackend)
    assert env.num_kilobots == 0 and env.kilobots == ()          # kilobots_env.py:67-68
    assert env.sim_step == 0.1 and env.world_bounds[1].tolist() == [1.0, 0.75]
    assert env.world_x_range == (-1.0, 1.0) and env.seed(5) == [5]
    obs = env.reset()
    assert env.num_kilobots == 6
    assert set(obs.keys()) == {'kilobots', 'objects', 'light'} and obs['kilobots'].shape == (6, 3)
    with pytest.raises(NotImplementedError):
        env.render()
    assert env.observation_space is NotImplemented


def test_direct_control_step_matches_the_kat_and_old_gym_signature():
    env = VelEnv(sim_factory=OracleBackend)
    env.reset()
    assert env.action_space.shape == (6, 2)
    np.testing.assert_allclose(env.action_space.high[0], [0.01, np.pi / 2])
    a = np.tile([0.008, 0.3], (6, 1))
    p0 = env.get_state()['kilobots']
    obs, reward, done, info = env.step(a)
    assert done is False and info == "" and reward > 0
    # contact-free closed form: theta advances by 10 * 0.1 * 0.3 / 1.08
    np.testing.assert_allclose(obs['kilobots'][:, 2] - p0[:, 2], 10 * 0.1 * 0.3 / 1.08, rtol=1e-5)
    # actions are clamped like kilobot.py:235-241; None -> zeros
    env.step(np.tile([0.5, -9.0], (6, 1)))
    np.testing.assert_allclose(env.kilobots[0].get_action(), [0.01, -np.pi / 2], rtol=1e-6)
    p1 = env.get_state()['kilobots']
    env.step(None)
    np.testing.assert_allclose(env.get_state()['kilobots'], p1, atol=1e-9)


def test_body_and_kilobot_views():
    env = VelEnv(sim_factory=OracleBackend)
    env.reset()
    kb = env.kilobots[2]
    x, y, th = kb.get_pose()
    np.testing.assert_allclose([x, y, th], [-0.05, 0.1, 0.6], atol=1e-6)
    assert kb.get_state() == kb.get_pose() and kb.get_radius() == 0.0165 and kb.width == 0.033
    np.testing.assert_allclose(kb.get_world_point((0.0, -0.0165)), kb.light_sensor_pos())
    np.testing.assert_allclose(kb.get_local_point(kb.get_world_point((0.01, 0.02))), (0.01, 0.02), atol=1e-7)
    assert abs(kb.get_local_orientation(1.0) - (1.0 - th)) < 1e-7
    kb.set_pose((0.3, -0.2, 1.0))
    np.testing.assert_allclose(kb.get_pose(), (0.3, -0.2, 1.0), atol=1e-6)
    np.testing.assert_allclose(env.get_state()['kilobots'][2], (0.3, -0.2, 1.0), atol=1e-6)
    a, b = env.kilobots[0], env.kilobots[1]
    assert a.collides_with(b) is None
    b.set_position(np.array(a.get_position()) + (0.03, 0.0))
    assert a.collides_with(b) is True
    with pytest.raises(NotImplementedError):
        Body(env.world)
    assert isinstance(kb, Circle)


def test_light_env_phototaxis_and_light_actions():
    env = PhotoEnv(sim_factory=OracleBackend)
    env.reset()
    assert env.action_space.shape == (2,) and env.action_space.high.tolist() == [0.01, 0.01]
    obs, r, d, i = env.step(np.array([0.01, -0.05]))
    # light moved by clamp(action, +-0.01) * 0.1 per substep, 10 substeps (light.py:59-75)
    np.testing.assert_allclose(obs['light'], [0.1 + 0.01, 0.0 - 0.01], atol=1e-6)
    np.testing.assert_allclose(env.get_light().get_state(), obs['light'])
    assert r == 1. and i == ""
    l0 = obs['light'].copy()
    obs, *_ = env.step(None)                          # action None: light not stepped
    np.testing.assert_allclose(obs['light'], l0)
    assert env.kilobots[0].get_motors() in ((255, 0), (0, 255))
    assert env.kilobots[0].get_ambientlight() > 0
    v, g = env.get_light().value_and_gradients(np.array([[0.0, 0.0], [2.0, 2.0]]))
    assert v[1] == 0 and np.allclose(g[1], 0)


def test_acceleration_env_state_has_five_columns():
    class AccEnv(DirectControlKilobotsEnv):
        def _configure_environment(self):
            for i in range(3):
                self._add_kilobot(SimpleAccelerationControlKilobot(self.world, position=(0.1 * i, 0.0), velocity=[0.0, 0.0]))

        def get_reward(self, *a):
            return 0.

    env = AccEnv(sim_factory=OracleBackend)
    env.reset()
    obs, *_ = env.step(np.tile([0.005, 0.1], (3, 1)))
    assert obs['kilobots'].shape == (3, 5)                      # kilobot.py:279-281
    np.testing.assert_allclose(obs['kilobots'][:, 3], 0.005, rtol=1e-5)
    assert len(env.kilobots[0].get_state()) == 5


def test_env_with_pushable_circles():
    class PushEnv(DirectControlKilobotsEnv):
        def _configure_environment(self):
            self._add_object(Circle(world=self.world, radius=0.075, position=(0.2, 0.0)))
            self._add_object(Circle(world=self.world, radius=0.05, position=(-0.3, 0.2), orientation=0.5))
            for i in range(5):
                self._add_kilobot(SimpleVelocityControlKilobot(self.world, position=(0.02, 0.035 * (i - 2)), velocity=[0.0, 0.0]))

        def get_reward(self, state, action, new_state):
            return float(new_state['objects'][0, 0] - state['objects'][0, 0])

    env = PushEnv(sim_factory=OracleBackend)
    obs = env.reset()
    assert obs['objects'].shape == (2, 3) and len(env.objects) == 2
    np.testing.assert_allclose(obs['objects'][1], (-0.3, 0.2, 0.5), atol=1e-6)
    total = 0.0
    for _ in range(15):
        obs, r, done, info = env.step(np.tile([0.01, 0.0], (5, 1)))
        total += r
    assert total > 0.01 and obs['objects'][0, 0] > 0.21          # the kilobots pushed the disc along +x
    np.testing.assert_allclose(env.objects[0].get_pose(), obs['objects'][0], atol=1e-6)
    assert env.objects[0].get_radius() == 0.075 and env.objects[0].width == 0.15
    assert env.kilobots[2].collides_with(env.objects[0]) in (True, None)


def test_env_with_a_pushable_box_and_triangle():
    from gym_kilobots_amd.lib import CornerQuad, Quad, Triangle, LForm

    class BoxEnv(DirectControlKilobotsEnv):
        def _configure_environment(self):
            # the reference's QuadPushingEnv object (kilobots_test_envs.py:11-21): a 0.15 m box
            self._add_object(CornerQuad(world=self.world, width=.15, height=.15, position=(.1, .02)))
            self._add_object(Triangle(world=self.world, width=.15, height=.15, position=(-.4, .3), orientation=0.3))
            for i in range(6):
                self._add_kilobot(SimpleVelocityControlKilobot(self.world, position=(-0.02, 0.035 * (i - 1)), velocity=[0.0, 0.0]))

        def get_reward(self, state, action, new_state):
            return float(new_state['objects'][0, 0] - state['objects'][0, 0])

    env = BoxEnv(sim_factory=OracleBackend)
    obs = env.reset()
    assert obs['objects'].shape == (2, 3)
    box, tri = env.objects
    assert isinstance(box, Quad) and box.width == .15 and box.get_height() == .15
    np.testing.assert_allclose(box.vertices[0][2], np.array([.1, .02]) + .075, atol=1e-6)      # SetAsBox order
    # Polygon bodies are recentred on their centroid (body.py:226-241): vertex mean of a triangle is the origin
    np.testing.assert_allclose(tri.local_vertices[0].mean(0), 0.0, atol=1e-12)
    (spec,) = tri._shape_spec()                       # one fixture
    assert spec[0] == 2 and len(spec[2]) == 3
    hull = np.array(spec[2])
    assert np.allclose(hull[0], hull[hull[:, 0].argmax()])                  # starts at the right-most (lowest) vertex
    e1, e2 = hull[1] - hull[0], hull[2] - hull[1]
    assert e1[0] * e2[1] - e1[1] * e2[0] > 0                                # counter-clockwise
    total = 0.0
    for _ in range(25):
        obs, r, done, info = env.step(np.tile([0.01, 0.0], (6, 1)))
        total += r
    assert total > 0.01 and obs['objects'][0, 0] > 0.11                     # pushed along +x ...
    assert abs(obs['objects'][0, 2]) > 1e-4                                 # ... and turned (off-centre crowd)
    np.testing.assert_allclose(box.get_pose(), obs['objects'][0], atol=1e-6)
    # Body.collides_with (body.py:87-90) for every shape: the pushing kilobots touch the box, nothing touches the triangle
    assert any(kb.collides_with(box) for kb in env.kilobots) and any(box.collides_with(kb) for kb in env.kilobots)
    assert all(kb.collides_with(tri) is None for kb in env.kilobots) and box.collides_with(tri) is None
    tri.set_pose((obs['objects'][0][0] + 0.12, obs['objects'][0][1], 0.0))
    assert box.collides_with(tri) is True and tri.collides_with(box) is True

    # multi-fixture bodies (the reference's TriangleTestEnv objects, kilobots_test_envs.py:98-103): 1 + 2 + 2 + 3 fixtures
    from gym_kilobots_amd.lib import TForm, CForm

    class ShapesEnv(BoxEnv):
        def _configure_environment(self):
            self._add_object(Triangle(world=self.world, width=.15, height=.15, position=(.0, .0)))
            self._add_object(LForm(world=self.world, width=.15, height=.15, position=(.0, .3)))
            self._add_object(TForm(world=self.world, width=.15, height=.15, position=(.0, -.3)))
            self._add_object(CForm(world=self.world, width=.15, height=.15, position=(.3, .0)))
            for i in range(6):
                self._add_kilobot(SimpleVelocityControlKilobot(self.world, position=(0.2, 0.035 * (i - 2.5)), velocity=[0.0, 0.0]))

        def get_reward(self, state, action, new_state):
            return float(new_state['objects'][3, 0] - state['objects'][3, 0])

    env = ShapesEnv(sim_factory=OracleBackend)
    obs = env.reset()
    assert obs['objects'].shape == (4, 3) and env.sim.cfg.num_fixtures == 8
    assert list(env.sim.cfg.obj_fixture_body) == [0, 1, 1, 2, 2, 3, 3, 3]
    np.testing.assert_allclose(obs['objects'][1], (.0, .3, .0), atol=1e-6)        # the state is the body origin
    total = 0.0
    for _ in range(30):
        obs, r, done, info = env.step(np.tile([0.01, 0.0], (6, 1)))
        total += r
    assert total > 0.005 and int(env.sim.status.max()) == 0                          # the C was pushed along +x
    assert env.objects[3].vertices.shape == (3, 4, 2)

    class TooManyEnv(BoxEnv):
        def _configure_environment(self):
            for i in range(3):
                self._add_object(CForm(world=self.world, width=.15, height=.15, position=(.3 * i - .3, .0)))
            self._add_kilobot(SimpleVelocityControlKilobot(self.world, position=(0, 0.4), velocity=[0.0, 0.0]))

    with pytest.raises(UnknownObjectException):
        TooManyEnv(sim_factory=OracleBackend).reset()


def test_unsupported_scenes_fail_loudly():
    class MixedEnv(KilobotsEnv):
        def _configure_environment(self):
            self._light = CircularGradientLight()
            self._add_kilobot(PhototaxisKilobot(self.world, position=(0, 0)))
            self._add_kilobot(SimplePhototaxisKilobot(self.world, position=(0.1, 0)))

        def get_reward(self, *a):
            return 0.

    # (kilobots of different classes in one env run since round 2: KB_DRIVE_MIXED, tests/test_mixed_laws.py)
    env = MixedEnv(sim_factory=OracleBackend)
    env.reset()
    assert env.sim.drive_mode == 5 and env.sim.bot_mode.tolist() == [[4, 3]]
    env.step(np.array([0.005, 0.0]))

    class OddLightEnv(KilobotsEnv):
        def _configure_environment(self):
            self._light = Light()
            self._add_kilobot(PhototaxisKilobot(self.world, position=(0, 0)))

        def get_reward(self, *a):
            return 0.

    with pytest.raises(UnknownLightTypeException):
        OddLightEnv(sim_factory=OracleBackend).reset()


def test_batched_env_and_replicated_scene():
    env = BatchedKilobotsEnv(5, 16, sim_factory=OracleBackend, seed=3,
                             reward_fn=lambda prev, a, obs: -(obs[..., :2] ** 2).sum(-1).mean(-1))
    obs = env.reset()
    assert obs.shape == (5, 16, 3)
    a = torch.zeros(5, 16, 2)
    a[..., 0] = 0.01
    obs2, r, done, info = env.step(a)
    assert r.shape == (5,) and not done.any() and obs2.shape == (5, 16, 3)
    assert torch.allclose(env.episode_returns, r)
    assert env.gather_episode_returns().shape == (5,)
    multi = VelEnv(num_envs=3, sim_factory=OracleBackend)
    o = multi.reset()
    assert o['kilobots'].shape == (3, 6, 3) and np.array_equal(o['kilobots'][0], o['kilobots'][2])


def test_env_shard_partition():
    for total, world in ((32768, 8), (10, 4), (7, 7), (5, 8)):
        parts = [kdist.env_shard(total, r, world) for r in range(world)]
        assert parts[0][0] == 0 and parts[-1][1] == total
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
        sizes = [hi - lo for lo, hi in parts]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        kdist.env_shard(8, 8, 8)


def test_other_lights_through_the_env_api():
    from gym_kilobots_amd.lib import GradientLight, MomentumLight, CompositeLight, SimplePhototaxisKilobot as SPK
    bounds = (np.array([-1.1, -0.825]), np.array([1.1, 0.825]))

    def make(light_factory):
        class E(KilobotsEnv):
            def _configure_environment(self):
                self._light = light_factory()
                for i in range(4):
                    self._add_kilobot(SPK(self.world, position=(0.05 * i, 0.1), light=self._light))

            def get_reward(self, *a):
                return 0.
        env = E(sim_factory=OracleBackend)
        env.reset()
        return env

    env = make(lambda: GradientLight(angle=0.0))
    assert env.action_space.shape == (1,)
    p0 = env.get_state()['kilobots'][:, :2]
    obs, *_ = env.step(np.array([np.pi / 2]))            # gradient now points along +y: kilobots climb it at 1 cm/s
    np.testing.assert_allclose(obs['light'], [np.pi / 2], atol=1e-6)
    np.testing.assert_allclose(obs['kilobots'][:, :2] - p0, np.tile([0.0, 0.01], (4, 1)), atol=2e-6)

    env = make(lambda: MomentumLight(position=np.array([0.0, 0.0]), velocity=np.array([0.006, 0.008]), max_velocity=0.01,
                                     radius=0.3, bounds=bounds))
    assert env.observation_space is NotImplemented and env.action_space.shape == (2,)
    obs, *_ = env.step(np.array([0.0, 0.0]))
    np.testing.assert_allclose(obs['light'], [0.006, 0.008, 0.006, 0.008], atol=1e-7)      # 10 substeps of 0.1 s

    env = make(lambda: CompositeLight([CircularGradientLight(position=np.array([-0.3, 0.1]), radius=0.3, bounds=bounds),
                                       MomentumLight(position=np.array([0.4, 0.1]), max_velocity=0.01, radius=0.3, bounds=bounds)]))
    assert env.action_space.shape == (4,)
    obs, *_ = env.step(np.array([0.01, 0.0, 0.0, 0.01]))
    np.testing.assert_allclose(obs['light'][:2], [-0.29, 0.1], atol=1e-6)
    assert obs['light'].shape == (6,) and obs['light'][3] > 0.1           # the momentum component picked up speed along +y


YAML_SCENE = """
!EvalEnv
width: 1.0
height: 0.8
resolution: 600
objects:
  - !ObjectConf {idx: 0, color: [10, 20, 300], shape: circle, width: 0.06, height: 0.06, init: [0.2, 0.1, 0.0], symmetry: none}
  - !ObjectConf {idx: 1, color: null, shape: circle, width: 0.05, height: 0.05, init: random, symmetry: none}
light: !LightConf
  type: composite
  init: fixed
  radius: null
  components:
    - !LightConf {type: circular, init: [-0.2, 0.0], radius: 0.3}
    - !LightConf {type: momentum, init: random, radius: 0.25}
kilobots: !KilobotsConf {num: 12, mean: light, std: 0.03, type: PhototaxisKilobot}
"""


def test_yaml_env_f3():
    import yaml
    from gym_kilobots_amd.envs import YamlKilobotsEnv, EnvConfiguration, UnknownObjectException
    conf = yaml.load(YAML_SCENE, Loader=yaml.Loader)
    assert isinstance(conf, EnvConfiguration) and conf == yaml.load(YAML_SCENE, Loader=yaml.Loader)
    assert conf.objects[0].object_type == 'circle' and conf.kilobots.num == 12
    np.random.seed(3)
    env = YamlKilobotsEnv(configuration=conf, sim_factory=OracleBackend)
    assert env.world_size == (1.0, 0.8) and env.world_bounds[1].tolist() == [0.5, 0.4] and env.screen_width == 600
    obs = env.reset()
    assert obs['kilobots'].shape == (12, 3) and obs['objects'].shape == (2, 3) and obs['light'].shape == (6,)
    # reference quirk kept: the configured kilobot type is ignored (yaml_kilobots_env.py:147,327)
    assert type(env.kilobots[0]).__name__ == 'SimplePhototaxisKilobot'
    assert env.objects[0].color.tolist() == [10, 20, 255] and env.objects[0].get_radius() == 0.06
    assert env.action_space.shape == (4,)
    assert env.kilobots_state_space.shape == (24,) and env.object_observation_space.shape == (8,)
    assert env.state_space.shape == (24 + 6 + 6,) and env.observation_space.shape == (24 + 6 + 8,)
    assert (np.abs(obs['kilobots'][:, 0]) <= 0.5 - 0.02 + 1e-3).all()       # spawn clipped 2 cm inside the arena (+ resolve step)
    o2, r, d, i = env.step(np.array([0.01, 0.0, 0.0, 0.0]))
    assert r == .0 and d is False and abs(o2['light'][0] - (-0.2 + 0.01)) < 1e-6
    env.iteration_counter = 3
    env.inc_iteration_counter()
    assert env.iteration_counter == 4
    YamlKilobotsEnv.honour_kilobot_type = True
    try:
        env.reset()
        assert type(env.kilobots[0]).__name__ == 'PhototaxisKilobot'
    finally:
        YamlKilobotsEnv.honour_kilobot_type = False
    # boxes and triangles run on the device; the multi-fixture shapes fail loudly
    conf.objects[0].shape = 'quad'
    conf.objects[1].shape = 'triangle'
    env2 = YamlKilobotsEnv(configuration=conf, sim_factory=OracleBackend)
    o3 = env2.reset()
    assert type(env2.objects[0]).__name__ == 'Quad' and type(env2.objects[1]).__name__ == 'Triangle'
    assert o3['objects'].shape == (2, 3) and env2.objects[0].vertices.shape == (1, 4, 2)
    env2.step(np.array([0.0, 0.0, 0.0, 0.0]))
    conf.objects[0].shape = 'l_shape'
    env3 = YamlKilobotsEnv(configuration=conf, sim_factory=OracleBackend)
    env3.reset()
    assert type(env3.objects[0]).__name__ == 'LForm' and env3.sim.cfg.num_fixtures == 3
    conf.objects[0].shape = 'pentagon'
    with pytest.raises(UnknownObjectException):
        YamlKilobotsEnv(configuration=conf, sim_factory=OracleBackend).reset()


def test_debug_view_f4(tmp_path):
    import matplotlib
    matplotlib.use('Agg')
    from gym_kilobots_amd import kb_plotting
    be = OracleBackend(2, 9, O_DRIVE_SP, 1, num_objects=1, light_radius=0.3)
    xy = np.stack([np.linspace(-0.3, 0.3, 9), np.zeros(9)], -1)[None].repeat(2, 0)
    be.set_poses_m(xy, np.zeros((2, 9)))
    be.set_objects_m(np.array([[[0.0, 0.3]], [[0.1, 0.3]]]))
    be.num_objects = 1
    kb, objs, light = kb_plotting.snapshot(be, 1)
    assert kb.shape == (9, 3) and objs.shape == (1, 3) and light.shape == (1, 2) and abs(objs[0, 0] - 0.1) < 1e-6
    out = kb_plotting.save_env_png(be, str(tmp_path / 'env.png'), env_index=1, light_radii=[0.3], title='env 1')
    import os
    assert os.path.getsize(out) > 2000


def test_state_is_read_once_per_change_and_never_stale():
    """env.step asks for the state three times; the device is read once per change.  Writes through the body / light
    views, steps and resets invalidate the cached read; callers get copies."""
    env = PhotoEnv(sim_factory=OracleBackend)
    env.reset()
    reads = []
    real = env._read_state
    env._read_state = lambda: (reads.append(1), real())[1]
    s0 = env.get_state()
    s0b = env.get_state()
    assert len(reads) <= 1 and np.array_equal(s0['kilobots'], s0b['kilobots'])
    s0['kilobots'][0, 0] = 123.0                                 # a copy: the next caller is not affected
    assert env.get_state()['kilobots'][0, 0] != 123.0
    n = len(reads)
    obs, *_ = env.step(np.array([0.01, 0.0]))
    assert len(reads) == n + 1                                   # one read for state, next_state and observation
    assert not np.array_equal(obs['kilobots'], s0b['kilobots']) and not np.array_equal(obs['light'], s0b['light'])
    env.kilobots[1].set_pose((0.2, 0.1, 0.5))                    # body view write
    np.testing.assert_allclose(env.get_state()['kilobots'][1], (0.2, 0.1, 0.5), atol=1e-6)
    env._light.set_position(np.array([-0.3, 0.2]))               # light view write
    np.testing.assert_allclose(env.get_state()['light'], (-0.3, 0.2), atol=1e-6)
    env.sim.x[0, 0] = 5.0                                        # direct tensor write: the caller says so
    env.world.touch()
    assert abs(env.get_state()['kilobots'][0, 0] - 0.2) < 1e-6
    first = env.reset()
    assert np.array_equal(first['kilobots'], env.get_state()['kilobots'])
    assert abs(first['kilobots'][0, 0] - 0.0) < 1e-6
