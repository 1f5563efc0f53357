"""GPU parity of the round-2 additions, through the C ABI against the oracle: IR-range neighbour sensing (kb_sense and the
pass fused into kb_step), kb_reset (Philox-keyed device spawn), the damping models, capacity overflow behaviour, and the
drop-in API pieces that round 1 only exercised on the oracle backend (Body.collides_with, kb_plotting)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import scenes
from tests.test_parity_gpu import make_pair, assert_same, assert_ws_same, cpu, dev

pytestmark = pytest.mark.gpu


def counts(g):
    return cpu(g).view(np.uint32) if cpu(g).dtype != np.uint32 else cpu(g)


# ---------------------------------------------------------------------------------------------- kb_sense
@pytest.mark.parametrize('E,N,R', [(8, 64, 0.07), (8, 64, 0.3), (4, 1024, 0.05), (4, 1024, 0.1), (3, 333, 0.034), (2, 7, 0.5), (5, 1, 0.1)])
def test_sense_equals_brute_force_oracle(E, N, R):
    """cfg2 / cfg3 slices, odd sizes; R from one cell (0.035 m) to R > 2 cells and the whole arena."""
    if N == 1024:
        xy, th = scenes.lattice_spawn(E, N, seed=3)
    else:
        xy, th = scenes.gaussian_spawn(E, N, sigma=0.2, seed=4)
    osim, gsim = make_pair(E, N, xy=xy, th=th)
    got = counts(gsim.sense(R))
    assert np.array_equal(got, osim.sense(R))
    assert got.max() > 0 or N == 1


def test_sense_at_walls_and_corners():
    """Kilobots pressed into the corners and along the walls, some outside the arena (cell indices clamp)."""
    N = 96
    rng = np.random.RandomState(9)
    xy = np.zeros((4, N, 2))
    corners = np.array([[-1.0, -0.75], [1.0, -0.75], [1.0, 0.75], [-1.0, 0.75]])
    for e in range(4):
        xy[e, :24] = corners[e] + rng.uniform(-0.03, 0.08, size=(24, 2)) * -np.sign(corners[e])
        xy[e, 24:48] = np.stack([rng.uniform(-1, 1, 24), np.full(24, 0.75 - 0.0165) + rng.uniform(-0.01, 0.03, 24)], -1)
        xy[e, 48:72] = np.stack([np.full(24, -1.0 + 0.0165) + rng.uniform(-0.03, 0.01, 24), rng.uniform(-0.75, 0.75, 24)], -1)
        xy[e, 72:] = rng.uniform(-0.2, 0.2, size=(24, 2))
    osim, gsim = make_pair(4, N, xy=xy, th=np.zeros((4, N)))
    for R in (0.04, 0.09, 0.15):
        assert np.array_equal(counts(gsim.sense(R)), osim.sense(R)), R


@pytest.mark.parametrize('E,N,R,kw', [(6, 64, 0.07, {}), (3, 1024, 0.07, {}), (4, 100, 0.12, dict(num_objects=1)),
                                      (4, 200, 0.05, dict(solver_mode=1)), (2, 1024, 0.1, dict(solver_mode=3))])
def test_fused_sensing_in_the_step(E, N, R, kw):
    """kb_config.sense_radius: kb_step leaves the counts of the last substep's sensing point in nbr_count -- equal to the
    oracle's, and equal to kb_sense of the poses before that substep; stepping itself is unchanged by the sensing."""
    xy, th = scenes.lattice_spawn(E, N, seed=6) if N == 1024 else scenes.gaussian_spawn(E, N, sigma=0.15, seed=6)
    objects = np.tile(np.array([[0.3, 0.2]])[None], (E, 1, 1)) if kw.get('num_objects') else None
    kw = {k: v for k, v in kw.items() if k != 'num_objects'}
    osim, gsim = make_pair(E, N, xy=xy, th=th, objects=objects, sense_radius=R, **kw)
    _, plain = make_pair(E, N, xy=xy, th=th, objects=objects, **kw)
    for k in range(3):
        a = scenes.random_actions(E, N, seed=40 + k)
        osim.set_actions(a)
        osim.step(4)
        gsim.step(3, actions=dev(a))
        before_last = counts(gsim.sense(R)).copy()
        gsim.step(1)
        plain.step(4, actions=dev(a))
        torch.cuda.synchronize()
        assert np.array_equal(counts(gsim.nbr_count), osim.nbr_count), k
        assert np.array_equal(counts(gsim.nbr_count), before_last), k
        assert_same(osim, gsim, 'sensing step %d' % k)
        for f in ('x', 'y', 'theta'):
            assert torch.equal(getattr(gsim, f), getattr(plain, f)), f
    assert int(cpu(gsim.status).max()) == 0


# ---------------------------------------------------------------------------------------------- kb_reset
@pytest.mark.parametrize('mode', [O.DRIVE_VELOCITY, O.DRIVE_ACCEL, O.DRIVE_MOTORS])
def test_device_reset_equals_oracle(mode):
    E, N = 16, 200
    osim, gsim = make_pair(E, N, mode)
    # (std 3.0: most of the cloud is clipped onto the bounds -- coincident kilobots in the corners, more contacts than the
    #  store holds, so that case compares the spawn itself and not the step to resolve)
    for seed, std, rt, rv, resolve in ((0, 0.1, False, False, True), (2 ** 40 + 5, 0.3, True, True, True), (7, 3.0, True, False, False)):
        osim.reset(seed=seed, mean=(0.05, -0.1), std=std, random_theta=rt, random_velocity=rv, resolve=resolve, env_offset=3)
        gsim.reset(seed=seed, mean=(0.05, -0.1), std=std, random_theta=rt, random_velocity=rv, resolve=resolve, env_offset=3)
        fields = ('x', 'y', 'theta') + (('v', 'w') if mode != O.DRIVE_MOTORS else ('motor_l', 'motor_r'))
        assert_same(osim, gsim, 'reset seed %d' % seed, fields)
        assert_ws_same(osim, gsim, 'reset seed %d' % seed)
        if not resolve:
            continue
        assert int(osim.status.max()) == 0
        a = scenes.random_actions(E, N, seed=1)
        if mode != O.DRIVE_MOTORS:
            osim.set_actions(a)
            gsim.set_actions(dev(a))
        osim.step(5)
        gsim.step(5)
        assert_same(osim, gsim, 'after reset seed %d' % seed)


def test_device_reset_full_size_properties():
    """cfg3 size: 4096 x 1024 draws on the device; moments of the cloud, determinism, shard == rows."""
    from gym_kilobots_amd.sim import KilobotSim
    E, N = 4096, 1024
    g = KilobotSim(E, N)
    g.reset(seed=5, std=0.25, resolve=False)
    x, y = g.x.double() / 25.0, g.y.double() / 25.0
    assert abs(float(x.mean())) < 1e-3 and abs(float(y.mean())) < 1e-3
    assert abs(float(x.std()) - 0.25) < 1e-3 and abs(float(y.std()) - 0.25) < 1e-3
    assert float(x.abs().max()) <= 1.0 - 0.02 + 1e-6 and float(y.abs().max()) <= 0.75 - 0.02 + 1e-6
    x1 = g.x.clone()
    part = KilobotSim(64, N)
    part.reset(seed=5, std=0.25, resolve=False, env_offset=1000)
    assert torch.equal(part.x, x1[1000:1064])
    g.reset(seed=5, std=0.25, resolve=False)
    assert torch.equal(g.x, x1)
    assert int(g.status.max().item()) == 0


def test_batched_env_resets_on_the_device():
    from gym_kilobots_amd.envs import BatchedKilobotsEnv
    from tests.oracle_backend import OracleBackend
    g = BatchedKilobotsEnv(6, 80, seed=3, spawn_std=0.12)
    o = BatchedKilobotsEnv(6, 80, seed=3, spawn_std=0.12, sim_factory=OracleBackend)
    assert np.array_equal(cpu(g.reset()), o.reset().numpy())
    a = scenes.random_actions(6, 80, seed=2)
    og, *_ = g.step(dev(a))
    oo, *_ = o.step(torch.from_numpy(a))
    assert np.array_equal(cpu(og), oo.numpy())
    assert type(g.sim).__name__ == 'KilobotSim'


# ---------------------------------------------------------------------------------------------- damping
def test_linear_damping_model_parity():
    E, N = 4, 128
    xy, th = scenes.gaussian_spawn(E, N, sigma=0.1, seed=8)
    osim, gsim = make_pair(E, N, xy=xy, th=th, damping_model=1)
    _, pade = make_pair(E, N, xy=xy, th=th)
    for k in range(3):
        a = scenes.random_actions(E, N, seed=60 + k)
        osim.set_actions(a)
        osim.step(10)
        gsim.step(10, actions=dev(a))
        pade.step(10, actions=dev(a))
        assert_same(osim, gsim, 'linear damping step %d' % k)
    assert not torch.equal(gsim.x, pade.x)


# ---------------------------------------------------------------------------------------------- overflow
def test_overlapping_1024_spawn_in_the_last_env():
    """a heavily overlapping 1024-kilobot spawn (about 14 000 touching pairs against a default capacity of
    4160) in the LAST env: the per-bot warm-start counts add up to more than the slice holds, reads behind it are
    guarded, status bit 0 is raised, the other envs are untouched -- and with a sized contact store the pile is solved
    without any flag and equals the oracle bit for bit."""
    from gym_kilobots_amd.sim import KilobotSim
    E, N = 3, 1024
    xy, th = scenes.lattice_spawn(E, N, seed=1)
    dense, _ = scenes.gaussian_spawn(1, N, sigma=0.1, seed=2)
    xy[E - 1] = dense[0]
    g = KilobotSim(E, N)
    g.set_poses_m(xy, th)
    a = dev(scenes.random_actions(E, N, seed=3))
    for _ in range(4):
        g.step(2, actions=a)
    torch.cuda.synchronize()
    st = cpu(g.status)
    assert st[E - 1] & 1 and st[0] == 0 and st[1] == 0
    assert bool(torch.isfinite(g.x).all()) and bool(torch.isfinite(g.y).all())
    ref = KilobotSim(2, N)
    ref.set_poses_m(xy[:2], th[:2])
    for _ in range(4):
        ref.step(2, actions=a[:2].contiguous())
    assert torch.equal(ref.x, g.x[:2]) and torch.equal(ref.y, g.y[:2])
    with pytest.raises(Exception, match='contact capacity overflow'):
        g.check_status('raise')
    # a pile that needs more than the default store but stays inside the staging limits of the device step (<= 255
    # contacts per cell pair): with a sized store no flag, bit-exact
    dense, _ = scenes.gaussian_spawn(1, N, sigma=0.15, seed=2)
    osim, big = make_pair(1, N, xy=dense, th=th[:1], contact_capacity=12000, ws_slots=64)
    an = scenes.random_actions(1, N, seed=3)
    for k in range(2):
        osim.set_actions(an)
        osim.step(2)
        big.step(2, actions=dev(an))
        assert_same(osim, big, 'dense pile step %d' % k)
        assert_ws_same(osim, big, 'dense pile step %d' % k)
    assert int(cpu(big.status).max()) == 0 and int(osim.status.max()) == 0
    assert osim.count_contacts(0)[0] > 4160


# ---------------------------------------------------------------------------------------------- drop-in API on the real sim
def _pushing_env(**kw):
    from gym_kilobots_amd.envs import KilobotsEnv
    from gym_kilobots_amd.lib import SimpleVelocityControlKilobot, Quad, Circle

    class Env(KilobotsEnv):
        def _configure_environment(self):
            self._add_object(Quad(world=self.world, width=0.15, height=0.15, position=(0.0, 0.0)))
            self._add_object(Circle(world=self.world, radius=0.05, position=(0.3, 0.0)))
            for i, p in enumerate([(-0.0915, 0.0), (0.0, 0.0915), (-0.4, -0.3), (-0.4 + 0.032, -0.3), (0.3 - 0.066, 0.0)]):
                self._add_kilobot(SimpleVelocityControlKilobot(self.world, position=p, orientation=0.0, velocity=(0.0, 0.0)))

        def get_reward(self, *a):
            return 0.0
    return Env(**kw)


def test_collides_with_on_the_gpu_equals_the_oracle_backend():
    """a12: Body.collides_with (body.py:87-90) evaluated on the device poses of a real KilobotSim."""
    from tests.oracle_backend import OracleBackend
    g, o = _pushing_env(), _pushing_env(sim_factory=OracleBackend)
    g.reset()
    o.reset()
    assert type(g.sim).__name__ == 'KilobotSim'
    for env in (g, o):
        kb, ob = env.kilobots, env.objects
        table = [[bool(a.collides_with(b)) for b in list(kb) + list(ob)] for a in list(kb) + list(ob)]
        env._table = table
    assert g._table == o._table
    kb, ob = g.kilobots, g.objects
    assert kb[0].collides_with(ob[0]) and kb[1].collides_with(ob[0]) and kb[2].collides_with(kb[3]) and kb[4].collides_with(ob[1])
    assert not kb[0].collides_with(kb[2]) and not kb[2].collides_with(ob[0]) and not ob[0].collides_with(ob[1])
    g.close()


def test_plotting_snapshot_of_a_real_sim(tmp_path):
    """f4: the matplotlib debug view draws env 0 of a KilobotSim (round 1 only ran it on the oracle backend)."""
    import matplotlib
    matplotlib.use('Agg')
    from gym_kilobots_amd import kb_plotting
    g = _pushing_env()
    g.reset()
    kb, objs, light = kb_plotting.snapshot(g.sim)
    assert kb.shape == (5, 3) and objs.shape == (2, 3) and light is None
    assert np.allclose(kb[:, :2], [k.get_position() for k in g.kilobots], atol=1e-7)
    h = 0.075
    ax = kb_plotting.plot_env(g.sim, object_radii=[0.0, 0.05], object_vertices=[[(-h, -h), (h, -h), (h, h), (-h, h)], None], title='env 0')
    assert len(ax.patches) == 1 + 2 + 5            # arena, two objects, five kilobots
    out = tmp_path / 'env0.png'
    kb_plotting.save_env_png(g.sim, str(out), object_radii=[0.075, 0.05])
    assert out.stat().st_size > 2000
    g.close()


def test_three_envs_of_1024_kilobots_resident_per_cu():
    """The benchmark instantiation holds THREE envs per CU (compact LDS image <= 52 KiB, 80 VGPRs): the HIP occupancy query
    through the C ABI says so; the other 1024-kilobot configurations keep the regular image and two."""
    from gym_kilobots_amd.sim import KilobotSim
    g = KilobotSim(8, 1024)
    assert g.lds_bytes <= 52 * 1024 and g.resident_envs_per_cu == 3
    assert KilobotSim(8, 1024, num_objects=4).resident_envs_per_cu == 2
    assert KilobotSim(8, 1024, contact_capacity=12000).resident_envs_per_cu == 3      # (round 3: the generic kernels stage 688 contacts too)
    # and it computes what the regular image computes: the same scene on both, bit for bit, over fused and single substeps
    xy, th = scenes.lattice_spawn(8, 1024, seed=21, pitch=0.04)
    big = KilobotSim(8, 1024, contact_capacity=4168)          # one entry more than the default: the generic kernel (run-time sizes)
    assert big.lds_bytes >= g.lds_bytes and big.resident_envs_per_cu == 3
    for s in (g, big):
        s.set_poses_m(xy, th)
    for k in range(6):
        a = dev(scenes.random_actions(8, 1024, seed=70 + k))
        n = (1, 3, 10)[k % 3]
        g.step(n, actions=a)
        big.step(n, actions=a)
        for f in ('x', 'y', 'theta'):
            assert torch.equal(getattr(g, f), getattr(big, f)), (k, f)
    assert int(g.status.max().item()) == 0


def test_status_bits_is_the_or_over_envs_in_one_read():
    """KilobotSim.status_bits: the small-batch path (one copy, OR on the host) and the large-batch path (masked reduction
    on the device) name the same bits; bits outside kb_status are not reported."""
    from gym_kilobots_amd.sim import KilobotSim
    from gym_kilobots_amd import _native as nat
    small = KilobotSim(8, 16)
    assert small.status_bits() == 0
    small.status[3] = 2
    small.status[6] = 8 | 1
    assert small.status_bits() == 11
    small.status[0] = 64                      # not a kb_status flag
    assert small.status_bits() == 11
    with pytest.raises(nat.KilobotsStatusError, match='warm-start slot overflow'):
        small.check_status('raise')
    big = KilobotSim((1 << 16) + 8, 4)
    assert big.status_bits() == 0
    big.status[70000 - 4500] = 4
    big.status[5] = 1
    assert big.status_bits() == 5


def test_host_state_is_poses_objects_and_status_in_one_copy():
    """kb_get_state: the packed read-back equals kb_get_poses, the object views divided by 25 and kb_buffers.status, bit for bit."""
    from gym_kilobots_amd.sim import KilobotSim
    gsim = KilobotSim(5, 48, num_objects=2, obj_radius=[0.06, 0.09] + [0.075] * 6)
    gsim.reset(seed=11, std=0.2)
    gsim.set_objects_m(np.tile(np.array([[0.5, 0.3], [-0.4, -0.2]], np.float32), (5, 1, 1)), theta=np.tile(np.array([0.2, 1.0], np.float32), (5, 1)))
    a = np.random.RandomState(2).uniform([0.0, -1.0], [0.01, 1.0], (5, 48, 2)).astype(np.float32)
    gsim.set_actions(dev(a))
    gsim.step(7)
    gsim.status[3] = 9
    kb, objs, status = gsim.host_state()
    assert kb.shape == (5, 48, 3) and objs.shape == (5, 2, 3) and status.dtype == np.int32
    assert np.array_equal(kb, cpu(gsim.poses()))
    assert np.array_equal(objs, cpu(gsim.object_poses()))
    assert np.array_equal(status, [0, 0, 0, 9, 0])
    plain = KilobotSim(3, 20)
    plain.reset(seed=1, std=0.2)
    kb, objs, status = plain.host_state()
    assert objs.shape == (3, 0, 3) and np.array_equal(kb, cpu(plain.poses())) and not status.any()
