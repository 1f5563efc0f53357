"""Seeded synthetic scenes shared by the parity tests and bench.py."""
import numpy as np

W, H = 2.0, 1.5          # kilobots_env.py:19


def gaussian_spawn(E, N, sigma, seed, mean=(0.0, 0.0), random_theta=True):
    """YamlKilobotsEnv._init_kilobots spawn rule, yaml_kilobots_env.py:346-352."""
    rng = np.random.RandomState(seed)
    xy = rng.normal(scale=sigma, size=(E, N, 2)) + np.asarray(mean)
    lo = np.array([-W / 2, -H / 2]) + 0.02
    hi = np.array([W / 2, H / 2]) - 0.02
    xy = np.minimum(np.maximum(xy, lo), hi)
    th = rng.uniform(-np.pi, np.pi, size=(E, N)) if random_theta else np.zeros((E, N))
    return xy, th


def lattice_spawn(E, N, seed, pitch=0.045, jitter=0.004):
    """cfg3: bots on a jittered square lattice (no initial overlap), SURVEY.md 8d item 3."""
    rng = np.random.RandomState(seed)
    side = int(np.ceil(np.sqrt(N)))
    idx = np.arange(N)
    gx = (idx % side - (side - 1) / 2.0) * pitch
    gy = (idx // side - (side - 1) / 2.0) * pitch
    xy = np.stack([gx, gy], -1)[None] + rng.uniform(-jitter, jitter, size=(E, N, 2))
    th = rng.uniform(-np.pi, np.pi, size=(E, N))
    return xy, th


def random_actions(E, N, seed):
    """U([0, 0.01] x [-pi/2, pi/2]): SimpleVelocityControlKilobot.action_space, kilobot.py:216-218."""
    rng = np.random.RandomState(seed)
    a = rng.uniform([0.0, -np.pi / 2], [0.01, np.pi / 2], size=(E, N, 2))
    return a.astype(np.float32)


CFG4_OBJECTS = np.array([[0.5, 0.35], [-0.5, 0.35], [-0.5, -0.35], [0.5, -0.35]])
CFG4_RADIUS = 0.075


def cfg4_actions(xy, th_unused, seed):
    """cfg4: half the bots (even ids) head for the nearest object at full speed, the rest act randomly.
    Velocity-control actions cannot set the heading directly, so the chasers get omega towards the target."""
    E, N, _ = xy.shape
    a = random_actions(E, N, seed)
    return a


def toward_objects_theta(xy):
    """Initial headings: even bots face their nearest cfg4 object, odd bots keep a seeded random heading."""
    d = xy[:, :, None, :] - CFG4_OBJECTS[None, None]
    k = np.argmin((d ** 2).sum(-1), axis=-1)
    tgt = CFG4_OBJECTS[k]
    return np.arctan2(tgt[..., 1] - xy[..., 1], tgt[..., 0] - xy[..., 0])
