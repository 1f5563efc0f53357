#!/usr/bin/env python3
"""Generate golden vectors from the reference's own NumPy code (build container only).

Runs ONLY where /root/reference exists (never on the GPU box, never at test time).
It imports the reference's ``gym_kilobots.lib.kilobot`` and ``gym_kilobots.lib.light``
unmodified and records inputs -> outputs for the parts of the hot path that are pure
Python/NumPy:

  a2  Kilobot.step motor law            (gym_kilobots/lib/kilobot.py:86-127)
  a3  SimpleVelocityControlKilobot      (gym_kilobots/lib/kilobot.py:213-263)
  a4  SimpleAccelerationControlKilobot  (gym_kilobots/lib/kilobot.py:266-300)
  a5  SimplePhototaxisKilobot.step      (gym_kilobots/lib/kilobot.py:171-210)
  a6  PhototaxisKilobot._loop           (gym_kilobots/lib/kilobot.py:303-333)
  a7  CircularGradientLight / SinglePositionLight / GradientLight / MomentumLight /
      CompositeLight                    (gym_kilobots/lib/light.py)

``gym`` and ``Box2D`` are not installed here (and cannot be), and every reference module
imports one of them at top level.  The two stand-in modules below are *data holders only*:
``spaces.Box`` stores low/high, ``b2Vec2`` is a 2-float (fp32-rounded) vector with the
arithmetic pybox2d's b2Vec2 exposes, and the fake body stores position / angle / velocities
and implements ``GetWorldVector`` / ``GetWorldPoint`` (a rotation by the body angle, in fp32
like Box2D's b2Rot).  No physics is stood in for: the Box2D solver (row a10) stays
"parity unpinned" (DESIGN.md).

Output: tests/golden/reference_vectors.json  (data only: inputs and expected outputs).
"""
import json
import math
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden',
                   'reference_vectors.json')


def f32(x):
    return float(np.float32(x))


# --------------------------------------------------------------------------- stand-ins
class b2Vec2:
    """2-vector with fp32 storage, like Box2D's b2Vec2 seen through pybox2d."""

    def __init__(self, x=0.0, y=0.0):
        if hasattr(x, '__len__'):
            x, y = x[0], x[1]
        self.x = f32(x)
        self.y = f32(y)

    def __iter__(self):
        return iter((self.x, self.y))

    def __len__(self):
        return 2

    def __getitem__(self, i):
        return (self.x, self.y)[i]

    def __mul__(self, a):
        return b2Vec2(self.x * a, self.y * a)

    __rmul__ = __mul__

    def __truediv__(self, a):
        return b2Vec2(self.x / a, self.y / a)

    def __repr__(self):
        return 'b2Vec2(%r,%r)' % (self.x, self.y)


class _FakeBody:
    def __init__(self, position, angle, linearDamping, angularDamping):
        self.position = b2Vec2(*position)
        self.angle = f32(angle)
        self.linearDamping = linearDamping
        self.angularDamping = angularDamping
        self.linearVelocity = b2Vec2(0, 0)
        self.angularVelocity = 0.0

    def _rot(self):
        return f32(math.cos(self.angle)), f32(math.sin(self.angle))

    def GetWorldVector(self, v):
        c, s = self._rot()
        v = b2Vec2(*v)
        return b2Vec2(f32(c * v.x) - f32(s * v.y), f32(s * v.x) + f32(c * v.y))

    def GetWorldPoint(self, p):
        w = self.GetWorldVector(p)
        return b2Vec2(w.x + self.position.x, w.y + self.position.y)

    def CreateCircleFixture(self, **kw):
        return types.SimpleNamespace(**kw)


class b2World:
    def __init__(self, *a, **k):
        pass

    def CreateDynamicBody(self, position, angle, linearDamping, angularDamping):
        return _FakeBody(position, angle, linearDamping, angularDamping)

    def DestroyBody(self, b):
        pass


class _Box:
    def __init__(self, low, high, dtype=np.float64):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = self.low.shape
        self.dtype = dtype


def install_standins():
    box2d = types.ModuleType('Box2D')
    box2d.b2Vec2 = b2Vec2
    box2d.b2World = b2World
    box2d.b2Body = _FakeBody
    box2d.b2ChainShape = object
    gym = types.ModuleType('gym')
    spaces = types.ModuleType('gym.spaces')
    spaces.Box = _Box
    gym.spaces = spaces
    gym.Env = object
    sys.modules['Box2D'] = box2d
    sys.modules['gym'] = gym
    sys.modules['gym.spaces'] = spaces


def body_vel(kb):
    b = kb._body
    return [b.linearVelocity.x, b.linearVelocity.y, float(b.angularVelocity)]


def main():
    install_standins()
    sys.path.insert(0, REF)
    # import the two leaf modules directly (gym_kilobots/lib/__init__ pulls them in)
    from gym_kilobots.lib import kilobot as K
    from gym_kilobots.lib import light as L

    rng = np.random.RandomState(1234)
    world = b2World()
    out = {'meta': {'generator': 'tools/gen_golden.py',
                    'source': 'reference gym_kilobots.lib.kilobot / .light executed with data-only '
                              'stand-ins for gym.spaces.Box and Box2D.b2Vec2/body',
                    'units': 'body velocities are Box2D world units (metres x 25)'}}

    # ---------------------------------------------------------------- a2 motor law
    class MotorBot(K.Kilobot):
        def _setup(self):
            pass

        def _loop(self):
            pass

    cases = []
    thetas = [0.0, 0.3, -1.2, 2.9, 7.5, -11.0]
    motors = [(0, 255), (255, 0), (0, 100), (37, 0), (0, 1), (200, 0)]
    for th in thetas:
        for (ml, mr) in motors:
            kb = MotorBot(world, position=(0.1, -0.2), orientation=th)
            kb.set_motors(ml, mr)
            kb.step(0.1)
            cases.append({'theta': f32(th), 'left': ml, 'right': mr, 'dt': 0.1, 'vel': body_vel(kb)})
    raises = []
    for (ml, mr) in [(255, 255), (128, 64), (0, 0)]:
        kb = MotorBot(world, position=(0, 0), orientation=0.3)
        kb.set_motors(ml, mr)
        try:
            kb.step(0.1)
            raises.append({'left': ml, 'right': mr, 'raises': None})
        except Exception as e:  # reference behaviour: TypeError at kilobot.py:127
            raises.append({'left': ml, 'right': mr, 'raises': type(e).__name__})
    out['a2_motor'] = {'cases': cases, 'raises': raises}

    # ---------------------------------------------------------------- a3 velocity control
    cases = []
    for _ in range(24):
        th = rng.uniform(-7, 7)
        act = rng.uniform([-0.005, -3.0], [0.02, 3.0])
        kb = K.SimpleVelocityControlKilobot(world, position=(0, 0), orientation=th, velocity=[0.0, 0.0])
        kb.set_action(act)
        clamped = [float(a) for a in kb.get_action()]
        kb.step(0.1)
        cases.append({'theta': f32(th), 'action': [float(a) for a in act], 'clamped': clamped,
                      'vel': body_vel(kb)})
    kb = K.SimpleVelocityControlKilobot(world, position=(0, 0), orientation=0.5, velocity=[0.0, 0.0])
    kb.set_action(np.array([0.02, 3.0]))
    kb.step(0.1)
    cases.append({'theta': f32(0.5), 'action': [0.02, 3.0], 'clamped': [float(a) for a in kb.get_action()],
                  'vel': body_vel(kb)})
    kb.set_action(None)
    kb.step(0.1)
    none_case = {'theta': f32(0.5), 'action': None, 'clamped': [float(a) for a in kb.get_action()],
                 'vel': body_vel(kb)}
    out['a3_velocity'] = {'cases': cases, 'none_action': none_case,
                          'action_low': K.SimpleVelocityControlKilobot.action_space.low.tolist(),
                          'action_high': K.SimpleVelocityControlKilobot.action_space.high.tolist(),
                          'density': K.SimpleVelocityControlKilobot._density}

    # ---------------------------------------------------------------- a4 acceleration control
    cases = []
    for _ in range(16):
        th = rng.uniform(-3, 3)
        v0 = rng.uniform([0.0, -1.5], [0.01, 1.5])
        # (`if velocity:` at kilobot.py:225 rejects ndarrays; construct with the random default and
        #  then plant the start velocity)
        kb = K.SimpleAccelerationControlKilobot(world, position=(0.3, 0.1), orientation=th)
        kb._velocity = v0.copy()
        seq = []
        for _s in range(4):
            act = rng.uniform([-0.01, -1.0], [0.01, 1.0])
            kb.set_action(act)
            kb.step(0.1)
            seq.append({'action': [float(a) for a in act],
                        'clamped': [float(a) for a in kb.get_action()],
                        'velocity': [float(v) for v in kb._velocity],
                        'vel': body_vel(kb),
                        'state': [float(s) for s in kb.get_state()]})
        cases.append({'theta': f32(th), 'v0': [float(v) for v in v0], 'steps': seq})
    out['a4_accel'] = {'cases': cases,
                       'action_low': K.SimpleAccelerationControlKilobot.action_space.low.tolist(),
                       'action_high': K.SimpleAccelerationControlKilobot.action_space.high.tolist()}

    # ---------------------------------------------------------------- a7 lights
    lights = {}
    pts = np.concatenate([rng.uniform(-0.5, 0.5, size=(20, 2)),
                          np.array([[0.1, 0.0], [0.3, 0.0], [0.0, -0.05], [0.2, 0.0]])])
    lt = L.CircularGradientLight(position=np.array([0.05, -0.02]), radius=0.2)
    v, g = lt.value_and_gradients(pts.copy())
    lights['circular'] = {'position': [0.05, -0.02], 'radius': 0.2, 'points': pts.tolist(),
                          'values': v.tolist(), 'gradients': g.tolist()}
    lt0 = L.CircularGradientLight(position=np.array([0.0, 0.0]), radius=0.2)
    p0 = np.array([[0.1, 0.0], [0.3, 0.0], [0.0, -0.05]])
    v, g = lt0.value_and_gradients(p0.copy())
    lights['circular_origin'] = {'position': [0.0, 0.0], 'radius': 0.2, 'points': p0.tolist(),
                                 'values': v.tolist(), 'gradients': g.tolist()}
    # light stepping: relative actions clamped to +-0.01, position clamped to bounds
    bounds = (np.array([-1.1, -0.825]), np.array([1.1, 0.825]))
    lt = L.CircularGradientLight(position=np.array([1.0995, 0.0]), radius=0.2, bounds=bounds,
                                 action_bounds=(np.array([-0.01, -0.01]), np.array([0.01, 0.01])))
    seq = []
    for _s in range(8):
        act = rng.uniform(-0.03, 0.03, size=2)
        lt.step(act, 0.1)
        seq.append({'action': act.tolist(), 'position': [float(p) for p in lt.get_state()]})
    lights['circular_step'] = {'start': [1.0995, 0.0], 'bounds': [b.tolist() for b in bounds],
                               'action_bounds': [[-0.01, -0.01], [0.01, 0.01]], 'dt': 0.1, 'steps': seq}
    # SinglePositionLight (negative distance / unit gradient everywhere)
    # SinglePositionLight.value_and_gradients divides (N,2) by (N,): it only broadcasts for N == 2
    sp = L.SinglePositionLight(position=np.array([0.05, -0.02]))
    p2 = pts[:2].copy()
    try:
        v, g = sp.value_and_gradients(p2.copy())
        lights['single_position'] = {'position': [0.05, -0.02], 'points': p2.tolist(),
                                     'values': v.tolist(), 'gradients_as_returned': g.tolist()}
    except Exception as e:
        lights['single_position'] = {'error': type(e).__name__}
    try:
        sp.value_and_gradients(pts.copy())
        lights['single_position']['n24_error'] = None
    except Exception as e:
        lights['single_position']['n24_error'] = type(e).__name__
    # GradientLight
    gl = L.GradientLight(angle=0.4)
    seq = []
    for a in [1.0, -2.5, 7.0, -7.0, 3.5]:
        gl.step(np.array([a]), 0.1)
        seq.append({'action': a, 'angle': float(np.asarray(gl.get_state()).ravel()[0]),
                    'gradient': [float(x) for x in np.asarray(gl.get_gradient(None)).ravel()]})
    lights['gradient'] = {'start_angle': 0.4, 'steps': seq}
    # MomentumLight
    ml = L.MomentumLight(position=np.array([0.0, 0.0]), velocity=np.array([0.006, 0.008]),
                         max_velocity=0.01, radius=0.2, bounds=bounds,
                         action_bounds=(np.array([-0.01, -0.01]), np.array([0.01, 0.01])))
    seq = []
    for _s in range(6):
        act = rng.uniform(-0.02, 0.02, size=2)
        ml.step(act, 0.1)
        seq.append({'action': act.tolist(), 'state': [float(s) for s in ml.get_state()]})
    lights['momentum'] = {'start': [0.0, 0.0, 0.006, 0.008], 'max_velocity': 0.01, 'dt': 0.1,
                          'bounds': [b.tolist() for b in bounds], 'steps': seq}
    # CompositeLight of two circular lights
    c1 = L.CircularGradientLight(position=np.array([-0.1, 0.0]), radius=0.3)
    c2 = L.CircularGradientLight(position=np.array([0.2, 0.1]), radius=0.25)
    cl = L.CompositeLight([c1, c2])
    v, g = cl.value_and_gradients(pts.copy())
    lights['composite'] = {'lights': [{'position': [-0.1, 0.0], 'radius': 0.3},
                                      {'position': [0.2, 0.1], 'radius': 0.25}],
                           'points': pts.tolist(), 'values': v.tolist(), 'gradients': g.tolist()}
    out['a7_light'] = lights

    # ---------------------------------------------------------------- a5 simple phototaxis
    cases = []
    lt0 = L.CircularGradientLight(position=np.array([0.0, 0.0]), radius=0.2)
    for p in [(0.1, 0.0), (0.3, 0.0), (0.0, -0.05), (-0.07, 0.11), (0.19, 0.05)]:
        kb = K.SimplePhototaxisKilobot(world, position=p, orientation=0.7, light=lt0)
        sp = np.array([kb.light_sensor_pos()])
        v, g = lt0.value_and_gradients(sp.copy())
        kb.set_light_value_and_gradient(v[0], g[0])
        kb.step(0.1)
        cases.append({'position': list(p), 'sensor': sp[0].tolist(), 'value': float(v[0]),
                      'gradient': g[0].tolist(), 'vel': body_vel(kb)[:2],
                      'linear_damping_after': float(kb._body.linearDamping)})
    out['a5_simple_phototaxis'] = {'light': {'position': [0.0, 0.0], 'radius': 0.2}, 'cases': cases}

    # ---------------------------------------------------------------- a6 phototaxis state machine
    def run_seq(values):
        kb = K.PhototaxisKilobot(world, position=(0, 0), orientation=0.0)
        trace = []
        for v in values:
            kb.set_light_value_and_gradient(v, np.zeros(2))
            kb.step(0.1)
            trace.append([int(kb._motor_left), int(kb._motor_right)])
        return trace

    seqs = {}
    s1 = []
    for v in [1, 2, 3] + [3] * 20:
        s1 += [float(v)] * 6
    seqs['rising_then_flat'] = s1
    s2 = [float(x) for x in rng.uniform(0, 255, size=150)]
    seqs['random'] = s2
    s3 = [0.0] * 120  # falsy light value -> get_ambientlight returns 0
    seqs['zeros'] = s3
    s4 = [float(255 - 0.5 * i) for i in range(150)]
    seqs['falling'] = s4
    out['a6_phototaxis'] = {name: {'values': vals, 'motors': run_seq(vals)} for name, vals in seqs.items()}

    # sensor position of the base Kilobot: world point of (0, -r)  (kilobot.py:54-55)
    sens = []
    for th in thetas:
        kb = K.PhototaxisKilobot(world, position=(0.2, -0.1), orientation=th)
        sens.append({'theta': f32(th), 'position': [0.2, -0.1],
                     'sensor': [float(s) for s in kb.light_sensor_pos()]})
    out['a7_sensor_pos'] = sens

    # geometry / material constants (kilobot.py:9-30, body.py:7,11-16)
    out['constants'] = {
        'radius': K.Kilobot._radius, 'max_linear_velocity': K.Kilobot._max_linear_velocity,
        'max_angular_velocity': float(K.Kilobot._max_angular_velocity),
        'leg_left': K.Kilobot._leg_left.tolist(), 'leg_right': K.Kilobot._leg_right.tolist(),
        'density': K.Kilobot._density, 'friction': K.Kilobot._friction,
        'restitution': K.Kilobot._restitution, 'linear_damping': K.Kilobot._linear_damping,
        'angular_damping': K.Kilobot._angular_damping, 'world_scale': K._world_scale,
    }

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote', os.path.normpath(OUT), os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
    main()
