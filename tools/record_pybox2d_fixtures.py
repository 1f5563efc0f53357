#!/usr/bin/env python3
"""Record the scenes of tests/golden/mini_solver.json in REAL Box2D (the `box2d-py` wheel the reference depends on, setup.py:5)
and write tests/golden/pybox2d_trajectories.json in the same format -- the one-command act that turns "parity unpinned" for
SURVEY row a10 into a pinned comparison wherever the wheel can be installed (it cannot in the build container: no network).

    pip install box2d-py numpy        # on any machine with network access
    python tools/record_pybox2d_fixtures.py
    python -m pytest tests/test_oracle_vs_pybox2d.py      # consumes the file when present, skips otherwise

What is recorded: for every scene the poses (x, y, theta in world units = metres x 25, body.py:7) of all kilobots and objects
after every world.Step(0.1, 10, 10), the bodies built exactly as gym_kilobots/lib/body.py and kilobot.py build them
(linearDamping = angularDamping = 0.8, circle fixtures of radius 0.4125 and density 1.0 / 2.0, friction 0.0; objects density 2,
friction 0.01, restitution 0; arena = static body with a b2ChainShape of the four corners like kilobots_env.py:46-51), the
commanded velocities re-assigned before every step like Kilobot.step does.  Also stored: Box2D's version string and which damping
formula it uses, so that the consumer can pick kb_config.damping_model."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    try:
        import Box2D
        from Box2D import b2World, b2ChainShape, b2PolygonShape
    except ImportError:
        sys.exit('box2d-py is not installed: pip install box2d-py (needs network access; not available in the build container)')
    import numpy as np
    src = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'mini_solver.json')))
    out = {'_about': 'trajectories of REAL Box2D (box2d-py %s) for the scenes of mini_solver.json; made by tools/record_pybox2d_fixtures.py; '
                     'world units (metres x 25)' % getattr(Box2D, '__version__', '?'), 'box2d_version': getattr(Box2D, '__version__', '?'), 'scenes': {}}
    W, H = 2.0 * 25.0, 1.5 * 25.0
    for name, sc in src['scenes'].items():
        if name.endswith('__linear_damping'):
            continue                       # (a real Box2D has one damping formula: recorded below, the consumer picks kb_config.damping_model)
        world = b2World(gravity=(0, 0), doSleep=bool(sc.get('sleep')))
        world.continuousPhysics = bool(sc.get('toi'))
        arena = world.CreateStaticBody(position=(0, 0))
        arena.CreateFixture(shape=b2ChainShape(vertices=[(-W / 2, -H / 2), (W / 2, -H / 2), (W / 2, H / 2), (-W / 2, H / 2)]))   # kilobots_env.py:46-51
        bots = []
        for kb in sc['kilobots']:
            x, y, th, v, w = kb[:5]
            body = world.CreateDynamicBody(position=(x, y), angle=th, linearDamping=0.8, angularDamping=0.8)       # body.py:32-36
            body.CreateCircleFixture(radius=0.0165 * 25.0, density=kb[5] if len(kb) > 5 else 2.0, friction=0.0, restitution=0.0)
            bots.append([body, v, w])
        objs = []
        for ob in sc.get('objects', []):
            body = world.CreateDynamicBody(position=(ob['x'], ob['y']), angle=ob.get('theta', 0.0), linearDamping=0.8, angularDamping=0.8)
            if ob['shape'] == 'circle':
                body.CreateCircleFixture(radius=ob['r'], density=2.0, friction=0.01, restitution=0.0)        # body.py:187-192
            else:
                body.CreatePolygonFixture(box=(ob['hx'], ob['hy']), density=2.0, friction=0.01, restitution=0.0)   # body.py:136-142
            body.linearVelocity = (ob.get('vx', 0.0), ob.get('vy', 0.0))
            body.angularVelocity = ob.get('w', 0.0)
            objs.append(body)
        traj = []
        for k in range(sc['steps']):
            for first, cmds in sc.get('commands', []):
                if first == k:
                    for bot, (v, w) in zip(bots, cmds):
                        bot[1], bot[2] = v, w
            for body, v, w in bots:        # SimpleVelocityControlKilobot.step, kilobot.py:253-258
                body.angularVelocity = w
                body.linearVelocity = (float(np.cos(body.angle)) * v * 25.0, float(np.sin(body.angle)) * v * 25.0)
            world.Step(0.1, 10, 10)
            world.ClearForces()
            rec = {'kilobots': [[b.position[0], b.position[1], b.angle] for b, _, _ in bots],
                   'objects': [[o.position[0], o.position[1], o.angle, o.linearVelocity[0], o.linearVelocity[1], o.angularVelocity] for o in objs]}
            if sc.get('sleep'):
                rec['asleep'] = [not b.awake for b, _, _ in bots] + [not o.awake for o in objs]
            traj.append(rec)
        out['scenes'][name] = dict({k_: v_ for k_, v_ in sc.items() if k_ != 'trajectory'}, trajectory=traj)
    # which damping formula does this Box2D use?  one free body, c = 0.8, h = 0.1: 0.9259259 (Pade) or 0.92 (linear)
    world = b2World(gravity=(0, 0))
    b = world.CreateDynamicBody(position=(0, 0), linearDamping=0.8)
    b.CreateCircleFixture(radius=0.4125, density=1.0)
    b.linearVelocity = (1.0, 0.0)
    world.Step(0.1, 10, 10)
    out['damping_factor_at_c0.8_h0.1'] = float(b.linearVelocity[0])
    dst = os.path.join(ROOT, 'tests', 'golden', 'pybox2d_trajectories.json')
    json.dump(out, open(dst, 'w'))
    print('wrote', dst, 'damping factor', out['damping_factor_at_c0.8_h0.1'])


if __name__ == '__main__':
    main()
