#!/usr/bin/env python3
"""Runs small scenes through the independent mini-solver (tools/box2d_mini.py) and writes their trajectories to
tests/golden/mini_solver.json.  tests/test_oracle_vs_mini_solver.py replays the same scenes on the oracle.

    python tools/gen_mini_solver_golden.py            # rewrite the fixture
Scene format (all lengths in Box2D world units = metres x 25, like the reference's b2Body values):
  kilobots: [[x, y, theta, v_cmd m/s, omega_cmd rad/s], ...]  SimpleVelocityControlKilobot (kilobot.py:213-263): every substep
            the env writes v = 25 v_cmd (cos theta, sin theta), omega = omega_cmd into the body, then world.Step
  objects:  [{shape: circle|box, r | hx, hy, x, y, theta, vx, vy, w}, ...]   density 2, friction 0.01, damping 0.8 (body.py:11-16)
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import box2d_mini as B  # noqa: E402

W, H = 2.0 * 25.0, 1.5 * 25.0
R_BOT = 0.0165 * 25.0

SCENES = {
    # single contacts: the order of the Gauss-Seidel sweep cannot matter -> tight tolerance
    'head_on': dict(kilobots=[[-0.5, 0.0, 0.0, 0.01, 0.0], [0.5, 0.02, math.pi, 0.01, 0.0]], objects=[], steps=40, tol=2e-5),
    'glancing': dict(kilobots=[[-0.45, -0.3, 0.6, 0.01, 0.2], [0.3, 0.25, -2.4, 0.008, -0.3]], objects=[], steps=40, tol=2e-5),
    'wall_left': dict(kilobots=[[-25.0 + 0.6, 3.0, math.pi - 0.5, 0.01, 0.1]], objects=[], steps=40, tol=2e-5),
    'wall_top': dict(kilobots=[[4.0, 18.75 - 0.55, 1.2, 0.01, -0.2]], objects=[], steps=40, tol=2e-5),
    'bot_pushes_box': dict(kilobots=[[-2.9, 0.7, 0.0, 0.01, 0.0]], objects=[dict(shape='box', hx=1.875, hy=1.875, x=0.0, y=0.0, theta=0.3, vx=0, vy=0, w=0)],
                           steps=60, tol=5e-5),
    'bot_pushes_disc': dict(kilobots=[[-2.6, 0.4, 0.0, 0.01, 0.0]], objects=[dict(shape='circle', r=1.875, x=0.0, y=0.0, theta=0.0, vx=0, vy=0, w=0)],
                            steps=60, tol=5e-5),
    'box_hits_wall': dict(kilobots=[[20.0, 15.0, 0.0, 0.0, 0.0]], objects=[dict(shape='box', hx=1.875, hy=1.25, x=3.0, y=-18.75 + 1.6, theta=0.25, vx=1.5, vy=-3.0, w=0.4)],
                          steps=40, tol=2e-4),
    'box_slides_on_wall': dict(kilobots=[[20.0, 15.0, 0.0, 0.0, 0.0]], objects=[dict(shape='box', hx=1.875, hy=1.25, x=-5.0, y=-18.75 + 1.25 + 0.015, theta=0.0, vx=4.0, vy=-0.5, w=0.0)],
                               steps=40, tol=2e-4),
    'box_slams_flat': dict(kilobots=[[20.0, 15.0, 0.0, 0.0, 0.0]], objects=[dict(shape='box', hx=1.875, hy=1.25, x=3.0, y=-18.75 + 1.25 + 0.25, theta=0.02, vx=2.5, vy=-8.0, w=1.2)],
                           steps=20, tol=2e-4),
    'disc_hits_wall': dict(kilobots=[[20.0, 15.0, 0.0, 0.0, 0.0]], objects=[dict(shape='circle', r=1.5, x=25.0 - 1.9, y=2.0, theta=0.0, vx=4.0, vy=2.0, w=0.0)],
                           steps=40, tol=5e-5),
    'disc_disc': dict(kilobots=[[20.0, 15.0, 0.0, 0.0, 0.0]], objects=[dict(shape='circle', r=1.5, x=-1.6, y=0.3, theta=0.0, vx=3.0, vy=0.0, w=0.0),
                                                                       dict(shape='circle', r=1.25, x=1.4, y=-0.5, theta=0.0, vx=-2.0, vy=0.5, w=1.0)], steps=40, tol=5e-5),
    'box_box': dict(kilobots=[[20.0, 15.0, 0.0, 0.0, 0.0]], objects=[dict(shape='box', hx=1.875, hy=1.875, x=-2.2, y=0.0, theta=0.0, vx=3.0, vy=0.0, w=0.0),
                                                                     dict(shape='box', hx=1.25, hy=1.25, x=1.3, y=0.6, theta=0.5, vx=-2.0, vy=0.0, w=0.0)], steps=40, tol=2e-4),
    # the continuous step (b2World::SolveTOI) against the walls: the same wall scenes with continuousPhysics on, a fast disc,
    # a kilobot that starts a hair outside the contact radius, a corner
    'wall_left__toi': dict(kilobots=[[-25.0 + 0.6, 3.0, math.pi - 0.5, 0.01, 0.1]], objects=[], steps=40, tol=2e-5, toi=True),
    'wall_top__toi': dict(kilobots=[[4.0, 18.75 - 0.55, 1.2, 0.01, -0.2]], objects=[], steps=40, tol=2e-5, toi=True),
    'wall_graze__toi': dict(kilobots=[[-25.0 + 0.4335, -2.0, math.pi / 2 + 0.05, 0.01, 0.0]], objects=[], steps=40, tol=2e-5, toi=True),
    'corner__toi': dict(kilobots=[[25.0 - 0.62, -18.75 + 0.6, -0.7, 0.01, 0.0]], objects=[], steps=60, tol=5e-5, toi=True),
    'disc_hits_wall__toi': dict(kilobots=[[20.0, 15.0, 0.0, 0.0, 0.0]], objects=[dict(shape='circle', r=1.5, x=25.0 - 1.9, y=2.0, theta=0.0, vx=4.0, vy=2.0, w=0.0)],
                                steps=40, tol=5e-5, toi=True),
    'fast_disc__toi': dict(kilobots=[[20.0, 15.0, 0.0, 0.0, 0.0]], objects=[dict(shape='circle', r=1.25, x=-25.0 + 3.2, y=-4.0, theta=0.0, vx=-16.0, vy=3.0, w=2.0)],
                           steps=30, tol=5e-5, toi=True),
    # several contacts: Box2D sweeps in contact-creation order, the oracle in its canonical key order -> looser tolerance
    'chain_of_three': dict(kilobots=[[-0.9, 0.0, 0.0, 0.01, 0.0], [-0.05, 0.03, 0.0, 0.0, 0.0], [0.8, -0.02, 0.0, 0.0, 0.0]], objects=[], steps=60, tol=2e-3),
    # sleeping (b2World doSleep = True, kilobots_env.py:45): `commands` = [[first step, [[v_cmd, omega_cmd] per kilobot]], ...]
    # replaces the per-kilobot commands from that step on.  Single contacts / no contacts: tight tolerance.
    'sleep_overlap_at_rest': dict(kilobots=[[0.0, 0.0, 0.0, 0.0, 0.0], [0.5, 0.0, 0.0, 0.0, 0.0], [12.5, 12.5, 0.0, 0.0, 0.0]], objects=[],
                                  steps=14, tol=2e-5, sleep=True),
    'sleep_and_wake_by_command': dict(kilobots=[[0.0, 0.0, 0.3, 0.01, 0.2], [6.0, 3.0, 0.0, 0.0, 0.0]], objects=[], steps=30, tol=2e-5, sleep=True, contact_free=True,
                                      commands=[[6, [[0.0, 0.0], [0.0, 0.0]]], [16, [[0.0003, 0.0], [0.0, 0.0]]], [24, [[0.008, -0.3], [0.0, 0.0]]]]),
    'sleeping_bot_woken_by_a_pusher': dict(kilobots=[[0.0, 0.0, 0.0, 0.0, 0.0], [1.5, 0.05, math.pi, 0.0, 0.0]], objects=[], steps=60, tol=5e-5, sleep=True,
                                           commands=[[8, [[0.0, 0.0], [0.01, 0.0]]]]),
    # b2Contact::Update: a contact that stops touching wakes both bodies -- kilobot 2 rests against kilobots 0 and 1, all three fall
    # asleep, then it is commanded away: its departure (the contacts end) wakes the two it leaves behind, whose sleep times restart
    'mover_leaves_a_sleeping_pair': dict(kilobots=[[0.0, 0.0, 0.0, 0.0, 0.0], [0.0, 0.82, 0.0, 0.0, 0.0], [0.71, 0.41, 0.0, 0.0, 0.0]], objects=[],
                                         steps=40, tol=5e-5, sleep=True, commands=[[14, [[0.0, 0.0], [0.0, 0.0], [0.01, 0.0]]]]),
    # ... and the case in which ONLY that rule wakes the sleeper: kilobot 0 creeps away from kilobot 1 below the sleep tolerance (its
    # tiny command re-wakes it every step, b2Body::SetLinearVelocity); both rest, fall asleep together at the end of step 10 -- the
    # step in which they stop touching -- and in step 11 the creeper's wake-up finds the contact gone: kilobot 1 is woken although no
    # island reaches it any more, and sleeps again five steps later
    'creeper_leaves_a_sleeper': dict(kilobots=[[0.0, 0.0, math.pi, 0.0002, 0.0], [0.8206, 0.0, 0.0, 0.0, 0.0]], objects=[], steps=20, tol=2e-5, sleep=True),
    'disc_coasts_to_sleep': dict(kilobots=[[20.0, 15.0, 0.0, 0.0, 0.0]], objects=[dict(shape='circle', r=1.5, x=0.0, y=0.0, theta=0.0, vx=0.6, vy=0.2, w=0.3)],
                                 steps=70, tol=5e-5, sleep=True, contact_free=True),
    # kilobots of different classes: fixture density 2 (SimpleVelocityControlKilobot, kilobot.py:214) against density 1 (Kilobot,
    # :25): a sixth column gives the density; the light one is commanded (0, 0) like a kilobot with both motors off
    'heavy_pushes_light': dict(kilobots=[[-0.6, 0.0, 0.0, 0.01, 0.0, 2.0], [0.45, 0.03, 0.0, 0.0, 0.0, 1.0]], objects=[], steps=50, tol=2e-5),
    'heavy_light_heavy_chain': dict(kilobots=[[-0.9, 0.0, 0.0, 0.01, 0.0, 2.0], [-0.05, 0.03, 0.0, 0.0, 0.0, 1.0], [0.8, -0.02, 0.0, 0.0, 0.0, 2.0]],
                                    objects=[], steps=60, tol=2e-3),
    'two_bots_push_box': dict(kilobots=[[-2.9, 0.9, 0.0, 0.01, 0.0], [-2.9, -0.8, 0.0, 0.01, 0.0]],
                              objects=[dict(shape='box', hx=1.875, hy=1.875, x=0.0, y=0.0, theta=0.0, vx=0, vy=0, w=0)], steps=60, tol=2e-3),
}


def run(scene, damping='pade', dt=0.1, vel_iters=10, pos_iters=10):
    w = B.World(damping=damping, continuous=bool(scene.get('toi')), allow_sleep=bool(scene.get('sleep')))
    table = w.create_body(dynamic=False)
    x0, x1, y0, y1 = -W / 2, W / 2, -H / 2, H / 2
    for a, b in (((x0, y1), (x0, y0)), ((x0, y0), (x1, y0)), ((x1, y0), (x1, y1)), ((x1, y1), (x0, y1))):   # kilobots_env.py:48-51
        table.create_fixture(B.Edge(a, b), density=0.0, friction=0.2)
    bots, objs = [], []
    for kb in scene['kilobots']:
        x, y, th, v, om = kb[:5]
        b = w.create_body(position=(x, y), angle=th, linear_damping=0.8, angular_damping=0.8)
        b.create_fixture(B.Circle(R_BOT), density=kb[5] if len(kb) > 5 else 2.0, friction=0.0, restitution=0.0)
        bots.append([b, np.float32(v), np.float32(om)])
    for o in scene['objects']:
        b = w.create_body(position=(o['x'], o['y']), angle=o['theta'], linear_damping=0.8, angular_damping=0.8)
        shape = B.Circle(o['r']) if o['shape'] == 'circle' else B.Polygon.box(o['hx'], o['hy'])
        b.create_fixture(shape, density=2.0, friction=0.01, restitution=0.0)
        b.v, b.w = B.V(o['vx'], o['vy']), np.float32(o['w'])
        objs.append(b)
    traj = []
    for k in range(scene['steps']):
        for first, cmds in scene.get('commands', []):
            if first == k:
                for bot, (v, om) in zip(bots, cmds):
                    bot[1], bot[2] = np.float32(v), np.float32(om)
        for b, v, om in bots:                 # SimpleVelocityControlKilobot.step, kilobot.py:253-258 (assignments through the
            sp = v * np.float32(25.0)         # b2Body setters, which wake a sleeping body iff the value is non-zero)
            b.set_angular_velocity(om)
            b.set_linear_velocity(B.V(np.float32(math.cos(float(b.a))) * sp, np.float32(math.sin(float(b.a))) * sp))
        w.step(dt, vel_iters, pos_iters)
        rec = {'kilobots': [[float(b.position.x), float(b.position.y), float(b.angle)] for b, _, _ in bots],
               'objects': [[float(b.position.x), float(b.position.y), float(b.angle), float(b.v.x), float(b.v.y), float(b.w)] for b in objs]}
        if scene.get('sleep'):                # m_sleepTime, -1 = asleep (the encoding of kb_buffers.sleep_time)
            rec['sleep'] = [float(b.sleep_time) if b.awake else -1.0 for b, _, _ in bots]
            rec['osleep'] = [float(b.sleep_time) if b.awake else -1.0 for b in objs]
        traj.append(rec)
    touched = any(c.touching for c in w.contacts)
    return traj, touched


def main():
    out = {'_about': 'trajectories of tools/box2d_mini.py (independent restatement of Box2D 2.3.1 b2World::Step, continuousPhysics off); '
                     'made by tools/gen_mini_solver_golden.py; world units (metres x 25)', 'scenes': {}}
    for name, sc in SCENES.items():
        for damping in ('pade', 'linear'):
            if damping == 'linear' and name not in ('head_on', 'box_hits_wall'):
                continue
            traj, _ = run(sc, damping)
            key = name if damping == 'pade' else name + '__linear_damping'
            out['scenes'][key] = dict(sc, damping=damping, trajectory=traj)
            print(key, 'final', traj[-1]['kilobots'][0], traj[-1]['objects'][:1])
    p = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'mini_solver.json')
    json.dump(out, open(p, 'w'))
    print('wrote', p, os.path.getsize(p), 'bytes')


if __name__ == '__main__':
    main()
