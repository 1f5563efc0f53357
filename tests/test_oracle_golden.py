"""Pins the CPU oracle against vectors produced by the reference's own code
(tests/golden/reference_vectors.json, generator tools/gen_golden.py) and against the
closed-form known-answer values of SURVEY.md section 8c.  CPU only."""
import math

import numpy as np
import pytest

from oracle import oracle as O

FAR = (0.0, 0.0)  # metres: arena centre, far from every wall


def one_bot(mode, light=O.LIGHT_NONE, pos=FAR, theta=0.0, **kw):
    sim = O.OracleSim(O.default_config(1, 1, mode, light, **kw))
    sim.set_poses_m(np.array([[pos]]), np.array([[theta]]))
    return sim


def test_sincos_matches_libm():
    xs = np.concatenate([np.linspace(-20, 20, 4001), np.random.RandomState(0).uniform(-2000, 2000, 2000)])
    for x in xs.astype(np.float32):
        s, c = O.sincosf(float(x))
        assert abs(s - math.sin(float(x))) < 2.5e-7 and abs(c - math.cos(float(x))) < 2.5e-7


def test_a2_motor_law(golden):
    for case in golden['a2_motor']['cases']:
        sim = one_bot(O.DRIVE_MOTORS, theta=case['theta'])
        sim.motor_l[...] = case['left']
        sim.motor_r[...] = case['right']
        sim.step(1)
        got = [sim.cmd_vx[0, 0], sim.cmd_vy[0, 0], sim.cmd_w[0, 0]]
        np.testing.assert_allclose(got, case['vel'], rtol=2e-5, atol=2e-7)
    # the reference raises TypeError for both-on / both-off (kilobot.py:127); recorded, not reproduced
    assert all(r['raises'] == 'TypeError' for r in golden['a2_motor']['raises'])


def test_a2_both_motors_intended_meaning():
    sim = one_bot(O.DRIVE_MOTORS, theta=0.3)
    sim.motor_l[...] = 128
    sim.motor_r[...] = 64
    sim.step(1)
    lin = (128 + 64) / 510. * 0.01
    np.testing.assert_allclose([sim.cmd_vx[0, 0], sim.cmd_vy[0, 0], sim.cmd_w[0, 0]],
                               [math.sin(0.3) * lin * 25, math.cos(0.3) * lin * 25, (64 - 128) / 510. * math.pi / 2],
                               rtol=1e-6)
    sim.motor_l[...] = 0
    sim.motor_r[...] = 0
    sim.step(1)
    assert sim.cmd_vx[0, 0] == 0 and sim.cmd_vy[0, 0] == 0 and sim.cmd_w[0, 0] == 0


def test_a3_velocity_control(golden):
    g = golden['a3_velocity']
    for case in g['cases']:
        sim = one_bot(O.DRIVE_VELOCITY, theta=case['theta'])
        sim.set_actions(np.array([[case['action']]]))
        np.testing.assert_allclose([sim.v[0, 0], sim.w[0, 0]], case['clamped'], rtol=1e-6, atol=1e-9)
        sim.step(1)
        np.testing.assert_allclose([sim.cmd_vx[0, 0], sim.cmd_vy[0, 0], sim.cmd_w[0, 0]], case['vel'],
                                   rtol=2e-6, atol=1e-7)
    sim = one_bot(O.DRIVE_VELOCITY, theta=0.5)
    sim.set_actions(np.array([[[0.02, 3.0]]]))
    sim.set_actions(None)   # kilobot.py:240-241
    sim.step(1)
    assert [sim.cmd_vx[0, 0], sim.cmd_vy[0, 0], sim.cmd_w[0, 0]] == g['none_action']['vel']
    assert g['density'] == 2.0


def test_a4_acceleration_control(golden):
    for case in golden['a4_accel']['cases']:
        sim = one_bot(O.DRIVE_ACCEL, theta=case['theta'], pos=(0.3, 0.1))
        sim.v[...] = case['v0'][0]
        sim.w[...] = case['v0'][1]
        th = case['theta']
        for st in case['steps']:
            sim.theta[...] = th          # the golden run has no world.Step, so the pose never moves
            sim.x[...] = np.float32(0.3 * 25)
            sim.y[...] = np.float32(0.1 * 25)
            sim.set_actions(np.array([[st['action']]]))
            np.testing.assert_allclose([sim.acc_v[0, 0], sim.acc_w[0, 0]], st['clamped'], rtol=1e-6)
            sim.step(1)
            np.testing.assert_allclose([sim.v[0, 0], sim.w[0, 0]], st['velocity'], rtol=2e-6, atol=1e-9)
            np.testing.assert_allclose([sim.cmd_vx[0, 0], sim.cmd_vy[0, 0], sim.cmd_w[0, 0]], st['vel'],
                                       rtol=3e-6, atol=1e-7)


def test_a7_circular_light_and_a5_simple_phototaxis(golden):
    g = golden['a7_light']['circular']
    pts = np.array(g['points'])
    sim = O.OracleSim(O.default_config(1, len(pts), O.DRIVE_SIMPLE_PHOTOTAXIS, O.LIGHT_CIRCULAR,
                                       light_radius=g['radius']))
    sim.set_poses_m(pts[None], np.zeros((1, len(pts))))
    sim.light_x[...] = g['position'][0]
    sim.light_y[...] = g['position'][1]
    sim.step(1)
    np.testing.assert_allclose(sim.light_value[0], g['values'], rtol=1e-5, atol=2e-4)
    np.testing.assert_allclose(np.stack([sim.light_gx[0], sim.light_gy[0]], -1), g['gradients'], atol=2e-6)
    # a5: commanded velocity = gradient capped to 0.01 m/s, x25
    g5 = golden['a5_simple_phototaxis']
    for case in g5['cases']:
        sim = one_bot(O.DRIVE_SIMPLE_PHOTOTAXIS, O.LIGHT_CIRCULAR, pos=case['position'], theta=0.7,
                      light_radius=g5['light']['radius'])
        sim.step(1)
        np.testing.assert_allclose([sim.cmd_vx[0, 0], sim.cmd_vy[0, 0]], case['vel'], atol=1e-6)
        np.testing.assert_allclose(sim.light_value[0, 0], case['value'], rtol=1e-5, atol=2e-4)
        assert case['linear_damping_after'] == 0.0


def test_a7_sensor_position_and_light_step(golden):
    for case in golden['a7_sensor_pos']:
        # light exactly 0.05 m right of the expected sensor point -> gradient must be (+1, 0), value 255*(1-0.25)
        sim = one_bot(O.DRIVE_MOTORS, O.LIGHT_CIRCULAR, pos=case['position'], theta=case['theta'])
        sim.light_x[...] = case['sensor'][0] + 0.05
        sim.light_y[...] = case['sensor'][1]
        sim.step(1)
        np.testing.assert_allclose([sim.light_gx[0, 0], sim.light_gy[0, 0]], [1.0, 0.0], atol=2e-5)
        np.testing.assert_allclose(sim.light_value[0, 0], 255 * 0.75, rtol=2e-5)
    g = golden['a7_light']['circular_step']
    sim = one_bot(O.DRIVE_MOTORS, O.LIGHT_CIRCULAR, light_lo=g['bounds'][0], light_hi=g['bounds'][1],
                  light_act_lo=g['action_bounds'][0], light_act_hi=g['action_bounds'][1])
    sim.light_x[...] = g['start'][0]
    sim.light_y[...] = g['start'][1]
    for st in g['steps']:
        sim.step(1, light_action=np.array([st['action']]))
        np.testing.assert_allclose([sim.light_x[0], sim.light_y[0]], st['position'], rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize('name', ['rising_then_flat', 'random', 'zeros', 'falling'])
def test_a6_phototaxis_state_machine(golden, name):
    g = golden['a6_phototaxis'][name]
    # drive the state machine through the light value: put the light so that the sensed value equals v
    sim = one_bot(O.DRIVE_PHOTOTAXIS, O.LIGHT_CIRCULAR, light_radius=1.0)
    for v, motors in zip(g['values'], g['motors']):
        sim.set_poses_m(np.array([[FAR]]), np.array([[0.0]]))
        # sensor sits at (0, -r); value = 255 (1 - d/R)  ->  d = (1 - v/255) R, light placed to the right
        d = (1.0 - v / 255.0) * 1.0
        sim.light_x[...] = d
        sim.light_y[...] = -0.0165
        sim.step(1)
        assert [int(sim.motor_l[0, 0]), int(sim.motor_r[0, 0])] == motors


# ---- closed-form known-answer tests, SURVEY.md 8c (contact-free world step) -------------------
def test_kat_pivot_right():
    sim = one_bot(O.DRIVE_MOTORS)
    sim.motor_l[...] = 0
    sim.motor_r[...] = 255
    exp = {1: (-0.00145182, 0.00178041, 0.14544410), 2: (-0.00314634, 0.00333161, 0.29088821),
           10: (-0.02045543, 0.00483729, 1.45444104)}
    for k in range(1, 11):
        sim.step(1)
        if k in exp:
            np.testing.assert_allclose(sim.poses_m()[0, 0], exp[k], rtol=3e-5, atol=3e-8)


def test_kat_velocity_control():
    sim = one_bot(O.DRIVE_VELOCITY, theta=0.5)
    sim.set_actions(np.array([[[0.008, 0.3]]]))
    sim.step(1)
    np.testing.assert_allclose(sim.poses_m()[0, 0], (0.00065006, 0.00035513, 0.52777778), rtol=2e-5)
    sim.step(9)
    np.testing.assert_allclose(sim.poses_m()[0, 0], (0.00598803, 0.00432027, 0.77777778), rtol=2e-5)


def test_kat_simple_phototaxis_moves_1mm_per_substep():
    sim = one_bot(O.DRIVE_SIMPLE_PHOTOTAXIS, O.LIGHT_CIRCULAR, pos=(0.15, 0.0))
    for k in range(1, 6):
        sim.step(1)
        np.testing.assert_allclose(sim.poses_m()[0, 0, :2], (0.15 - 0.001 * k, 0.0), atol=2e-7)


# ---- the other light models: GradientLight, MomentumLight, CompositeLight ---------------
def test_f2_momentum_light_steps(golden):
    g = golden['a7_light']['momentum']
    sim = one_bot(O.DRIVE_MOTORS, O.LIGHT_MOMENTUM, light_lo=g['bounds'][0], light_hi=g['bounds'][1],
                  light_max_velocity=g['max_velocity'])
    sim.light_x[...], sim.light_y[...] = g['start'][0], g['start'][1]
    sim.light_vx[...], sim.light_vy[...] = g['start'][2], g['start'][3]
    for st in g['steps']:
        sim.step(1, light_action=np.array([st['action']]))
        got = [sim.light_x[0], sim.light_y[0], sim.light_vx[0], sim.light_vy[0]]
        np.testing.assert_allclose(got, st['state'], rtol=2e-6, atol=1e-9)


def test_f2_gradient_light_steps(golden):
    g = golden['a7_light']['gradient']
    sim = one_bot(O.DRIVE_SIMPLE_PHOTOTAXIS, O.LIGHT_GRADIENT, pos=(0.2, -0.1))
    sim.light_x[...] = g['start_angle']
    for st in g['steps']:
        sim.step(1, light_action=np.array([[st['action']]]))
        np.testing.assert_allclose(sim.light_x[0], st['angle'], atol=3e-7)
        np.testing.assert_allclose([sim.light_gx[0, 0], sim.light_gy[0, 0]], st['gradient'], atol=3e-7)
        # value = projection of the sensor position on the gradient direction (intent of light.py:255-257)
        np.testing.assert_allclose(sim.light_value[0, 0], 0.2 * st['gradient'][0] - 0.1 * st['gradient'][1], atol=1e-5)
        sim.set_poses_m(np.array([[(0.2, -0.1)]]), np.array([[0.0]]))
    assert golden['a7_light']['single_position']['n24_error'] == 'ValueError'   # reference bug recorded, not reproduced


def test_f2_composite_light(golden):
    g = golden['a7_light']['composite']
    pts = np.array(g['points'])
    n = len(g['lights'])
    sim = O.OracleSim(O.default_config(1, len(pts), O.DRIVE_SIMPLE_PHOTOTAXIS, O.LIGHT_COMPOSITE, light_count=n,
                                       light_kind=[O.LIGHT_CIRCULAR] * n,
                                       lightc_radius=[l['radius'] for l in g['lights']]))
    sim.set_poses_m(pts[None], np.zeros((1, len(pts))))
    for i, l in enumerate(g['lights']):
        sim.light_x[0, i], sim.light_y[0, i] = l['position']
    sim.step(1)
    np.testing.assert_allclose(sim.light_value[0], g['values'], rtol=1e-5, atol=3e-4)
    np.testing.assert_allclose(np.stack([sim.light_gx[0], sim.light_gy[0]], -1), g['gradients'], atol=3e-6)
    # composite action = concatenated component actions (light.py:122-127)
    sim.step(1, light_action=np.array([[0.01, 0.0, 0.0, -0.02]]))
    np.testing.assert_allclose(sim.light_x[0], [g['lights'][0]['position'][0] + 0.001, g['lights'][1]['position'][0]], atol=1e-7)
    np.testing.assert_allclose(sim.light_y[0], [g['lights'][0]['position'][1], g['lights'][1]['position'][1] - 0.001], atol=1e-7)


def test_oracle_semantics_regression():
    """The oracle's own end states for three small scenes (tests/golden/oracle_regression.npz, written by
    tools/gen_oracle_regression.py).  Not reference data: it makes a change of the specification a deliberate act."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('gen_oracle_regression', os.path.join(root, 'tools', 'gen_oracle_regression.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = np.load(os.path.join(root, 'tests', 'golden', 'oracle_regression.npz'))
    got = mod.compute()
    assert sorted(want.files) == sorted(got)
    for k in want.files:
        assert np.array_equal(want[k], got[k]), k
