"""The one-line JSON contract of bench.py, checked on the committed line of the last measurement pass (no GPU needed): every
key the driver reads is there, the roofline figures follow from each other by the stated formula, and the rocprofv3
summary committed next to it agrees with the live launch time."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest(name):
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_%s' % name)))
    assert files, 'no profiles/r*_%s committed' % name
    return files[-1]


def test_driver_line_has_the_contract_keys_and_consistent_roofline():
    d = json.loads(open(_latest('bench_line_driver.json')).readline())
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['metric'] == 'kilobot-steps/sec' and d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert d['dtype'] == 'f32' and d['data'] == 'synthetic' and d['n_gpus'] == 1 and d['steps'] == 20 and d['warmup'] == 5
    assert 'workload' in d['config'] and '4096 envs x 1024 kilobots' in d['config']['workload'] and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    # achieved = algorithmic bytes per launch / average launch duration; frac = achieved / peak
    assert abs(r['achieved'] - r['algorithmic_bytes_per_launch'] / (r['avg_launch_ms'] * 1e-3) / 1e9) < 1e-6 * r['achieved']
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    assert r['algorithmic_bytes_per_launch'] == 48.0 * 4096 * 1024
    assert r['traffic'] is None or r['traffic'] > r['algorithmic_bytes_per_launch']       # HBM bytes from the PMC passes
    # value = kilobot-steps of the timed region / wall time, and the wall time per step is not shorter than the kernel
    assert abs(d['value'] - 4096 * 1024 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    assert d['ms_per_step'] >= 0.98 * r['avg_launch_ms']
    c = d['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in c, k
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0
    assert d['contacts_per_env'] > 450 and d['status_flags'] == 0 and d['returns_gathered'] == 4096


def test_rocprof_summary_agrees_with_the_live_launch_time():
    d = json.loads(open(_latest('bench_line.json')).readline())
    txt = open(_latest('bench_kernel_trace_summary.txt')).read()
    m = re.search(r'timed launches \(index \d+\.\.\d+\): avg ([\d.]+) us', txt)
    assert m, txt
    assert abs(float(m.group(1)) * 1e-3 - d['roofline']['avg_launch_ms']) < 0.05 * d['roofline']['avg_launch_ms']
    assert 'kb_step_kernel' in txt


def test_bench_line_reports_the_legs_next_to_the_headline():
    """with_sleep_state, fused_env_step and jammed_swarm ride along, none of them is the value."""
    d = json.loads(open(_latest('bench_line.json')).readline())
    for k in ('with_sleep_state', 'fused_env_step', 'jammed_swarm'):
        assert d.get(k), k
    j = d['jammed_swarm']
    assert j['contacts_per_env'] > 2000 and j['status_flags'] == 0          # one island, far beyond the LDS staging
    envs_j = int(j['workload'].split(' envs')[0])
    assert envs_j in (1024, 4096) and abs(j['kilobot_steps_per_s_one_gpu'] - envs_j * 1024 / (j['ms_per_launch'] * 1e-3)) < 1e-6 * j['kilobot_steps_per_s_one_gpu']
    assert j['kilobot_steps_per_s_one_gpu'] < d['value']                    # the slow corner, reported as such


def test_bench_line_reports_the_small_and_the_object_configuration():
    """cfg2 (the reference's own CPU-sized case) and cfg4 (cfg3 + pushable objects) are legs of the same line since round 3;
    each states its launch time, the rate that follows from it and the share of the HBM roofline, none of them is the value."""
    d = json.loads(open(_latest('bench_line.json')).readline())
    c2 = d['cfg2']
    assert '256 envs x 64 kilobots' in c2['workload'] and c2['status_flags'] == 0
    assert abs(c2['kilobot_steps_per_s_one_gpu'] - 256 * 64 / (c2['ms_per_launch'] * 1e-3)) < 1e-6 * c2['kilobot_steps_per_s_one_gpu']
    for leg in ('after_settle', 'after_500_substeps'):
        c4 = d['cfg4'][leg]
        assert c4['status_flags'] == 0 and c4['contacts_per_env'] > 450
        assert abs(c4['kilobot_steps_per_s_one_gpu'] - 4096 * 1024 / (c4['ms_per_launch'] * 1e-3)) < 1e-6 * c4['kilobot_steps_per_s_one_gpu']
        assert abs(c4['roofline_frac'] - 48.0 * c4['kilobot_steps_per_s_one_gpu'] / 8e12) < 1e-3 * c4['roofline_frac']
        assert c4['kilobot_steps_per_s_one_gpu'] < d['value']
    assert d['returns_expected'] == d['returns_gathered'] == 4096 and d['collective_world_size'] == 1
