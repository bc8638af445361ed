"""bench.py prints ONE stdout line the driver must be able to parse: < 8000 bytes, valid JSON, with the contract's keys,
`roofline` and `cpu_baseline` (the round-5 line was 22.7 KB and was not parsed).  The formatter is run on the canned full
record of round 5 (profiles/r05_w_bench.json) and on an inflated one."""
import copy
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def canned():
    with open(os.path.join(ROOT, 'profiles', 'r05_w_bench.json')) as f:
        return json.load(f)


def _check(line, full):
    assert len(line) < 8000, len(line)
    assert '\n' not in line
    got = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in got, k
    assert got['value'] == pytest.approx(full['value'], rel=1e-5)
    assert got['config']['workload'] == full['config']['workload'] or got['config']['workload'].endswith('...')
    rf = got['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert rf['frac'] == pytest.approx(rf['achieved'] / rf['peak'], rel=1e-4)
    assert rf['hbm_nominal']['frac'] == pytest.approx(rf['hbm_nominal']['achieved'] / rf['hbm_nominal']['peak'], rel=1e-4)
    cb = got['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in cb, k
    return got


def test_the_round_5_record_fits_one_parseable_line(canned):
    got = _check(bench.compact_line(canned, full_record='bench_legs.json'), canned)
    # one scalar group per extra leg
    for leg in ('config3_4096', 'config5_8192', 'config4_grasp_2048', 'lockstep_env_step', 'async_rollout'):
        assert got['legs'][leg]['value'] == pytest.approx(canned[leg]['value'], rel=1e-3)
    assert got['legs']['reference_semantics']['gpu']['early_exit_effort_limited_motor']['value'] > 0
    assert got['full_record'] == 'bench_legs.json'


def test_an_inflated_record_still_fits(canned):
    big = copy.deepcopy(canned)
    for i in range(200):
        big['extra_leg_%d' % i] = {'value': 1.0 * i, 'unit': 'env_steps/s', 'note': 'x' * 500, 'sim_steps_per_s': 3.0}
    big['config']['mode'] = 'y' * 5000
    got = _check(bench.compact_line(big), big)
    assert 'dropped_for_size' in got


def test_the_multi_gpu_line_carries_its_like_for_like_reference(canned):
    multi = copy.deepcopy(canned)
    multi.update({'n_gpus': 8, 'n1_same_workload': 150000.0, 'scaling_efficiency_vs_n1_same_workload': multi['value'] / (8 * 150000.0)})
    got = _check(bench.compact_line(multi), multi)
    assert got['n1_same_workload'] == 150000.0
    assert 'scaling_efficiency_vs_n1_same_workload' in got


def test_no_constant_claims_about_pybullet_in_the_bench():
    """whether pybullet can be imported is asked of the machine (tools/pybullet_parity.probe), never asserted by a string"""
    with open(os.path.join(ROOT, 'bench.py')) as f:
        src = f.read()
    for phrase in ('pybullet not importable', 'module not available'):
        assert phrase not in src, phrase
    assert 'pybullet_probe()' in src
