"""The ONE stdout line of bench.py must stay parseable by the driver (it keeps ~8 KB of stdout tail; round 4's 37 KB
line was recorded as `parsed: null`): bench.compact_line on recorded full records, N = 1 and N > 1."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RECORDED = [f for f in ('profiles/r04_d_bench.json', 'profiles/r04_a_bench.json', 'profiles/r03_bench.json')
            if os.path.exists(os.path.join(ROOT, f))]

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline')


def _full(name):
    with open(os.path.join(ROOT, name)) as f:
        return json.load(f)


@pytest.mark.parametrize('name', RECORDED)
def test_default_line_is_short_and_round_trips(name):
    import bench
    out = _full(name)
    text = bench.compact_line(out, 'bench_detail.json')
    assert '\n' not in text and len(text) < 4096, len(text)
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line['value'] == pytest.approx(out['value'], rel=1e-5)
    assert line['ms_per_step'] == pytest.approx(out['ms_per_step'], rel=1e-4)
    assert line['config']['workload'] == out['config']['workload']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in line['roofline'], k
    assert line['roofline']['frac'] == pytest.approx(out['roofline']['frac'], rel=1e-3)
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in line['cpu_baseline'], k
    assert line['parity']['rel_err'] <= line['parity']['tolerance']
    assert 'dropped' not in line                       # nothing had to be cut on a real record
    assert 'kernels' not in line and 'gemm_shapes' not in line


def test_multi_gpu_line_with_scaling_companions_is_short():
    import bench
    out = copy.deepcopy(_full(RECORDED[0]))
    out.update({'n_gpus': 8, 'global_batch': 8192, 'rccl_ranks_seen': 8, 'cpu_baseline': None, 'parity': None,
                'other_configs': None, 'value_f16x3': None, 'value_exact_f32': None})
    out['config']['parallelism'] = 'dp8'
    for sc, gb in (('scaling_strong', 1024), ('scaling_exact', 1024)):
        out[sc] = {'value': 1.234567e6, 'ms_per_step': 0.8299, 'global_batch': gb, 'batch_per_gpu': gb // 8,
                   'last_loss': 16.123456}
    text = bench.compact_line(out, 'bench_detail_n8.json')
    assert len(text) < 4096
    line = json.loads(text)
    assert line['n_gpus'] == 8 and line['config']['global_batch'] == 8192
    assert line['scaling_strong']['global_batch'] == 1024 and line['scaling_exact']['batch_per_gpu'] == 128
    assert line['cpu_baseline'] is None and line['roofline']['frac'] > 0


def test_oversized_optional_blocks_are_dropped_not_truncated():
    import bench
    out = copy.deepcopy(_full(RECORDED[0]))
    out['other_configs'] = {'cfg%d' % i: {'value': 1.0 * i, 'ms_per_step': 2.0, 'dtype': 'f32', 'gemm_mode': 'bf16x6',
                                           'roofline': {'frac': 0.3}, 'parity': {'rel_err': 1e-7, 'grad_rel_err': 1e-6}}
                            for i in range(60)}
    text = bench.compact_line(out, 'bench_detail.json')
    line = json.loads(text)
    assert len(text) < 4096 and line['dropped'] >= 1 and 'other_configs' not in line
    for k in CONTRACT:
        assert k in line


def test_oversized_mandatory_fields_are_shortened_and_a_plain_record_still_gives_a_line():
    """the line must come out even when the MANDATORY part alone is too long (free-text fields are shortened, never
    an AssertionError after the whole benchmark has run), and for a --plain record (no kernel table, no roofline, no CPU
    baseline, no parity)."""
    import bench
    out = copy.deepcopy(_full(RECORDED[0]))
    out['config']['workload'] = 'w' * 6000
    out['cpu_baseline'] = dict(out.get('cpu_baseline') or {'value': 1.0, 'unit': 'triples/s', 'cores': 1, 'kind': 'port'},
                               sample='s' * 3000)
    text = bench.compact_line(out, 'bench_detail.json')
    line = json.loads(text)
    assert len(text) < 4096 and line.get('truncated', 0) >= 1
    for k in CONTRACT:
        assert k in line
    plain = copy.deepcopy(_full(RECORDED[0]))
    plain.update({'roofline': None, 'cpu_baseline': None, 'parity': None, 'kernels': {}, 'gemm_shapes': [],
                  'roofline_rgcn_gather': {}, 'roofline_gru': None, 'kernel_only': {'ms_per_step': 0.0, 'glue_ms_per_step': 0.0},
                  'other_configs': None, 'value_f16x3': None, 'value_exact_f32': None, 'launches_per_step': None,
                  'value_median': {'value': 3.7e5, 'ms_per_step': 2.76, 'steps': 100, 'p10_ms': 2.7, 'p90_ms': 2.8},
                  'value_list_api': {'value': 2.2e5, 'ms_per_step': 4.6, 'steps': 30}, 'plan_entries_per_step': 61})
    line = json.loads(bench.compact_line(plain, 'bench_detail.json'))
    assert line['roofline'] is None and line['value_median']['steps'] == 100 and line['value_list_api']['steps'] == 30
    assert line['plan_entries_per_step'] == 61
    for k in CONTRACT:
        assert k in line
