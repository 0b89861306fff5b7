# coding=utf-8
"""The ONE JSON line bench.py prints must stay small enough for the driver to parse (round 4's 22.5 kB line was not; round
3's 11.9 kB line was) and must carry the contract's keys at top level.  compact_line() is a pure function of the full
record, so this runs without a GPU: the canned record is round 4's own complete output (tests/golden/bench_full_r04.json)."""
import json
import os

import bench

HERE = os.path.dirname(os.path.abspath(__file__))


def canned():
    with open(os.path.join(HERE, 'golden', 'bench_full_r04.json')) as f:
        return json.loads(f.readline())


def test_single_gpu_line_is_small_and_complete():
    full = canned()
    assert len(json.dumps(full)) > 20000                   # the record that broke the driver's parser
    out = bench.compact_line(full, 'gpurun_out/bench_detail_n1.json')
    text = json.dumps(out)
    assert len(text) < bench.LINE_BUDGET < 12000
    assert '\n' not in text
    back = json.loads(text)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in back, k
    assert back['metric'] == full['metric'] and back['n_gpus'] == 1 and back['steps'] == full['steps']
    assert abs(back['value'] / full['value'] - 1) < 1e-5 and abs(back['ms_per_step'] / full['ms_per_step'] - 1) < 1e-5
    assert isinstance(back['config']['workload'], str) and 'model' not in back['config']
    rf = back['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in rf, k
    assert rf['bound'] == 'hbm' and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-4
    cb = back['cpu_baseline']
    for k in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert k in cb, k
    assert cb['cores'] == 1 and cb['kind'] == 'port'
    # no per-kernel tables on the line
    assert 'traffic_per_kernel_bytes' not in text
    # the clustered+phased reading of the metric stays on the line, with its own roofline and counter traffic
    assert back['value_clustered_and_phased'] > 0 and back['roofline_clustered_and_phased']['traffic'] > 0
    # round 6: what `value` is stays on the line; the roofline's launch time in both readings when the record has them
    full['value_is'] = bench.VALUE_IS
    full['roofline'].update({'launch_ms_in_run': 0.008, 'launch_ms_profiles': 0.0097, 'frac_in_run': 0.24})
    back = json.loads(json.dumps(bench.compact_line(full, None)))
    assert back['value_is'].startswith('phased_only') and back['roofline']['launch_ms_profiles'] == 0.0097 and back['roofline']['frac_in_run'] == 0.24


def test_multi_gpu_line_is_small():
    full = canned()
    world = 8
    full['n_gpus'] = world
    full['per_rank'] = [{'rank': r, 'contigs': list(range(3)), 'marks': 2500000 + r, 'candidates': 250000, 'reads': 400000,
                         'ef_classify_ms': 0.0101 + r * 1e-4, 'ef_classify_launches_timed': 25, 'ef_classify_algorithmic_bytes': 38000000,
                         'ef_classify_GBs': 3700.123456, 'ef_classify_frac_of_8TBs': 0.46251234,
                         'kernels_us_isolated': {'ef_classify': 10.1, 'ef_seed_sort': 6.2, 'ef_finalize': 5.9},
                         'kernels_only_ms_per_step': 0.0251234, 'gather_only_us': 31.4} for r in range(world)]
    full['topology'] = {'backend': 'duet_comm (RCCL)', 'world_size': world, 'rccl_version': '2.26.6', 'one_gpu_plumbing_mode': False,
                        'distinct_devices': world,
                        'ranks': [{'rank': r, 'device': r, 'name': 'AMD Instinct MI355X', 'cus': 256, 'hbm_GiB': 287.98,
                                   'pci_bus_id': '0000:%02x:00.0' % (5 + r), 'pid': 1000 + r} for r in range(world)]}
    full['sharding'] = {'lpt_imbalance_max_over_mean_marks': 1.0123, 'candidates_per_rank': [250000] * world, 'record_bytes_per_rank': 1250000}
    full['gather'] = {'collectives_per_problem': 1, 'bytes_contributed_per_rank': 1250000, 'us_isolated_max_over_ranks': 33.0,
                      'kernels_only_ms_per_step_max_over_ranks': 0.026}
    full['same_problem_on_1_gpu'] = {'ms_per_step': 0.108, 'marks_per_s': 1.85e11, 'parity_vs_oracle': True, 'note': 'x' * 300}
    out = bench.compact_line(full, None)
    text = json.dumps(out)
    assert len(text) < bench.LINE_BUDGET
    back = json.loads(text)
    assert back['topology']['distinct_devices'] == world and len(back['topology']['devices']) == world
    # round 6: what RCCL itself says about the communicator travels to the line's top level
    full['topology']['rccl_ranks_seen'] = world
    full['topology']['comm_selftest'] = 'ok'
    back2 = json.loads(json.dumps(bench.compact_line(full, None)))
    assert back2['rccl_ranks_seen'] == world and back2['topology']['rccl_ranks_seen'] == world
    assert len(back['per_rank']['marks']) == world


def test_line_sheds_blocks_rather_than_growing():
    full = canned()
    full['extra'] = {'point_%d' % i: {'marks': i, 'ms_per_run': 1.5, 'marks_per_s': 1e9, 'kernels_ms': {'k%d' % j: 0.1 for j in range(20)}}
                     for i in range(60)}
    out = bench.compact_line(full, None)
    assert len(json.dumps(out)) < bench.LINE_BUDGET
    assert 'roofline' in out and 'cpu_baseline' in out and 'value' in out
