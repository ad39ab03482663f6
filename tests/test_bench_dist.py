"""bench.py's N > 1 control flow under gloo, without a GPU (`--dist-backend gloo --stub-step-ms`): every branch that only runs with
more than one rank -- process-group init under torch.distributed.run, the rank-count all-reduce, barrier + MAX-over-ranks timing,
the per-rank all_gather, the train leg's all-reduce record, and the propagation of a parity-gate failure on rank 0 to every
rank -- executes here, so that the first multi-GPU lease of the driver does not debug them (the reference itself
refuses multi-GPU, trainer.py:76-81)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, extra, timeout=240):
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', str(world), '--steps', '4', '--warmup', '1',
           '--dist-backend', 'gloo', '--stub-step-ms', '20', '--train-steps', '2', '--train-warmup', '1'] + extra
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_two_ranks_print_one_line_with_the_multi_rank_records():
    r = _run(2, [])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                       # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2 and d['dist_backend'] == 'gloo'
    assert d['scaling'] == 'weak' and d['config']['global_batch'] == 64
    assert len(d['per_rank_scenes_per_s']) == 2 and all(v > 0 for v in d['per_rank_scenes_per_s'])
    # whole-job value = scenes of all ranks / the slowest rank's time: 2 x 32 scenes per ~20 ms step
    assert 0.5 * 64 / 0.021 < d['value'] < 64 / 0.020
    assert abs(d['value'] - 64 / (d['ms_per_step'] * 1e-3)) < 1e-3 * d['value']
    assert d['roofline']['frac'] <= 1.0 and d['roofline']['bound'] == 'mfma'
    t = d['train_step']
    assert t['n_gpus'] == 2 and t['global_batch'] == 16
    assert t['allreduce']['bytes'] == 4 * (4659307 + 1) and t['allreduce']['ms'] > 0 and t['allreduce']['GBps_bus'] > 0
    # the N = 1 only legs stay out of a multi-rank line
    assert 'cpu_baseline' not in d and 'with_backbones' not in d and 'f32_mfma_build' not in d


def test_a_failed_parity_gate_on_rank_0_stops_every_rank_before_timing():
    r = _run(2, ['--stub-parity-fail'])
    assert r.returncode != 0
    assert 'parity gate FAILED' in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]      # no bench line


def test_gpus_flag_must_match_the_launch():
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--stub-step-ms', '1'], capture_output=True, text=True,
                       timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and 'torch.distributed.run' in (r.stderr + r.stdout)


def test_recorded_arbiter_readers_survive_every_file_their_patterns_match():
    """bench.py quotes the float64 arbiters from the newest committed profiles/*.json its glob patterns match; a file of another layout
    under a matching name (the parity log of the backward arbiter matched both patterns in round 6 and broke the line) must be passed
    over, not raised on."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_under_test', os.path.join(ROOT, 'bench.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    fwd, bwd = m.recorded_arbiter(), m.recorded_bwd_arbiter()
    assert fwd is not None and fwd['rms_ratio_max'] <= 1.5 and fwd['p99_ratio_max'] <= 2.0
    assert bwd is not None and any(isinstance(v, dict) and 'median_rms_ratio' in v for v in bwd.values())
