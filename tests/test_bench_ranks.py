"""`bench.py --gpus N` starts its own N ranks (reference semantics: the ranks of bin/run_vega_mc_mpi.py:17-25,54-65).

CPU part: the parent is a process that makes no GPU call and relays failure - in a container without a GPU every rank
stops with the loud "needs a GPU" error and the launcher must come back non-zero, not hang.  GPU part: the whole N > 1 path
(event-ordered gather on its own stream, two batches in flight, barriers, max over ranks) rehearsed with two ranks on the
one GPU of the box over gloo.
"""
import json
import subprocess
import sys

import pytest

from conftest import REPO, run_programs


def _has_gpu():
    import torch
    return torch.cuda.device_count() > 0


def test_world_size_must_match_gpus():
    out = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--gpus', '2', '--core-only'], cwd=str(REPO), text=True,
                         env={**__import__('os').environ, 'WORLD_SIZE': '1', 'RANK': '0'}, capture_output=True, timeout=300)
    assert out.returncode != 0
    assert 'WORLD_SIZE = 1' in out.stderr and out.stdout == ''


def test_share_gpu_needs_gloo():
    out = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--gpus', '2', '--ranks-share-gpu'], cwd=str(REPO), text=True,
                         capture_output=True, timeout=300)
    assert out.returncode != 0 and '--dist-backend gloo' in out.stderr


@pytest.mark.skipif(_has_gpu(), reason='the failure path of a box without a GPU')
def test_launcher_reports_a_failed_rank_and_does_not_hang():
    out = subprocess.run([sys.executable, str(REPO / 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--ranks-share-gpu',
                          '--core-only', '--steps', '2'], cwd=str(REPO), text=True, capture_output=True, timeout=600)
    assert out.returncode != 0
    assert out.stdout == ''                                 # no JSON line from a failed job
    # (a rank says why it stopped; the launcher ends the other one as soon as the first has failed - whether that one got to say
    # it too is a matter of timing)
    assert 1 <= out.stderr.count('needs a GPU') <= 2 and 'exited with code 1' in out.stderr
    assert 'exited with code' in out.stderr


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_over_gloo():
    argv = [sys.executable, str(REPO / 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--ranks-share-gpu', '--core-only',
            '--steps', '4', '--warmup', '2', '--ramp-steps', '10']
    (rc, output), = run_programs([argv], [{'HSA_ENABLE_IPC_MODE_LEGACY': '0'}], timeout=900)
    assert rc == 0, output[-4000:]
    lines = [ln for ln in output.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, output[-4000:]
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 4 and line['scaling'] == 'weak'
    coll = line['config']['collective']
    assert coll != 'none'
    assert coll['backend'] == 'gloo' and coll['world_size'] == 2 and coll['ranks'] == [0, 1]
    assert coll['distinct_rank_blocks_in_last_gather'] == 2 and coll['ranks_share_gpu'] is True
    assert line['value'] > 0 and line['config']['batches_in_flight'] == 2
    # the line explains itself at N > 1: every rank's own time, the gather's own time, the slowest rank's kernels, the CPUs
    own = coll['rank_elapsed_s']
    assert len(own['all']) == 2 and own['min'] <= own['median'] <= own['max'] <= line['ms_per_step'] * 4e-3 * 1.001
    assert coll['gather_us_per_step']['mean_over_ranks'] > 0 and coll['gather_us_per_step']['max_over_ranks'] >= coll['gather_us_per_step']['mean_over_ranks']
    slow = coll['slowest_rank']
    assert slow['rank'] in (0, 1) and slow['elapsed_s'] == own['max'] and slow['kernel_ms_in_timed_region']
    assert all(v['launches'] > 0 and v['ms'] > 0 for v in slow['kernel_ms_in_timed_region'].values())
    assert len(coll['cpu_affinity']) == 2 and all('cpus_before' in a for a in coll['cpu_affinity'])
    # whole-job aggregate: both ranks' walkers over the slowest rank's time
    assert abs(line['value'] - 2 * line['config']['batch_per_gpu'] * 4 / (line['ms_per_step'] * 4e-3)) < 1e-6 * line['value']
