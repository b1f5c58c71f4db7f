"""The library's host-side planners under sanitizers, on the CPU (no GPU, no HIP): the tape cut of the quadratic-form launch
(bisection, skew, slot numbering, block queues), the CSR checks and the host Cholesky factor live in
vega_amd/csrc/vmx_plan.h - plain C++ that decides determinism and correctness before any kernel runs.
tests/helpers/planner_driver.cpp is built with g++ -fsanitize=address,undefined and asserts, for B in {9, 64, 256, 512, 4096},
several problem sets and block counts: every (item, row tile, walker tile) K range covered exactly once, pieces of equal
cost to within one entry, slots in tape order, lock-step group members identical, identical output for identical inputs."""
import shutil
import subprocess

import pytest

from conftest import REPO


@pytest.fixture(scope='module')
def driver(tmp_path_factory):
    gxx = shutil.which('g++')
    if gxx is None:
        pytest.skip('g++ is not installed')
    exe = tmp_path_factory.mktemp('planner') / 'planner_driver'
    cmd = [gxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-Wall', '-Wextra',
           '-o', str(exe), str(REPO / 'tests' / 'helpers' / 'planner_driver.cpp')]
    built = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert built.returncode == 0, built.stderr[-4000:]
    assert 'warning' not in built.stderr, built.stderr[-4000:]
    return exe


def test_planners_under_address_and_ub_sanitizers(driver):
    run = subprocess.run([str(driver)], capture_output=True, text=True, timeout=600,
                         env={'ASAN_OPTIONS': 'detect_leaks=1', 'UBSAN_OPTIONS': 'print_stacktrace=1'})
    assert run.returncode == 0, (run.stdout[-3000:], run.stderr[-3000:])
    assert 'FAIL' not in run.stdout and 'runtime error' not in run.stderr and 'AddressSanitizer' not in run.stderr
    assert 'all planner checks passed' in run.stdout
    assert run.stdout.count('ok tape') == 7 * 5 * 4          # shapes x batch sizes x block counts


def test_the_library_uses_the_tested_planner():
    """libvegamx takes its tape, CSR check and Cholesky factor from the header the driver tests - no second copy."""
    hip = (REPO / 'vega_amd' / 'csrc' / 'vegamx.hip').read_text()
    dev = (REPO / 'vega_amd' / 'csrc' / 'vmx_device.h').read_text()
    assert '#include "vmx_plan.h"' in dev
    assert 'vmx_plan::plan_quad_tape(' in hip and 'vmx_plan::csr_problem(' in hip
    assert 'struct GemmWork' not in hip and 'struct GemmWork' not in dev
    plan = (REPO / 'vega_amd' / 'csrc' / 'vmx_plan.h').read_text()
    assert 'hip' not in plan.lower().replace('vegamx.hip', '').replace('no hip types', '')
