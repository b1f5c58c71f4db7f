import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
GOLDEN = REPO / 'tests' / 'golden'
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # torch bundles its own HIP runtime: when a test uses both torch device tensors and libvegamx.so, torch must
    # bring the runtime up first (as bench.py does), otherwise it finds no device in a process that already loaded
    # the system runtime through the engine.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:       # pragma: no cover - CPU-only container
        pass


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


_PROBLEMS = {}


def load_problem(name, **kw):
    """Cached Problem for tests/golden/configs/<name>/main.ini (or an explicit main file)."""
    from vega_amd.setup import build_problem
    key = (name, tuple(sorted(kw.items())))
    if key not in _PROBLEMS:
        main = name if name.endswith('.ini') else f'configs/{name}/main.ini'
        _PROBLEMS[key] = build_problem(main, search_dirs=[GOLDEN], **kw)
    return _PROBLEMS[key]


@pytest.fixture(scope='session')
def problem_loader():
    return load_problem
