"""CPU checks of the drop-in boundary: the C-ABI library builds, loads, exports every symbol that
include/vegamx.h declares, agrees with the ctypes struct layouts, and fails loudly without a GPU."""
import ctypes as C
import re

import pytest

from conftest import REPO, load_problem


@pytest.fixture(scope='module')
def lib():
    import __graft_entry__ as g
    g.build()
    from vega_amd import engine
    return engine.load_library()


def test_every_declared_symbol_is_exported(lib):
    header = (REPO / 'include' / 'vegamx.h').read_text()
    declared = set(re.findall(r'\b(vmx_[a-z_]+)\s*\(', header))
    assert len(declared) >= 25
    for sym in declared:
        assert hasattr(lib, sym), sym
    from vega_amd import engine
    assert declared == set(engine.EXPORTED_SYMBOLS)


def test_struct_layouts_match(lib):
    from vega_amd import engine
    for which, struct in enumerate((engine.Tracer, engine.PipeDesc, engine.MetalDesc, engine.ItemDesc)):
        assert lib.vmx_struct_size(which) == C.sizeof(struct)


def test_no_gpu_is_a_loud_error(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    handle = C.c_void_p()
    rc = lib.vmx_create(C.byref(handle), 0)
    assert rc < 0
    assert b'device' in lib.vmx_last_error().lower()
    from vega_amd import VegaInterface
    from vega_amd.engine import EngineError
    with pytest.raises(EngineError):
        VegaInterface(None, problem=load_problem('auto'), max_batch=1)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under vega_amd/ may import it."""
    for path in (REPO / 'vega_amd').rglob('*.py'):
        text = path.read_text()
        assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), path
    for path in (REPO / 'vega_amd' / 'csrc').iterdir():
        assert 'oracle' not in path.read_text().lower(), path
