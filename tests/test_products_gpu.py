"""The stand-alone product D^T[n][m] = sum_k A[m][k] X[n][k] (`vmx_matvec_device`: the kernels the distortion, metal
and FFTLog steps use) on ragged shapes - matrix rows and walker counts that are not multiples of the 64 x 64 block tile,
K tails, every batch-size regime (streaming kernels for B <= 8, MFMA kernels above) and forced split-K factors - against
a plain fp64 torch product of the same operands.  Tolerance: 1e-13 of the result's scale (fp64 sums of ~1e3 terms in a
different order)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

SHAPES = [(70, 96), (257, 300), (1000, 1000), (2500, 2500), (5000, 5000)]
BATCHES = [1, 3, 8, 9, 37, 64, 130, 256]


def _engine(max_batch=1):
    from vega_amd import VegaInterface
    return VegaInterface('configs/auto/main.ini', search_dirs=[GOLDEN], max_batch=max_batch)


def _check(eng, torch, m, k, batch, seed):
    dev = torch.device('cuda', 0)
    gen = torch.Generator(device=dev).manual_seed(seed)
    ld = (k + 31) // 32 * 32                       # operands are zero-padded to a multiple of 32 columns
    a = torch.zeros(m, ld, dtype=torch.float64, device=dev)
    a[:, :k] = torch.rand(m, k, dtype=torch.float64, device=dev, generator=gen) - 0.5
    x = torch.zeros(batch, ld, dtype=torch.float64, device=dev)
    x[:, :k] = torch.rand(batch, k, dtype=torch.float64, device=dev, generator=gen) - 0.5
    ldy = (m + 31) // 32 * 32
    y = torch.full((batch, ldy), float('nan'), dtype=torch.float64, device=dev)
    eng.matvec_device(a.data_ptr(), m, ld, x.data_ptr(), batch, y.data_ptr())
    eng.sync()
    ref = x[:, :k] @ a[:, :k].T
    err = float((y[:, :m] - ref).abs().max() / ref.abs().max())
    assert err <= 1e-13, f'M={m} K={k} B={batch}: scaled error {err:.2e}'


def test_ragged_products_every_batch_regime():
    import torch
    vega = _engine()
    for i, (m, k) in enumerate(SHAPES):
        for batch in BATCHES:
            _check(vega.engine, torch, m, k, batch, seed=17 * i + batch)
    vega.close()
