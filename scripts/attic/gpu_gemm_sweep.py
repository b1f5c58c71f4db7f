"""Distortion-style product y[b] = A x[b] over walker-batch sizes and shapes, through vmx_matvec_device
(the kernels the evaluation uses): time per launch, TFLOP/s (fp64 MFMA peak 78.6) and GB/s of algorithmic bytes.
usage: python scripts/gpu_gemm_sweep.py [B ...]"""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch  # noqa: E402
from vega_amd import VegaInterface  # noqa: E402

batches = [int(a) for a in sys.argv[1:]] or [16, 64, 256]
vega = VegaInterface('configs/auto/main.ini', search_dirs=[REPO / 'tests' / 'golden'], max_batch=1)
eng = vega.engine
dev = torch.device('cuda', 0)
for n in (2500, 5000, 1590, 3180):
    ld = (n + 31) // 32 * 32
    A = torch.zeros(n, ld, dtype=torch.float64, device=dev)
    A[:, :n] = torch.rand(n, n, dtype=torch.float64, device=dev) - 0.5
    for B in batches:
        x = torch.zeros(B, ld, dtype=torch.float64, device=dev)
        x[:, :n] = torch.rand(B, n, dtype=torch.float64, device=dev) - 0.5
        y = torch.zeros(B, ld, dtype=torch.float64, device=dev)
        for _ in range(3):
            eng.matvec_device(A.data_ptr(), n, ld, x.data_ptr(), B, y.data_ptr())
        eng.sync()
        ref = x[:, :n] @ A[:, :n].T
        err = float((y[:, :n] - ref).abs().max() / ref.abs().max())
        eng.timings(reset=True)
        eng.set_profiling(True)
        for _ in range(20):
            eng.matvec_device(A.data_ptr(), n, ld, x.data_ptr(), B, y.data_ptr())
        eng.sync()
        ms, launches = eng.timings(reset=True)['matvec_api']
        eng.set_profiling(False)
        t = ms / launches * 1e-3
        print(f'n={n} B={B}: {t * 1e6:8.1f} us  {2.0 * n * n * B / t / 1e12:6.2f} TF  '
              f'{8.0 * (n * n + 2 * B * n) / t / 1e9:8.1f} GB/s  err={err:.1e}', flush=True)
vega.close()
