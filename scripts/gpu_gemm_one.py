"""One stand-alone product shape (default 5000^2, B = 256) launched a few times: the target of counter passes on the
product kernels.  usage: python scripts/gpu_gemm_one.py [n] [B] [launches]"""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch  # noqa: E402
from vega_amd import VegaInterface  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 5
vega = VegaInterface('configs/auto/main.ini', search_dirs=[REPO / 'tests' / 'golden'], max_batch=1)
eng = vega.engine
dev = torch.device('cuda', 0)
ld = (n + 31) // 32 * 32
A = torch.zeros(n, ld, dtype=torch.float64, device=dev)
A[:, :n] = torch.rand(n, n, dtype=torch.float64, device=dev) - 0.5
x = torch.zeros(B, ld, dtype=torch.float64, device=dev)
x[:, :n] = torch.rand(B, n, dtype=torch.float64, device=dev) - 0.5
y = torch.zeros(B, ld, dtype=torch.float64, device=dev)
for _ in range(launches):
    eng.matvec_device(A.data_ptr(), n, ld, x.data_ptr(), B, y.data_ptr())
eng.sync()
ref = x[:, :n] @ A[:, :n].T
print('err', float((y[:, :n] - ref).abs().max() / ref.abs().max()))
vega.close()
