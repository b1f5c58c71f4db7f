import sys, json
sys.path.insert(0, '.')
import torch
torch.cuda.init()
import bench
print(json.dumps(bench.distortion_csr(0), indent=1))
