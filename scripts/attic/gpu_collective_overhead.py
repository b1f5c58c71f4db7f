"""Per-step cost of the chi2 all_gather in bench.py's step loop (world size 1, RCCL): variants of stream placement."""
import os, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402
from vega_amd import VegaInterface, synthetic  # noqa: E402

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
B = 256
prob = bench.build_problem('joint')
vega = VegaInterface(None, problem=prob, max_batch=B)
eng = vega.engine
eng.set_constant_nl_hint(True)
pools = [torch.from_numpy(synthetic.walkers(eng.low.theta0, eng.names, B, varied=bench.VARIED, seed=100 + i)).to(dev) for i in range(8)]
bufs = [torch.zeros(B, dtype=torch.float64, device=dev) for _ in range(2)]
gath = [torch.zeros(B, dtype=torch.float64, device=dev) for _ in range(2)]
eng_stream = torch.cuda.ExternalStream(eng.stream_handle(), device=dev)
comm = torch.cuda.Stream(device=dev)
done = [None, None]

def step(i, mode):
    s = i % 2
    if mode in ('side', 'side_noop') and done[s] is not None:
        eng_stream.wait_event(done[s])
    eng.eval_device(pools[i % 8].data_ptr(), B, bufs[s].data_ptr())
    if mode == 'inline':
        with torch.cuda.stream(eng_stream):
            dist.all_gather_into_tensor(gath[s], bufs[s])
    elif mode == 'inline_async':
        with torch.cuda.stream(eng_stream):
            done[s] = dist.all_gather_into_tensor(gath[s], bufs[s], async_op=True)
    elif mode in ('side', 'side_noop'):
        comm.wait_event(eng_stream.record_event())
        with torch.cuda.stream(comm):
            if mode == 'side':
                dist.all_gather_into_tensor(gath[s], bufs[s])
            done[s] = comm.record_event()

for mode in ('none', 'inline', 'side', 'side_noop', 'none', 'inline'):
    done[:] = [None, None]
    for i in range(6):
        step(i, mode)
    eng.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(60):
        step(i, mode)
    eng.sync(); torch.cuda.synchronize()
    print(f'{mode}: {(time.perf_counter() - t0) / 60 * 1e3:.4f} ms/step', flush=True)
vega.close()
dist.destroy_process_group()
