"""What the per-step collective of `bench.py --gpus N` costs on ONE GPU (an RCCL group of one rank), piece by piece:
  plain      two batches in flight, no collective
  events     + the event on the lane's stream and the wait on the communication stream (no collective)
  copy       + a 2 KB device copy on the communication stream in place of the collective
  gather     + all_gather_into_tensor (RCCL)
  lane       the collective enqueued on the LANE's own stream, behind the evaluation: no event, no second stream
Regions of 40 steps, median of 7, evaluations / s.
"""
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def main():
    import torch
    import torch.distributed as dist
    torch.cuda.init()
    from bench import build_problem, VARIED
    from vega_amd import VegaInterface, synthetic
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29541')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    B = 256
    vega = VegaInterface(None, problem=build_problem('joint'), max_batch=B, device=0)
    eng = vega.engine
    eng.set_constant_nl_hint(True, gaussian=True)
    eng.set_lanes(2)
    pools = [torch.from_numpy(synthetic.walkers(eng.low.theta0, eng.names, B, varied=VARIED, seed=synthetic.SEED + i)).to(dev)
             for i in range(8)]
    chi2 = [torch.zeros(B, dtype=torch.float64, device=dev) for _ in range(4)]
    gathered = [torch.zeros(B, dtype=torch.float64, device=dev) for _ in range(4)]
    comm = torch.cuda.Stream(device=dev)
    done = [None] * 4
    ext = {}

    def last_stream():
        h = eng.last_stream_handle()
        if h not in ext:
            ext[h] = torch.cuda.ExternalStream(h, device=dev)
        return ext[h]

    def step(i, mode):
        slot = i % 4
        if mode not in ('plain', 'lane') and done[slot] is not None:
            done[slot].synchronize()
        eng.eval_device(pools[i % 8].data_ptr(), B, chi2[slot].data_ptr())
        if mode == 'plain':
            return
        if mode == 'lane':
            with torch.cuda.stream(last_stream()):
                dist.all_gather_into_tensor(gathered[slot], chi2[slot])
            return
        comm.wait_event(last_stream().record_event())
        with torch.cuda.stream(comm):
            if mode == 'copy':
                gathered[slot].copy_(chi2[slot], non_blocking=True)
            elif mode == 'gather':
                dist.all_gather_into_tensor(gathered[slot], chi2[slot])
            done[slot] = comm.record_event()

    for i in range(400):
        step(i, 'plain')
    eng.sync()
    for rnd in range(2):
        for mode in ('plain', 'events', 'copy', 'gather', 'lane'):
            for i in range(40):
                step(i, mode)
            eng.sync(); torch.cuda.synchronize()
            rates = []
            for _ in range(7):
                t0 = time.perf_counter()
                for i in range(40):
                    step(i, mode)
                eng.sync(); torch.cuda.synchronize()
                rates.append(B * 40 / (time.perf_counter() - t0))
            # host time of the enqueue alone
            t0 = time.perf_counter()
            for i in range(40):
                step(i, mode)
            host = (time.perf_counter() - t0) / 40
            eng.sync(); torch.cuda.synchronize()
            print(f'{mode:7s} median {np.median(rates):9.0f} evals/s  min {min(rates):9.0f} max {max(rates):9.0f}   host enqueue {host * 1e6:6.1f} us / step',
                  flush=True)
    dist.destroy_process_group()
    vega.close()


if __name__ == '__main__':
    main()
