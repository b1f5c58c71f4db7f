"""Block timeline of the persistent quadratic-form launch (VMX_QUAD_TRACE) against the tape's own accounting: per block the K
stages and entries of its piece (the host's bisection, restated here), a least-squares fit duration = a stages + b entries + c,
and the spread of the blocks' end times (development aid for the `quad_overhead` constant)."""
import math, os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / 'tests'))
out = REPO / 'gpurun_out' / 'quad_trace.bin'
os.environ['VMX_QUAD_TRACE'] = str(out)
import numpy as np
from conftest import synth_joint_problem
from vega_amd import VegaInterface, synthetic
import bench
B = int(os.environ.get('PKB', '256'))
OVH = float(os.environ.get('VMX_QUAD_OVH', '4.0'))
prob = synth_joint_problem()
vega = VegaInterface(None, problem=prob, max_batch=B)
eng = vega.engine
theta = synthetic.walkers(eng.low.theta0, eng.names, B, seed=3, varied=bench.VARIED)
eng.set_profiling(True)          # (eager launches: the trace buffer is allocated at the launch)
for _ in range(6):
    eng.eval(theta)
eng.sync()
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 4).astype(np.float64)
live = t[:, 0] > 0
t0 = t[live, 0].min()
us = (t - t0) / 100.0

BM, BK, MIN_SEG = 64, 32, 2
nqs = [it.model_grid.size + 12 * (i == 0) for i, it in enumerate(prob.items.values())]     # (auto: + its broadband columns)
nqs = [int(os.environ.get('NQ0', nqs[0])), int(os.environ.get('NQ1', nqs[1]))]
ranges = []
for q, nq in enumerate(nqs):
    nq_pad = (nq + 31) // 32 * 32
    for mt in range((nq + BM - 1) // BM):
        ranges.append((q, mt, min(((mt + 1) * BM + BK - 1) // BK, nq_pad // BK)))
ranges.sort(key=lambda r: -r[2])
tn = (B + 63) // 64
gs = 4 if tn % 4 == 0 else 2 if tn % 2 == 0 else 1
n_groups, P = tn // gs, len(t)
n_pieces = P // gs


def fill(cap):
    pieces, pc, cur = [[0, 0]], 0, 0.0
    for r in ranges:
        for grp in range(n_groups):
            left = r[2]
            while left > 0:
                avail = cap - cur - OVH
                if cur > 0 and avail < min(left, MIN_SEG):
                    pc += 1; cur = 0.0; pieces.append([0, 0]); continue
                take = min(left, max(MIN_SEG, int(math.floor(avail + 1e-9))))
                rem = left - take
                if 0 < rem < MIN_SEG:
                    take = take - (MIN_SEG - rem) if take - (MIN_SEG - rem) >= MIN_SEG else left
                pieces[-1][0] += take; pieces[-1][1] += 1
                cur += take + OVH; left -= take
    return pieces


lo = sum(r[2] for r in ranges) * n_groups / n_pieces
hi = lo + OVH * len(ranges) * n_groups / n_pieces + 4 * OVH + 256
for _ in range(48):
    mid = 0.5 * (lo + hi)
    if len(fill(mid)) <= n_pieces: hi = mid
    else: lo = mid
pieces = fill(hi)
pieces += [[0, 0]] * (n_pieces - len(pieces))
per_xcd = P // 8 // gs
stages = np.array([pieces[(p % 8) * per_xcd + (p // 8) // gs][0] for p in range(P)], dtype=float)
entries = np.array([pieces[(p % 8) * per_xcd + (p // 8) // gs][1] for p in range(P)], dtype=float)
dur = us[:, 3] - us[:, 0]
ok = live & (stages > 0)
A = np.stack([stages[ok], entries[ok], np.ones(ok.sum())], axis=1)
coef, res, *_ = np.linalg.lstsq(A, dur[ok], rcond=None)
print('blocks', P, 'live', int(live.sum()), 'pieces', n_pieces, 'cap', round(hi, 2), 'stages per piece', stages[ok].min(), stages[ok].max(), 'entries', entries[ok].min(), entries[ok].max())
print('fit: duration = %.3f us * stages + %.2f us * entries + %.1f us   (charge = %.2f stages; rms residual %.2f us)' % (coef[0], coef[1], coef[2], coef[1] / coef[0], float(np.sqrt(np.mean((A @ coef - dur[ok])**2)))))
print('start quantiles', np.percentile(us[live, 0], [0, 50, 100]).round(1))
print('end quantiles  ', np.percentile(us[live, 3], [0, 10, 50, 90, 100]).round(1))
print('duration quantiles', np.percentile(dur[ok], [0, 10, 50, 90, 100]).round(1))
for ne in sorted(set(entries[ok].astype(int))):
    m = ok & (entries == ne)
    print(f'entries {ne}: blocks {int(m.sum()):4d}  stages {stages[m].mean():6.1f}  duration {dur[m].mean():7.1f} us  end {us[m, 3].mean():7.1f}  (min {us[m, 3].min():.1f} max {us[m, 3].max():.1f})')
# co-resident pairs (blocks p and p + P / 2 land on the same CU when the dispatcher fills an XCD's CUs in order)
half = P // 2
pair_stage = stages[:half] + stages[half:]
pair_end = np.maximum(us[:half, 3], us[half:, 3])
print('pairs p, p + P/2: stage sums', pair_stage.min(), pair_stage.max(), ' end of the later one: quantiles', np.percentile(pair_end, [0, 50, 100]).round(1))
by_xcd = [us[(np.arange(P) % 8) == x, 3].max() for x in range(8)]
print('last end per XCD', np.round(by_xcd, 1))
old, young = np.arange(P) < half, np.arange(P) >= half
for name, m in (('blocks 0 .. P/2-1 (dispatched first)', old & ok & (entries <= 2)), ('blocks P/2 .. P-1', young & ok & (entries <= 2))):
    print(f'{name}: n {int(m.sum())} stages {stages[m].mean():.1f} end mean {us[m, 3].mean():.1f} quantiles', np.percentile(us[m, 3], [0, 10, 50, 90, 100]).round(1),
          ' K loop start', np.percentile(us[m, 1], [50]).round(1))
