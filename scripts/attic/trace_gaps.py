"""Per-kernel durations and inter-kernel gaps from a rocprofv3 kernel-trace CSV (last N kernels)."""
import csv, sys, collections
path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n_chain = int(sys.argv[2]) if len(sys.argv) > 2 else 7
tail = rows[-n_chain * 100:]
dur = collections.defaultdict(list); gaps = []
for a, b in zip(tail[:-1], tail[1:]):
    gaps.append((int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3)
for r in tail:
    dur[r['Kernel_Name'][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in dur.items():
    print(f'{k:60s} n={len(v):4d} mean {sum(v)/len(v):7.2f} us  min {min(v):7.2f}')
gaps.sort()
print('gaps: median', gaps[len(gaps)//2], 'p10', gaps[len(gaps)//10], 'p90', gaps[9*len(gaps)//10])
# chain period: start-to-start of the same kernel
first = tail[0]['Kernel_Name']
starts = [int(r['Start_Timestamp']) for r in tail if r['Kernel_Name'] == first]
per = sorted((b - a) / 1e3 for a, b in zip(starts[:-1], starts[1:]))
print('period of', first[:40], 'median', per[len(per)//2])
busy = sum(sum(v) for v in dur.values()) / max(1, len(starts))
print('kernel time per evaluation', busy)
