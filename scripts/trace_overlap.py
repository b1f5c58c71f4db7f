"""Summarise a rocprofv3 --kernel-trace CSV: per kernel name count / mean duration, and the busy / overlapped time of the
last steps (development aid)."""
import csv
import sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = defaultdict(list)
for r in rows:
    names[r['Kernel_Name'][:60]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for n, d in sorted(names.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f'{n:60s} n={len(d):5d} mean {sum(d)/len(d):9.2f} us  total {sum(d)/1e3:9.2f} ms')
# timeline of the last ~2 steps
tail = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -16:]
t0 = int(tail[0]['Start_Timestamp'])
for r in tail:
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} -> {(int(r['End_Timestamp'])-t0)/1e3:9.1f} us  q{r.get('Queue_Id','?'):>3s} {r['Kernel_Name'][:70]}")
