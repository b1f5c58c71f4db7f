#!/bin/bash
# k_pk_l3 / k_pk_tab2 averages of the bench's one-lane steps under a library build ($1, default the shipped one) (development aid)
R=$PWD
O=$R/gpurun_out/l3_quick
rm -rf $O && mkdir -p $O
[ -n "$1" ] && export VEGAMX_LIBRARY=$R/$1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --core-only --lanes 1 --steps 20 --warmup 5 > $O/line.json 2> $O/err.txt
python3 - <<PY
import csv, glob
f = glob.glob('$O/t/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(f"{r['Name'][:50]:52s} {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:7.1f}")
PY
rm -rf $O/t
