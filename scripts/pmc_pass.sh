#!/bin/bash
# one rocprofv3 counter pass: scripts/pmc_pass.sh NAME "COUNTERS" program args...  -> gpurun_out/pmc/NAME_summary.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc
name=$1; ctr=$2; shift 2
mkdir -p $O && cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc $ctr --output-format csv -d $O/$name -o $name -- python3 "$@" > $O/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $O/$name.log; exit 1; }
python3 - "$O" "$name" <<'PY'
import csv, glob, os, sys, collections
O, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/{name}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    os.remove(f)
with open(f"{O}/{name}_summary.csv", "w") as out:
    out.write("kernel,counter,dispatches,mean,sum\n")
    for k, cs in sorted(agg.items()):
        for c, v in cs.items():
            out.write(f"\"{k}\",{c},{len(v)},{sum(v)/len(v):.6g},{sum(v):.6g}\n")
PY
