#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of bench.py, then HBM-traffic counters
# (FETCH_SIZE and WRITE_SIZE in separate passes) and SQ busy counters on one GTSRBNet forward.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
mkdir -p $O && cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench.err || exit 1
rm -f $O/bench/*kernel_trace.csv
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 $R/scripts/tune_conv.py 8192 > $O/fetch.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 $R/scripts/tune_conv.py 8192 > $O/write.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq -o sq -- python3 $R/scripts/tune_conv.py 8192 > $O/sq.log 2>&1 || echo "sq pass failed (non-fatal)"
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/prof"
for name in ("fetch", "write", "sq"):
    files = glob.glob(f"{O}/{name}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        os.remove(f)
    with open(f"{O}/{name}_summary.csv", "w") as out:
        out.write("kernel,counter,dispatches,mean,sum\n")
        for k, cs in sorted(agg.items()):
            for c, v in cs.items():
                out.write(f"\"{k}\",{c},{len(v)},{sum(v)/len(v):.6g},{sum(v):.6g}\n")
PY
find $O -type f | head -40
