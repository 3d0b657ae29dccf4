#!/bin/bash
# rocprofv3 evidence for the bf16 throughput mode: kernel trace of a GTSRBNet forward in both precisions
# (tune_bf16.py) and of a WideResNet-34-10 forward in bf16 mode, then SQ counter passes (issue / wait / LDS).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_bf16
rm -rf $O; mkdir -p $O && cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gtsrb -o gtsrb -- python3 $R/scripts/tune_bf16.py 32000 3 > $O/gtsrb_under_rocprof.log 2> $O/gtsrb.err || exit 1
rm -f $O/gtsrb/*kernel_trace.csv
export GRAPHITE_PREC=bf16
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wrn -o wrn -- python3 $R/scripts/tune_wrn.py 1024 > $O/wrn_under_rocprof.log 2> $O/wrn.err || exit 1
rm -f $O/wrn/*kernel_trace.csv
pass() {   # name, counters, program...
  local name=$1 ctr=$2; shift 2
  timeout -s KILL 150 rocprofv3 --pmc $ctr --output-format csv -d $O/$name -o $name -- python3 "$@" > $O/$name.log 2>&1 || { echo "pass $name failed"; return 1; }
}
pass sq_issue "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" $R/scripts/tune_bf16.py 8192 2 || echo non-fatal
pass sq_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT" $R/scripts/tune_bf16.py 8192 2 || echo non-fatal
pass sq_issue_wrn "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" $R/scripts/tune_wrn.py 512 || echo non-fatal
pass sq_lds_wrn "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT" $R/scripts/tune_wrn.py 512 || echo non-fatal
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/prof_bf16"
for d in sorted(glob.glob(f"{O}/*/")):
    name = os.path.basename(d.rstrip("/"))
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        os.remove(f)
    with open(f"{O}/{name}_summary.csv", "w") as out:
        out.write("kernel,counter,dispatches,mean,sum\n")
        for k, cs in sorted(agg.items()):
            for c, v in cs.items():
                out.write(f"\"{k}\",{c},{len(v)},{sum(v)/len(v):.6g},{sum(v):.6g}\n")
PY
find $O -type f | head -40
