#!/bin/bash
# counter passes for the warp kernel only (scripts/prof_tr_hits.py): texture-path and issue counters
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profw
rm -rf $O; mkdir -p $O && cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1 ctr=$2; shift 2
  timeout -s KILL 150 rocprofv3 --pmc $ctr --output-format csv -d $O/$name -o $name -- python3 "$@" > $O/$name.log 2>&1 || { echo "pass $name failed"; return 1; }
}
pass ta "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" $R/scripts/prof_tr_hits.py || exit 1
pass sqv "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" $R/scripts/prof_tr_hits.py || exit 1
pass tcp "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" $R/scripts/prof_tr_hits.py || echo non-fatal
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/profw"
for d in sorted(glob.glob(f"{O}/*/")):
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_warp" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in agg.items():
            print(f"{c:36s} n={len(v)} mean={sum(v)/len(v):.6g}")
        os.remove(f)
PY
