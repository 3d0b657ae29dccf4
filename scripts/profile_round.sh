#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of bench.py, then counter passes (each counter group in a run
# of its own, as the guide prescribes) on: one GTSRBNet forward (tune_conv.py), one small tr_hits job (prof_tr_hits.py:
# warp + composite kernels) and one WideResNet-34-10 forward (tune_wrn.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O && cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_under_rocprof.json 2> $O/bench.err || exit 1
rm -f $O/bench/*kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wrn -o wrn -- python3 $R/scripts/tune_wrn.py 1024 > $O/wrn_under_rocprof.log 2> $O/wrn.err || exit 1
rm -f $O/wrn/*kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bagnet -o bagnet -- python3 $R/scripts/tune_bagnet.py 128 > $O/bagnet_under_rocprof.log 2> $O/bagnet.err || echo non-fatal
rm -f $O/bagnet/*kernel_trace.csv
pass() {   # name, counters, program...
  local name=$1 ctr=$2; shift 2
  timeout -s KILL 150 rocprofv3 --pmc $ctr --output-format csv -d $O/$name -o $name -- python3 "$@" > $O/$name.log 2>&1 || { echo "pass $name failed"; return 1; }
}
pass fetch_gtsrb "FETCH_SIZE" $R/scripts/tune_conv.py 32000 || exit 2
pass write_gtsrb "WRITE_SIZE" $R/scripts/tune_conv.py 32000 || exit 3
pass sq_gtsrb "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" $R/scripts/tune_conv.py 32000 || echo non-fatal
pass fetch_trhits "FETCH_SIZE" $R/scripts/prof_tr_hits.py || exit 4
pass write_trhits "WRITE_SIZE" $R/scripts/prof_tr_hits.py || exit 5
pass tcc_trhits "TCC_HIT_sum TCC_MISS_sum" $R/scripts/prof_tr_hits.py || echo non-fatal
pass tcp_trhits "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" $R/scripts/prof_tr_hits.py || echo non-fatal
pass ta_trhits "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" $R/scripts/prof_tr_hits.py || echo non-fatal
pass sqv_trhits "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" $R/scripts/prof_tr_hits.py || echo non-fatal
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16 -o bf16 -- python3 $R/scripts/tune_bf16.py 32000 3 > $O/bf16_under_rocprof.log 2> $O/bf16.err || echo non-fatal
rm -f $O/bf16/*kernel_trace.csv
pass sq_bf16 "SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $R/scripts/tune_bf16.py 32000 2 || echo non-fatal
pass fetch_wrn "FETCH_SIZE" $R/scripts/tune_wrn.py 1024 || exit 6
pass write_wrn "WRITE_SIZE" $R/scripts/tune_wrn.py 1024 || exit 7
pass sq_wrn "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" $R/scripts/tune_wrn.py 1024 || echo non-fatal
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/prof"
for d in sorted(glob.glob(f"{O}/*/")):
    name = os.path.basename(d.rstrip("/"))
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        os.remove(f)
    with open(f"{O}/{name}_summary.csv", "w") as out:
        out.write("kernel,counter,dispatches,mean,sum\n")
        for k, cs in sorted(agg.items()):
            for c, v in cs.items():
                out.write(f"\"{k}\",{c},{len(v)},{sum(v)/len(v):.6g},{sum(v):.6g}\n")
PY
find $O -type f | head -60
