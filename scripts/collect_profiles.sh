#!/bin/bash
# copies the summaries of the last scripts/profile_round.sh run from gpurun_out/prof into profiles/ under a round tag
tag=${1:-r03}; P=gpurun_out/prof; D=profiles
cp $P/bench/bench_kernel_stats.csv $D/${tag}_final_bench_kernel_stats.csv
cp $P/bench_under_rocprof.json $D/${tag}_final_bench_under_rocprof.json
cp $P/wrn/wrn_kernel_stats.csv $D/${tag}_final_wrn34_10_kernel_stats.csv
cp $P/wrn_under_rocprof.log $D/${tag}_final_wrn34_10_under_rocprof.txt
cp $P/bagnet/bagnet_kernel_stats.csv $D/${tag}_final_bagnet17_kernel_stats.csv
cp $P/bagnet_under_rocprof.log $D/${tag}_final_bagnet17_under_rocprof.txt
cp $P/bf16/bf16_kernel_stats.csv $D/${tag}_final_bf16_gtsrb_kernel_stats.csv 2>/dev/null
cp $P/bf16_under_rocprof.log $D/${tag}_final_bf16_gtsrb_under_rocprof.txt 2>/dev/null
for n in fetch_gtsrb write_gtsrb sq_gtsrb fetch_trhits write_trhits tcc_trhits tcp_trhits ta_trhits sqv_trhits sq_bf16 fetch_wrn write_wrn sq_wrn; do
  [ -f $P/${n}_summary.csv ] && cp $P/${n}_summary.csv $D/${tag}_final_pmc_${n}.csv
done
python3 scripts/make_traffic_json.py $tag
