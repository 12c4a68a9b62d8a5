#!/bin/bash
# GPU box: rocprofv3 evidence for the current build -> gpurun_out/prof_<tag>/ (copy the summaries into profiles/ afterwards).
#   bash tools/profile_round.sh r04
# Kernel trace and the PMC passes are separate runs (a --pmc run never carries a trace domain); the program follows `--` directly.
tag=${1:-r04}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
common="bench.py --steps 60 --warmup 15 --streams 1 --inflight 1 --no-graph --no-cpu-baseline --no-extras --settle-seconds 1.0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $common > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 5 --warmup 3 --streams 1 --inflight 1 --no-graph --no-cpu-baseline --no-extras --settle-seconds 0.1 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 5 --warmup 3 --streams 1 --inflight 1 --no-graph --no-cpu-baseline --no-extras --settle-seconds 0.1 > $out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 5 --warmup 3 --streams 1 --inflight 1 --no-graph --no-cpu-baseline --no-extras --settle-seconds 0.1 > $out/pmc_sq.log 2>&1
stats=$(find $out/trace -name "*kernel_stats.csv" | head -1)
cp "$stats" $out/${tag}_kernel_stats.csv
python3 tools/collect_traffic.py $out/pmc_fetch $out/pmc_write $out/${tag}_hbm_traffic.json > $out/${tag}_hbm_traffic.txt
python3 tools/pmc_summary.py $out/pmc_sq > $out/${tag}_pmc_sq.txt
# the training step (BASELINE configs[2] shape, B = 8): kernel trace of 10 steps per math mode, PMC passes of 2 eager steps
for math in bf16x3 bf16x6 fp32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_trace_$math -- python3 tools/train_bench.py --math $math --steps 10 --warmup 3 --no-graph > $out/train_trace_$math.log 2>&1
  cp "$(find $out/train_trace_$math -name "*kernel_stats.csv" | head -1)" $out/${tag}_train_kernel_stats_$math.csv
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/train_fetch_$math -- python3 tools/train_bench.py --math $math --steps 2 --warmup 0 --no-graph > $out/train_fetch_$math.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/train_write_$math -- python3 tools/train_bench.py --math $math --steps 2 --warmup 0 --no-graph > $out/train_write_$math.log 2>&1
  python3 tools/collect_train_traffic.py $out/train_fetch_$math $out/train_write_$math 2 $math $out/${tag}_train_hbm_traffic_$math.json > $out/${tag}_train_hbm_traffic_$math.txt
  cat $out/${tag}_train_hbm_traffic_$math.txt | head -6
done
head -12 $out/${tag}_kernel_stats.csv | cut -c1-150
cat $out/${tag}_hbm_traffic.txt
