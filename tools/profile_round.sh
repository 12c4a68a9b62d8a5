#!/bin/bash
# GPU box: rocprofv3 evidence for the current build -> gpurun_out/prof_<tag>/ (copy the summaries into profiles/ afterwards).
#   bash tools/profile_round.sh r03
# Kernel trace and the PMC passes are separate runs (a --pmc run never carries a trace domain); the program follows `--` directly.
tag=${1:-r03}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
common="bench.py --steps 60 --warmup 15 --streams 1 --inflight 1 --no-graph --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $common > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 5 --warmup 3 --streams 1 --inflight 1 --no-graph --no-cpu-baseline --no-extras > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 5 --warmup 3 --streams 1 --inflight 1 --no-graph --no-cpu-baseline --no-extras > $out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/pmc_sq -- python3 bench.py --steps 5 --warmup 3 --streams 1 --inflight 1 --no-graph --no-cpu-baseline --no-extras > $out/pmc_sq.log 2>&1
stats=$(find $out/trace -name "*kernel_stats.csv" | head -1)
cp "$stats" $out/${tag}_kernel_stats.csv
python3 tools/collect_traffic.py $out/pmc_fetch $out/pmc_write $out/${tag}_hbm_traffic.json > $out/${tag}_hbm_traffic.txt
python3 tools/pmc_summary.py $out/pmc_sq > $out/${tag}_pmc_sq.txt
head -12 $out/${tag}_kernel_stats.csv | cut -c1-150
cat $out/${tag}_hbm_traffic.txt
