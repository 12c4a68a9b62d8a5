#!/bin/bash
# GPU box: the round's committed evidence on the CURRENT sources -> gpurun_out/final_<tag>/ (copy into profiles/ afterwards).
#   bash tools/final_evidence.sh r04
# 1. rocprofv3 kernel trace + PMC passes of inference and of the training step (tools/profile_round.sh)
# 2. bench.py lines: default, the driver's call (--steps 20 --warmup 5), the other BASELINE shapes
# 3. micro-benchmarks and the clock probe behind two statements of DESIGN.md
tag=${1:-r04}
out=gpurun_out/final_$tag
mkdir -p $out
bash tools/profile_round.sh $tag > $out/profile.log 2>&1
cp gpurun_out/prof_$tag/${tag}_* $out/ 2>/dev/null
# bench.py reports roofline.traffic only when profiles/<tag>_hbm_traffic.json carries the hash of the sources being timed: put the
# files of step 1 in place before the bench lines are taken (the same files are committed afterwards)
cp gpurun_out/prof_$tag/${tag}_hbm_traffic.json gpurun_out/prof_$tag/${tag}_train_hbm_traffic_*.json profiles/ 2>/dev/null
python3 bench.py > $out/${tag}_bench.log 2> $out/${tag}_bench.err
python3 bench.py --steps 20 --warmup 5 > $out/${tag}_bench_driver_call.log 2> $out/${tag}_bench_driver_call.err
for cfg in cfg1 cfg4 cfg5; do python3 bench.py --config $cfg --steps 50 --warmup 10 --no-cpu-baseline > $out/${tag}_bench_$cfg.log 2> $out/${tag}_bench_$cfg.err; done
python3 bench.py --config cfg3 --steps 20 --warmup 5 > $out/${tag}_bench_cfg3.log 2> $out/${tag}_bench_cfg3.err
python3 bench.py --config cfg3 --batch 1 --steps 40 --warmup 10 > $out/${tag}_bench_cfg3_b1.log 2> $out/${tag}_bench_cfg3_b1.err
python3 bench.py --config cfg3 --train-math bf16x6 --steps 20 --warmup 5 > $out/${tag}_bench_cfg3_bf16x6.log 2> $out/${tag}_bench_cfg3_bf16x6.err
python3 bench.py --config cfg3 --train-math bf16x6 --batch 1 --steps 40 --warmup 10 > $out/${tag}_bench_cfg3_b1_bf16x6.log 2> $out/${tag}_bench_cfg3_b1_bf16x6.err
[ -x ab_so/mfma_small ] && ab_so/mfma_small > $out/${tag}_mfma_small.txt 2>&1
python3 tools/clock_probe.py 5 > $out/${tag}_clock_probe.txt 2>&1
for f in $out/${tag}_bench*.log; do echo "== $f"; tail -c 300 $f; echo; done
