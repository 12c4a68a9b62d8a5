set -e
mkdir -p gpurun_out/r3w
timeout -k 10 300 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/r3w/test.log 2>&1 || { tail -30 gpurun_out/r3w/test.log; exit 1; }
tail -2 gpurun_out/r3w/test.log
timeout -k 10 200 python tools/train_families.py > gpurun_out/r3w/fam.log 2>&1
grep "step ms" gpurun_out/r3w/fam.log | cut -c1-150
