set -e
mkdir -p gpurun_out/r4n
timeout -k 10 600 python -m pytest tests/test_gpu_train.py -x -q > gpurun_out/r4n/test.log 2>&1 || { tail -40 gpurun_out/r4n/test.log; exit 1; }
tail -2 gpurun_out/r4n/test.log
timeout -k 10 200 python tools/train_families.py > gpurun_out/r4n/fam.log 2>&1
grep "step ms" gpurun_out/r4n/fam.log | cut -c1-130
grep "k_lin:" gpurun_out/r4n/fam.log | head -12
