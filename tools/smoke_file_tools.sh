#!/bin/bash
# GPU box: the file-based tools end to end on trees built from tests/golden/h5/ (test_sets.py, train_dp.py --path_for_train --batch_metrics, a checkpoint round trip).
set -e
T=$(mktemp -d)
mkdir -p $T/data_for_test/SR_2x2_2x/SetA $T/data_for_train/SR_5x5_2x/S1
cp tests/golden/h5/scene_a2_2x.h5 $T/data_for_test/SR_2x2_2x/SetA/s1.h5
cp tests/golden/h5/train_000001.h5 $T/data_for_train/SR_5x5_2x/S1/000001.h5
cp tests/golden/h5/chunked_gzip.h5 $T/data_for_train/SR_5x5_2x/S1/000002.h5
python tools/test_sets.py --angRes 2 --scale_factor 2 --path_for_test $T/data_for_test/ --precision fp16
python tools/train_dp.py --angRes 5 --scale_factor 2 --batch_size 2 --epoch 2 --path_for_train $T/data_for_train/ --path_log $T/log/ --batch_metrics
ls $T/log/SR_5x5_2x/LFT/checkpoints/
