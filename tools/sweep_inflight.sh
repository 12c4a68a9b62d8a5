#!/bin/bash
# GPU box: throughput of bench.py's workload over streams-per-step x steps-in-flight, and the other BASELINE shapes.
run() { python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*:', round(j['value']), 'patches/s', j['roofline']['kernel'], round(j['roofline']['launch_ms']*1e3,1), 'us')"; }
for cfg in "1 1" "1 2" "1 3" "1 4" "2 2"; do set -- $cfg; run --streams $1 --inflight $2; done
run --batch 8
run --batch 16
run --lr 64 --batch 8          # BASELINE configs[3] shape per GPU
run --ang 9 --batch 2          # BASELINE configs[4]
run --precision fp32
