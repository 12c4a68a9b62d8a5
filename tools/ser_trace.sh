#!/bin/bash
# GPU box: serial (one step in flight, one stream) rocprofv3 kernel durations of ab_so/ variants, one line per inference kernel.
#   bash tools/ser_trace.sh outdir name1 name2 ...
out=$1; shift
mkdir -p $out
export TMPDIR=/tmp LFT_AB_ANY_ABI=1
for v in "$@"; do
  export LFT_LIB_PATH=ab_so/liblft_$v.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ser_$v -- python3 bench.py --steps 200 --warmup 20 --streams 1 --inflight 1 --no-cpu-baseline --no-extras > $out/ser_$v.log 2>&1 || { echo "$v failed"; tail -3 $out/ser_$v.log; continue; }
  f=$(ls $out/ser_$v/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" "$v" <<'PY' >> $out/summary.txt
import csv, sys
rows = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    for k in ("k_conv0", "k_conv64", "k_ang", "k_spa1", "k_spa_b", "k_up", "k_assemble"):
        if k in n and int(r["Calls"]) > 100:
            c, t = rows.get(k, (0, 0.0))
            rows[k] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]))
print(sys.argv[2], " ".join(f"{k}={t / c / 1e3:.2f}" for k, (c, t) in sorted(rows.items())))
PY
done
cat $out/summary.txt
