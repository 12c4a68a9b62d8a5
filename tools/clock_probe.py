#!/usr/bin/env python3
"""Why is the fp16 path a few per cent slower than bf16 only under sustained load (two steps in flight), when every kernel
times the same on its own?  Sample the card's shader clock and socket power (rocm-smi, read-only) while bench.py's headline step
runs in each precision for several seconds.
  python tools/clock_probe.py [seconds]"""
import json, os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0


def sample(stop, out):
    while not stop.is_set():
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True)
        try:
            j = json.loads(r.stdout)
            card = next(iter(j.values()))
            sclk = next((v for k, v in card.items() if "sclk" in k.lower()), None)
            pw = next((v for k, v in card.items() if "power" in k.lower() and "W" in k), None)
            m = re.search(r"(\d+)\s*Mhz", str(sclk), flags=re.I)
            out.append((time.time(), int(m.group(1)) if m else None, float(pw) if pw not in (None, "N/A") else None))
        except Exception as e:            # noqa: BLE001
            out.append((time.time(), None, None))
        time.sleep(0.15)


for prec in ("bf16", "fp16", "bf16", "fp16"):
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    steps = int(secs / 0.00042)
    th.start()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--precision", prec, "--steps", str(steps), "--warmup", "5", "--no-cpu-baseline", "--no-extras"],
                       capture_output=True, text=True)
    stop.set(); th.join()
    try:
        v = json.loads(r.stdout.strip().splitlines()[-1])["value"]
    except Exception:                     # noqa: BLE001
        v = None
    clk = [c for _, c, _ in out if c]
    pw = [p for _, _, p in out if p]
    busy = clk[len(clk) // 3:]            # the last two thirds: under load
    print(f"{prec}: {v and round(v)} patches/s over {steps} steps; sclk under load mean {sum(busy) / max(1, len(busy)):.0f} MHz (min {min(busy, default=0)}, max {max(busy, default=0)}), "
          f"power mean {sum(pw[len(pw) // 3:]) / max(1, len(pw[len(pw) // 3:])):.0f} W, {len(clk)} samples", flush=True)
