#!/bin/bash
# round 5: what a shard spends per component beside its pass -- an eighth of config 3; PLS_HIP_TAIL = 1 (round 4: pass with its
# tail -> update, two launches) against 2 (the update as the last act of the tail: ONE launch per component); one process with
# and without event brackets in the timed region, two processes over the device-side exchange sharing the GPU
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5; mkdir -p $O; rm -f $O/fixed_*
for rep in 1 2 3; do
  for t in 1 2; do
    PLS_HIP_TAIL=$t python3 bench.py --workload C3eighth --steps 20 --warmup 5 --no-cpu --no-alt --profile-after > $O/fixed_tail${t}_plain_$rep.json 2>/dev/null
    PLS_HIP_TAIL=$t python3 bench.py --workload C3eighth --algo kernel --steps 20 --warmup 5 --no-cpu --no-alt --profile-after > $O/fixed_tail${t}_kernelplan_$rep.json 2>/dev/null
    PLS_HIP_TAIL=$t timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --reducer ipc --workload C3eighth --steps 20 --warmup 5 --no-cpu --no-alt > $O/fixed_tail${t}_ipc2_$rep.json 2> $O/fixed_tail${t}_ipc2_$rep.err
  done
done
python3 - <<'PY'
import glob, json, os
print("an eighth of config 3 (131,072 x 512 fp64, A = 20), bench.py --workload C3eighth --steps 20 --warmup 5")
for f in sorted(glob.glob("gpurun_out/r5/fixed_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1]); r = d["roofline"]
        per = d["ms_per_step"] * 1e3 / 20
        print("%-26s %8.1f comp/s  %7.1f us per component;  dominant pass %6.1f us by HIP events (%s)  -> beside the pass %5.1f us   reducer %s" % (
            os.path.basename(f)[6:-5], d["value"], per, r["avg_launch_ms"] * 1e3, r.get("measured_over", "?")[:30], per - r["avg_launch_ms"] * 1e3, d["config"].get("reducer")))
    except Exception as e:
        print(f, "failed", e)
PY
