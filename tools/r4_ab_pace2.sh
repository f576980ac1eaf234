#!/bin/bash
# round 4: workgroups per CU of retile_xty (copy + X^T Y sweep), KERNEL plan, alternating runs
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O; rm -f $O/rx_*.json
for rep in 1 2 3; do for wl in C3 C4 wide8k C5rank; do for wgs in 1 2; do
    PLS_HIP_RX_WGS=$wgs python3 bench.py --workload $wl --algo kernel --steps 10 --warmup 3 --no-alt --no-cpu > $O/rx_${wl}_w${wgs}_r${rep}.json 2>/dev/null
done; done; done
python3 - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r4/rx_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d["roofline"]
        print("%-28s %9.1f comp/s  %9.4f ms/fit  pass %.5f ms frac %.4f  retile_xty %.4f ms" % (os.path.basename(f), d["value"], d["ms_per_step"], r["avg_launch_ms"], r["frac"], r["families_ms_per_fit"]["deflate"]))
    except Exception as e:
        print(f, "failed", e)
PY
