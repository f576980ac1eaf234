#!/bin/bash
# round 4: forms of the deflating pass on the tiled copy, same box, alternating:
#   PLS_HIP_TILED=0  a descriptor per column group (round 3)   1  one descriptor per tile   2  + t_prev first, p_prev
#   preloaded, every column group stored as soon as it has arrived
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
for rep in 1 2 3; do for v in 0 1 2; do
  PLS_HIP_TILED=$v python3 bench.py --steps 10 --warmup 3 --no-alt --no-cpu > $O/tiled${v}_$rep.json 2>/dev/null
done; done
for wl in C3eighth; do for v in 0 1 2; do
  PLS_HIP_TILED=$v python3 bench.py --workload $wl --steps 10 --warmup 3 --no-alt --no-cpu > $O/tiled${v}_$wl.json 2>/dev/null
done; done
python3 - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r4/tiled[012]_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d["roofline"]
        print("%-28s %9.1f comp/s  %9.4f ms/fit  %s %.5f ms frac %.4f" % (os.path.basename(f), d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"]))
    except Exception as e:
        print(f, "failed", e)
PY
