#!/bin/bash
# round 4: pacing of the deflating pass on the tiled copy (PLS_HIP_PACE x 64 cycles of s_sleep per tile and wave)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O; rm -f $O/pace_*.json
for wl in C3 C3eighth C5rank C4; do for pace in 0 8 16 24 32 40 48; do
  PLS_HIP_PACE=$pace python3 bench.py --workload $wl --steps 8 --warmup 3 --no-alt --no-cpu > $O/pace_${wl}_p${pace}.json 2>/dev/null
done; done
python3 - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r4/pace_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d["roofline"]
        print("%-28s %9.1f comp/s  %9.4f ms/fit  %s %.5f ms frac %.4f" % (os.path.basename(f), d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"]))
    except Exception as e:
        print(f, "failed", e)
PY
