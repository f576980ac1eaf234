#!/bin/bash
# round 4: does the caller's column stride matter to the passes that read the caller's matrix?  config 3 (ld = N = 2^20: columns 8 MiB
# apart) against ld = N + 64, 8 components (no copy into tiles: every pass of the KERNEL plan reads X itself) and 20 (both plans)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O; rm -f $O/pad_*
for rep in 1 2; do for wl in C3a8 C3a8pad C3 C3pad; do for algo in kernel nipals gram; do
  python3 bench.py --algo $algo --workload $wl --steps 10 --warmup 3 --no-cpu --no-alt > $O/pad_${wl}_${algo}_r$rep.json 2>/dev/null
done; done; done
python3 - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r4/pad_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1]); r = d["roofline"]
        print("%-24s %9.1f comp/s  %9.4f ms/fit  %s avg %.4f ms  %s" % (os.path.basename(f)[4:-5], d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["families_ms_per_fit"]))
    except Exception as e:
        print(f, "failed", e)
PY
