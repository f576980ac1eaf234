#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
for mode in 0 1 2; do for st in 0 16 32 48 72 96; do
  PLS_HIP_SYRK_STAGM=$mode PLS_HIP_SYRK_STAG=$st python3 bench.py --algo gram --steps 10 --warmup 3 --no-alt --no-cpu > $O/stag_m${mode}_s${st}.json 2>/dev/null
done; done
python3 - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r4/stag_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1]); r = d["roofline"]
        print("%-20s %9.1f comp/s  %9.4f ms/fit  families %s" % (os.path.basename(f), d["value"], d["ms_per_step"], r["families_ms_per_fit"]))
    except Exception as e:
        print(f, "failed", e)
PY
