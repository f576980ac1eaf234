#!/bin/bash
# round 4: the reduction of the partial rows in the tail of the pass (PLS_HIP_TAIL=1, default) against the separate
# reduce launch (0): identical bits, wall time per component on an eighth of config 3, one rank and two ranks over IPC
set -o pipefail
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
PLS_HIP_TAIL=0 python3 tools/tail_ab.py > $O/tail_digest_0.txt 2> $O/tail_digest_0.err || { tail -5 $O/tail_digest_0.err; exit 1; }
PLS_HIP_TAIL=1 python3 tools/tail_ab.py > $O/tail_digest_1.txt 2> $O/tail_digest_1.err || { tail -5 $O/tail_digest_1.err; exit 1; }
if cmp -s $O/tail_digest_0.txt $O/tail_digest_1.txt; then echo "tail A/B: bit-identical on $(wc -l < $O/tail_digest_1.txt) fits"; else echo "tail A/B: DIFFERENT"; diff $O/tail_digest_0.txt $O/tail_digest_1.txt; exit 1; fi
for rep in 1 2; do for t in 0 1; do
  PLS_HIP_TAIL=$t python3 bench.py --workload C3eighth --steps 20 --warmup 5 --no-cpu --no-alt > $O/tail${t}_eighth_$rep.json 2>/dev/null
  PLS_HIP_TAIL=$t timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --reducer ipc --workload C3eighth --steps 20 --warmup 5 --no-cpu --no-alt > $O/tail${t}_eighth_ipc2_$rep.json 2> $O/tail${t}_eighth_ipc2_$rep.err
done; done
for t in 0 1; do
  PLS_HIP_TAIL=$t python3 bench.py --steps 10 --warmup 3 --no-cpu --no-alt > $O/tail${t}_C3.json 2>/dev/null
  PLS_HIP_TAIL=$t python3 bench.py --algo kernel --steps 10 --warmup 3 --no-cpu --no-alt > $O/tail${t}_C3_kernel.json 2>/dev/null
done
python3 - <<'PY'
import glob, json, os
for f in sorted(glob.glob("gpurun_out/r4/tail[01]_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        r = d["roofline"]
        A = 20
        print("%-32s %9.1f comp/s  %9.4f ms/fit = %7.1f us/component  %s %.5f ms  reducer %s" % (os.path.basename(f), d["value"], d["ms_per_step"], d["ms_per_step"] * 1e3 / A, r["kernel"], r["avg_launch_ms"], d["config"]["reducer"]))
    except Exception as e:
        print(f, "failed", e)
PY
