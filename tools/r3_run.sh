#!/bin/bash
# round-3 measurement helper (one gpurun call): TESTS="pytest args", BENCH="wl:algo wl:algo ...", TAG=name
set -o pipefail
O=gpurun_out/r3
mkdir -p $O
TAG=${TAG:-run}
if [ -n "$TESTS" ]; then
  eval timeout -k 10 ${TEST_TIMEOUT:-900} python -m pytest $TESTS -x -q -m gpu > $O/${TAG}_tests.txt 2>&1
  echo "tests rc=$?" | tee -a $O/${TAG}_tests.txt
  tail -${TEST_TAIL:-6} $O/${TAG}_tests.txt
fi
for wa in $BENCH; do
  wl=${wa%%:*}; algo=${wa##*:}
  timeout -k 10 300 python bench.py --workload $wl --algo $algo --steps ${STEPS:-6} --warmup 2 --no-cpu --no-alt $BENCH_ARGS > $O/${TAG}_${wl}_${algo}.json 2> $O/${TAG}_${wl}_${algo}.err || echo "bench $wl $algo failed"
  python - <<PY
import json
try:
    d = json.load(open("$O/${TAG}_${wl}_${algo}.json"))
    r = d["roofline"]
    print("$wl $algo", d["value"], "comp/s", d["ms_per_step"], "ms/fit; dominant", r["kernel"], r["avg_launch_ms"], "ms", r["frac"], r["families_ms_per_fit"])
except Exception as e:
    print("$wl $algo: no line", e)
PY
done
