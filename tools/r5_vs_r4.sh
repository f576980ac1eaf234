#!/bin/bash
# round 5 against the round-4 library (build/r4/libpls_hip.so, built from commit ebbe18d) on the SAME box, alternating
mkdir -p gpurun_out/r5
one() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']; print('%9.1f comp/s  %9.3f ms/fit  dominant %-22s %8.4f ms  frac %.4f' % (d['value'], d['ms_per_step'], r.get('kernel','?')[:22], r['avg_launch_ms'], r['frac']))"; }
R4=$GRAFT_REPO_ROOT/build/r4/libpls_hip.so
for cfg in "C3 nipals" "C3 kernel" "C3 gram" "C3eighth nipals" "C3eighth kernel" "C4 nipals" "C4 kernel" "C5rank nipals" "C5rank kernel"; do
  set -- $cfg
  for lib in r4 r5 r4 r5; do
    echo -n "$1 $2 $lib   "
    if [ $lib = r4 ]; then export PLS_AMD_LIBRARY=$R4; else unset PLS_AMD_LIBRARY; fi
    timeout -k 10 200 python3 bench.py --workload $1 --algo $2 --steps 10 --warmup 3 --no-cpu --no-alt --profile-after 2>/dev/null | one || exit 1
  done
done
