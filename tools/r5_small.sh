#!/bin/bash
# round 5: small and mid-size fits, and the KERNEL plan on a shard, against the round-4 library on the same box
mkdir -p gpurun_out/r5
R4=$GRAFT_REPO_ROOT/build/r4/libpls_hip.so
one() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']; print('%9.1f comp/s  %9.2f us/component  pass %8.2f us' % (d['value'], d['ms_per_step']*50, r['avg_launch_ms']*1e3))"; }
for lib in r4 r5 r4 r5; do
  if [ $lib = r4 ]; then export PLS_AMD_LIBRARY=$R4; else unset PLS_AMD_LIBRARY; fi
  echo -n "C3eighth kernel $lib  "; timeout -k 10 200 python3 bench.py --workload C3eighth --algo kernel --steps 20 --warmup 5 --no-cpu --no-alt --profile-after 2>/dev/null | one || exit 1
done
PLS_AMD_LIBRARY=$R4 timeout -k 10 300 python3 tools/small_scan.py gpurun_out/r5/small_scan_r4lib.txt > /dev/null 2>&1 || exit 1
unset PLS_AMD_LIBRARY
timeout -k 10 300 python3 tools/small_scan.py gpurun_out/r5/small_scan.txt > /dev/null 2>&1 || exit 1
paste -d'|' <(cut -c1-50 gpurun_out/r5/small_scan.txt) <(cut -c27-40 gpurun_out/r5/small_scan_r4lib.txt) <(cut -c41-100 gpurun_out/r5/small_scan.txt)
