#!/bin/bash
# round 5: PLS_HIP_TAIL = 1 (pass with its tail -> update kernel) against 2 (update in the tail), same box, alternating
mkdir -p gpurun_out/r5
one() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%8.1f comp/s  %8.2f us/component  pass %8.2f us' % (d['value'], d['ms_per_step']*1e3/20, d['roofline']['avg_launch_ms']*1e3))"; }
for rep in 1 2 3; do
  for t in 1 2; do
    echo -n "TAIL=$t C3eighth  "; PLS_HIP_TAIL=$t timeout -k 10 200 python3 bench.py --workload C3eighth --steps 20 --warmup 5 --no-cpu --no-alt --profile-after 2>/dev/null | one || exit 1
  done
done
for t in 1 2; do echo -n "TAIL=$t C3  "; PLS_HIP_TAIL=$t timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu --no-alt --profile-after 2>/dev/null | one || exit 1; done
