#!/bin/bash
# NOTE: the experiment switch this script sweeps (PLS_HIP_EXP_* / the pair weight) was compiled out once its value was fixed;
# kept as the record of how the file of the same name under profiles/r5/ was produced (check out the commit named there to re-run).
# round 5: the deflating pass on a shard -- operand vectors to LDS behind the first tile's loads (LATE), pacing (PACE x 64 cycles per tile)
mkdir -p gpurun_out/r5
one() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%8.1f comp/s  %8.2f us/component  pass %8.2f us' % (d['value'], d['ms_per_step']*1e3/20, d['roofline']['avg_launch_ms']*1e3))"; }
for rep in 1 2; do
for cfg in "0 32" "1 32" "1 16" "1 8" "1 0" "0 0"; do
  set -- $cfg
  echo -n "LATE=$1 PACE=$2 C3eighth  "; PLS_HIP_EXP_LATE=$1 PLS_HIP_EXP_PACE=$2 timeout -k 10 200 python3 bench.py --workload C3eighth --steps 20 --warmup 5 --no-cpu --no-alt --profile-after 2>/dev/null | one || exit 1
done
done
for cfg in "0 32" "1 32" "1 16"; do set -- $cfg; echo -n "LATE=$1 PACE=$2 C3  "; PLS_HIP_EXP_LATE=$1 PLS_HIP_EXP_PACE=$2 timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu --no-alt --profile-after 2>/dev/null | one || exit 1; done
