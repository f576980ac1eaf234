#!/bin/bash
# NOTE: the experiment switch this script sweeps (PLS_HIP_EXP_* / the pair weight) was compiled out once its value was fixed;
# kept as the record of how the file of the same name under profiles/r5/ was produced (check out the commit named there to re-run).
# round 5: shares of the tiles by XCD class in the deflating pass (PLS_HIP_EXP_RHO: time per tile of the odd XCDs over the even ones)
mkdir -p gpurun_out/r5
one() { python3 -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%8.1f comp/s  %8.2f us/component  pass %8.2f us  frac %.4f' % (d['value'], d['ms_per_step']*1e3/{'C3':20,'C3eighth':20,'C4':50,'C5rank':20}.get(d['config']['workload'].split(':')[0].split()[0],20), d['roofline']['avg_launch_ms']*1e3, d['roofline']['frac']))"; }
for rho in 1.0 1.05 1.08 1.11; do
  echo "== rho $rho: C3eighth, C3eighth, C3, C4, C5rank"
  for w in C3eighth C3eighth C3 C4 C5rank; do
    PLS_HIP_EXP_RHO=$rho timeout -k 10 200 python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu --no-alt --profile-after 2>/dev/null | one || exit 1
  done
done
