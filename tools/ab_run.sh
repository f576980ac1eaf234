#!/bin/bash
# A/B of two builds of the library on one box: pls_amd/csrc/libpls_hip.so (A) and pls_amd/csrc/ab/libpls_hip.so (B)
cp pls_amd/csrc/libpls_hip.so /tmp/A.so; cp pls_amd/csrc/ab/libpls_hip.so /tmp/B.so
for rep in 1 2; do for v in A B; do
  cp /tmp/$v.so pls_amd/csrc/libpls_hip.so
  for wa in ${BENCH:-C3:nipals C4:nipals C5rank:nipals}; do
    wl=${wa%%:*}; algo=${wa##*:}
    python bench.py --workload $wl --algo $algo --steps 8 --warmup 3 --no-cpu --no-alt 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$v rep$rep $wl $algo', d['value'], 'comp/s; fused avg', r['avg_launch_ms'], 'ms', r['frac'])"
  done
done; done
