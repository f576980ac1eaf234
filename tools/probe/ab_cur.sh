#!/bin/bash
# A/B on one box: the library of the tree against build/cur/libpls_hip.so (a copy taken before the change under test)
mkdir -p gpurun_out/r5
{
for rep in 1 2; do
for lib in "" build/cur/libpls_hip.so; do
  echo "== library: ${lib:-tree}"
  PLS_AMD_LIBRARY=${lib:+$PWD/$lib} timeout -k 10 200 python bench.py --algo kernel --steps 10 --warmup 3 --no-cpu --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C3 kernel plan', d['value'], d['ms_per_step'], d['roofline']['families_ms_per_fit'])"
  PLS_AMD_LIBRARY=${lib:+$PWD/$lib} timeout -k 10 200 python bench.py --workload C4 --algo kernel --steps 10 --warmup 3 --no-cpu --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('C4 kernel plan', d['value'], d['ms_per_step'], d['roofline']['families_ms_per_fit'])"
  PLS_AMD_LIBRARY=${lib:+$PWD/$lib} timeout -k 10 200 python tools/xty_m8.py 2>/dev/null
done
done
} 2>&1 | tee gpurun_out/r5/ab_cur.txt
