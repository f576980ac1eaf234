#!/bin/bash
# NOTE: the experiment switch this script sweeps (PLS_HIP_EXP_* / the pair weight) was compiled out once its value was fixed;
# kept as the record of how the file of the same name under profiles/r5/ was produced (check out the commit named there to re-run).
# round 5: SYRK with the diagonal blocks in pairs (PAIRS=1) against one per workgroup (0), the pair's weight in full blocks (DW2)
mkdir -p gpurun_out/r5
echo -n "PAIRS=0           "; PLS_HIP_SYRK_PAIRS=0 timeout -k 10 120 python3 tools/syrk_time.py 60 || exit 1
for dw in 0.95 1.0 1.05 1.1 1.15 1.2 1.3; do echo -n "PAIRS=1 DW2=$dw  "; PLS_HIP_SYRK_PAIRS=1 timeout -k 10 120 python3 tools/syrk_time.py 60 || exit 1; done
echo -n "PAIRS=0           "; PLS_HIP_SYRK_PAIRS=0 timeout -k 10 120 python3 tools/syrk_time.py 60 || exit 1
