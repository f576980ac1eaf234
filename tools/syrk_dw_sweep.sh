#!/bin/bash
# diagonal-workgroup weight of the SYRK row split (PLS_HIP_SYRK_DIAGW), config 3 with one and with four responses
t() { python bench.py --workload $2 --algo gram --steps 10 --warmup 3 --no-cpu --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$2 $1', d['value'], 'comp/s', d['ms_per_step'], 'ms/fit')"; }
for wl in C3 C3m4; do
for dw in 0.62 0.66 0.70 0.74 0.78 0.82; do PLS_HIP_SYRK_DIAGW=$dw t dw_$dw $wl; done
t default $wl
PLS_HIP_SYRK_W8=0 t four_waves $wl
done
