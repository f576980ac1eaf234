#!/bin/bash
# SYRK A/B on config 3 (GRAM plan end to end, the SYRK launch by HIP events): 8 waves per workgroup (default) vs
# PLS_HIP_SYRK_W8=0 (4 waves), alternating processes
O=gpurun_out/r3; mkdir -p $O
t() { python bench.py --workload C3 --algo gram --steps 20 --warmup 4 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['alt'].get('type2_mfma_syrk',{}); print('$1 gram', d['value'], 'comp/s', d['ms_per_step'], 'ms/fit; KERNEL_TYPE2', r.get('value'), 'comp/s; SYRK launch', r.get('roofline',{}).get('avg_launch_ms'), 'ms =', r.get('roofline',{}).get('frac'), 'of 78.6 TF')"; }
for i in 1 2 3; do
PLS_HIP_SYRK_W8=1 t eight_waves
PLS_HIP_SYRK_W8=0 t four_waves
done
for dw in 0.66 0.70 0.74; do PLS_HIP_SYRK_DIAGW=$dw PLS_HIP_SYRK_W8=1 t eight_waves_diagw_$dw; done
