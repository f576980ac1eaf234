t() { python bench.py --workload C3 --algo gram --steps 10 --warmup 3 --no-cpu --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], 'comp/s', d['ms_per_step'], 'ms/fit', d['roofline']['families_ms_per_fit'])"; }
for dw in 0.58 0.62 0.66 0.70; do PLS_HIP_SYRK_XY=0 PLS_HIP_SYRK_DIAGW=$dw t noY_dw_$dw; done
