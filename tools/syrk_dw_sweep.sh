t() { python bench.py --workload C3m4 --algo gram --steps 8 --warmup 3 --no-cpu --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], 'comp/s', d['ms_per_step'], 'ms/fit')"; }
for dw in 0.74 0.78 0.80 0.84 0.88; do PLS_HIP_SYRK_DIAGW=$dw t dw_$dw; done
PLS_HIP_SYRK_W8=0 t four_waves
t default
