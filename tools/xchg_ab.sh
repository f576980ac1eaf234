O=gpurun_out/r3; mkdir -p $O
run() { # fused flag, tag
  PLS_HIP_XCHG_FUSED=$1 timeout -k 10 300 python bench.py --gpus 2 --backend gloo --reducer ipc --workload C3eighth --steps 20 --warmup 5 --no-cpu --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$2', round(d['ms_per_step']*1e3/20,1))"
}
one() { timeout -k 10 200 python bench.py --workload C3eighth --steps 20 --warmup 5 --no-cpu --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('one', round(d['ms_per_step']*1e3/20,1))"; }
for i in 1 2 3; do one; run 1 fused; run 0 two; done
