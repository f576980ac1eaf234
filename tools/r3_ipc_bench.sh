#!/bin/bash
# per-component cost of the cross-process device-side exchange: ranks sharing the one GPU (gloo carries the set-up only)
O=gpurun_out/r3; mkdir -p $O
for spec in "1:torch" "2:ipc" "2:torch" "4:ipc" "4:torch"; do
  n=${spec%%:*}; red=${spec##*:}
  if [ "$n" = "1" ]; then
    timeout -k 10 200 python bench.py --workload C3eighth --steps 10 --warmup 3 --no-cpu --no-alt > $O/ipc_bench_${n}_${red}.json 2> $O/ipc_bench_${n}_${red}.err
  else
    timeout -k 10 300 python bench.py --gpus $n --backend gloo --reducer $red --workload C3eighth --steps 10 --warmup 3 --no-cpu --no-alt > $O/ipc_bench_${n}_${red}.json 2> $O/ipc_bench_${n}_${red}.err
  fi
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$O/ipc_bench_${n}_${red}.json") if l.startswith("{")][-1])
    print("ranks $n reducer", d["config"]["reducer"], ":", d["value"], "comp/s", d["ms_per_step"], "ms/fit ->", round(d["ms_per_step"]*1e3/20,1), "us per component")
except Exception as e:
    print("ranks $n $red: no line", e)
PY
done
