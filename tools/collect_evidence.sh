#!/bin/bash
# Collects the measurement evidence of a round on the GPU box into gpurun_out/evidence/ (copy what is to be judged
# into profiles/rNN/ afterwards).  rocprofv3: the program goes directly after `--`; counters in their own passes.
# usage: bash tools/collect_evidence.sh [part ...]   parts: c3 gram deflate c4 c5 eighth edge narrow wide group host roctx small fuzz   (default: all but fuzz)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
E=gpurun_out/evidence; mkdir -p $E
PARTS="${*:-c3 gram deflate c4 c5 eighth edge narrow wide group host roctx small}"
stats() {  # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $E/tmp_$name -o p -- "$@" > $E/${name}_bench_under_rocprof.json 2> $E/tmp_$name.err
  cp $E/tmp_$name/p_kernel_stats.csv $E/${name}_kernel_stats.csv; rm -rf $E/tmp_$name $E/tmp_$name.err
}
pmc() {  # name, command...
  local name=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $E/tmp_${name}_$ctr -o p -- "$@" > /dev/null 2> $E/tmp_pmc.err
  done
  python3 tools/summarize_pmc.py $E/tmp_${name}_FETCH_SIZE/p_counter_collection.csv $E/tmp_${name}_WRITE_SIZE/p_counter_collection.csv $E/pmc_traffic_$name.json > $E/pmc_traffic_$name.txt
  rm -rf $E/tmp_${name}_FETCH_SIZE $E/tmp_${name}_WRITE_SIZE $E/tmp_pmc.err
}
for part in $PARTS; do case $part in
c3)
  python3 bench.py --steps 20 --warmup 5 > $E/bench_default_run.json 2> /dev/null
  stats final_nipals_fused python3 bench.py --steps 5 --warmup 2 --no-alt --no-cpu
  pmc C3_nipals_fused python3 bench.py --steps 2 --warmup 1 --no-alt --no-cpu
  stats c3_kernel_fused python3 bench.py --algo kernel --steps 5 --warmup 2 --no-alt --no-cpu
  pmc C3_kernel_fused python3 bench.py --algo kernel --steps 2 --warmup 1 --no-alt --no-cpu ;;
gram)
  # the SYRK (X^T X + X^T Y of config 3 in one launch): 100 timed launches behind 5 warm-ups, so that the rocprofv3 average
  # means what bench.py's alt.gram_mfma_syrk / alt.type2_mfma_syrk mean; matrix-pipe utilisation in a counter pass of its own
  python3 bench.py --algo gram --steps 20 --warmup 5 --no-cpu --no-alt > $E/bench_C3_gram_plan.json 2> /dev/null
  python3 tools/syrk_time.py 100 > $E/syrk_hip_events.txt 2> /dev/null
  stats c3_gram python3 tools/syrk_time.py 100
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $E/tmp_mfma -o p -- python3 tools/syrk_time.py 10 > /dev/null 2> $E/tmp_mfma.err
  python3 tools/mfma_util.py $E/tmp_mfma/p_counter_collection.csv syrk > $E/syrk_mfma_pmc.txt 2>&1; rm -rf $E/tmp_mfma $E/tmp_mfma.err
  # the plan's last sweep, T = X R with 20 columns (xb_mfma4_kernel): HIP events and the rocprofv3 average over 60 launches, HBM traffic
  python3 tools/xb_time.py 60 > $E/xb_hip_events.txt 2> /dev/null
  stats c3_xb python3 tools/xb_time.py 60
  pmc C3_xb python3 tools/xb_time.py 3 ;;
deflate)
  python3 tools/deflate_run.py > $E/deflate_piece_hip_events.txt 2> /dev/null
  stats deflate_piece python3 tools/deflate_run.py
  pmc deflate_piece python3 tools/deflate_run.py ;;
c4)
  python3 bench.py --workload C4 --steps 10 --warmup 3 --no-cpu > $E/bench_C4_all_plans.json 2> /dev/null
  python3 bench.py --workload C4 --algo kernel --steps 10 --warmup 3 --no-cpu --no-alt > $E/bench_C4_kernel_plan.json 2> /dev/null
  stats c4_nipals python3 bench.py --workload C4 --steps 3 --warmup 1 --no-alt --no-cpu
  pmc C4_nipals_fused python3 bench.py --workload C4 --steps 2 --warmup 1 --no-alt --no-cpu
  stats c4_kernel python3 bench.py --workload C4 --algo kernel --steps 3 --warmup 1 --no-alt --no-cpu
  pmc C4_kernel_fused python3 bench.py --workload C4 --algo kernel --steps 2 --warmup 1 --no-alt --no-cpu
  python3 tools/xty_m8.py > $E/xty_m8.txt 2> /dev/null ;;
c5)
  python3 bench.py --workload C5rank --steps 5 --warmup 2 --no-cpu --no-alt > $E/bench_C5_one_shard.json 2> /dev/null
  python3 bench.py --workload C5rank --algo kernel --steps 5 --warmup 2 --no-cpu --no-alt > $E/bench_C5_one_shard_kernel_plan.json 2> /dev/null
  stats c5rank_nipals python3 bench.py --workload C5rank --steps 2 --warmup 1 --no-alt --no-cpu
  pmc C5rank_nipals_fused python3 bench.py --workload C5rank --steps 1 --warmup 1 --no-alt --no-cpu ;;
eighth)
  python3 bench.py --workload C3eighth --steps 20 --warmup 5 --no-cpu --no-alt > $E/bench_C3eighth.json 2> /dev/null
  stats c3eighth python3 bench.py --workload C3eighth --steps 10 --warmup 3 --no-alt --no-cpu ;;
edge)
  # the shapes that used to fall off the one-sweep kernels, next to their aligned twins (round 3)
  for wl in C3 C3odd- C3odd+ tall64 tall64a C4 C4odd; do for algo in nipals kernel; do
    python3 bench.py --workload $wl --algo $algo --steps 10 --warmup 3 --no-cpu --no-alt > $E/edge_bench_${wl}_${algo}.json 2> /dev/null
  done; done
  python3 - <<'PY'
import glob, json, os
E = "gpurun_out/evidence"
rows = []
for f in sorted(glob.glob(E + "/edge_bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception:
        continue
    r = d["roofline"]
    rows.append((os.path.basename(f)[11:-5], d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"], r["families_ms_per_fit"]))
with open(E + "/edge_shapes_summary.txt", "w") as out:
    out.write("workload_plan  components/s  ms/fit  dominant family: avg launch ms, fraction of 8 TB/s; family ms per fit\n")
    for r in rows:
        out.write("%-18s %9.1f %9.3f  %s %.4f ms %.4f  %s\n" % r)
PY
  stats c3odd_kernel python3 bench.py --workload C3odd- --algo kernel --steps 3 --warmup 1 --no-alt --no-cpu
  pmc C3odd_kernel python3 bench.py --workload C3odd- --algo kernel --steps 2 --warmup 1 --no-alt --no-cpu
  stats c3odd_nipals python3 bench.py --workload C3odd- --steps 3 --warmup 1 --no-alt --no-cpu
  ./pls_amd/csrc/tune/unaligned_probe > $E/unaligned_probe.txt 2>&1 ;;
narrow)
  # narrow matrices (taller tiles) and 9-32 responses on many columns (the previous forms, measured in round 3, are deleted)
  for wl in narrow32 tall64a narrow128 narrow256; do for algo in nipals kernel; do
    python3 bench.py --workload $wl --algo $algo --steps 8 --warmup 3 --no-cpu --no-alt > $E/narrow_bench_${wl}_${algo}.json 2> /dev/null
  done; done
  for wl in C4m16 C4m32; do
    python3 bench.py --workload $wl --algo kernel --steps 3 --warmup 2 --no-cpu --no-alt > $E/narrow_bench_${wl}_kernel.json 2> /dev/null
  done
  python3 - <<'PY'
import glob, json, os
E = "gpurun_out/evidence"
with open(E + "/narrow_and_many_responses_summary.txt", "w") as out:
    out.write("workload_plan[_variant]  components/s  ms/fit  dominant family: avg launch ms, fraction of 8 TB/s\n")
    for f in sorted(glob.glob(E + "/narrow_bench_*.json")):
        try:
            d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        except Exception:
            continue
        r = d["roofline"]
        out.write("%-46s %9.1f %9.3f  %s %.4f ms %.4f\n" % (os.path.basename(f)[13:-5], d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"], r["frac"]))
PY
  ;;
wide)
  # beyond 4096 columns (row-pack tiles) and few-rows-many-columns shapes, with the previous forms beside them; z-scores
  for wl in wide8k wide6k64 wide8k64; do for algo in kernel nipals; do
    python3 bench.py --workload $wl --algo $algo --steps 6 --warmup 2 --no-cpu --no-alt > $E/wide512_bench_${wl}_${algo}.json 2> /dev/null
  done; done
  for wl in wide16k wide12k64; do
    python3 bench.py --workload $wl --algo kernel --steps 6 --warmup 2 --no-cpu --no-alt > $E/wide512_bench_${wl}_kernel.json 2> /dev/null
  done
  for wl in shortwide genes genes4; do for algo in kernel nipals; do
    python3 bench.py --workload $wl --algo $algo --steps 6 --warmup 2 --no-cpu --no-alt > $E/shortwide_bench_${wl}_${algo}.json 2> /dev/null
  done; done
  rm -f $E/zscore_time_raw.txt
  for shape in "1048576 512" "1048576 512 f32" "4000000 64"; do
    python3 tools/zscore_time.py $shape 2> /dev/null >> $E/zscore_time_raw.txt
  done ;;
group)
  # the two in-process exchanges with virtual members (one hardware queue per member), the cross-process exchange with
  # ranks sharing the GPU, config 5 at its own size
  export GPU_MAX_HW_QUEUES=16
  for mode in host device; do
    PLS_HIP_GROUP_EXCHANGE=$mode python3 tools/group_overhead.py 131072 512 20 2> /dev/null | grep members > $E/group_overhead_eighth_$mode.txt
    PLS_HIP_GROUP_EXCHANGE=$mode python3 tools/group_overhead.py 2> /dev/null | grep members > $E/group_overhead_C3_$mode.txt
  done
  unset GPU_MAX_HW_QUEUES
  bash tools/r3_ipc_bench.sh > $E/ipc_exchange_shared_gpu.txt 2>&1
  cp gpurun_out/r3/ipc_bench_*.json $E/ 2> /dev/null
  python3 -m pytest tests/test_gpu_configs.py -q -m gpu -k own_size -s 2>&1 | grep -E "config 5|passed|failed" > $E/config5_full_size_one_gpu.txt ;;
fuzz)
  python3 tools/fuzz_parity.py 1500 3 > $E/fuzz_parity.txt 2>&1 ;;
host)
  ./tools/host_entry_time 1048576 512 1 20 5 > $E/host_entry_time_C3.txt 2>&1
  PLS_HIP_ALGO=kernel ./tools/host_entry_time 1048576 512 1 20 3 > $E/host_entry_time_C3_kernel_plan.txt 2>&1
  ./tools/h2d_probe > $E/h2d_probe.txt 2>&1 ;;
small)
  python3 tools/small_fit_time.py $E/small_fits.json > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $E/tmp_small -o p -- python3 tools/small_fit_time.py /dev/null > /dev/null 2> $E/tmp_small.err
  cp $E/tmp_small/p_kernel_stats.csv $E/small_fits_kernel_stats.csv; rm -rf $E/tmp_small $E/tmp_small.err ;;
roctx)
  PLS_HIP_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $E/tmp_roctx -o p -- python3 tools/roctx_demo.py > /dev/null 2> $E/tmp_roctx.err
  python3 - <<'PY'
import csv, glob
E = "gpurun_out/evidence"
rows = list(csv.DictReader(open(glob.glob(E + "/tmp_roctx/p_marker_api_trace.csv")[0])))
ker = list(csv.DictReader(open(glob.glob(E + "/tmp_roctx/p_kernel_trace.csv")[0])))
with open(E + "/roctx_marker_trace.txt", "w") as f:
    f.write("rocprofv3 --marker-trace --kernel-trace of tools/roctx_demo.py with PLS_HIP_ROCTX=1: %d marker records, %d kernel dispatches\n" % (len(rows), len(ker)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    for r in rows[:60]:
        f.write("%10.1f us  %-8s %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, r.get("Function", ""), r.get("Message", r.get("Name", ""))))
PY
  rm -rf $E/tmp_roctx $E/tmp_roctx.err ;;
esac; echo "done $part"; done
ls $E
