# phase bisect of coop_update_kernel (needs the temporary `stop` hooks; not part of the product build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in 2 3 30 31 4 5 6 7 0; do
  PLS_HIP_COOP_STOP=$s rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stop$s -o p -- python3 bench.py --workload C4 --algo kernel --steps 2 --warmup 1 --no-cpu --no-alt > /dev/null 2>&1
  python3 -c "
import csv
for r in csv.DictReader(open('gpurun_out/prof_stop$s/p_kernel_stats.csv')):
    if 'coop_update' in r['Name']: print('stop=$s', r['Calls'], r['AverageNs'])
"
  rm -rf gpurun_out/prof_stop$s
done
