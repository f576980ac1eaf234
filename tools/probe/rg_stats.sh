# rocprofv3 kernel stats of the one-launch mid-size fits (AUTO), one shape per run: bash tools/probe/rg_stats.sh (on the GPU box)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/rg && cd /tmp && export TMPDIR=/tmp
for s in "5000 128 1 10" "1025 26 1 5" "100000 40 8 12"; do
  tag=$(echo $s | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rg/$tag -o p -- python3 $R/tools/probe/rg_one.py $s > $R/gpurun_out/rg/$tag.txt 2> $R/gpurun_out/rg/$tag.err
  cp $R/gpurun_out/rg/$tag/p_kernel_stats.csv $R/gpurun_out/rg/${tag}_kernel_stats.csv
  rm -rf $R/gpurun_out/rg/$tag $R/gpurun_out/rg/$tag.err
  tail -1 $R/gpurun_out/rg/$tag.txt
  head -3 $R/gpurun_out/rg/${tag}_kernel_stats.csv | cut -c1-200
done
