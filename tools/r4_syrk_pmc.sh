#!/bin/bash
# round 4: the memory side of the SYRK (X^T X on the fp64 matrix cores): HBM bytes, L2 hit rate, LDS bank conflicts and
# where the waves wait -- each counter set in a pass of its own (rocprofv3: program directly after --)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4/syrk_pmc; mkdir -p $O
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/tmp_$name -o p -- python3 tools/syrk_run.py > /dev/null 2> $O/$name.err
  cp $O/tmp_$name/p_counter_collection.csv $O/$name.csv 2>/dev/null; rm -rf $O/tmp_$name
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE GRBM_GUI_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
pass sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64
python3 tools/pmc_table.py syrk $O/fetch.csv $O/write.csv $O/tcc.csv $O/sq1.csv $O/sq2.csv > $O/../syrk_memory_side_pmc.txt 2>&1
cat $O/../syrk_memory_side_pmc.txt; tail -3 $O/*.err | head -40
