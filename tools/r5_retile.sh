#!/bin/bash
# round 5: retile_xty against the placement of the library's copy (offset of the copy inside its allocation; XPAD shifts X itself)
mkdir -p gpurun_out/r5
for rep in 1 2 3 4 5 6; do PLS_HIP_EXP_WORK_PRINT=1 timeout -k 10 100 python3 tools/retile_offset.py 0 2>&1 | grep -v amdgpu | sort -u | tr '\n' ' '; echo; done
for off in 256 4096 65536 131072 262144 524288 1048576 2097152 4194304 8388608 16777216 33554432 1310720 5505024; do
  PLS_HIP_EXP_WORK_PRINT=1 PLS_HIP_EXP_WORK_OFF=$off timeout -k 10 100 python3 tools/retile_offset.py $off 2>&1 | grep -v amdgpu | sort -u | tr '\n' ' '; echo
done
