#!/bin/bash
# the stand-alone tuner and the library's xb on the same box, back to back: is the library slower than the bare kernel?
mkdir -p gpurun_out/r5
{
timeout -k 10 100 pls_amd/csrc/tune/xb4_tune | sed -n 1,4p
timeout -k 10 120 python tools/probe/xb_one.py
PLS_HIP_XB4=0 timeout -k 10 120 python tools/probe/xb_one.py
timeout -k 10 100 pls_amd/csrc/tune/xb4_tune | sed -n 1,4p
} 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5/xb_pair.txt
