#!/bin/bash
# round 5: the single-launch fits (configs 1 and 2) against the round-4 library on the same box
R4=$GRAFT_REPO_ROOT/build/r4/libpls_hip.so
for lib in r4 r5 r4 r5; do
  if [ $lib = r4 ]; then export PLS_AMD_LIBRARY=$R4; else unset PLS_AMD_LIBRARY; fi
  timeout -k 10 200 python3 tools/small_fit_time.py /tmp/sf_$lib.json > /dev/null 2>&1 || exit 1
  python3 -c "
import json; d=json.load(open('/tmp/sf_$lib.json'))
print('$lib', {k: (v['single_launch']['back_to_back_us'], v['single_launch']['latency_us']) for k, v in d.items()})"
done
