# the block form of the one-launch X^T X fit's phase 1 against the row form (PLS_HIP_RESIDENT_GRAM=4), shape by shape
for s in "5000 128 1 10" "8000 100 1 5" "10000 64 1 8" "5000 128 4 10" "4000 48 1 5" "20000 64 2 6" "100000 64 1 8" "3001 77 1 7 f32" "50000 100 8 10"; do
  timeout -k 10 120 python tools/probe/rg_one.py $s || exit 1
  PLS_HIP_RESIDENT_GRAM=4 timeout -k 10 120 python tools/probe/rg_one.py $s | sed 's/^/   row form: /' || exit 1
done
