mkdir -p gpurun_out/r5
run() { echo "#### $*"; timeout -k 10 200 python tools/pass_stamps.py "$@" 2>&1 | grep -v "amdgpu.ids\|^start\|tail entered\|^exit\|still in their" || exit 1; }
run 131072 nipals && run 131072 kernel && run 1048576 kernel && run 131072 nipals 4096 8 10 f32 && run 131072 kernel 4096 8 10 f32 && run 262144 nipals 1024 4 6 && run 262144 kernel 1024 4 6 && run 131072 nipals 512 1 20 f32 && run 1048576 nipals 64 1 6 && run 131072 nipals
