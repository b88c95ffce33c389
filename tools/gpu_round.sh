set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
timeout -k 10 600 python tools/bench_p32_long.py > gpurun_out/$T/p32.txt 2>&1; echo "rc=$?"; grep -v amdgpu gpurun_out/$T/p32.txt
