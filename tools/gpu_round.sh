set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "reject_bad or native_rccl or scan_plan" > gpurun_out/$T/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/$T/pytest.log
