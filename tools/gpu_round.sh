set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
python -m pytest tests/test_gpu_configs.py -m gpu -q --durations=8 > gpurun_out/$T/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest.log
tail -40 gpurun_out/$T/pytest.log
