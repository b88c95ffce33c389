set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or scan_plan" > gpurun_out/$T/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest.log
tail -15 gpurun_out/$T/pytest.log
timeout -k 10 300 python bench.py --steps 500 --no-cpu --no-stress 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('step', d['ms_per_step'], 'value', d['value'], 'kern', d['roofline']['kernel_ms'])
"
