set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest.log
tail -25 gpurun_out/$T/pytest.log
python bench.py --steps 200 --no-cpu --no-stress 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('step', d['ms_per_step'], 'value', d['value'])
"
