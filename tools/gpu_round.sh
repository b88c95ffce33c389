set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest.log
tail -5 gpurun_out/$T/pytest.log
timeout -k 10 300 python bench.py --steps 500 --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('step', d['ms_per_step'], 'value', d['value'], 'kern', d['roofline']['kernel_ms'], 'sweeps', d['config']['mean_sweeps_per_solve'], 'stress', d['stress']['solves_per_s'], d['stress']['mean_sweeps'], d['stress_rough']['solves_per_s'], 'c3', d['ncsx_c3']['scan_ms'], d['ncsx_c3']['mean_sweeps'], 'refb', d['reference_batch']['scan_ms'], d['reference_batch']['refine_ms'], d['reference_batch']['mean_sweeps'], 'large', d['scan_large']['solves_per_s'])
"
