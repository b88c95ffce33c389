set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$T/pytest.log
tail -12 gpurun_out/$T/pytest.log
python bench.py --steps 300 --no-cpu 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"value\"], d[\"ms_per_step\"], d[\"roofline\"][\"kernel_ms\"], d[\"stress\"][\"solves_per_s\"])"
