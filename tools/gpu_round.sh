set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
IBS_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 500 --no-cpu --no-stress > gpurun_out/$T/bench_nat.json 2> gpurun_out/$T/bench_nat.err; echo "rc=$?"
tail -1 gpurun_out/$T/bench_nat.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['n_gpus'], d['ms_per_step'], d['config']['allgather_roundtrip_ok'], d['ncsx_c2_sharded']['ms_per_pass'], d['ncsx_c2_sharded']['checks_passed'])
"
