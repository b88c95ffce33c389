set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
IBS_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 200 --no-cpu --no-stress > gpurun_out/$T/bench_dist1.json 2> gpurun_out/$T/bench_dist1.err; echo "dist1 rc=$?"
tail -1 gpurun_out/$T/bench_dist1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['n_gpus'], d['ms_per_step'], d['config']['allgather_roundtrip_ok'], d.get('ncsx_c2_sharded'))
"
IBS_BENCH_SHARE_GPU=1 timeout -k 10 300 python bench.py --gpus 3 --steps 100 --warmup 10 > gpurun_out/$T/bench3.json 2> gpurun_out/$T/bench3.err; echo "bench3 rc=$?"
tail -1 gpurun_out/$T/bench3.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['n_gpus'], d['ms_per_step'], d['config']['allgather_roundtrip_ok'], {k:v for k,v in d.get('ncsx_c2_sharded').items() if k!='workload'})
"
