set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
IBS_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 200 --no-cpu --no-stress > gpurun_out/$T/bench_dist1.json 2> gpurun_out/$T/bench_dist1.err; echo "dist1 rc=$?"; tail -3 gpurun_out/$T/bench_dist1.err
tail -1 gpurun_out/$T/bench_dist1.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['n_gpus'], d['ms_per_step'], d['config']['workload'][-90:], d.get('ncsx_c2_sharded',{}).get('ms_per_pass'), d.get('ncsx_c2_sharded',{}).get('gathered_equals_one_gpu_bitwise'))
"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu --no-stress > gpurun_out/$T/bench_tr1.json 2> gpurun_out/$T/bench_tr1.err; echo "torchrun1 rc=$?"; tail -1 gpurun_out/$T/bench_tr1.json | cut -c1-200
