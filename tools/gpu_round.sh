set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "native_rccl or new_entry" > gpurun_out/$T/pt.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/$T/pt.log
for ov in 1 0; do
IBS_BENCH_OVERLAP=$ov IBS_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 2000 --no-cpu --no-stress > gpurun_out/$T/bench_ov$ov.json 2> gpurun_out/$T/bench_ov$ov.err; echo "rc=$?"
tail -1 gpurun_out/$T/bench_ov$ov.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('overlap=$ov', d['value'], d['n_gpus'], d['ms_per_step'], d['config']['allgather_roundtrip_ok'], d['config']['workload'][-90:], d['ncsx_c2_sharded']['ms_per_pass'], d['ncsx_c2_sharded']['checks_passed'])
"
done
