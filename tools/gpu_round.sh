set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
python tests/tools/refine_trace_compare.py > gpurun_out/$T/refine_compare.log 2>&1; echo "compare rc=$?"
IBS_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 200 --warmup 20 > gpurun_out/$T/bench2.json 2> gpurun_out/$T/bench2.err; echo "bench2 rc=$?"
tail -3 gpurun_out/$T/bench2.err
bash tools/run_profiles.sh ${T} > gpurun_out/$T/profiles.log 2>&1; echo "profiles rc=$?"
