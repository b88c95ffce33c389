set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
bash tools/run_profiles.sh ${T} > gpurun_out/$T/profiles.log 2>&1; echo "profiles rc=$?"
python tools/bench_c5.py > gpurun_out/$T/c5.txt 2>&1; echo "c5 rc=$?"
IBS_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 200 --warmup 20 > gpurun_out/$T/bench2.json 2> gpurun_out/$T/bench2.err; echo "bench2 rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/$T/bench20.json 2> gpurun_out/$T/bench20.err; echo "bench20 rc=$?"
