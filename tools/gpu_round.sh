set -o pipefail
T=${1:-r02a}
mkdir -p gpurun_out/$T
bash tools/run_profiles.sh ${T} > gpurun_out/$T/profiles.log 2>&1; echo "profiles rc=$?"
python bench.py --steps 20 --warmup 5 > gpurun_out/$T/bench20.json 2> gpurun_out/$T/bench20.err; echo "bench20 rc=$?"
python tools/bench_pipeline.py > gpurun_out/$T/pipeline.txt 2>&1; echo "pipeline rc=$?"; tail -4 gpurun_out/$T/pipeline.txt
