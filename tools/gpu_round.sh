set -o pipefail
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log
tail -15 gpurun_out/r02a/pytest.log
python bench.py --steps 1000 --warmup 50 > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --no-cpu --no-stress > gpurun_out/r02a/bench20.json 2> gpurun_out/r02a/bench20.err; echo "bench20 rc=$?"
IBS_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 200 --warmup 20 > gpurun_out/r02a/bench2.json 2> gpurun_out/r02a/bench2.err; echo "bench2 rc=$?"
tail -3 gpurun_out/r02a/bench2.err
