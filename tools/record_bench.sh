#!/bin/bash
# The bench records of one round, on the GPU box:  bash tools/record_bench.sh r06a
#   1. `python bench.py` (the default run): the ONE stdout line, the full record beside it
#   2. the driver's command (`--gpus 1 --steps 20 --warmup 5`)
#   3. the N > 1 code path over RCCL with a one-rank group (IBS_BENCH_FORCE_DIST=1)
#   4. `--gpus 2` on the one GPU with the shared-memory stand-in for librccl (tests/cabi/fake_rccl.c) bound as the library's RCCL:
#      the library's in-stream and overlapped gathers with two ranks in the collective
# Results under gpurun_out/bench_TAG/; copy what is to be kept into profiles/.
set -eo pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/bench_$TAG
mkdir -p $OUT
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
IBS_BENCH_DETAIL=$OUT/${TAG}_bench_detail.json python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/bench_default.err
echo "default run done: $(wc -c < $OUT/${TAG}_bench_line.json) bytes"
IBS_BENCH_DETAIL=$OUT/${TAG}_bench_driver_command_detail.json python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_command_line.json 2> $OUT/bench_driver.err
echo "driver command done"
IBS_BENCH_DETAIL=$OUT/${TAG}_rehearsal_rccl_1rank.json IBS_BENCH_FORCE_DIST=1 python bench.py --steps 200 --no-stress --no-cpu > $OUT/rehearsal_1rank_line.json 2> $OUT/rehearsal_1rank.err
echo "one-rank RCCL rehearsal done"
gcc -O1 -shared -fPIC -I ${ROCM_PATH:-/opt/rocm}/include tests/cabi/fake_rccl.c -o $OUT/libfake_rccl.so -lrt -lpthread
IBS_BENCH_DETAIL=$OUT/${TAG}_rehearsal_standin_2ranks.json IBS_BENCH_SHARE_GPU=1 IBS_RCCL_LIB=$OUT/libfake_rccl.so FAKE_RCCL_TIMEOUT_S=60 \
  python bench.py --gpus 2 --steps 200 --warmup 20 --no-stress --no-cpu > $OUT/rehearsal_2ranks_line.json 2> $OUT/rehearsal_2ranks.err
rm -f $OUT/libfake_rccl.so
echo "two-rank stand-in rehearsal done"
