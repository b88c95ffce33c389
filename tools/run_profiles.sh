#!/bin/bash
# Refresh the rocprofv3 evidence of one round on the GPU box:  bash tools/run_profiles.sh r01_f
# (kernel trace of bench.py, then three separate PMC passes of tools/profile_run.py; summaries under gpurun_out/)
set -eo pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 $R/bench.py --steps 100 --no-cpu > $OUT/bench_under_rocprof.log 2>&1
# PMC passes: the bench's own legs at the bench's sizes (entries keyed by kernel and launch size); the refinement, whose
# rounds launch the geometry kernel at many batch sizes, gets its own set
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o fetch -- python3 $R/tools/profile_run.py > $OUT/pmc_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc -o write -- python3 $R/tools/profile_run.py > $OUT/pmc_write.log 2>&1
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o sq -- python3 $R/tools/profile_run.py > $OUT/pmc_sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_refine -o sq -- python3 $R/tools/profile_run.py refine > $OUT/pmc_refine_sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_refine -o refine -- python3 $R/tools/refine_profile_run.py both 3 > $OUT/refine_under_rocprof.log 2>&1
cd $R
cp $OUT/kt_refine/refine_kernel_stats.csv $OUT/${TAG}_kernel_stats_refine.csv
python tools/kt_timeline.py $OUT/kt_refine > $OUT/${TAG}_refine_timeline.txt
# (pmc_summary leaves non-zero when the byte passes could not be matched launch by launch: the set is written, bench.py withholds `traffic`)
python tools/pmc_summary.py $OUT/pmc $OUT/${TAG}_pmc.json $TAG > $OUT/pmc_summary.txt || echo "pmc_summary: unmatched byte passes (see $OUT/pmc_summary.txt)"
python tools/pmc_summary.py $OUT/pmc_refine $OUT/${TAG}_pmc_refine.json $TAG > /dev/null || true
cp $OUT/kt/bench_kernel_stats.csv $OUT/${TAG}_kernel_stats_bench.csv
# the plain run quotes the counters just taken: the new set becomes profiles/pmc_current.json for it (copy both into profiles/ to keep them)
cp $OUT/${TAG}_pmc.json profiles/pmc_current.json
IBS_BENCH_DETAIL=$OUT/${TAG}_bench_detail.json timeout -k 10 300 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
wc -c $OUT/${TAG}_bench.json
head -12 $OUT/${TAG}_kernel_stats_bench.csv
