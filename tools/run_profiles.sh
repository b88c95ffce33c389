#!/bin/bash
# Refresh the rocprofv3 evidence of one round on the GPU box:  bash tools/run_profiles.sh r05a [a|b|all]
#   part a: kernel trace of bench.py + the FETCH_SIZE / WRITE_SIZE passes of tools/profile_run.py
#   part b: the SQ passes (bench legs, refinement), the refinement's kernel trace
# (two gpurun calls of <= 20 min each; the CSVs come back under gpurun_out/prof_TAG/ and are summarised HERE -- no GPU needed --
#  by `bash tools/run_profiles.sh TAG summary`, which writes profiles/TAG_* and profiles/pmc_current.json)
set -eo pipefail
TAG=${1:-rXX}
PART=${2:-all}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
SQ="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE"
if [ $PART = a ] || [ $PART = all ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 $R/bench.py --steps 100 --no-cpu > $OUT/bench_under_rocprof.log 2>&1
  echo "kernel trace done"
  # PMC passes: the bench's own legs at the bench's sizes (entries keyed by kernel and launch size)
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -o fetch -- python3 $R/tools/profile_run.py > $OUT/pmc_fetch.log 2>&1
  echo "fetch pass done"
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc -o write -- python3 $R/tools/profile_run.py > $OUT/pmc_write.log 2>&1
  echo "write pass done"
  cd $R
  cp $OUT/kt/bench_kernel_stats.csv $OUT/${TAG}_kernel_stats_bench.csv
fi
if [ $PART = b ] || [ $PART = all ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 400 rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc -o sq -- python3 $R/tools/profile_run.py > $OUT/pmc_sq.log 2>&1
  echo "sq pass done"
  # the refinement, whose rounds launch the geometry kernel at many batch sizes, gets its own set
  timeout -k 10 300 rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc_refine -o sq -- python3 $R/tools/profile_run.py refine > $OUT/pmc_refine_sq.log 2>&1
  echo "refine sq pass done"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_refine -o refine -- python3 $R/tools/refine_profile_run.py both 3 > $OUT/refine_under_rocprof.log 2>&1
  cd $R
  cp $OUT/kt_refine/refine_kernel_stats.csv $OUT/${TAG}_kernel_stats_refine.csv
  python tools/kt_timeline.py $OUT/kt_refine > $OUT/${TAG}_refine_timeline.txt
fi
if [ $PART = summary ] || [ $PART = all ]; then
  cd $R
  # (pmc_summary leaves non-zero when the byte passes could not be matched launch by launch: the set is written, bench.py withholds `traffic`)
  python tools/pmc_summary.py $OUT/pmc $OUT/${TAG}_pmc.json $TAG > $OUT/pmc_summary.txt || echo "pmc_summary: unmatched byte passes (see $OUT/pmc_summary.txt)"
  python tools/pmc_summary.py $OUT/pmc_refine $OUT/${TAG}_pmc_refine.json $TAG > /dev/null || true
  for f in ${TAG}_pmc.json ${TAG}_pmc_refine.json ${TAG}_kernel_stats_bench.csv ${TAG}_kernel_stats_refine.csv ${TAG}_refine_timeline.txt; do
    [ -f $OUT/$f ] && cp $OUT/$f profiles/$f
  done
  cp $OUT/${TAG}_pmc.json profiles/pmc_current.json
  head -12 $OUT/${TAG}_kernel_stats_bench.csv
fi
