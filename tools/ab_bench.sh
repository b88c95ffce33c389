#!/bin/bash
# A/B of two builds of the library inside ONE gpurun call (boxes differ by up to 15 %):  bash tools/ab_bench.sh <old.so> [reps]
# prints ms_per_step / kernel_ms / mean sweeps of the headline leg for the old and the current library, alternating.
OLD=$1; REPS=${2:-2}
for i in $(seq $REPS); do
  for L in old new; do
    if [ $L = old ]; then export IBS_LIB_PATH=$OLD; else unset IBS_LIB_PATH; fi
    timeout -k 10 200 python bench.py --steps 2000 --warmup 100 --no-cpu --no-stress 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['ms_per_step']*1e3,3), 'us/step  kernel', round(d['roofline'].get('kernel_ms',0)*1e3,3), 'us  sweeps', d['config']['mean_sweeps_per_solve'], ' max|dgam| vs oracle', d['config'].get('max_abs_dgam_vs_oracle'))" || exit 1
  done
done
