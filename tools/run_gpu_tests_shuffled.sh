#!/bin/bash
# state-dependence check: the GPU tests in reversed and in a seeded random order (pytest keeps command-line order)
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$PWD}
python -m pytest tests -m gpu --collect-only -q 2>/dev/null | grep "::" > /tmp/gpu_ids.txt
echo "collected $(wc -l < /tmp/gpu_ids.txt) tests"
tac /tmp/gpu_ids.txt > /tmp/gpu_ids_rev.txt
python - <<'PY'
import random
ids = open('/tmp/gpu_ids.txt').read().split()
random.Random(7).shuffle(ids)
open('/tmp/gpu_ids_shuf.txt', 'w').write("\n".join(ids))
PY
timeout -k 10 400 python -m pytest -q -p no:cacheprovider $(cat /tmp/gpu_ids_rev.txt) 2>&1 | tail -3
timeout -k 10 400 python -m pytest -q -p no:cacheprovider $(cat /tmp/gpu_ids_shuf.txt) 2>&1 | tail -3
