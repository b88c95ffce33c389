#!/bin/bash
# Host code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU box; GPU sanitizers are not available on this pool):
#   * the C oracle (oracle/ibs_oracle.c) rebuilt with -fsanitize=address,undefined, driven by its ctypes marshalling tests
#   * the bounded quasi-Newton state machine (csrc/ibs_lbfgsb2.hpp) compiled for the host with the same flags and driven
#     through a tiny C harness over the test functions of tests/test_lbfgsb2.py
# usage: bash tools/run_sanitizers.sh      (exit code 0 = clean)
set -eo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
make -C $R/oracle asan > /dev/null
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
cd $R
LD_PRELOAD="$ASAN $UBSAN" IBS_ORACLE_SO=$R/oracle/_build/libibs_oracle_asan.so OMP_NUM_THREADS=2 \
  python -m pytest tests/test_oracle_c.py -x -q -p no:cacheprovider
# the optimizer state machine, host build
mkdir -p /tmp/ibs_san
cat > /tmp/ibs_san/harness.cpp <<'CPP'
#include <cstdio>
#include <cmath>
#include "ibs_lbfgsb2.hpp"
using namespace ibs::lbfgsb2;
static void fun(int k, const double* x, double& f, double* g) {
  if (k == 0) { double u = x[0], v = x[1]; f = 100 * (v - u * u) * (v - u * u) + (1 - u) * (1 - u); g[0] = -400 * u * (v - u * u) - 2 * (1 - u); g[1] = 200 * (v - u * u); }
  else if (k == 1) { f = 1e-3 * (std::sin(2.1 * x[0] + 0.3) + std::cos(1.7 * x[1]) + 0.3 * std::sin(x[0] * x[1])); g[0] = 1e-3 * (2.1 * std::cos(2.1 * x[0] + 0.3) + 0.3 * x[1] * std::cos(x[0] * x[1])); g[1] = 1e-3 * (-1.7 * std::sin(1.7 * x[1]) + 0.3 * x[0] * std::cos(x[0] * x[1])); }
  else { f = 0.5 * (x[0] - 5) * (x[0] - 5) + 0.5 * (x[1] + 1) * (x[1] + 1); g[0] = x[0] - 5; g[1] = x[1] + 1; }
}
int main() {
  const double lo[2] = {0, 0}, hi[2] = {3.141592653589793, 1.5707963267948966};
  for (int k = 0; k < 3; ++k)
    for (int s = 0; s < 25; ++s) {
      State st;
      const double x0[2] = {0.13 * s, 0.06 * s};
      init(st, x0, lo, hi, 5e-11, 2e-8, 30, 20);
      double f, g[2]; int n = 0;
      do { fun(k, st.x, f, g); ++n; } while (step(st, f, g) && n < 2000);
      if (!(n < 2000) || !std::isfinite(st.f)) { std::printf("FAIL k=%d s=%d\n", k, s); return 1; }
    }
  std::printf("lbfgsb2 host harness clean\n");
  return 0;
}
CPP
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    -I $R/ideal-ballooning-solver_amd/csrc /tmp/ibs_san/harness.cpp -o /tmp/ibs_san/harness
/tmp/ibs_san/harness
echo "sanitizers: clean"
