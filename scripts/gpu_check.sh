#!/bin/bash
# Run on the GPU box via gpurun: GPU test-suite, smoke, bench (+ optional rocprof).  Logs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
STAGE=${1:-all}
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt
if [[ $STAGE == all || $STAGE == tests ]]; then
  timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
  echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
  tail -25 gpurun_out/pytest_gpu.log
fi
if [[ $STAGE == all || $STAGE == smoke ]]; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" >> gpurun_out/smoke.log
  tail -5 gpurun_out/smoke.log
fi
if [[ $STAGE == all || $STAGE == bench ]]; then
  timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit: $?" >> gpurun_out/bench.err
  tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
fi
