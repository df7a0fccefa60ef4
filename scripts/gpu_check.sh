#!/bin/bash
# Run on the GPU box via gpurun: GPU test-suite, smoke, bench (+ optional rocprof).  Logs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
STAGE=${1:-all}
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt
if [[ $STAGE == all || $STAGE == tests ]]; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -rf --tb=short 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
  echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
  tail -60 gpurun_out/pytest_gpu.log
fi
if [[ $STAGE == all || $STAGE == smoke ]]; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" >> gpurun_out/smoke.log
  tail -5 gpurun_out/smoke.log
fi
if [[ $STAGE == all || $STAGE == bench ]]; then
  timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit: $?" >> gpurun_out/bench.err
  tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json
fi
if [[ $STAGE == prof ]]; then
  # per-kernel time of the bench command (rocprofv3 kernel trace + stats); summaries are copied to profiles/ by hand
  REPO_DIR=$PWD
  rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO_DIR/gpurun_out/prof -o bench -- python $REPO_DIR/bench.py --steps 2 --warmup 1 --cpu-images 0 ${BENCH_ARGS:-} > $REPO_DIR/gpurun_out/prof/bench.json 2> $REPO_DIR/gpurun_out/prof/bench.err)
  echo "prof exit: $?"; cat gpurun_out/prof/bench.json; find gpurun_out/prof -name "*stats*" | head
  # keep only the small summaries (the raw kernel trace is large)
  find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
  python scripts/rocpd_summary.py gpurun_out/prof/bench_results.db > gpurun_out/prof/kernel_stats.csv; head -14 gpurun_out/prof/kernel_stats.csv
fi
if [[ $STAGE == pmc ]]; then
  # HBM traffic counters, one rocprofv3 pass per counter (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2;
  # MI355X_MICROARCH.md "rocprofv3 PMC slots"); kernel-trace only, no other trace domains.
  REPO_DIR=$PWD
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_$C && mkdir -p gpurun_out/pmc_$C
    (cd /tmp && timeout 900 rocprofv3 --pmc $C --kernel-trace -d $REPO_DIR/gpurun_out/pmc_$C -o pmc -- python $REPO_DIR/bench.py --steps 2 --warmup 1 --cpu-images 0 ${BENCH_ARGS:-} > $REPO_DIR/gpurun_out/pmc_$C/bench.json 2> $REPO_DIR/gpurun_out/pmc_$C/bench.err)
    echo "pmc $C exit: $?"
    python scripts/rocpd_pmc.py gpurun_out/pmc_$C/pmc_results.db $C > gpurun_out/pmc_$C/pmc_$C.csv; head -12 gpurun_out/pmc_$C/pmc_$C.csv
    [ -s gpurun_out/pmc_$C/pmc_$C.csv ] && [ $(wc -l < gpurun_out/pmc_$C/pmc_$C.csv) -gt 2 ] && rm -f gpurun_out/pmc_$C/pmc_results.db
  done
fi
