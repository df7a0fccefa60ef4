#!/bin/bash
# Round-2 GPU session script (run through gpurun): A/B of kernel variants, GPU test-suite, bench lines.  Logs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt; nproc >> gpurun_out/gpu.txt
for STAGE in "$@"; do
  case $STAGE in
    attn_ab) timeout 300 python scripts/debug/attn_ab.py > gpurun_out/attn_ab.log 2>&1; echo "attn_ab exit $?"; tail -12 gpurun_out/attn_ab.log;;
    tests) timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rf --tb=short -x 2>&1 | tail -80 > gpurun_out/pytest_gpu.log; tail -40 gpurun_out/pytest_gpu.log;;
    tests_all) timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rf --tb=short 2>&1 | tail -150 > gpurun_out/pytest_gpu.log; tail -60 gpurun_out/pytest_gpu.log;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?"; tail -4 gpurun_out/smoke.log;;
    bench) timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit: $?"; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json;;
    bench_resident) timeout 900 python bench.py --resident --cpu-images 0 --companion-steps 0 > gpurun_out/bench_resident.json 2> gpurun_out/bench_resident.err; echo "bench_resident exit: $?"; cat gpurun_out/bench_resident.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step')})";;
    bench_c3) timeout 900 python bench.py --model dino_vitb8 --K 15 --batch 512 --vit-batch 16 --cpu-images 2 --companion-steps 0 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; echo "bench_c3 exit: $?"; tail -3 gpurun_out/bench_c3.err; cat gpurun_out/bench_c3.json;;
    bench_dataset) timeout 900 python bench.py --dataset 10000 --cpu-images 0 > gpurun_out/bench_dataset.json 2> gpurun_out/bench_dataset.err; echo "bench_dataset exit: $?"; tail -3 gpurun_out/bench_dataset.err; cat gpurun_out/bench_dataset.json;;
    bench_spawn2) DSS_DIST_BACKEND=nccl timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --cpu-images 0 > gpurun_out/bench_spawn2.json 2> gpurun_out/bench_spawn2.err; echo "bench_spawn2 exit: $? (expected to fail on a 1-GPU box unless both ranks share the GPU)"; tail -5 gpurun_out/bench_spawn2.err; cat gpurun_out/bench_spawn2.json;;
    bench_spawn2_gloo) DSS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --cpu-images 0 --companion-steps 0 > gpurun_out/bench_spawn2_gloo.json 2> gpurun_out/bench_spawn2_gloo.err; echo "bench_spawn2_gloo exit: $? (two ranks sharing the one GPU, host-side collectives)"; tail -5 gpurun_out/bench_spawn2_gloo.err; cat gpurun_out/bench_spawn2_gloo.json;;
    cli) timeout 240 python scripts/cli_throughput.py ${CLI_N:-4096} > gpurun_out/cli.log 2>&1; echo "cli exit $?"; grep "images/s" gpurun_out/cli.log;;
    cli_threads) DSS_IO_PROCESSES=0 timeout 300 python scripts/cli_throughput.py ${CLI_N:-4096} > gpurun_out/cli_threads.log 2>&1; echo "cli_threads exit $?"; grep "images/s" gpurun_out/cli_threads.log;;
    prof)   # per-kernel time of the bench command (rocprofv3 kernel trace); the summary is copied to profiles/ by hand
      REPO_DIR=$PWD; rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO_DIR/gpurun_out/prof -o bench -- python $REPO_DIR/bench.py --steps 2 --warmup 1 --min-warmup-seconds 0 --cpu-images 0 --companion-steps 0 ${BENCH_ARGS:-} > $REPO_DIR/gpurun_out/prof/bench.json 2> $REPO_DIR/gpurun_out/prof/bench.err)
      echo "prof exit: $?"; python scripts/rocpd_summary.py gpurun_out/prof/bench_results.db > gpurun_out/prof/kernel_stats.csv; head -16 gpurun_out/prof/kernel_stats.csv
      find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete; rm -f gpurun_out/prof/bench_results.db;;
    pmc)    # hardware counters, ONE rocprofv3 pass per group (SQ: 8 slots; FETCH_SIZE / WRITE_SIZE need their own passes);
            # --kernel-trace only, no other trace domains
      REPO_DIR=$PWD
      declare -A PMCG=( [mfma]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
                          [wait]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES"
                          [fetch]="FETCH_SIZE" [write]="WRITE_SIZE" )
      for G in mfma wait fetch write; do
        rm -rf gpurun_out/pmc_$G && mkdir -p gpurun_out/pmc_$G
        (cd /tmp && timeout 900 rocprofv3 --pmc ${PMCG[$G]} --kernel-trace -d $REPO_DIR/gpurun_out/pmc_$G -o pmc -- python $REPO_DIR/bench.py --steps 1 --warmup 1 --min-warmup-seconds 0 --cpu-images 0 --companion-steps 0 ${BENCH_ARGS:-} > $REPO_DIR/gpurun_out/pmc_$G/bench.json 2> $REPO_DIR/gpurun_out/pmc_$G/bench.err)
        echo "pmc $G exit: $?"
        python scripts/rocpd_pmc_multi.py gpurun_out/pmc_$G/pmc_results.db 2 > gpurun_out/pmc_$G/pmc_$G.csv; head -8 gpurun_out/pmc_$G/pmc_$G.csv | cut -c1-260
        [ $(wc -l < gpurun_out/pmc_$G/pmc_$G.csv) -gt 2 ] && rm -f gpurun_out/pmc_$G/pmc_results.db
      done;;
    *) echo "unknown stage $STAGE";;
  esac
done
