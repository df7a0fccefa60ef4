#!/bin/bash
# Round-3 GPU session script (run through gpurun): kernel labs, PMC passes, GPU test-suite, bench lines.  Logs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
REPO_DIR=$PWD
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt; nproc >> gpurun_out/gpu.txt
pmc_pass() {   # pmc_pass <tag> <counters...> -- <command...>
  local tag=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rm -rf gpurun_out/pmc_$tag && mkdir -p gpurun_out/pmc_$tag
  (cd /tmp && timeout 300 rocprofv3 --pmc "${ctrs[@]}" --kernel-trace -d $REPO_DIR/gpurun_out/pmc_$tag -o pmc -- "$@" > $REPO_DIR/gpurun_out/pmc_$tag/run.log 2>&1)
  local db=$(find gpurun_out/pmc_$tag -name "*.db" | head -1)
  if [ -n "$db" ]; then python scripts/rocpd_pmc_multi.py $db 2 > gpurun_out/pmc_$tag.csv; rm -rf gpurun_out/pmc_$tag; head -4 gpurun_out/pmc_$tag.csv | cut -c1-400; else echo "pmc $tag: no db"; tail -5 gpurun_out/pmc_$tag/run.log; fi
}
for STAGE in "$@"; do
  case $STAGE in
    lab1)   # main shape only, both builds; LAB_ONLY=<variant> restricts
      rm -f gpurun_out/attn_lab1.log
      for BIN in attn_lab; do
        echo "--- $BIN" >> gpurun_out/attn_lab1.log
        timeout 120 scripts/probes/$BIN ${LAB_SHAPE:-290 901 6 1} ${LAB_ONLY:--1} 5 2>&1 | grep -v "ablation / not expected" >> gpurun_out/attn_lab1.log; echo "$BIN exit $?"
      done
      cat gpurun_out/attn_lab1.log;;
    lab_clock) timeout 120 scripts/probes/attn_lab_clock ${LAB_SHAPE:-290 901 6 1} ${LAB_ONLY:--1} 3 2>&1 | grep -v "vs product" > gpurun_out/attn_lab_clock.log; echo "lab_clock exit $?"; cat gpurun_out/attn_lab_clock.log;;
    lab_pmc)   # counters of single variants (LAB_VARIANTS="0 8"), shape 290 x 901 x 6 planar
      for V in ${LAB_VARIANTS:-0 8}; do
        pmc_pass lds_v$V SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT -- $REPO_DIR/scripts/probes/attn_lab 290 901 6 1 $V 2
        pmc_pass wait_v$V SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE -- $REPO_DIR/scripts/probes/attn_lab 290 901 6 1 $V 2
        pmc_pass l2_v$V TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -- $REPO_DIR/scripts/probes/attn_lab 290 901 6 1 $V 2
      done;;
    eigs_lab)   # product first (reference results), then every lab build in LAB_LIBS (scripts/build_lablib.sh tags)
      rm -f gpurun_out/eigs_lab.log
      timeout 300 python scripts/debug/eigs_lab.py --tag product --save /tmp/eigs_ref.npz ${EIGS_LAB_ARGS:-} 2>&1 | tail -1 >> gpurun_out/eigs_lab.log
      for T in ${LAB_LIBS:-}; do
        DSS_HIP_LIBRARY=$REPO_DIR/scripts/lablib/libdss_hip_$T.so timeout 300 python scripts/debug/eigs_lab.py --tag $T --ref /tmp/eigs_ref.npz ${EIGS_LAB_ARGS:-} 2>&1 | tail -1 >> gpurun_out/eigs_lab.log
      done
      cat gpurun_out/eigs_lab.log;;
    host_enqueue) timeout 600 python scripts/debug/host_enqueue.py > gpurun_out/host_enqueue.log 2>&1; echo "host_enqueue exit $?"; cat gpurun_out/host_enqueue.log;;
    tests) timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -rf --tb=short -x 2>&1 | tail -80 > gpurun_out/pytest_gpu.log; tail -40 gpurun_out/pytest_gpu.log;;
    tests_all) timeout 1800 python -m pytest tests -m gpu -q --timeout 900 -rf --tb=short 2>&1 | tail -150 > gpurun_out/pytest_gpu.log; tail -60 gpurun_out/pytest_gpu.log;;
    tests_attn) timeout 900 python -m pytest tests -m gpu -q --timeout 600 -rf --tb=short -k "attention or vit" 2>&1 | tail -40 > gpurun_out/pytest_attn.log; tail -25 gpurun_out/pytest_attn.log;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?"; tail -4 gpurun_out/smoke.log;;
    bench_quick) timeout 600 python bench.py --cpu-images 0 --dino-like-steps 0 --companion-steps 0 --distinct 256 ${BENCH_ARGS:-} 2> gpurun_out/bench_quick.err | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d[k] for k in ('value', 'ms_per_step', 'host_enqueue_ms_per_step', 'host_in_loop_ms_per_step')})
for k, v in d['kernels'].items(): print(k, v.get('launches'), v.get('avg_ms'), v.get('frac'), v.get('passes_per_image', ''))
"; tail -2 gpurun_out/bench_quick.err;;
    bench_ds) for DS in ${DATASETS:-1250 10000}; do timeout 600 python bench.py --dataset $DS --cpu-images 0 --dino-like-steps 0 --companion-steps 0 2> gpurun_out/bench_ds$DS.err > gpurun_out/bench_ds$DS.json; python -c "
import sys, json
d = json.loads(open('gpurun_out/bench_ds$DS.json').read())
print('dataset $DS:', {k: d[k] for k in ('value', 'ms_per_step', 'steps', 'scaling')}, d['config']['workload'])
"; done;;
    bench) timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit: $?"; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json;;
    bench_c3) timeout 900 python bench.py --model dino_vitb8 --K 15 --batch 512 --vit-batch 16 --cpu-images 2 --parity-images 2 --companion-steps 0 --dino-like-steps 0 > gpurun_out/bench_c3.json 2> gpurun_out/bench_c3.err; echo "bench_c3 exit: $?"; tail -2 gpurun_out/bench_c3.err; cut -c1-600 gpurun_out/bench_c3.json;;
    bench_c1) timeout 600 python bench.py --size 224 --cpu-images 8 --parity-images 8 --companion-steps 0 --dino-like-steps 0 > gpurun_out/bench_c1.json 2> gpurun_out/bench_c1.err; echo "bench_c1 exit: $?"; tail -2 gpurun_out/bench_c1.err; cut -c1-400 gpurun_out/bench_c1.json;;
    prof)   # per-kernel time of the bench command (rocprofv3 kernel trace + stats); PROF_TAG names the output
      T=${PROF_TAG:-c2}; rm -rf gpurun_out/prof_$T && mkdir -p gpurun_out/prof_$T
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO_DIR/gpurun_out/prof_$T -o bench -- python $REPO_DIR/bench.py --steps 2 --warmup 1 --min-warmup-seconds 0 --cpu-images 0 --companion-steps 0 --dino-like-steps 0 ${BENCH_ARGS:-} > $REPO_DIR/gpurun_out/prof_$T/bench.json 2> $REPO_DIR/gpurun_out/prof_$T/bench.err)
      echo "prof exit: $?"; python scripts/rocpd_summary.py gpurun_out/prof_$T/bench_results.db > gpurun_out/kernel_stats_$T.csv; head -14 gpurun_out/kernel_stats_$T.csv | cut -c1-200
      rm -rf gpurun_out/prof_$T;;
    pmc)    # hardware counters, ONE rocprofv3 pass per group (--kernel-trace only); PMC_GROUPS restricts, PROF_TAG names
      T=${PROF_TAG:-c2}
      declare -A PMCG=( [mfma]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
                          [wait]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES"
                          [fetch]="FETCH_SIZE" [write]="WRITE_SIZE" )
      for G in ${PMC_GROUPS:-mfma wait fetch write}; do
        rm -rf gpurun_out/pmcrun_$G && mkdir -p gpurun_out/pmcrun_$G
        (cd /tmp && timeout 600 rocprofv3 --pmc ${PMCG[$G]} --kernel-trace -d $REPO_DIR/gpurun_out/pmcrun_$G -o pmc -- python $REPO_DIR/bench.py --steps 1 --warmup 1 --min-warmup-seconds 0 --cpu-images 0 --companion-steps 0 --dino-like-steps 0 ${BENCH_ARGS:-} > $REPO_DIR/gpurun_out/pmc_${T}_$G.bench.json 2> $REPO_DIR/gpurun_out/pmcrun_$G/bench.err)
        echo "pmc $G exit: $?"
        python scripts/rocpd_pmc_multi.py gpurun_out/pmcrun_$G/pmc_results.db 2 > gpurun_out/pmc_${T}_$G.csv; head -6 gpurun_out/pmc_${T}_$G.csv | cut -c1-220
        rm -rf gpurun_out/pmcrun_$G
      done;;
    *) echo "unknown stage $STAGE";;
  esac
done
