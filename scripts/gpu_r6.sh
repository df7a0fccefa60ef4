#!/bin/bash
# Round-6 GPU session script (run through gpurun): stages given as arguments; logs -> gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
REPO_DIR=$PWD
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt; nproc >> gpurun_out/gpu.txt
QUICK="--cpu-images 0 --dino-like-steps 0 --companion-steps 0 --distinct 256"
bench_summary() { python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value', 'ms_per_step', 'steps', 'host_enqueue_ms_per_step', 'host_in_loop_ms_per_step')}, 'images/step', d['config']['images_per_step'], 'vit_batch', d['config']['vit_batch'])
tot = sum(v.get('total_ms', 0) for v in d['kernels'].values()) / d['steps']
print('kernel ms/step', round(tot, 1), 'other', round(d['ms_per_step'] - tot, 1), 'us/img', round(1e3 * d['ms_per_step'] / d['config']['images_per_step'], 2), 'lib sites', d.get('library_gemm_ms_per_step_by_site'))
for k, v in d['kernels'].items(): print(' ', k, v.get('launches'), v.get('avg_ms'), v.get('frac'), v.get('passes_per_image', ''))
"; }
for STAGE in "$@"; do
  echo "=== $STAGE"
  case $STAGE in
    bisect)   # ARMS="tag:arm:opts:ENV=..;ENV2=.. ..." forward_bisect arms one process each; N forwards per arm
      for A in ${ARMS:-cap:capture::}; do
        IFS=':' read -r TAG ARM OPTS ENVS <<< "$A"
        echo "--- arm $TAG ($ARM; $OPTS; $ENVS)"
        ( IFS=';'; for e in $ENVS; do export "$e"; done; unset IFS
          timeout ${ARM_TIMEOUT:-480} python scripts/debug/forward_bisect.py ${N:-6000} ${BISECT_MODEL:-dino_vitb8} ${BISECT_BATCH:-24} ${BISECT_SIZE:-480} $ARM "$OPTS" > gpurun_out/r06_bisect_$TAG.txt 2>&1; echo "exit $?" >> gpurun_out/r06_bisect_$TAG.txt )
        grep -v "^$" gpurun_out/r06_bisect_$TAG.txt | cut -c1-400 | tail -${BISECT_LINES:-30}
      done;;
    attn_lab)   # LIBS="tag1 tag2": attention tests against each lab library, then the same-process A/B of all of them
      for L in ${LIBS:-}; do
        echo "--- attention tests with lab library $L"
        DSS_HIP_LIBRARY=$REPO_DIR/scripts/lablib/libdss_hip_$L.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_determinism.py -m gpu -q --timeout 600 -rf --tb=short -k "attention" 2>&1 | tail -${LAB_TEST_LINES:-6}
      done
      LABS=""; for L in ${LIBS:-}; do LABS="$LABS:$REPO_DIR/scripts/lablib/libdss_hip_$L.so"; done
      DSS_LAB_LIBRARY=${LABS#:} timeout 900 python scripts/debug/attn_ab2.py 2>&1 | tee gpurun_out/r06_attn_ab.txt;;
    lt_describe)   # hipBLASLt's candidates for the library GEMM shapes, as dss_linear_lt walks them
      timeout 600 python -c "
import torch, dss_amd
from dss_amd import hip
for m, n, k in [(3601*24, 768, 3072), (3601*24, 768, 768), (3600*24, 768, 192), (3601*291, 768, 3072), (3601*291, 768, 768), (3600*291, 768, 192), (901*2473, 384, 1536), (901*291, 384, 1536)]:
    print('=== M', m, 'N', n, 'K', k)
    print('\\n'.join(l[:260] for l in hip.linear_lt_describe(m, n, k).splitlines()[:10]))
" > gpurun_out/r06_lt_describe.txt 2>&1; cut -c1-230 gpurun_out/r06_lt_describe.txt | head -${LT_LINES:-60};;
    det_tests) timeout 1500 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_kernels.py -m gpu -q --timeout 900 -rf --tb=short -k "${DET_K:-repeated or linear_lt or reproducible or bit_for_bit}" 2>&1 | tail -40 > gpurun_out/pytest_det.log; tail -25 gpurun_out/pytest_det.log;;
    eigs_ab)   # LIBS="tag ...": spectral-stage A/B (scripts/debug/eigs_ab.py), EIGS_AB_CASES="B,N,D,K;..."
      LABS=""; for L in ${LIBS:-}; do LABS="$LABS:$REPO_DIR/scripts/lablib/libdss_hip_$L.so"; done
      DSS_LAB_LIBRARY=${LABS#:} timeout 900 python scripts/debug/eigs_ab.py 2>&1 | tee gpurun_out/r06_eigs_ab.txt;;
    mlp_lab)   # the fused-MLP lab kernel (prebuilt: scripts/probes/mlp_fused_lab[_nogelu]) against the product's pair, same box
      ( for B in mlp_fused_lab mlp_fused_lab_nogelu; do echo "--- $B"; timeout 600 scripts/probes/$B ${MLP_IMAGES:-2473} 901 10; done
        timeout 600 python scripts/debug/mlp_pair_time.py ${MLP_IMAGES:-2473} 901 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_mlp_lab.txt;;
    two_ranks)   # the N > 1 code path on the one GPU of the box (gloo; both ranks share the GPU): weak line, then --dataset 2500
      ( DSS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --cpu-images 0 --dino-like-steps 0 --companion-steps 0 --steps 6 2> gpurun_out/two_ranks.err
        DSS_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --dataset 2500 --cpu-images 0 --dino-like-steps 0 --companion-steps 0 2>> gpurun_out/two_ranks.err ) > gpurun_out/r06_bench_2ranks_one_gpu_gloo.json
      tail -2 gpurun_out/two_ranks.err; cut -c1-200 gpurun_out/r06_bench_2ranks_one_gpu_gloo.json;;
    ln_tests) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -rf --tb=short -k "lnlinear or linear_kres or layernorm or patch_embed" 2>&1 | tail -30 > gpurun_out/pytest_ln.log; tail -15 gpurun_out/pytest_ln.log;;
    attn_tests) timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -rf --tb=short -k "attention" 2>&1 | tail -30 > gpurun_out/pytest_attn.log; tail -12 gpurun_out/pytest_attn.log;;
    vit_tests) timeout 1200 python -m pytest tests/test_gpu_e2e.py -m gpu -q --timeout 900 -rf --tb=short -k "vit or indexing or fp16_path or end_to_end_eigenvectors or config3" 2>&1 | tail -30 > gpurun_out/pytest_vit.log; tail -15 gpurun_out/pytest_vit.log;;
    eigs_tests) timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 900 -rf --tb=short -k "eigs or golden or symmetric or sign_rule or starved or affinity" 2>&1 | tail -30 > gpurun_out/pytest_eigs.log; tail -12 gpurun_out/pytest_eigs.log;;
    tests_all) timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -rf --tb=short 2>&1 | tail -150 > gpurun_out/pytest_gpu.log; tail -40 gpurun_out/pytest_gpu.log;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?"; tail -4 gpurun_out/smoke.log;;
    bench_quick) timeout 600 python bench.py $QUICK ${BENCH_ARGS:-} 2> gpurun_out/bench_quick.err | tee gpurun_out/bench_quick.json | bench_summary; tail -2 gpurun_out/bench_quick.err;;
    bench_sweep)   # SWEEP="args1|args2|...": one quick bench per entry
      IFS='|' read -ra SW <<< "${SWEEP:-}"
      for i in "${!SW[@]}"; do echo "--- sweep $i: ${SW[$i]}"; timeout 500 python bench.py $QUICK ${SW[$i]} 2> gpurun_out/bench_sweep$i.err | tee gpurun_out/bench_sweep$i.json | bench_summary | head -${SWEEP_LINES:-3}; tail -1 gpurun_out/bench_sweep$i.err; done;;
    lib_ab)   # LIBS="tag1 tag2": quick bench with the product library, then with each scripts/lablib/libdss_hip_<tag>.so, REPS times
      for rep in $(seq 1 ${REPS:-2}); do
        for L in product ${LIBS:-}; do
          if [ $L = product ]; then unset DSS_HIP_LIBRARY; else export DSS_HIP_LIBRARY=$REPO_DIR/scripts/lablib/libdss_hip_$L.so; fi
          echo "--- rep $rep lib $L"; timeout 400 python bench.py $QUICK --steps ${AB_STEPS:-6} ${BENCH_ARGS:-} 2> gpurun_out/lib_ab_$L.err | tee gpurun_out/lib_ab_${L}_$rep.json | bench_summary | head -${SWEEP_LINES:-12}
        done
      done; unset DSS_HIP_LIBRARY;;
    bench) timeout 900 python bench.py ${BENCH_ARGS:-} > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench exit: $?"; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.json | bench_summary; cut -c1-300 gpurun_out/bench.json;;
    bench_c1) timeout 600 python bench.py --size 224 --cpu-images 8 --parity-images 8 --companion-steps 0 --dino-like-steps 0 > gpurun_out/bench_c1.json 2> gpurun_out/bench_c1.err; echo "bench_c1 exit: $?"; tail -2 gpurun_out/bench_c1.err; cut -c1-400 gpurun_out/bench_c1.json;;
    bench_ds) for DS in ${DATASETS:-1250 10000}; do timeout 600 python bench.py --dataset $DS --cpu-images 0 --dino-like-steps 0 --companion-steps 0 ${DS_ARGS:-} 2> gpurun_out/bench_ds$DS.err > gpurun_out/bench_ds$DS${DS_TAG:-}.json; cat gpurun_out/bench_ds$DS${DS_TAG:-}.json | bench_summary; done;;
    bench_c3) timeout 1200 python bench.py --model dino_vitb8 --K 15 --cpu-images ${C3_CPU:-2} --parity-images ${C3_CPU:-2} --companion-steps 0 --dino-like-steps 0 --steps ${C3_STEPS:-6} ${C3_ARGS:-} > gpurun_out/bench_c3${C3_TAG:-}.json 2> gpurun_out/bench_c3.err; echo "bench_c3 exit: $?"; tail -2 gpurun_out/bench_c3.err; cat gpurun_out/bench_c3${C3_TAG:-}.json | bench_summary;;
    prof)   # per-kernel time of the bench command (rocprofv3 kernel trace + stats); PROF_TAG names the output
      T=${PROF_TAG:-c2}; rm -rf gpurun_out/prof_$T && mkdir -p gpurun_out/prof_$T
      (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $REPO_DIR/gpurun_out/prof_$T -o bench -- python $REPO_DIR/bench.py --steps 2 --warmup 1 --min-warmup-seconds 0 --cpu-images 0 --companion-steps 0 --dino-like-steps 0 ${BENCH_ARGS:-} > $REPO_DIR/gpurun_out/prof_$T/bench.json 2> $REPO_DIR/gpurun_out/prof_$T/bench.err)
      echo "prof exit: $?"; python scripts/rocpd_summary.py gpurun_out/prof_$T/bench_results.db > gpurun_out/kernel_stats_$T.csv; head -20 gpurun_out/kernel_stats_$T.csv | cut -c1-200
      rm -rf gpurun_out/prof_$T;;
    pmc)    # hardware counters, ONE rocprofv3 pass per group (--kernel-trace only); PMC_GROUPS restricts, PROF_TAG names
      T=${PROF_TAG:-c2}
      declare -A PMCG=( [mfma]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
                          [wait]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM SQ_VALU_MFMA_COEXEC_CYCLES"
                          [fetch]="FETCH_SIZE" [write]="WRITE_SIZE"
                          [l2a]="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
                          [l2b]="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_GATE_EN1_sum" )
      for G in ${PMC_GROUPS:-mfma wait fetch write}; do
        rm -rf gpurun_out/pmcrun_$G && mkdir -p gpurun_out/pmcrun_$G
        (cd /tmp && timeout 600 rocprofv3 --pmc ${PMCG[$G]} --kernel-trace -d $REPO_DIR/gpurun_out/pmcrun_$G -o pmc -- python $REPO_DIR/bench.py --steps 1 --warmup 1 --min-warmup-seconds 0 --cpu-images 0 --companion-steps 0 --dino-like-steps 0 ${BENCH_ARGS:-} > $REPO_DIR/gpurun_out/pmc_${T}_$G.bench.json 2> $REPO_DIR/gpurun_out/pmcrun_$G/bench.err)
        echo "pmc $G exit: $?"
        python scripts/rocpd_pmc_multi.py gpurun_out/pmcrun_$G/pmc_results.db 2 > gpurun_out/pmc_${T}_$G.csv; head -6 gpurun_out/pmc_${T}_$G.csv | cut -c1-220
        rm -rf gpurun_out/pmcrun_$G
      done;;
    cli) df -h /tmp /dev/shm | tail -2; timeout 1500 python scripts/cli_throughput.py ${CLI_N:-20480} > gpurun_out/cli_throughput.log 2>&1; echo "cli exit $?"; grep -v "Skipping\|^{" gpurun_out/cli_throughput.log | tail -25;;
    attn_ab) timeout 300 python scripts/debug/attn_ab.py > gpurun_out/attn_ab.log 2>&1; tail -30 gpurun_out/attn_ab.log;;
    *) echo "unknown stage $STAGE";;
  esac
done
