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
    *) echo "unknown stage $STAGE";;
  esac
done
