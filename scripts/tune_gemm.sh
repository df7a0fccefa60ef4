# One-off: let PyTorch TunableOp pick the fastest hipBLASLt/rocBLAS solution for the ViT GEMM shapes of the bench.
mkdir -p gpurun_out
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=$PWD/gpurun_out/tunableop_results.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=15 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5
t0=$(date +%s)
timeout 1200 python bench.py --steps 2 --warmup 2 --cpu-images 0 --gemm-tuning off ${TUNE_ARGS:-} 2> gpurun_out/tune.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('TUNING RUN img/s',d['value'],'ms/step',d['ms_per_step'])"
echo "tuning took $(( $(date +%s) - t0 )) s"; ls -la gpurun_out/tunableop_results*.csv; wc -l gpurun_out/tunableop_results*.csv
export PYTORCH_TUNABLEOP_TUNING=0
python bench.py --steps 3 --warmup 2 --cpu-images 0 --gemm-tuning off ${TUNE_ARGS:-} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('TUNED img/s',d['value'],'ms/step',d['ms_per_step'], {k:round(v['total_ms']/d['steps'],2) for k,v in d['kernels'].items()})"
