# BASELINE.json configs[2]: dino_vitb8 480x480 (3600 patches), K=15
python bench.py --model dino_vitb8 --K 15 --batch 512 --vit-batch 16 --steps 2 --warmup 1 --cpu-images 1 --distinct 64 2>gpurun_out/c3.err | tee gpurun_out/c3.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C3 img/s',d['value'],'ms/step',d['ms_per_step'],'host',d['host_enqueue_ms_per_step']); print({k:(round(v['total_ms']/d['steps'],2),v.get('achieved'),v.get('passes_per_image')) for k,v in d['kernels'].items()}); print(d.get('cpu_baseline'), d.get('parity'), d['unconverged_images'])"
tail -3 gpurun_out/c3.err
