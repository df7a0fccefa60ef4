for vb in 32 64 128 256; do
  python bench.py --steps 3 --warmup 2 --cpu-images 0 --vit-batch $vb 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('vit_batch',$vb,'img/s',d['value'],'ms/step',d['ms_per_step'],'host',d['host_enqueue_ms_per_step'], {k:round(v['total_ms']/d['steps'],2) for k,v in d['kernels'].items()})"
done
