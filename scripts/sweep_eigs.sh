for th in 256 512 1024; do
  DSS_EIGS_THREADS=$th python bench.py --steps 3 --warmup 2 --cpu-images 0 --batch 1024 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('threads',$th,'img/s',d['value'],'ms/step',d['ms_per_step'], {k:(round(v['total_ms']/d['steps'],2),v.get('achieved'),v.get('passes_per_image')) for k,v in d['kernels'].items()})"
done
