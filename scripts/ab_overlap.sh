for i in 1 2; do
for flag in "" "--no-overlap"; do
python bench.py --cpu-images 0 $flag 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('overlap' if d['config']['stage_overlap'] else 'serial ', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], {k:round(v['total_ms']/d['steps'],2) for k,v in d['kernels'].items()})"
done; done
python bench.py --cpu-images 6 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['parity'], d['unconverged_images'])"
