cd /root/repo
for sg in 0.005 0; do
  echo "feature sigma $sg"; python bench.py --feature-sigma $sg --cpu-sample 8 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['parity'], d['cpu_baseline']['value']); print({k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()}); print(d['work_per_step'])"
done
