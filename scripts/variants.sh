cd /root/repo
for g in 1 2 4 8; do
  echo "groups $g noprofile"; python bench.py --groups $g --no-profile --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
echo "groups 1 profile"; python bench.py --groups 1 --cpu-sample 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
