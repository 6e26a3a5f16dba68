cd /root/repo
for ex in peer allgather a2a_fixed a2a; do
  timeout 300 python bench.py --sharded --exchange $ex --steps 200 --warmup 20 --no-profile 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$ex', d['value'], d['ms_per_step'], d['config'].get('exchange'))"
done
