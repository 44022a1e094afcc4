cd /root/repo
for a in "--steps 10 --warmup 5" "--steps 10 --warmup 5" "--steps 10 --warmup 20" "--steps 20 --warmup 3"; do
BH_FORCE_PG=1 timeout 200 python bench.py --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('$a', d['ms_per_step'])"
done
for a in "--steps 10 --warmup 5" "--steps 20 --warmup 3"; do
timeout 200 python bench.py --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print('nopg $a', d['ms_per_step'])"
done
