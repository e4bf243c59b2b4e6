export TMPDIR=/tmp
run() { python bench.py --steps 80 --warmup 10 --no-cpu-baseline --no-corr-roofline "$@" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('pairs/s', round(d['value'], 2))"; }
for q in 4 8; do for s in 3 4 6; do echo "GPU_MAX_HW_QUEUES=$q streams=$s"; GPU_MAX_HW_QUEUES=$q run --streams $s; done; done
