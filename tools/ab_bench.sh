set -e
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  echo "ROWSTREAM_AUTO=$1 FUSE_LN=$2" >> gpurun_out/r3_ab1.txt
  ST_ROWSTREAM_AUTO=$1 ST_FUSE_LN=$2 timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-corr-roofline 2>gpurun_out/r3_ab1.err | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print({k: d[k] for k in ('value','ms_per_step','value_1_in_flight') if k in d}, d.get('roofline', {}).get('kernel_ms_per_step'))" >> gpurun_out/r3_ab1.txt
done
