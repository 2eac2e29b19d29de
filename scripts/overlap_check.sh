#!/bin/bash
# on the GPU box: frame overlap on/off and probe-stream priority, same library
cd "$(dirname "$0")/.." || exit 1
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['passes']; print('%-28s %8.1f Mpix/s %7.3f ms  rad %.3f irr %.3f scr %.3f' % ('$1', d['value'], d['ms_per_step'], p.get('radiance',{}).get('ms_avg',0), p.get('irradiance',{}).get('ms_avg',0), p['screen']['ms_avg']))"; }
for r in 1 2; do
  timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --serial 2>/dev/null | tail -1 | show serial
  for o in 1 2; do for p in -1 0; do
    MADARCH_HIP_PROBE_PRIORITY=$p timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --overlap $o 2>/dev/null | tail -1 | show "overlap $o prio $p"
  done; done
done
