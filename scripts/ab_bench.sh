#!/bin/bash
# on the GPU box: bench every variant library (interleaved rounds), print value/ms/passes
cd "$(dirname "$0")/.." || exit 1
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do
  for lib in madarch_amd/csrc/variants/libmadarch_hip_*.so; do
    tag=$(basename $lib .so); tag=${tag#libmadarch_hip_}
    MADARCH_HIP_LIBRARY=$PWD/$lib timeout -k 10 200 python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['passes']; q=d.get('passes_serial') or {}; print('%-28s %8.1f Mpix/s %7.3f ms  rad %.3f irr %.3f scr %.3f | serial %7.1f Mpix/s rad %.4f irr %.4f scr %.4f' % ('$tag', d['value'], d['ms_per_step'], p.get('radiance',{}).get('ms_avg',0), p.get('irradiance',{}).get('ms_avg',0), p['screen']['ms_avg'], d.get('value_serial',0), q.get('radiance',{}).get('ms_avg',0), q.get('irradiance',{}).get('ms_avg',0), q.get('screen',{}).get('ms_avg',0)))"
  done
done
