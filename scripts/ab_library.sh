#!/bin/bash
# on the GPU box: the shipped library against other builds of the same sources (LIB="path ..."), bench workloads, two rounds
cd "$(dirname "$0")/.." || exit 1
LIB=${LIB:?LIB="paths of the other builds"}
for r in ${ROUNDS:-1 2}; do
for w in ${WORKLOADS:-global_illumination_1080p_ddgi8x8x8}; do
for l in "" $LIB; do
  MADARCH_HIP_LIBRARY=$l timeout -k 10 120 python bench.py --workload $w --no-cpu-baseline $BENCH_ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-38s %-24s: %8.1f Mpix/s in flight, %8.1f serial | in flight %s | serial %s' % ('$w', '${l##*/}' or 'shipped', d['value'], d['value_serial'], {k: v['ms_avg'] for k, v in d['passes'].items()}, {k: v['ms_avg'] for k, v in d['passes_serial'].items()}))" || exit 1
done; done; done
