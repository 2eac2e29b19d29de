#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for f in 2000 0 1000 4000 100000 2000; do
  MADARCH_HIP_ORDER_FLOOR=$f timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('floor %6d: %8.1f in flight | serial %8.1f' % ($f, d['value'], d.get('value_serial',0)))"
done
for o in 0 1; do
  timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --screen-order $o 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('screen order $o: %8.1f in flight | serial %8.1f' % (d['value'], d.get('value_serial',0)))"
done
