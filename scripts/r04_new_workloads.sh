#!/bin/bash
# on the GPU box: one bench line per workload added in round 4 (the reference's examples as they actually run)
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/${1:-r04_new_workloads}
mkdir -p $OUT
for wl in simple_scene_1080p_full global_illumination_1080p_default_probes ball_game_1080p ball_game_1080p_ddgi8x8x8 ${EXTRA_WL}; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline > $OUT/$wl.json 2> $OUT/$wl.err || { echo "FAILED $wl"; tail -5 $OUT/$wl.err; exit 1; }
  python - $OUT/$wl.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p, q = d["passes"], d.get("passes_serial") or {}
print("%-44s %8.1f Mpix/s %7.3f ms | serial %8.1f median %8.1f | in flight %s | serial %s" % (d["config"]["workload"], d["value"], d["ms_per_step"], d.get("value_serial", 0),
      d.get("steady_state", {}).get("value_median", 0), {k: v["ms_avg"] for k, v in p.items()}, {k: v["ms_avg"] for k, v in q.items()}))
PY
done
