#!/bin/bash
# On the GPU box: bench line + rocprofv3 kernel stats (pipelined and serial) + PMC traffic for every bench workload
# (BASELINE configs 2-5 at their stated sizes and, since round 4, the reference's examples as they run) ->
# gpurun_out/profiles_<TAG>_<workload>/ ; scripts/keep_profiles.sh copies what is kept into profiles/.
TAG=${1:-r04}
cd "$(dirname "$0")/.." || exit 1
for w in ${WORKLOADS:-global_illumination_1080p_ddgi8x8x8 simple_scene_1080p_direct simple_scene_1080p_full light_shafts_1080p global_illumination_4096sq_ddgi8x8x8 global_illumination_1080p_default_probes ball_game_1080p}; do
  bash scripts/collect_profiles.sh ${TAG}_$w $w > /dev/null 2>&1 || echo "collect failed for $w"
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/profiles_${TAG}_$w/bench.json').read().strip().splitlines()[-1])
print('%-42s %8.1f Mpix/s %.4f ms | serial %8.1f Mpix/s %.4f ms | median %8.1f | %s' % ('$w', d['value'], d['ms_per_step'], d.get('value_serial',0), d.get('ms_per_step_serial',0), d.get('steady_state',{}).get('value_median',0), {k:v['ms_avg'] for k,v in (d.get('passes_serial') or {}).items()}))"
done
