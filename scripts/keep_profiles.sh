#!/bin/bash
# Here, after scripts/collect_all_workloads.sh TAG, pmc_breakdown.sh TAG and collect_bench_variants.sh TAG ran on the box:
# copies what gpurun merged back under gpurun_out/ to the names profiles/ keeps per round.  keep_profiles.sh TAG ROUND
TAG=${1:?tag}; R=${2:-r02}
cd "$(dirname "$0")/.." || exit 1
declare -A NAME=( [global_illumination_1080p_ddgi8x8x8]=c3_gi1080p [simple_scene_1080p_direct]=c2_simple1080p [light_shafts_1080p]=c4_shafts1080p [global_illumination_4096sq_ddgi8x8x8]=c5_gi4096sq
                  [simple_scene_1080p_full]=c2full_simple1080p [global_illumination_1080p_default_probes]=c3dp_gi1080p [ball_game_1080p]=ballgame1080p )
for w in "${!NAME[@]}"; do
  d=gpurun_out/profiles_${TAG}_$w; n=${NAME[$w]}
  [ -d $d ] || { echo "missing $d"; continue; }
  tail -1 $d/bench.json > profiles/${R}_${n}_bench.json
  cp $d/kernel_stats.csv profiles/${R}_${n}_kernel_stats.csv
  cp $d/kernel_stats_serial.csv profiles/${R}_${n}_kernel_stats_serial.csv
  cp $d/traffic.json profiles/${R}_${n}_traffic.json
done
[ -f gpurun_out/all_workloads_$TAG.log ] && cp gpurun_out/all_workloads_$TAG.log profiles/${R}_all_workloads.log
[ -f gpurun_out/pmc_breakdown_$TAG/breakdown.json ] && cp gpurun_out/pmc_breakdown_$TAG/breakdown.json profiles/${R}_c3_gi1080p_wave_cycle_breakdown.json
[ -f gpurun_out/pmc_breakdown_${TAG}_c2full/breakdown.json ] && cp gpurun_out/pmc_breakdown_${TAG}_c2full/breakdown.json profiles/${R}_c2full_simple1080p_wave_cycle_breakdown.json
[ -f gpurun_out/pmc_breakdown_${TAG}_c4/breakdown.json ] && cp gpurun_out/pmc_breakdown_${TAG}_c4/breakdown.json profiles/${R}_c4_shafts1080p_wave_cycle_breakdown.json
v=gpurun_out/variants_$TAG
if [ -d $v ]; then
  for f in driver_20steps f32_atlases animated_light swap_buffers rehearse_rccl; do cp $v/bench_$f.json profiles/${R}_c3_gi1080p_bench_$f.json; done
  cp $v/shards_4096.json profiles/${R}_shards_4096.json; cp $v/shards_1080p.json profiles/${R}_shards_1080p.json
  [ -f $v/ball_game.log ] && cp $v/ball_game.log profiles/${R}_ball_game.log
fi
git status --short profiles | head -40
