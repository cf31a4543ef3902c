#!/bin/bash
# Run in the repository after `bash tools/collect_round.sh <tag>` on the GPU box: copies the summaries kept in the repository
# from gpurun_out/ into profiles/<tag>/ and refreshes profiles/traffic.json.
# usage: bash tools/publish_round.sh <tag>
set -eu
tag=${1:?tag}
R=gpurun_out/round_$tag
P=profiles/$tag
mkdir -p "$P/ingest" "$P/histogram"
cp $R/bench_c*.json $R/bench_as_the_driver_runs_it.json $R/band_times.txt $R/share_8.txt $R/small_configs.txt $R/frame_stats.txt \
   $R/translucent.txt $R/orbit_hand_off.txt $R/sortlast_balance_*.txt $R/sortlast_rank_*.json $R/ingest_times.txt \
   $R/upload_times.txt $R/lane_march_timeline.txt $R/lane_march_wave_counts.json "$P/"
cp $R/dense_profile.txt "$P/dense_marches.txt"
cp gpurun_out/prof_$tag/bench_kernel_stats.csv gpurun_out/prof_$tag/bench_under_rocprof.json gpurun_out/prof_$tag/pmc_summary.txt "$P/"
python3 tools/timed_region_avg.py gpurun_out/prof_$tag/bench > "$P/bench_timed_region.txt"
cp gpurun_out/ingest_$tag/kernel_times.txt gpurun_out/ingest_$tag/pmc.txt "$P/ingest/"
cp gpurun_out/ingest_$tag/kt/kt_kernel_stats.csv "$P/ingest/"
cp gpurun_out/hist_$tag/kernel_times.txt gpurun_out/hist_$tag/pmc.txt "$P/histogram/"
cp gpurun_out/hist_$tag/kt/kt_kernel_stats.csv "$P/histogram/"
cp $R/histogram_times.txt "$P/histogram/call_times.txt"
python3 tools/make_traffic_json.py gpurun_out/prof_$tag 2048_3840x2160_fast "$P/pmc_summary.txt"
python3 tools/make_traffic_json.py gpurun_out/dense_$tag 512_1920x1080_fast_translucent "$P/dense_marches.txt" k_march_exact 2
python3 tools/valu_attribution.py "$P/lane_march_wave_counts.json" > "$P/valu_attribution.txt"
echo published "$P"
