#!/bin/bash
# Everything the measurements kept under profiles/ come from, on the GPU box:
# usage: bash tools/collect_round.sh <tag>   -> gpurun_out/round_<tag>/ and gpurun_out/prof_<tag>/
set -u
tag=${1:-rX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/round_$tag
mkdir -p "$out"
cd "$root"
bash tools/collect_profiles.sh "$tag" > "$out/collect_profiles.log" 2>&1 || { tail -5 "$out/collect_profiles.log"; exit 1; }
timeout -k 10 400 python3 bench.py > "$out/bench_c4.json" 2> "$out/bench_c4.err" || exit 1
for c in c1 c2 c3 c5; do
  timeout -k 10 300 python3 bench.py --config $c --no-cpu-baseline > "$out/bench_$c.json" 2> "$out/bench_$c.err" || exit 1
done
timeout -k 10 300 python3 bench.py --config c2 --table translucent --steps 60 --no-cpu-baseline > "$out/bench_c2_translucent.json" 2> "$out/bench_c2_translucent.err" || exit 1
timeout -k 10 300 python3 bench.py --shading --no-cpu-baseline --steps 120 > "$out/bench_c4_shaded.json" 2> "$out/bench_c4_shaded.err" || exit 1
timeout -k 10 300 python3 tools/band_times.py > "$out/band_times.txt" 2>&1 || exit 1
for p in 8; do bash tools/share_profile.sh $p > "$out/share_$p.txt" 2>&1 || exit 1; done
bash tools/small_config_profile.sh > "$out/small_configs.txt" 2>&1 || exit 1
timeout -k 10 200 python3 tools/frame_stats.py > "$out/frame_stats.txt" 2>&1 || exit 1
timeout -k 10 200 python3 tools/translucent.py all > "$out/translucent.txt" 2>&1 || exit 1
bash tools/dense_profile.sh "$tag" > "$out/dense_profile.txt" 2>&1 || exit 1
INGEST_REPS=8 bash tools/ingest_profile.sh "$tag" 2048 > "$out/ingest_profile.txt" 2>&1 || exit 1
timeout -k 10 100 python3 tools/ingest_time.py 256 512 1024 2048 > "$out/ingest_times.txt" 2>&1 || exit 1
for e in cabi; do
  timeout -k 10 300 python3 bench.py --exchange $e --steps 120 --no-cpu-baseline > "$out/bench_c4_$e.json" 2> "$out/bench_c4_$e.err" || exit 1
  timeout -k 10 300 python3 bench.py --config c5 --exchange $e --steps 120 --no-cpu-baseline > "$out/bench_c5_$e.json" 2> "$out/bench_c5_$e.err" || exit 1
done
bash tools/histogram_profile.sh "$tag" 2048 > "$out/histogram_profile.txt" 2>&1 || exit 1
timeout -k 10 100 python3 tools/histogram_time.py 256 512 1024 2048 > "$out/histogram_times.txt" 2>&1 || exit 1
timeout -k 10 200 python3 tools/upload_time.py 512 1024 1536 > "$out/upload_times.txt" 2>&1 || exit 1
timeout -k 10 200 python3 tools/timeline.py > "$out/lane_march_timeline.txt" 2>&1 || exit 1
cp gpurun_out/timeline_counts.json "$out/lane_march_wave_counts.json"
timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_as_the_driver_runs_it.json" 2> "$out/bench_as_the_driver_runs_it.err" || exit 1
for m in "" "--in-flight 2" "--pageable"; do
  timeout -k 10 200 ./examples/orbit --phantom 2048 --size 3840x2160 --frames 360 --samples 7094 --mode 1 $m >> "$out/orbit_hand_off.txt" 2>&1 || exit 1
done
for e in 2048 4096; do bash tools/sortlast_balance.sh $e "$out" > "$out/sortlast_balance_$e.txt" 2>&1 || exit 1; done
echo done
