#!/bin/bash
# Quick look at one build on the GPU box: kernel times of a few unpipelined benchmark frames, L2 hit
# and fetch size per kernel, then a short bench.  usage: bash tools/quick_profile.sh <tag> [bench steps]
set -u
tag=${1:-q}
steps=${2:-120}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/quick_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -o kt -- python3 "$root/tools/profile_frame.py" --frames 8 > "$out/kt.log" 2>&1 || exit 1
pass() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o pmc -- python3 "$root/tools/profile_frame.py" --frames 6 > "$out/$name.log" 2>&1 || exit 1
}
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass fetch FETCH_SIZE
cd "$root"
python3 tools/kernel_times.py "$out/kt" | tee "$out/kernel_times.txt"
python3 tools/pmc_summary.py "$out/tcc" "$out/fetch" | tee "$out/pmc.txt"
timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps "$steps" > "$out/bench.json" 2> "$out/bench.err" || exit 1
python3 - "$out/bench.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench: %.3f ms/step, %.1f fps, march kernel %.3f ms, single frame %s" % (j["ms_per_step"], j["frames_per_s"], j["roofline"]["kernel_ms_avg"], j.get("single_frame_ms")))
PY
