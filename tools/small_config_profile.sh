cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/c1prof
mkdir -p $out
for cfg in c1 c2; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$cfg -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 60 --no-cpu-baseline > $out/$cfg.json 2> $out/$cfg.err || exit 1
python3 $GRAFT_REPO_ROOT/tools/kernel_times.py $out/$cfg
python3 - $out/$cfg.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["config"]["name"], "ms/step %.3f"%j["ms_per_step"], "single", j["single_frame_ms"], "hand-over %.4f replay %.4f"%(j["roofline"]["handed_over_ray_fraction"], j["roofline"]["replay_ray_fraction"]))
PY
done
