cd $GRAFT_REPO_ROOT
for b in even occupied; do
  timeout -k 10 400 python tools/sortlast_rank_time.py --volume ${1:-2048} --frames 8 --step-deg 45 --balance $b > gpurun_out/r2_sl_${1:-2048}_$b.json 2>gpurun_out/r2_sl_$b.err || { tail -5 gpurun_out/r2_sl_$b.err; exit 1; }
  python3 - gpurun_out/r2_sl_${1:-2048}_$b.json <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(j["balance"], [round(b["partial_ms_mean"],2) for b in j["boxes"]], "max/mean %.2f"%j["partial_ms_max_over_mean"], "max of max %.2f" % max(b["partial_ms_max"] for b in j["boxes"]))
print("   ", [b["bricks"] for b in j["boxes"]])
PY
done
