"""Turns the PMC passes of tools/collect_profiles.sh into the entry of profiles/traffic.json that bench.py
quotes in its roofline object: HBM traffic per launch of the march kernel (FETCH_SIZE doubled, as gfx950
tallies 128-byte reads at 64 bytes, plus WRITE_SIZE), L2 hit rate, lanes active and vector-ALU busy.

  python tools/make_traffic_json.py gpurun_out/prof_r3 2048_3840x2160_fast profiles/r3
  python tools/make_traffic_json.py gpurun_out/dense_r3 512_1920x1080_fast_translucent profiles/r3 k_march_exact 2
      (the passes of tools/dense_profile.sh carry the mode as a suffix: sq12, fetch2, kt2 ...)"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import kernel_key, summarise  # noqa: E402

prof, key, source = sys.argv[1], sys.argv[2], sys.argv[3]
kernel = sys.argv[4] if len(sys.argv) > 4 else "k_march_fast"
suffix = sys.argv[5] if len(sys.argv) > 5 else ""
c = {}
for name in ("sq1", "sq2", "fetch", "write", "tcc"):
    if os.path.isdir(os.path.join(prof, name + suffix)):
        c.update(summarise(os.path.join(prof, name + suffix)).get(kernel, {}))
dur_ns = None
stats = glob.glob(prof + "/kt" + suffix + "/**/*_kernel_stats.csv", recursive=True) if suffix else \
    glob.glob(prof + "/**/*_kernel_stats.csv", recursive=True) + glob.glob(prof + "/bench_kernel_stats.csv")
for f in stats:
    for r in csv.DictReader(open(f)):
        if kernel_key(r["Name"]) == kernel and (suffix or "false, false" in r["Name"]):
            dur_ns = float(r["AverageNs"])
entry = {
    "traffic": int(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024) if "WRITE_SIZE" in c else None,
    "fetched": int(c["FETCH_SIZE"] * 1024 * 2),
    "source": "%s (rocprofv3 --pmc passes over tools/profile_frame.py, %s; FETCH_SIZE x2 + WRITE_SIZE)" % (source, kernel),
    "l2_hit": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
    "lanes_active": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_INSTS_VALU"]),
    "valu_insts_per_launch": c["SQ_INSTS_VALU"],
    "limited_by": "vector ALU under lane divergence" if kernel == "k_march_fast" else "vector ALU",
}
if dur_ns:
    # a wave64 vector instruction keeps its SIMD's pipe busy for 4 cycles; 1024 SIMDs at 2.4 GHz
    entry["valu_busy"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / (dur_ns * 1e-9 * 2.4e9 * 1024)
    entry["kernel_ms_rocprof"] = dur_ns / 1e6
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
table = json.load(open(path)) if os.path.exists(path) else {}
table[key] = entry
json.dump(table, open(path, "w"), indent=1, sort_keys=True)
print(key, json.dumps(entry))
