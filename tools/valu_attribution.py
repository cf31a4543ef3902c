"""Where the vector instructions of the lane march go, part by part.

  python tools/valu_attribution.py gpurun_out/timeline_counts.json [wave-level VALU instructions per launch from the counters]

Static side: the device code is compiled to assembly with line tables (hipcc -S -gline-tables-only, same flags as the
library), the loops of k_march_fast<unsigned short, false, false> are found from its backward branches, and every
vector instruction is filed under the part of the march it belongs to by where it sits (before the round loop, in the
skipping loop, in the round outside it, behind the loop) and by the source line it was compiled from.
Dynamic side: tools/timeline.py (diagnostic build) counts how often a WAVE ran each part.  Instructions x executions
is an upper estimate for parts with branches inside them (a wave skips a branch none of its lanes takes); the
counters' total (SQ_INSTS_VALU per launch) is printed beside the sum.
"""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vulkanrenderer_amd", "csrc")
SYM = "_ZN2vr12k_march_fastItLb0ELb0EEE"


def line_of(lines, pattern, after=0):
    for i in range(after, len(lines)):
        if re.search(pattern, lines[i]):
            return i + 1
    raise SystemExit("marker not found: " + pattern)


def main():
    counts = json.load(open(sys.argv[1]))
    measured = float(sys.argv[2]) if len(sys.argv) > 2 else json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["2048_3840x2160_fast"]["valu_insts_per_launch"]
    asm = "/tmp/vr_lines.s"
    flags = "--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-result".split()
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-gline-tables-only", "-S", "--cuda-device-only", "-o", asm, "vr_abi.hip"],
                   cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    text = open(asm).read().split("\n")
    files = {}
    for l in text:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m:
            files[int(m.group(1))] = m.group(2)
    start = next(i for i, l in enumerate(text) if l.startswith(SYM) and l.rstrip().endswith(":") is False and ":" in l)
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    labels, insts, cur = {}, [], ("", 0)
    for l in text[start + 1:end]:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if s.startswith(".loc"):
            p = s.split()
            cur = (files.get(int(p[1]), "?"), int(p[2]))
            continue
        if not s or s.startswith((".", ";")) or s.endswith(":"):
            continue
        insts.append((s.split()[0], s, cur))
    loops = []
    for i, (mn, s, _) in enumerate(insts):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            t = s.split()[1]
            if t in labels and labels[t] <= i:
                loops.append((labels[t], i))
    big = max(loops, key=lambda ab: ab[1] - ab[0])
    round_lo = min(a for a, b in loops if a <= big[1] and b >= big[0] and b - a > 100)
    round_hi = max(b for a, b in loops if a <= big[1] and b >= big[0] and b - a > 100)
    inner = [ab for ab in loops if round_lo < ab[0] and ab[1] < round_hi]
    skip_lo, skip_hi = max(inner, key=lambda ab: ab[1] - ab[0])

    march = open(os.path.join(CSRC, "vr_march.hpp")).read().split("\n")
    decider = (line_of(march, r"^struct Decider \{"), line_of(march, r"^\};", line_of(march, r"^struct Decider \{")))
    lanes = line_of(march, r"void march_lanes\(")
    decide_call = (line_of(march, r"// 3\. early termination", lanes), line_of(march, r"^    if \(marching\) \{", lanes) - 1)
    hand_over = (line_of(march, r"^    if \(marching\) \{", lanes), line_of(march, r"^    \} else if \(PARTIAL\) \{", lanes) - 1)
    model = [(line_of(march, r"^struct RayModel"), line_of(march, r"void k_occupied_bounds\(")),
             (line_of(march, r"void clip_to_tile\("), line_of(march, r"void k_pack_skip\("))]

    def part(i, loc):
        f, ln = loc
        in_march = f == "vr_march.hpp"
        if i < round_lo:
            if in_march and any(a <= ln < b for a, b in model):
                return "set-up: line model of the ray and its clip to the tile's depth range"
            return "set-up: ray, table into LDS, the rest in front of the loop"
        if i > round_hi:
            if (in_march and hand_over[0] <= ln <= hand_over[1]) or f == "amd_hip_atomic.h":
                return "behind the loop: hand-over record and queue slots"
            return "behind the loop: back tap, its decision test, pixel store"
        if skip_lo <= i <= skip_hi:
            return "a pass of the skipping loop"
        if in_march and (decider[0] <= ln <= decider[1] or decide_call[0] <= ln <= decide_call[1]):
            return "a round's decision test"
        if i < skip_lo:
            return "a round: tap positions of lanes inside material"
        return "a round: fetch, filter, table, blend"

    static = collections.Counter()
    for i, (mn, _, loc) in enumerate(insts):
        if mn.startswith("v_"):
            static[part(i, loc)] += 1
    c = counts
    times = {
        "set-up: line model of the ray and its clip to the tile's depth range": c["waves"] - c["waves_nothing_behind_tile"],
        "set-up: ray, table into LDS, the rest in front of the loop": c["waves"],
        "a pass of the skipping loop": c["skip_passes"],
        "a round: tap positions of lanes inside material": c["rounds_preparing_material_taps"],
        "a round: fetch, filter, table, blend": c["rounds"],
        "a round's decision test": c["rounds_with_decision_test"],
        "behind the loop: hand-over record and queue slots": c["waves_handing_over"],
        "behind the loop: back tap, its decision test, pixel store": c["waves_with_back_tap"],
    }
    print("k_march_fast<unsigned short, false, false>: %d instructions, %d of them vector; round loop [%d, %d], skipping loop [%d, %d]" % (
        len(insts), sum(static.values()), round_lo, round_hi, skip_lo, skip_hi))
    print("%-78s %8s %12s %10s %7s" % ("part", "VALU", "wave runs", "G VALU", "share"))
    total = sum(static[k] * times[k] for k in times)
    for k in times:
        print("%-78s %8d %12.0f %10.4f %6.1f%%" % (k, static[k], times[k], static[k] * times[k] / 1e9, 100.0 * static[k] * times[k] / total))
    print("%-78s %8s %12s %10.4f" % ("sum (upper estimate: a wave skips branches none of its lanes takes)", "", "", total / 1e9))
    print("%-78s %8s %12s %10.4f  (%.2f of the sum)" % ("counters: SQ_INSTS_VALU per launch", "", "", measured / 1e9, measured / total))
    cy = c["cycles"]
    print("cycles of the diagnostic build by part (shader clock, all waves): set-up %.3f, skipping loop %.3f, fetch rounds %.3f "
          "(of which decision test %.3f), behind the loop and bookkeeping %.3f" % (
              cy["setup"] / cy["all"], cy["skip_loop"] / cy["all"], cy["fetch_rounds_with_decision"] / cy["all"], cy["decision_test"] / cy["all"],
              1.0 - (cy["setup"] + cy["skip_loop"] + cy["fetch_rounds_with_decision"]) / cy["all"]))


if __name__ == "__main__":
    main()
