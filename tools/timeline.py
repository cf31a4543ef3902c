"""Diagnostic: per-wave start/end times of the fast march (needs `make -C vulkanrenderer_amd/csrc diag`
and VR_LIB_NAME=libvr_hip_diag.so).  Prints how many waves are resident over time."""
import ctypes as C
import math
import os
import sys

os.environ.setdefault("VR_LIB_NAME", "libvr_hip_diag.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import vulkanrenderer_amd as vr  # noqa: E402

n, w, h = 2048, 3840, 2160
ctx = vr.VolumeRenderer(0)
raw = torch.empty(n ** 3, dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
ctx.phantom_generate_device(raw.data_ptr(), 2, n)
ctx.upload_volume_device(raw.data_ptr(), 2, n, n, n)
ctx.synchronize()
del raw
ctx.update_transfer_function(bench.TF_GRAPHS, reset_state=True)
ctx.set_sampling(bench.samples_for(n))
ctx.set_mode(1)
out = torch.empty(w * h * 4, dtype=torch.uint8, device="cuda")
waves = ((w + 7) // 8) * ((h + 15) // 16) * 2 + 8
WORDS = 10   # kTimelineWords of the diagnostic build
tl = torch.zeros(waves * WORDS, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
L = vr.lib()
L.vr_diag_set_timeline.argtypes = [C.c_void_p, C.c_void_p]
for i in range(3):
    ctx.set_camera(*bench.orbit_camera(30))
    ctx.render_device(w, h, out.data_ptr())
ctx.synchronize()
assert L.vr_diag_set_timeline(ctx._h, tl.data_ptr()) == 0
ctx.set_camera(*bench.orbit_camera(30))
ctx.render_device(w, h, out.data_ptr())
ctx.synchronize()
print("kernel ms", ctx.march_kernel_times(1))
t = tl.cpu().numpy().reshape(-1, WORDS)
t = t[t[:, 0] != 0]
t0, t1 = t[:, 0].astype(np.float64), t[:, 1].astype(np.float64)
base = t0.min()
t0 = (t0 - base) / 100.0   # us
t1 = (t1 - base) / 100.0
dur = t1 - t0
print("waves", len(t), "span us", t1.max(), "dur us: mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % (
    dur.mean(), np.median(dur), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
edges = np.linspace(0, t1.max(), 41)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    resident = int(np.count_nonzero((t0 <= mid) & (t1 > mid)))
    started = int(np.count_nonzero((t0 >= a) & (t0 < b)))
    print("t=%7.0f us resident %6d (%.2f per SIMD) started %6d" % (mid, resident, resident / 1024.0, started))
info = t[:, 3].astype(np.uint64)
rounds = (info & np.uint64(0xfff)).astype(np.float64)
mmodel = ((info >> np.uint64(12)) & np.uint64(0xfff)).astype(np.float64)
mexact = ((info >> np.uint64(24)) & np.uint64(0xfff)).astype(np.float64)
mboth = ((info >> np.uint64(36)) & np.uint64(0xfff)).astype(np.float64)
ssum = ((info >> np.uint64(48)) & np.uint64(0xffff)).astype(np.float64)
print("per wave: rounds mean %.1f p90 %.0f max %.0f | max-lane model iters mean %.1f | exact iters mean %.1f | both mean %.1f p90 %.0f | "
      "sampled lane-rounds mean %.0f -> lane use in sample rounds %.2f" % (
          rounds.mean(), np.percentile(rounds, 90), rounds.max(), mmodel.mean(), mexact.mean(), mboth.mean(),
          np.percentile(mboth, 90), ssum.mean(), ssum.sum() / max(1.0, (rounds * 64).sum())))
busy = rounds > 0
print("waves with rounds:", int(busy.sum()), "their dur mean %.1f us, rounds mean %.1f, both mean %.1f" % (dur[busy].mean(), rounds[busy].mean(), mboth[busy].mean()))
print("waves without rounds: dur mean %.1f us, both mean %.1f" % (dur[~busy].mean(), mboth[~busy].mean()))
wave_passes = (t[:, 2] & 0xffff).astype(np.float64)
first_passes = ((t[:, 2] >> 16) & 0xffff).astype(np.float64)
lane_passes = ((t[:, 2] >> 32) & 0xffff).astype(np.float64)
pyramid_passes = ((t[:, 2].astype(np.uint64) >> np.uint64(48)) & np.uint64(0xffff)).astype(np.float64)
print("passes of the skipping loop the WAVE executed: mean %.1f (busy waves %.1f, p90 %.0f); lanes active in a pass: %.1f of 64" % (
    wave_passes.mean(), wave_passes[busy].mean(), np.percentile(wave_passes[busy], 90), lane_passes.sum() / max(1.0, wave_passes.sum())))
print("of them before the first fetch round (busy waves): %.1f" % first_passes[busy].mean())
print("passes summed over the lanes of busy waves: %.0f per wave, of them over empty cubes of the pyramid %.0f, over empty 2^3 blocks of occupied bricks or onto a tap %.0f" % (
    lane_passes[busy].mean(), pyramid_passes[busy].mean(), lane_passes[busy].mean() - pyramid_passes[busy].mean()))
for lo, hi in ((1, 8), (8, 16), (16, 32), (32, 64), (64, 128), (128, 1e9)):
    sel = busy & (rounds >= lo) & (rounds < hi)
    if sel.any():
        print("busy waves with %3.0f..%-4.0f rounds: %6d, wave passes mean %.1f, max-lane passes %.1f, lane use in rounds %.2f, wave-time share %.3f" % (
            lo, hi, sel.sum(), wave_passes[sel].mean(), mboth[sel].mean(), ssum[sel].sum() / (rounds[sel] * 64).sum(), dur[sel].sum() / dur.sum()))
# total work by duration: sum of durations vs span*slots
print("sum dur (wave-us)", dur.sum(), "-> avg resident waves", dur.sum() / t1.max())
# the longest waves: where are they in the image
order = np.argsort(-dur)[:10]
tiles_x = (w + 7) // 8
ids = np.nonzero(tl.cpu().numpy().reshape(-1, WORDS)[:, 0])[0]
for k in order:
    wid = ids[k]
    print("wave", wid, "tile", wid % tiles_x, wid // tiles_x, "dur us %.1f start %.0f" % (dur[k], t0[k]),
          "rounds %d model %d exact %d both %d sampled %d" % (rounds[k], mmodel[k], mexact[k], mboth[k], ssum[k]))
late = np.argsort(-t1)[:12]
print("last waves to finish:")
for k in late:
    wid = ids[k]
    print("wave", wid, "tile", wid % tiles_x, wid // tiles_x, "dur us %.1f start %.0f end %.0f" % (dur[k], t0[k], t1[k]),
          "rounds %d model %d exact %d both %d sampled %d" % (rounds[k], mmodel[k], mexact[k], mboth[k], ssum[k]))
for lo, hi in ((0, 50), (50, 100), (100, 200), (200, 400), (400, 800), (800, 1e9)):
    sel = (dur >= lo) & (dur < hi)
    print("dur %4.0f..%-6.0f us: %6d waves, %.3f of wave-time, mean rounds %.1f both %.1f" % (
        lo, hi, sel.sum(), dur[sel].sum() / dur.sum(), rounds[sel].mean() if sel.any() else 0, mboth[sel].mean() if sel.any() else 0))


# cycles by phase (shader clock, elapsed while the wave shares its SIMD with others)
cyc = t[:, 4:8].astype(np.float64)
word7 = t[:, 7].astype(np.uint64)
cyc[:, 3] = (word7 & np.uint64(0xffffffff)).astype(np.float64)
hist = np.stack([((word7 >> np.uint64(32 + 8 * k)) & np.uint64(0xff)).astype(np.float64) for k in range(4)], axis=1)
total_rounds = rounds.sum()
shares = [hist[:, k].sum() / total_rounds for k in range(4)]
print("fetch rounds by lanes that fetch in them: 1..8 %.3f, 9..16 %.3f, 17..32 %.3f, 33..48 %.3f, 49..64 %.3f" % (
    shares[0], shares[1], shares[2], shares[3], 1.0 - sum(shares)))
tot = cyc[:, 3].sum()
print("cycle share: setup %.3f skip-loop %.3f sample %.3f rest %.3f" % (
    cyc[:, 0].sum() / tot, cyc[:, 1].sum() / tot, cyc[:, 2].sum() / tot, 1 - cyc[:, :3].sum() / tot))
b = busy
totb = cyc[b, 3].sum()
print("busy waves (%.2f of all cycles): setup %.3f skip-loop %.3f sample %.3f; per round skip %.0f sample %.0f cycles" % (
    totb / tot, cyc[b, 0].sum() / totb, cyc[b, 1].sum() / totb, cyc[b, 2].sum() / totb,
    cyc[b, 1].sum() / rounds[b].sum(), cyc[b, 2].sum() / rounds[b].sum()))
print("idle waves: setup %.0f skip-loop %.0f total %.0f cycles per wave" % (cyc[~b, 0].mean(), cyc[~b, 1].mean(), cyc[~b, 3].mean()))

# how often a WAVE ran each part of the lane march (what tools/valu_attribution.py multiplies the parts' instruction counts with)
more = t[:, 8].astype(np.uint64)
preps = (more & np.uint64(0xffff)).astype(np.float64)
decides = ((more >> np.uint64(16)) & np.uint64(0xffff)).astype(np.float64)
flags = (more >> np.uint64(32)).astype(np.int64)
counts = {
    "waves": int(len(t)), "waves_nothing_behind_tile": int(np.count_nonzero(flags & 1)), "skip_passes": float(wave_passes.sum()),
    "rounds": float(rounds.sum()), "rounds_preparing_material_taps": float(preps.sum()), "rounds_with_decision_test": float(decides.sum()),
    "waves_handing_over": int(np.count_nonzero(flags & 2)), "waves_with_back_tap": int(np.count_nonzero(flags & 4)),
    "waves_queueing_replay": int(np.count_nonzero(flags & 8)),
    "cycles": {"all": float(cyc[:, 3].sum()), "setup": float(t[:, 4].sum()),
               "skip_loop": float(t[:, 5].sum()), "fetch_rounds_with_decision": float(t[:, 6].sum()), "decision_test": float(t[:, 9].sum())},
}
print("wave-level executions:", counts)
import json  # noqa: E402
os.makedirs("gpurun_out", exist_ok=True)
json.dump(counts, open("gpurun_out/timeline_counts.json", "w"), indent=1)
