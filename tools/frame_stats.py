"""Diagnostic: counters of a few frames of the benchmark workload (taps fetched, table look-ups,
rays handed to the wave kernels) per hit ray."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
torch.cuda.synchronize()
for step in (0, 45, 90, 200):
    ctx.set_camera(*bench.orbit_camera(step))
    st = ctx.render_bands(w, h, 0, 1, ptr=out.data_ptr(), stats=True)
    hit = max(1, st["rays_hit"])
    print("step %3d  hit rays %d  taps/ray %.2f  look-ups/ray %.2f  long rays %.2f%%  replayed %.3f%%  bricks %d  kernel %.3f ms" % (
        step, hit, st["samples_evaluated"] / hit, st["cells_read"] / hit, 100.0 * st["long_rays"] / hit,
        100.0 * st["replay_rays"] / hit, st["bricks_touched"], st["kernel_ms"]))
    lane_taps = st["samples_evaluated"] - st["long_samples"] - st["replay_samples"]
    print("   taps: lane march %.1f M, wave march %.1f M (%.0f per handed-over ray), replay %.1f M; bricks of the lane march alone %d" % (
        lane_taps / 1e6, st["long_samples"] / 1e6, st["long_samples"] / max(1, st["long_rays"]), st["replay_samples"] / 1e6, st["main_bricks"]))
    print("   rays by taps in the lane march (bin k: 2^k - 1 .. 2^(k+1) - 2):", [c for c in st["rays_by_lane_taps"][:12]])
    print("   handed-over rays by taps in the wave march:                     ", [c for c in st["rays_by_wave_taps"][:16]])
ctx.close()
