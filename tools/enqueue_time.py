"""Diagnostic: how far the host runs ahead of the GPU in a pipelined run of the benchmark frame: time to
enqueue the frames against the time until they are finished."""
import os
import sys
import time

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
ctx.set_pipeline(2)
outs = [torch.empty(w * h * 4, dtype=torch.uint8, device="cuda") for _ in range(3)]
torch.cuda.synchronize()
for i in range(6):
    ctx.set_camera(*bench.orbit_camera(i))
    ctx.render_device(w, h, outs[i % 3].data_ptr())
ctx.synchronize()
frames = 120
t0 = time.perf_counter()
stamps = []
for i in range(frames):
    ctx.set_camera(*bench.orbit_camera(i))
    ctx.render_device(w, h, outs[i % 3].data_ptr())
    stamps.append(time.perf_counter() - t0)
t1 = time.perf_counter()
ctx.synchronize()
t2 = time.perf_counter()
print("enqueue of %d frames: %.3f ms each; finished after %.3f ms each" % (frames, (t1 - t0) / frames * 1e3, (t2 - t0) / frames * 1e3))
print("host time of frames 0..9 (ms):", ["%.2f" % (s * 1e3) for s in stamps[:10]])
print("host time of frames 100..109 (ms):", ["%.2f" % (s * 1e3) for s in stamps[100:110]])
ctx.close()
