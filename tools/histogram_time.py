"""Time of vr_histogram (the reference's 100-bin intensity histogram, main.cpp:166-181) on a device-resident phantom.

python tools/histogram_time.py [edge ...]   (default 1024 2048; u16)
Prints the wall time of the call on the host (kernel + read-back + the host's bin sums) and the rate against the
voxel bytes the kernel has to read (2 B per voxel of the brick storage's payload).
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkanrenderer_amd as vr  # noqa: E402


def main():
    edges = [int(a) for a in sys.argv[1:]] or [1024, 2048]
    for n in edges:
        with vr.VolumeRenderer(0) as ctx:
            buf = torch.empty(n ** 3, dtype=torch.int16, device="cuda")
            torch.cuda.synchronize()
            ctx.phantom_generate_device(buf.data_ptr(), 2, n)
            ctx.upload_volume_device(buf.data_ptr(), 2, n, n, n)
            ctx.synchronize()
            del buf
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                h = ctx.prepare_histogram_volume_data()
                best = min(best, time.perf_counter() - t0)
            print("edge %d: vr_histogram %.2f ms, %.2f TB/s of voxel payload (2 B per voxel); bins 0..4: %s" % (
                n, best * 1e3, 2.0 * n ** 3 / best / 1e12, [round(float(x), 4) for x in h[:5]]), flush=True)


if __name__ == "__main__":
    main()
