"""Time of the volume upload's device work (re-tiling into bricks + macrocells + maximum), HIP events.

python tools/ingest_time.py [edge ...]      (default 2048; dtype u16; INGEST_BE=1: big-endian input)
Prints per edge: ms of vr_upload_volume_device on a device-resident phantom (best of the uploads after the
first, which allocates; the later ones find the buffers in place), the bytes it has to move (u16: 2 B read per voxel, 1472 B of voxels + 4 B of macrocell written per brick) and the rate.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vulkanrenderer_amd as vr  # noqa: E402


def main():
    edges = [int(a) for a in sys.argv[1:]] or [2048]
    for n in edges:
        with vr.VolumeRenderer(0) as ctx:
            buf = torch.empty(n ** 3, dtype=torch.int16, device="cuda")
            torch.cuda.synchronize()
            ctx.phantom_generate_device(buf.data_ptr(), 2, n)
            ctx.synchronize()
            best = 1e9
            for rep in range(int(os.environ.get("INGEST_REPS", "4"))):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                stream = torch.cuda.ExternalStream(ctx.stream)
                e0.record(stream)
                ctx.upload_volume_device(buf.data_ptr(), 2, n, n, n)
                e1.record(stream)
                ctx.synchronize()
                ms = e0.elapsed_time(e1)
                if rep > 0:
                    best = min(best, ms)
            nb = (n + 4 + 7) // 8
            moved = 2 * n ** 3 + (1472 + 4) * nb ** 3
            print("edge %d: upload %.2f ms, %.1f GB algorithmic (raw once + bricks + macrocells), %.2f TB/s = %.0f %% of 8 TB/s"
                  % (n, best, moved / 1e9, moved / best / 1e9, 100 * moved / best / 1e9 / 8.0), flush=True)


if __name__ == "__main__":
    main()
