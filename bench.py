#!/usr/bin/env python3
"""Headline benchmark: Gsamples/s of the volume ray-march, 2048^3 uint16 phantom at 3840x2160.

    python bench.py --gpus N --steps K --warmup W

A step is one frame of the orbit (camera h = -180 deg + step * 1 deg, v = 20 deg) rendered by the
HIP path through the C ABI with the volume resident in HBM.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) the frame is split sort-first into 16-row bands, the
volume is replicated, and the bands are gathered on rank 0 over RCCL inside the timed region.
`value` counts the taps the reference shader would execute for these frames (rays that hit the box
x (samples + 2)) per second of wall time, max over ranks; total work is fixed, so scaling is strong.
Rank 0 prints one JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TF_X = (0.0, 120.0, 300.0, 483.0)
TF_GRAPHS = [list(zip(TF_X, c)) for c in ((0, .9, 1, 1), (0, .3, .8, 1), (0, .1, .5, 1), (0, 0, .05, .6))]
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def samples_for(n):
    """Taps per ray so that the longest ray (the cube diagonal) is sampled every half voxel."""
    return int(math.ceil(2.0 * math.sqrt(3.0) * n)) - 1


def orbit_camera(step):
    return math.radians(-180.0 + (step % 360) * 1.0), math.radians(20.0)


def cpu_baseline(threads=0):
    """The CPU checker (a port of the reference shader) timed on a bounded sample of the workload:
    512^3 uint16 phantom, 1920x1080, half-voxel sampling, one whole frame (about 10 s on 16 cores)."""
    from oracle import vr_oracle as o
    o.build()
    n, w, h = 512, 1920, 1080
    samples = samples_for(n)
    vol, _ = o.normalize(o.phantom(n, np.uint16))
    lut = np.zeros((256, 4), np.float32)
    o.lut_update(lut, TF_GRAPHS)
    cam = (math.radians(30.0), math.radians(20.0))
    vp, m = o.camera(cam[0], cam[1], w, h)
    rows = (0, h)
    t0 = time.perf_counter()
    img, col = o.render(vol, lut, vp[4:8], m, w, h, samples, rows=rows, want_colour=True, threads=threads)
    dt = time.perf_counter() - t0
    strip_hits = o.count_hits(vp[4:8], m, w, h, rows=rows)   # every such ray takes samples + 2 taps
    taps = strip_hits * (samples + 2)
    return {"value": taps / dt / 1e9, "unit": "Gsamples/s", "cores": o.threads() if threads <= 0 else threads,
            "kind": "port",
            "sample": "CPU restatement of the shader (oracle/, OpenMP): 512^3 u16 phantom, 1920x1080, %d samples/ray, "
                      "rows %d-%d (%d rays hit, %.2f s)" % (samples, rows[0], rows[1] - 1, strip_hits, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=360, help="frames timed; 360 = the whole orbit at 1 degree a step")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--volume", type=int, default=2048, help="phantom edge length in voxels")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--mode", choices=("fast", "reference", "reference_skip"), default="fast")
    ap.add_argument("--stats-frames", type=int, default=12, help="frames measured with counters for the roofline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tune", action="store_true",
                    help="let the library pick its per-lane work budget by timing one view (vr_tune); the default "
                         "value is the one measured best for the default workload")
    ap.add_argument("--shading", action="store_true",
                    help="switch on the shader's commented-out lighting (not the headline workload: the reference renders without it)")
    args = ap.parse_args()

    if os.environ.get("VR_BENCH_WATCHDOG"):
        # rehearsals only: a rank stuck for this many seconds prints its Python stacks and exits
        import faulthandler
        log = open(os.environ["VR_BENCH_WATCHDOG_FILE"] + "." + os.environ.get("RANK", "0"), "w") if os.environ.get("VR_BENCH_WATCHDOG_FILE") else sys.stderr
        faulthandler.dump_traceback_later(float(os.environ["VR_BENCH_WATCHDOG"]), exit=True, file=log)
    import torch
    import torch.distributed as dist

    import vulkanrenderer_amd as vr
    from vulkanrenderer_amd.sortfirst import SortFirstFrame

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    # Rehearsal switches (not used by a normal run): several ranks on one GPU, exchanging over gloo
    # through host memory, to exercise the multi-rank control flow where only one GPU is at hand.
    backend = os.environ.get("VR_BENCH_BACKEND", "nccl")
    if os.environ.get("VR_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n, w, h = args.volume, args.width, args.height
    samples = samples_for(n)
    mode = {"fast": vr.MODE_FAST, "reference": vr.MODE_REFERENCE_ORDER, "reference_skip": vr.MODE_REFERENCE_SKIP}[args.mode]

    ctx = vr.VolumeRenderer(local_rank)
    t_setup = time.perf_counter()
    raw = torch.empty(n ** 3, dtype=torch.int16, device=device)
    torch.cuda.synchronize()
    ctx.phantom_generate_device(raw.data_ptr(), vr.DTYPE_U16, n)
    ctx.upload_volume_device(raw.data_ptr(), vr.DTYPE_U16, n, n, n)   # bricks + macrocells, volume replicated per rank
    ctx.synchronize()
    del raw
    torch.cuda.empty_cache()
    ctx.update_transfer_function(TF_GRAPHS, reset_state=True)
    ctx.set_sampling(samples, (0.0, 0.0, 0.0, 0.0))
    ctx.set_mode(mode)
    if args.shading:
        ctx.set_shading(True)
    lane_budget = None
    if args.tune and mode == vr.MODE_FAST:
        # part of the set-up: the library measures its one tuning parameter on a view of the orbit
        ctx.set_camera(*orbit_camera(45))
        lane_budget = ctx.tune(w, h, rank if world > 1 else 0, world if world > 1 else 1)
    t_setup = time.perf_counter() - t_setup

    sf = SortFirstFrame(w, h, rank, world, device, host_staging=backend != "nccl")
    lib_stream = torch.cuda.ExternalStream(ctx.stream, device=device)
    target = sf.frame if world == 1 else sf.local
    # Two frames in flight: the few rays a frame's main march hands over are finished beside the next
    # frame's march, so consecutive frames go to two output buffers in turn.  Every frame is complete
    # when the timed region ends (fence() below waits for both streams).
    ctx.set_pipeline(2)
    targets = [target, torch.zeros_like(target)] if world == 1 else None
    gathered = [None, None]      # per band buffer: event after the gather that last read it

    def frame(step):
        ctx.set_camera(*orbit_camera(step))
        if world == 1:
            ctx.render_bands(w, h, 0, 1, ptr=targets[step & 1].data_ptr())
            return
        # two band buffers in turn: this frame is rendered while the previous one is still in the gather
        turn = sf.turn
        if gathered[turn] is not None:
            lib_stream.wait_event(gathered[turn])
        ctx.render_bands(w, h, rank, world, ptr=sf.local.data_ptr())
        ctx.wait_frame(torch.cuda.current_stream().cuda_stream)      # the gather reads the finished frame
        sf.exchange()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        gathered[turn] = ev

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        frame(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = ctx.march_kernel_times(min(args.steps, 1024))

    # latency of single frames (untimed pass, one frame at a time on one stream): the spread over the orbit
    latency = None
    if world == 1:
        ctx.set_pipeline(1)
        marks = []
        for i in range(args.steps):
            ctx.set_camera(*orbit_camera(i))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(lib_stream)
            ctx.render_bands(w, h, 0, 1, ptr=target.data_ptr())
            b.record(lib_stream)
            marks.append((a, b))
        ctx.synchronize()
        ms = np.array([a.elapsed_time(b) for a, b in marks])
        latency = {"mean": float(ms.mean()), "p5": float(np.percentile(ms, 5)), "p50": float(np.percentile(ms, 50)),
                   "p95": float(np.percentile(ms, 95)), "frames": int(ms.size)}
        ctx.set_pipeline(2)

    # work of the timed frames: what the reference shader would execute
    nominal = 0
    if rank == 0:
        for i in range(args.steps):
            ctx.set_camera(*orbit_camera(i))
            nominal += ctx.count_hit_rays(w, h) * (samples + 2)

    # counters of a few frames of this rank's share (untimed): algorithmic bytes of the march kernel
    stats_steps = sorted(set(int(round(k * (args.steps - 1) / max(1, args.stats_frames - 1))) for k in range(args.stats_frames))) if args.steps > 1 else [0]
    agg = {"algorithmic_bytes": 0, "samples_evaluated": 0, "bricks_touched": 0, "replay_rays": 0, "rays_hit": 0}
    for s in stats_steps:
        ctx.set_camera(*orbit_camera(s))
        st = ctx.render_bands(w, h, rank if world > 1 else 0, world if world > 1 else 1,
                              ptr=(target if world == 1 else sf.local).data_ptr(), stats=True)
        for k in agg:
            agg[k] += st[k]
    ctx.synchronize()
    for k in agg:
        agg[k] /= len(stats_steps)

    if rank == 0:
        kern_s = float(np.mean(kernel_ms)) / 1e3
        achieved = agg["algorithmic_bytes"] / kern_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        # the counters were collected on whole-frame launches: no figure for a rank's share of the bands
        if world == 1 and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("%d_%dx%d_%s%s" % (n, w, h, args.mode, "_shaded" if args.shading else ""))
            except Exception:
                traffic = None
        line = {
            "metric": "Gsamples/s + frames/s, 2048^3 u16 @ 4K, 1/2/4/8 MI355X; % HBM roofline",
            "value": nominal / elapsed / 1e9,
            "unit": "Gsamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "frames_per_s": args.steps / elapsed,
            "single_frame_ms": latency,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%d^3 uint16 Shepp-Logan phantom, %dx%d, %d samples/ray (half-voxel on the diagonal), 4-point "
                            "transfer function, 360-frame orbit at 1 deg/step, mode %s%s" % (n, w, h, samples, args.mode, ", gradient lighting on" if args.shading else ""),
                "volume": [n, n, n], "image": [w, h], "samples_per_ray": samples, "mode": args.mode,
                "parallelism": "sort-first x%d (16-row bands, replicated volume, RCCL gather)" % world if world > 1 else "single GPU",
                "frames_in_flight": 2,
                "shading": bool(args.shading),
                "lane_budget": lane_budget,
                "setup_s": round(t_setup, 2),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_march_fast" if args.mode == "fast" else "k_march_exact",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": agg["algorithmic_bytes"],
                "kernel_ms_avg": kern_s * 1e3,
                "kernel_ms_min": float(np.min(kernel_ms)),
                "kernel_ms_max": float(np.max(kernel_ms)),
                "bricks_touched_per_launch": agg["bricks_touched"],
                "samples_evaluated_per_launch": agg["samples_evaluated"],
                "evaluated_gsamples_per_s": agg["samples_evaluated"] / kern_s / 1e9,
                "gather_gb_per_s": 16.0 * agg["samples_evaluated"] / kern_s / 1e9,
                "replay_ray_fraction": agg["replay_rays"] / max(1.0, agg["rays_hit"]),
                "frames_with_counters": len(stats_steps),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
