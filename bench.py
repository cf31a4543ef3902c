#!/usr/bin/env python3
"""Headline benchmark: Gsamples/s and frames/s of the volume ray-march.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c1|c2|c3|c4|c5]

A step is one frame rendered by the HIP path through the C ABI with the volume resident in HBM and
the frame left in device memory.  The configurations are the benchmark's named workloads:

    c1  256^3 uint8 phantom,  512x512,   200 samples/ray, fixed camera (30, 20 degrees)
    c2  512^3 uint16 phantom, 1920x1080, half-voxel sampling, orbit
    c3  1024^3 uint16,        3840x2160, orbit                       (sort-first when N > 1)
    c4  2048^3 uint16,        3840x2160, 360-frame orbit (default: the configuration the metric is quoted on)
    c5  sort-last: the volume cut into one box of bricks per GPU, 16 GiB of voxels per GPU
        (2048^3 on 1 GPU ... 4096^3 on 8), 3840x2160, partial images merged over RCCL

With N > 1 and c1-c4 the frame is split sort-first into 16-row bands, the volume is replicated and
the bands are gathered on rank 0 over RCCL inside the timed region; total work is fixed, so the
scaling is strong.  With c5 every rank holds its box only; the volume grows with N (weak scaling).
`value` counts the taps the reference shader would execute for the timed frames (rays that hit the
box x (samples + 2)) per second of wall time, max over ranks.  Rank 0 prints one JSON line.

Launch: `python bench.py --gpus N` starts its N ranks itself (torch.distributed.run as a child
process, before this process touches a GPU); under a launcher that has already set WORLD_SIZE it is
one of the ranks.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TF_X = (0.0, 120.0, 300.0, 483.0)
TF_GRAPHS = [list(zip(TF_X, c)) for c in ((0, .9, 1, 1), (0, .3, .8, 1), (0, .1, .5, 1), (0, 0, .05, .6))]
# a second table, not the headline workload: low alpha everywhere, no ray becomes opaque
TRANSLUCENT_X = (0.0, 60.0, 300.0, 483.0)
TRANSLUCENT_GRAPHS = [list(zip(TRANSLUCENT_X, c)) for c in ((0.2, .9, 1, 1), (0.1, .3, .8, 1), (0.3, .1, .5, 1), (0.0, 0.004, 0.01, 0.02))]
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak
METRIC = "Gsamples/s + frames/s, 2048^3 u16 @ 4K, 1/2/4/8 MI355X; % HBM roofline"

CONFIGS = {
    "c1": dict(volume=256, dtype=1, width=512, height=512, samples=200, camera="fixed", split="sort-first"),
    "c2": dict(volume=512, dtype=2, width=1920, height=1080, samples=None, camera="orbit", split="sort-first"),
    "c3": dict(volume=1024, dtype=2, width=3840, height=2160, samples=None, camera="orbit", split="sort-first"),
    "c4": dict(volume=2048, dtype=2, width=3840, height=2160, samples=None, camera="orbit", split="sort-first"),
    "c5": dict(volume=None, dtype=2, width=3840, height=2160, samples=None, camera="orbit", split="sort-last"),
}
# c5: 16 GiB of uint16 voxels per GPU
SORT_LAST_VOLUME = {1: 2048, 2: 2560, 4: 3200, 8: 4096}


def samples_for(n):
    """Taps per ray so that the longest ray (the cube diagonal) is sampled every half voxel."""
    return int(math.ceil(2.0 * math.sqrt(3.0) * n)) - 1


def orbit_camera(step):
    return math.radians(-180.0 + (step % 360) * 1.0), math.radians(20.0)


def fixed_camera(step):
    return math.radians(30.0), math.radians(20.0)


def cpu_baseline(config="c2", threads=0):
    """The CPU checker (a port of the reference shader) timed on one whole frame of a small configuration:
    c2 = 512^3 uint16, 1920x1080, half-voxel sampling (about 10 s on 16 cores); c1 = 256^3 uint8, 512x512,
    200 samples (the configuration the CPU leg is quoted on; well under a second)."""
    from oracle import vr_oracle as o
    o.build()
    cfg = CONFIGS[config]
    n, w, h = cfg["volume"], cfg["width"], cfg["height"]
    samples = cfg["samples"] or samples_for(n)
    vol, _ = o.normalize(o.phantom(n, np.uint8 if cfg["dtype"] == 1 else np.uint16))
    lut = np.zeros((256, 4), np.float32)
    o.lut_update(lut, TF_GRAPHS)
    cam = (math.radians(30.0), math.radians(20.0))
    vp, m = o.camera(cam[0], cam[1], w, h)
    rows = (0, h)
    reps = 3 if config == "c1" else 1
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        o.render(vol, lut, vp[4:8], m, w, h, samples, rows=rows, want_colour=True, threads=threads)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    hits = o.count_hits(vp[4:8], m, w, h, rows=rows)   # every such ray takes samples + 2 taps
    taps = hits * (samples + 2)
    return {"value": taps / best / 1e9, "unit": "Gsamples/s", "cores": o.threads() if threads <= 0 else threads,
            "kind": "port",
            "sample": "CPU restatement of the shader (oracle/, OpenMP), configuration %s: %d^3 %s phantom, %dx%d, %d samples/ray, "
                      "one whole frame (%d rays hit, %.2f s%s)" % (config, n, "u8" if cfg["dtype"] == 1 else "u16", w, h, samples, hits, best,
                                                                 ", best of %d" % reps if reps > 1 else "")}


def frame_graphs(step, kind):
    """Control graphs of frame `step` for --tf-every-frame: the reference rebuilds its table from the editor's
    graphs every frame (main.cpp:800-804, 643-736); `same` hands over unchanged graphs, `values` moves the
    opacity of the third alpha point (table values change, the set of transparent entries does not), `threshold`
    moves the second alpha point along x (the transparent range itself moves, so the occupancy tables follow; it
    stays between the phantom's tissue levels 0.2 and 0.3, i.e. x in (97, 145), so that the object on screen --
    and with it the work of a frame -- is the static run's and the difference is the cost of the table change)."""
    if kind == "values":
        a3 = 0.05 + 0.04 * math.sin(2.0 * math.pi * step / 60.0)
        return TF_GRAPHS[:3] + [list(zip(TF_X, (0, 0, a3, .6)))]
    if kind == "threshold":
        x2 = 120.0 + 18.0 * math.sin(2.0 * math.pi * step / 90.0)
        return TF_GRAPHS[:3] + [list(zip((0.0, x2, 300.0, 483.0), (0, 0, .05, .6)))]
    return TF_GRAPHS


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=360, help="frames timed; 360 = the whole orbit at 1 degree a step")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c4")
    ap.add_argument("--volume", type=int, default=None, help="phantom edge length in voxels (default: the configuration's)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--mode", choices=("fast", "reference", "reference_skip"), default="fast")
    ap.add_argument("--schedule", choices=("direct_send", "binary_swap"), default="binary_swap",
                    help="c5: how the partial images are exchanged (binary swap is the schedule the benchmark names)")
    ap.add_argument("--balance", choices=("occupied", "even"), default="occupied",
                    help="c5: how the brick grid is cut into boxes: planes chosen so that every box holds about the same "
                         "number of occupied bricks (counted on the GPUs first), or an even split")
    ap.add_argument("--table", choices=("benchmark", "translucent"), default="benchmark",
                    help="transfer function: the benchmark's 4-point table, or one with low alpha everywhere (not the headline "
                         "workload: every tap of every ray has to be marched)")
    ap.add_argument("--tf-every-frame", choices=("off", "same", "values", "threshold"), default="same",
                    help="hand the control graphs to the library in front of every frame, as the reference's renderCommands does "
                         "(main.cpp:800-804): the same graphs (default: the reference's loop with an idle editor), graphs whose "
                         "opacity values move, graphs whose transparent range moves; off: table set once in the set-up")
    ap.add_argument("--exchange", choices=("python", "cabi"), default="python",
                    help="who drives the multi-GPU exchange of a frame: vulkanrenderer_amd/sortfirst.py and sortlast.py over "
                         "torch.distributed, or libvr_mg.so (include/vr_mg.h), the same sequence behind the C ABI over an RCCL "
                         "communicator of its own; with one GPU the latter still issues every transfer (to itself)")
    ap.add_argument("--stats-frames", type=int, default=12, help="frames measured with counters for the roofline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tune", action="store_true",
                    help="skip vr_tune in the set-up (the library then picks its per-lane work budget from the size of the launch); "
                         "frames do not depend on the value")
    ap.add_argument("--shading", action="store_true",
                    help="switch on the shader's commented-out lighting (not the headline workload: the reference renders without it)")
    return ap.parse_args(argv)


def launch_ranks(args):
    """--gpus N without a launcher: start the N ranks as a child process tree.  Nothing here has touched
    a GPU (torch is not even imported), so no GPU state is carried into the children."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


class Setup:
    """What both splits share: process group, device, configuration."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        # Rehearsal switches (not used by a normal run): several ranks on one GPU, exchanging over gloo
        # through host memory, to exercise the multi-rank control flow where only one GPU is at hand.
        self.backend = os.environ.get("VR_BENCH_BACKEND", "nccl")
        if os.environ.get("VR_BENCH_ONE_DEVICE"):
            local_rank = 0
        self.local_rank = local_rank
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)
        if self.world > 1:
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.device)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
        cfg = dict(CONFIGS[args.config])
        if cfg["volume"] is None:
            if self.world not in SORT_LAST_VOLUME and args.volume is None:
                raise SystemExit("c5 is defined for 1, 2, 4 or 8 GPUs (or give --volume)")
            cfg["volume"] = SORT_LAST_VOLUME.get(self.world)
        self.overridden = [k for k in ("volume", "width", "height") if getattr(args, k) is not None]
        for k in self.overridden:
            cfg[k] = getattr(args, k)
        if cfg["samples"] is None:
            cfg["samples"] = samples_for(cfg["volume"])
        self.cfg = cfg
        self.camera = fixed_camera if cfg["camera"] == "fixed" else orbit_camera

    def fence(self, ctx):
        if getattr(self, "layer", None) is not None:
            self.layer.synchronize()     # libvr_mg.so's gather runs on a stream of its own
        ctx.synchronize()
        self.torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.world == 1:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.device if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def workload(self, args, extra):
        c = self.cfg
        cam = "fixed camera (30, 20 degrees)" if c["camera"] == "fixed" else \
            "%d frames of the 360-frame orbit at 1 deg/step" % args.steps if args.steps < 360 else \
            "%d frames round the 360-frame orbit at 1 deg/step" % args.steps
        text = "%s: %d^3 %s Shepp-Logan phantom, %dx%d, %d samples/ray%s, 4-point transfer function, %s, mode %s%s%s" % (
            args.config, c["volume"], "uint8" if c["dtype"] == 1 else "uint16", c["width"], c["height"], c["samples"],
            " (half-voxel on the diagonal)" if CONFIGS[args.config]["samples"] is None else "", cam,
            args.mode + (", translucent table (not the headline workload)" if args.table == "translucent" else ""),
            ", gradient lighting on" if args.shading else "", extra)
        if self.overridden:
            text += " [not the named configuration: %s overridden]" % ", ".join(self.overridden)
        return text


def multi_gpu_layer(su, ctx):
    """This rank's end of libvr_mg.so's communicator: rank 0 makes the RCCL unique id, torch.distributed hands it round."""
    from vulkanrenderer_amd import multigpu as mgl
    torch = su.torch
    if su.world > 1 and su.backend != "nccl":
        raise SystemExit("--exchange cabi needs one GPU per rank (RCCL refuses two ranks on one device)")
    uid = mgl.unique_id() if su.rank == 0 else bytes(mgl.UNIQUE_ID_BYTES)
    if su.world > 1:
        t = torch.tensor(list(uid), dtype=torch.uint8, device=su.device)
        su.dist.broadcast(t, 0)
        uid = bytes(t.cpu().tolist())
    return mgl, mgl.MultiGpu(ctx, uid, su.rank, su.world)


def counters_of_profiles(key):
    """Counter figures of the march kernel collected with rocprofv3 --pmc on the configuration `key`
    (profiles/traffic.json, written from the round's PMC passes); None where nothing was collected."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(path)).get(key)
    except Exception:
        return None


def run_sort_first(args, su):
    torch = su.torch
    import vulkanrenderer_amd as vr
    from vulkanrenderer_amd.sortfirst import SortFirstFrame
    c, world, rank, device = su.cfg, su.world, su.rank, su.device
    n, w, h, samples = c["volume"], c["width"], c["height"], c["samples"]
    mode = {"fast": vr.MODE_FAST, "reference": vr.MODE_REFERENCE_ORDER, "reference_skip": vr.MODE_REFERENCE_SKIP}[args.mode]

    ctx = vr.VolumeRenderer(su.local_rank)
    t_setup = time.perf_counter()
    raw = torch.empty(n ** 3, dtype=torch.uint8 if c["dtype"] == 1 else torch.int16, device=device)
    torch.cuda.synchronize()
    ctx.phantom_generate_device(raw.data_ptr(), c["dtype"], n)
    ctx.upload_volume_device(raw.data_ptr(), c["dtype"], n, n, n)   # bricks + macrocells, volume replicated per rank
    ctx.synchronize()
    del raw
    torch.cuda.empty_cache()
    ctx.update_transfer_function(TRANSLUCENT_GRAPHS if args.table == "translucent" else TF_GRAPHS, reset_state=True)
    ctx.set_sampling(samples, (0.0, 0.0, 0.0, 0.0))
    ctx.set_mode(mode)
    if args.shading:
        ctx.set_shading(True)
    lane_budget = None
    ctx.set_pipeline(2)
    if not args.no_tune and mode == vr.MODE_FAST:
        # part of the set-up: the library measures its one tuning parameter on a view of the orbit, on this
        # rank's share of the frame (frames are the same for every value)
        ctx.set_camera(*su.camera(45))
        lane_budget = ctx.tune(w, h, rank if world > 1 else 0, world if world > 1 else 1)
    t_setup = time.perf_counter() - t_setup

    sf = SortFirstFrame(w, h, rank, world, device, host_staging=su.backend != "nccl")
    lib_stream = torch.cuda.ExternalStream(ctx.stream, device=device)
    target = sf.frame if world == 1 else sf.local
    # Frames in flight: the few rays a frame's main march hands over are finished beside the marches of the
    # next two frames, so consecutive frames go to three output buffers in turn (a frame that reuses the
    # buffer of one still being finished would wait for it).  Every frame is complete when the timed region
    # ends (fence() waits for all streams).
    ctx.set_pipeline(2)
    targets = [target, torch.zeros_like(target), torch.zeros_like(target)] if world == 1 else None
    torch.cuda.synchronize()     # the fill ran on torch's stream, the frames are written on the context's
    gathered = [None] * len(sf.locals)   # per band buffer: event after the gather that last read it

    # the editor's graphs as the C arrays the library takes, packed once: the timed loops measure the library's
    # call, not Python's conversion of sixteen tuples (50 us a frame)
    base_graphs = vr.PackedGraphs(TRANSLUCENT_GRAPHS if args.table == "translucent" else TF_GRAPHS)
    moving = {kind: [vr.PackedGraphs(frame_graphs(i, kind)) for i in range(max(args.steps, args.warmup, 120))] for kind in ("values", "threshold")}
    tf_kind = [args.tf_every_frame]
    layer = None
    if args.exchange == "cabi":
        _, layer = multi_gpu_layer(su, ctx)
        if targets is None and rank == 0:
            targets = [sf.frame, torch.zeros_like(sf.frame), torch.zeros_like(sf.frame)]
            torch.cuda.synchronize()

    su.layer = layer

    def frame(step):
        if tf_kind[0] != "off":   # the reference's frame: table, uniforms, dispatch (main.cpp:800-828)
            ctx.update_transfer_function(base_graphs if tf_kind[0] == "same" else moving[tf_kind[0]][step])
        ctx.set_camera(*su.camera(step))
        if layer is not None:     # bands, gather into the frame's rows and the waits between them behind one call
            layer.render_sort_first(w, h, targets[step % 3].data_ptr() if rank == 0 else 0)
            return
        if world == 1:
            ctx.render_bands(w, h, 0, 1, ptr=targets[step % 3].data_ptr())
            return
        # band buffers in turn: this frame is rendered while the previous ones are still finished and gathered
        turn = sf.turn
        if gathered[turn] is not None:
            lib_stream.wait_event(gathered[turn])
        ctx.render_bands(w, h, rank, world, ptr=sf.local.data_ptr())
        ctx.wait_frame(torch.cuda.current_stream().cuda_stream)      # the gather reads the finished frame
        sf.exchange()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        gathered[turn] = ev

    for i in range(args.warmup):
        frame(i)
    su.fence(ctx)
    t0 = time.perf_counter()
    for i in range(args.steps):
        frame(i)
    su.fence(ctx)
    elapsed = su.max_over_ranks(time.perf_counter() - t0)
    kernel_ms = ctx.march_kernel_times(min(args.steps, 1024))

    # the same loop with the table set once, with the same graphs handed over, and with graphs that change every
    # frame: short runs over the same views for all four, so that they compare (outside the timed region)
    tf_extras = None
    if world == 1 and args.table == "benchmark":
        short = min(args.steps, 120)
        per_kind = {}
        for kind in ("off", "same", "values", "threshold"):
            tf_kind[0] = kind
            for i in range(args.warmup):
                frame(i)
            su.fence(ctx)
            t1 = time.perf_counter()
            for i in range(short):
                frame(i)
            su.fence(ctx)
            per_kind[kind] = (time.perf_counter() - t1) / short * 1e3
        tf_kind[0] = args.tf_every_frame
        ctx.update_transfer_function(base_graphs, reset_state=True)
        tf_extras = {"ms_per_step": per_kind, "frames": short, "timed_region_uses": args.tf_every_frame,
                     "note": "off: table set once in the set-up; same: unchanged control graphs handed to the library in front of every "
                             "frame (the reference's loop with an idle editor); values / threshold: graphs whose opacity values / whose "
                             "transparent range move every frame"}

    # latency of single frames (untimed pass, one frame at a time on one stream): the spread over the views
    latency = None
    frame_ms = None
    if world == 1:
        ctx.set_pipeline(1)
        marks = []
        for i in range(args.steps):
            ctx.set_camera(*su.camera(i))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(lib_stream)
            ctx.render_bands(w, h, 0, 1, ptr=target.data_ptr())
            b.record(lib_stream)
            marks.append((a, b))
        ctx.synchronize()
        ms = np.array([a.elapsed_time(b) for a, b in marks])
        frame_ms = float(ms.mean())
        latency = {"mean": frame_ms, "p5": float(np.percentile(ms, 5)), "p50": float(np.percentile(ms, 50)),
                   "p95": float(np.percentile(ms, 95)), "frames": int(ms.size)}
        ctx.set_pipeline(2)

    # work of the timed frames: what the reference shader would execute
    nominal = 0
    if rank == 0:
        per_view = {}
        for i in range(args.steps):
            cam = su.camera(i)
            if cam not in per_view:
                ctx.set_camera(*cam)
                per_view[cam] = ctx.count_hit_rays(w, h) * (samples + 2)
            nominal += per_view[cam]

    # counters of a few frames of this rank's share (untimed): algorithmic bytes of the march kernel
    stats_steps = sorted(set(int(round(k * (args.steps - 1) / max(1, args.stats_frames - 1))) for k in range(args.stats_frames))) if args.steps > 1 else [0]
    keys = ("algorithmic_bytes", "samples_evaluated", "bricks_touched", "main_bricks", "replay_rays", "long_rays", "rays_hit",
            "long_samples", "replay_samples")
    agg = dict.fromkeys(keys, 0)
    for s in stats_steps:
        ctx.set_camera(*su.camera(s))
        st = ctx.render_bands(w, h, rank if world > 1 else 0, world if world > 1 else 1,
                              ptr=(target if world == 1 else sf.local).data_ptr(), stats=True)
        for k in agg:
            agg[k] += st[k]
    ctx.synchronize()
    for k in agg:
        agg[k] /= len(stats_steps)

    line = None
    if rank == 0:
        kern_s = float(np.mean(kernel_ms)) / 1e3
        # fast mode gives way to the reference-order march with skipping under tables whose rays never go dark: the
        # counting frames then show no brick touched by the lane march
        gave_way = args.mode == "fast" and agg["main_bricks"] == 0 and agg["samples_evaluated"] > 0
        fast = args.mode == "fast" and not gave_way
        brick_bytes = 512 * c["dtype"]
        # bytes the dominant kernel needs: the bricks IT touches (the tail kernels' bricks are not its), the
        # occupancy tables, the table and the frame it stores
        overhead = agg["algorithmic_bytes"] - brick_bytes * agg["bricks_touched"]
        kernel_bytes = brick_bytes * (agg["main_bricks"] if fast else agg["bricks_touched"]) + overhead
        achieved = kernel_bytes / kern_s / 1e9
        # PMC figures exist for whole-frame launches of configurations profiled with rocprofv3
        pmc = counters_of_profiles("%d_%dx%d_%s%s%s" % (n, w, h, args.mode, "_shaded" if args.shading else "",
                                                         "_translucent" if args.table == "translucent" else "")) if world == 1 else None
        if isinstance(pmc, (int, float)):
            pmc = {"traffic": pmc}
        pmc = pmc or {}
        roofline = {
            "bound": "hbm",
            "kernel": "k_march_fast" if fast else "k_march_exact (reference order with skipping: fast mode gave way under this table)" if gave_way else "k_march_exact",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc.get("traffic"),
            "traffic_source": pmc.get("source", "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE + WRITE_SIZE passes of this workload)") if pmc.get("traffic") else None,
            # what the counters say limits the kernel: not the memory system
            "limited_by": pmc.get("limited_by", "vector ALU under lane divergence" if fast else "vector ALU"),
            "valu_busy": pmc.get("valu_busy"),
            "lanes_active": pmc.get("lanes_active"),
            # the roofline that binds this kernel: share of the chip's vector-ALU lane-cycles that do useful work
            # (pipes busy x lanes active in an issued instruction), from the same counter passes
            "roofline_valu": {"bound": "valu", "frac": pmc["valu_busy"] * pmc["lanes_active"], "valu_busy": pmc["valu_busy"],
                              "lanes_active": pmc["lanes_active"], "wave_valu_insts_per_launch": pmc.get("valu_insts_per_launch"),
                              "source": pmc.get("source")} if pmc.get("valu_busy") and pmc.get("lanes_active") else None,
            # the chip's lane-cycles (1024 SIMDs x 16 lanes at 2.4 GHz) over the kernel's time per tap: what a tap costs
            # whatever fills the lanes; per tap the shader would execute and per tap this kernel fetched
            "lane_cycles_per_nominal_tap": kern_s * 2.4e9 * 16384.0 * args.steps / max(1.0, float(nominal)),
            "lane_cycles_per_fetched_tap": kern_s * 2.4e9 * 16384.0 / max(1.0, agg["samples_evaluated"] - agg["long_samples"] - agg["replay_samples"]),
            "l2_hit": pmc.get("l2_hit"),
            "algorithmic_bytes_per_launch": kernel_bytes,
            "kernel_ms_avg": kern_s * 1e3,
            "kernel_ms_min": float(np.min(kernel_ms)),
            "kernel_ms_max": float(np.max(kernel_ms)),
            # the same bytes over the device time of a whole frame (all kernels, one frame at a time)
            "frame_algorithmic_bytes": agg["algorithmic_bytes"],
            "frac_whole_frame": (agg["algorithmic_bytes"] / (frame_ms / 1e3) / 1e9 / HBM_PEAK_GBS) if frame_ms else None,
            "frac_per_step": agg["algorithmic_bytes"] / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
            "bricks_touched_per_launch": agg["main_bricks"] if fast else agg["bricks_touched"],
            "bricks_touched_per_frame": agg["bricks_touched"],
            "samples_evaluated_per_frame": agg["samples_evaluated"],
            "samples_evaluated_per_launch": agg["samples_evaluated"] - agg["long_samples"] - agg["replay_samples"],
            "evaluated_gsamples_per_s": (agg["samples_evaluated"] - agg["long_samples"] - agg["replay_samples"]) / kern_s / 1e9,
            "gather_gb_per_s": 16.0 * (agg["samples_evaluated"] - agg["long_samples"] - agg["replay_samples"]) / kern_s / 1e9,
            "handed_over_ray_fraction": agg["long_rays"] / max(1.0, agg["rays_hit"]),
            "replay_ray_fraction": agg["replay_rays"] / max(1.0, agg["rays_hit"]),
            "frames_with_counters": len(stats_steps),
        }
        line = {
            "metric": METRIC,
            "value": nominal / elapsed / 1e9,
            "unit": "Gsamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "frames_per_s": args.steps / elapsed,
            "single_frame_ms": latency,
            "tf_every_frame": tf_extras,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": su.workload(args, "" if args.tf_every_frame == "off" else ", control graphs handed to the library every frame (%s)" %
                                        {"same": "unchanged", "values": "opacity values moving", "threshold": "transparent range moving"}[args.tf_every_frame]),
                "name": args.config,
                "volume": [n, n, n], "image": [w, h], "samples_per_ray": samples, "mode": args.mode,
                "parallelism": "sort-first x%d (16-row bands, replicated volume, RCCL gather)" % world if world > 1 else "single GPU",
                "frames_in_flight": 2,
                "exchange": "libvr_mg.so (C ABI, own RCCL communicator)" if layer is not None else "vulkanrenderer_amd/sortfirst.py (torch.distributed)",
                "tf_every_frame": args.tf_every_frame,
                "shading": bool(args.shading),
                "lane_budget": lane_budget,
                "setup_s": round(t_setup, 2),
            },
            "roofline": roofline,
        }
    if layer is not None:
        su.layer = None
        layer.close()
    ctx.close()
    return line


def run_sort_last(args, su):
    torch = su.torch
    import vulkanrenderer_amd as vr
    from vulkanrenderer_amd import sortlast as sl
    c, world, rank, device = su.cfg, su.world, su.rank, su.device
    n, w, h, samples = c["volume"], c["width"], c["height"], c["samples"]
    if args.mode != "fast" or args.shading:
        raise SystemExit("c5 renders partial images: no --mode, no --shading")
    grid = sl.box_grid(world)
    box = sl.box_of_rank(rank, grid)
    schedule = args.schedule
    ctx = vr.VolumeRenderer(su.local_rank)
    t_setup = time.perf_counter()
    lo, hi = sl.upload_phantom_box(ctx, n, c["dtype"], grid, box, device)
    torch.cuda.empty_cache()
    ctx.update_transfer_function(TF_GRAPHS, reset_state=True)
    ctx.set_sampling(samples, (0.0, 0.0, 0.0, 0.0))
    if args.balance == "occupied" and world > 1:
        # the boxes of the even split count their occupied bricks, one all-reduce sums the grids, every rank
        # derives the same cut planes and loads its box of that split instead
        nb = vr.brick_grid(n, n, n)
        cell = max(1, min(8, min(b // (2 * g) for b, g in zip(nb, grid))))
        counts = sl.occupied_counts(ctx, nb, cell, device, host_staging=su.backend != "nccl")
        cuts = sl.balanced_cuts(counts, cell, nb, grid)
        lo, hi = sl.upload_phantom_box(ctx, n, c["dtype"], grid, box, device, cuts=cuts)
        torch.cuda.empty_cache()
    lane_budget = None
    if not args.no_tune:
        # part of the set-up: the library measures the lane budget of this box's partial march on a view of the orbit
        ctx.set_camera(*su.camera(45))
        lane_budget = ctx.tune(w, h)
    t_setup = time.perf_counter() - t_setup
    engine = sl.HipEngine(ctx, w, h)
    fr = sl.SortLastFrame(engine, w, h, rank, world, device, grid=grid, schedule=schedule, host_staging=su.backend != "nccl")

    layer, out_frame = None, None
    if args.exchange == "cabi":
        mgl, layer = multi_gpu_layer(su, ctx)
        layer.set_box_grid(grid)
        su.layer = layer
        out_frame = torch.zeros((h, w, 4), dtype=torch.uint8, device=device) if rank == 0 else None
        torch.cuda.synchronize()
        cabi_schedule = mgl.BINARY_SWAP if schedule == "binary_swap" else mgl.DIRECT_SEND

    def frame(step, mark=None):
        ctx.set_camera(*su.camera(step))
        if layer is not None and mark is None:    # the timed frames; the per-stage pass below uses the Python driver's marks
            fr.remarched = layer.render_sort_last(w, h, cabi_schedule, out_frame.data_ptr() if rank == 0 else 0)
            return out_frame
        return fr.render(mark=mark)

    for i in range(args.warmup):
        frame(i)
    su.fence(ctx)
    t0 = time.perf_counter()
    remarched = 0
    for i in range(args.steps):
        frame(i)
        remarched += fr.remarched
    su.fence(ctx)
    elapsed = su.max_over_ranks(time.perf_counter() - t0)

    # device time per stage on this rank (untimed pass over a few views, events on the context's stream)
    stage_names = ("partial_march", "exchange_merge", "decide", "remarch", "gather")
    views = sorted(set(int(round(k * (args.steps - 1) / max(1, min(args.stats_frames, args.steps) - 1))) for k in range(min(args.stats_frames, args.steps)))) \
        if args.steps > 1 else [0]
    stage_ms = np.zeros((len(views), len(stage_names)))
    for vi, s in enumerate(views):
        evs = []

        def mark(_name):
            e = torch.cuda.Event(enable_timing=True)
            e.record(engine.stream)
            evs.append(e)
        mark("start")
        frame(s, mark=mark)
        ctx.synchronize()
        torch.cuda.synchronize()
        for k in range(len(stage_names)):
            stage_ms[vi, k] = evs[k].elapsed_time(evs[k + 1])
    mine = torch.tensor(stage_ms.mean(axis=0), dtype=torch.float64)
    per_rank = [mine.clone() for _ in range(world)]
    if world > 1:
        if su.backend == "nccl":
            got = [torch.zeros_like(mine, device=device) for _ in range(world)]
            su.dist.all_gather(got, mine.to(device))
            per_rank = [g.cpu() for g in got]
        else:
            su.dist.all_gather(per_rank, mine)
    nominal = 0
    if rank == 0:
        for i in range(args.steps):
            ctx.set_camera(*su.camera(i))
            nominal += ctx.count_hit_rays(w, h) * (samples + 2)
    line = None
    if rank == 0:
        stages = {name: {"mean_over_ranks_ms": float(np.mean([p[k] for p in per_rank])), "max_over_ranks_ms": float(np.max([p[k] for p in per_rank]))}
                  for k, name in enumerate(stage_names)}
        march = [float(p[0]) for p in per_rank]
        line = {
            "metric": METRIC,
            "value": nominal / elapsed / 1e9,
            "unit": "Gsamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "frames_per_s": args.steps / elapsed,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": su.workload(args, ", sort-last over %d box%s (%dx%dx%d), %s" % (world, "" if world == 1 else "es", grid[0], grid[1], grid[2], schedule.replace("_", " "))),
                "name": args.config,
                "volume": [n, n, n], "image": [w, h], "samples_per_ray": samples,
                "parallelism": "sort-last x%d (boxes of bricks %dx%dx%d, %s over RCCL, byte-exact re-march of undecided pixels)" % (world, grid[0], grid[1], grid[2], schedule),
                "bricks_of_rank0": [list(lo), list(hi)], "balance": args.balance if world > 1 else "one box", "lane_budget_rank0": lane_budget,
                "exchange": "libvr_mg.so (C ABI, own RCCL communicator)" if layer is not None else "vulkanrenderer_amd/sortlast.py (torch.distributed)",
                "setup_s": round(t_setup, 2),
            },
            "stages": stages,
            "partial_march_ms_per_rank": march,
            "partial_march_max_over_mean": max(march) / max(1e-9, float(np.mean(march))),
            "remarched_pixel_fraction": remarched / float(args.steps) / float(w * h),
            "exchange_bytes_per_rank": (world - 1) * fr.span * 20,
            "roofline": {
                "bound": "hbm", "kernel": "k_march_partial", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "note": "the partial march has no counting variant: no algorithmic byte count for this configuration",
                "kernel_ms_avg": float(np.mean(march)),
            },
        }
    if layer is not None:
        su.layer = None
        layer.close()
    ctx.close()
    return line


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))

    if os.environ.get("VR_BENCH_WATCHDOG"):
        # rehearsals only: a rank stuck for this many seconds prints its Python stacks and exits
        import faulthandler
        log = open(os.environ["VR_BENCH_WATCHDOG_FILE"] + "." + os.environ.get("RANK", "0"), "w") if os.environ.get("VR_BENCH_WATCHDOG_FILE") else sys.stderr
        faulthandler.dump_traceback_later(float(os.environ["VR_BENCH_WATCHDOG"]), exit=True, file=log)

    su = Setup(args)
    line = run_sort_last(args, su) if su.cfg["split"] == "sort-last" else run_sort_first(args, su)
    if su.rank == 0:
        if su.world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline("c2")
            line["cpu_baseline_c1"] = cpu_baseline("c1")
        print(json.dumps(line), flush=True)
    if su.world > 1:
        su.dist.destroy_process_group()


if __name__ == "__main__":
    main()
