#!/usr/bin/env python3
"""bench.py - throughput of the render loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4] [--spp-per-step S]

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks
(one process per GPU, `torch.distributed.run` on 127.0.0.1) before anything touches HIP,
relays rank 0's JSON line and fails if any rank fails - the stand-in for the reference's
thread fan-out (src/raytracer.rs:146-171).  Launched under torch.distributed.run it is one rank.

A step is one pass of the hot path over the whole frame: by default all samples of every pixel
(the whole of calculate_some_pxls, src/raytracer.rs:232-271, for one frame; `--spp-per-step` renders
a part of them per step, the progressive form), followed by the quantiser and, with N > 1 GPUs, one
RCCL all-gather of the packed bands.  The default workload is the one the headline metric is quoted
on: 10,000 random spheres at 3840x2160, 256 spp - one kernel launch per step, so `ms_per_step` is
the metric's ms per frame.  The frame is split over the N ranks by interleaved 8-row bands (strong
scaling: the total work is fixed).  With N > 1 the line also carries, per rank, kernel time, segments
and the time of the fan-in (`config.ranks`).

Rank 0 prints ONE JSON line.  `value` is Mrays/s over all ranks, a ray being one
path segment (one closest-hit query over the list), counted on the device.

`roofline`: the render kernel is bound by VALU issue.  `achieved` is the rate of executed
VALU lane-operations (one lane of one vector instruction; a fused multiply-add counts once):
lane-operations per segment from the committed counter pass (profiles/sq_counters.json,
SQ_INSTS_VALU x SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU of this kernel on this workload)
x the segments of this run / the kernel time of this run (HIP events).  `peak` is the rate at
which the chip issues them, 256 CUs x 4 SIMDs x 32 lanes per clock x 2.4 GHz = 78.6 T/s (the
157.3 TFLOP/s FP32 vector peak counts a fused multiply-add as two).  The work of the reference's
linear scan that the box tree skips is reported separately (`algorithmic_equivalent`).
The counter passes are used only if they were taken on the library build that is running
(`library_build`, rt_build_id); otherwise `frac` is null and `frac_null_because` says why.
`traffic`, `hbm_gbps`, `hbm_frac_of_8TBps`: HBM bytes of the launch from the same passes, and their
rate over this run's kernel time against 8 TB/s.  `config.others` holds the same figures for the
other configurations, measured briefly on a single GPU.

Rehearsal of the multi-rank path on a box with one GPU (not a measurement):
RT_BENCH_REHEARSAL=1 puts every rank on device 0, gathers through gloo and host
memory instead of RCCL, and has rank 0 render the same samples alone and compare
the gathered frame with its own, pixel for pixel.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# FP32 vector peak of MI355X: 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz, FMA = 2 FLOP
PEAK_FP32_VECTOR_TFLOPS = 157.3
# VALU lane-operations per second the chip can issue: 1024 SIMDs x 32 lanes per clock x 2.4 GHz
# (= the FP32 peak for code without fused multiply-add, which the reference's arithmetic is)
PEAK_VALU_LANE_TOPS = PEAK_FP32_VECTOR_TFLOPS / 2.0
METRIC = "Mrays/s (whole node) + ms/frame at 4K 256spp, 1/2/4/8 MI355X"


def launch_ranks(args):
    """Parent of a multi-GPU run: starts one rank per GPU and relays rank 0's line.  Runs before
    anything initialises HIP in this process, and execs nothing."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    for l in proc.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if proc.returncode != 0 or not lines:
        sys.exit(proc.returncode or 1)
    print(lines[-1], flush=True)
    sys.exit(0)


def host_cores():
    """Threads the CPU baseline may use: the CPUs this process is allowed on (its affinity mask, what
    `nproc` prints), cut to a cgroup CPU quota if one is set, and to the 256 threads the oracle takes."""
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(allowed, quota or allowed, 256)), allowed, quota


def cpu_baseline(wl, tables, budget_s=15.0):
    """The CPU oracle (a C restatement of the reference's loop, row-interleaved
    threads like src/raytracer.rs:245, one per host core) on a bounded sample of the same workload:
    1 spp of every `stride`-th band of the frame, band set after band set until the time budget is used."""
    from raytracing_amd import make_params
    from tests import oracle_lib

    orc = oracle_lib.load()
    cores, allowed, quota = host_cores()
    w, h = wl["width"], wl["height"]
    total_bands = (h + 7) // 8
    # a call renders the rows of total_bands / stride bands; every thread should get a few of them
    stride = 64
    while stride > 1 and (total_bands // stride) * 8 < 2 * cores:
        stride //= 2
    accum = np.zeros((h, w, 4), dtype=np.float32)
    seg = 0
    bands = 0
    t0 = time.perf_counter()
    # (bit-reversed order: the band sets taken so far are spread evenly over the frame)
    order = sorted(range(stride), key=lambda b: int(format(b, "06b")[::-1], 2))
    for begin in order:
        if begin >= total_bands:
            continue
        p = make_params(w, h, spp=wl["spp"], sample_begin=0, sample_count=1, band_begin=begin,
                        band_stride=stride)
        _, _, s = orc.render(wl["flat"], wl["camera"], tables, p, threads=cores, accum=accum, want_ids=False)
        seg += s
        bands += len(range(begin, total_bands, stride))
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {
        "value": round(seg / dt / 1e6, 5), "unit": "Mrays/s", "cores": cores, "kind": "port",
        "host": {"cpus_allowed": allowed, "cpu_count": os.cpu_count(), "cgroup_cpu_quota": quota},
        "sample": "%d of %d 8-row bands of the %dx%d frame at 1 of %d spp, %d segments in %.1f s on %d threads"
                  % (bands, total_bands, w, h, wl["spp"], seg, dt, cores),
    }


def library_build_id():
    from raytracing_amd import _lib
    return _lib.load().rt_build_id().decode()


def profile_record(fname, key):
    """One workload's record of a committed counter file under profiles/ (tools/collect_counters.py), and
    why it cannot be used, if so: absent, or taken on a library built from other kernel sources than the
    one being benched (rt_build_id)."""
    try:
        with open(os.path.join(ROOT, "profiles", fname)) as f:
            rec = json.load(f)[key]
    except (OSError, ValueError, KeyError, TypeError):
        return None, "no committed counter pass for this workload in profiles/" + fname
    have, want = rec.get("library_build"), library_build_id()
    if have != want:
        return None, "profiles/%s[%s] was taken on library build %s, this is %s" % (fname, key, have, want)
    return rec, None


def counters_from_profile(key):
    """Per-segment VALU figures of the render kernel on one workload from the committed rocprofv3
    --pmc passes (profiles/sq_counters.json); (None, reason) if they do not belong to this build."""
    rec, why = profile_record("sq_counters.json", key)
    if rec is not None and not rec.get("valu_lane_ops_per_segment"):
        return None, "record without lane operations"
    return rec, why


def hbm_traffic_from_profile(key, spp_per_launch):
    """HBM bytes per launch of render_tiles from the committed rocprofv3 --pmc passes
    (profiles/hbm_traffic.json); None if absent, from another build or another launch size."""
    rec, _ = profile_record("hbm_traffic.json", key)
    if rec is None or rec.get("spp_per_launch") != spp_per_launch:
        return None
    return rec["bytes_per_launch"]


def valu_roofline(key, kernel, segments_per_launch, kernel_ms, spp_per_launch, flops_seg, flat):
    """`roofline` object for one kernel launch size: executed VALU lane-operations against the rate
    the chip issues them (see the module docstring); the counter-derived parts are None when no
    committed counter pass covers this workload."""
    prof, why_not = counters_from_profile(key)
    seg_per_s = segments_per_launch / (kernel_ms * 1e-3)
    traffic = hbm_traffic_from_profile(key, spp_per_launch)
    if traffic is not None and prof is not None and prof.get("segments_per_launch"):
        # the counter pass rendered the whole frame in one launch; a rank of a multi-GPU run renders its
        # share of the bands, and the traffic (nearly all of it per-sample staging) goes with the segments
        traffic = int(traffic * segments_per_launch / prof["segments_per_launch"])
    out = {"bound": "valu", "achieved": None, "peak": PEAK_VALU_LANE_TOPS, "unit": "TFLOP/s", "frac": None,
           "traffic": traffic,
           # the HBM side of the same launch: bytes moved / kernel time against the 8 TB/s peak
           "hbm_gbps": None if traffic is None else round(traffic / (kernel_ms * 1e-3) / 1e9, 1),
           "hbm_frac_of_8TBps": None if traffic is None else round(traffic / (kernel_ms * 1e-3) / 8e12, 4),
           "kernel": kernel, "kernel_ms": round(float(kernel_ms), 3), "library_build": library_build_id(),
           "definition": "achieved = executed VALU lane-operations per second in units of 1e12 (a lane-operation "
                         "is one lane of one vector instruction = at most one unfused flop; an FMA counts once); "
                         "peak = 1024 SIMDs x 32 lanes/clk x 2.4 GHz"}
    if prof is None:
        out["frac_null_because"] = why_not
    else:
        lane_ops = prof["valu_lane_ops_per_segment"] * seg_per_s / 1e12
        out["achieved"] = round(lane_ops, 3)
        out["frac"] = round(lane_ops / PEAK_VALU_LANE_TOPS, 4)
        # the same split in two: share of the VALU issue slots used (one wave instruction per SIMD every
        # two clocks), and lanes active per instruction; frac = issue_slots_used x active_lanes / 64
        issue = prof["valu_wave_insts_per_wave_segment"] / 64.0 * seg_per_s / (1024 * 2.4e9 / 2.0)
        out["valu_issue_slots_used"] = round(issue, 4)
        out["counters"] = {
            "valu_lane_ops_per_segment": prof["valu_lane_ops_per_segment"],
            "valu_wave_insts_per_wave_segment": prof["valu_wave_insts_per_wave_segment"],
            "active_lanes_per_valu_inst": prof["active_lanes_per_valu_inst"],
            "fma_share_of_valu_insts": prof.get("fma_share_of_valu_insts"),
            "source": prof.get("source"),
        }
        if prof.get("fma_lane_ops_per_segment") is not None:
            # flops with a fused multiply-add counted twice, against the 157.3 TFLOP/s peak
            fl = (prof["valu_lane_ops_per_segment"] + prof["fma_lane_ops_per_segment"]) * seg_per_s / 1e12
            out["frac_of_fp32_peak_fma2"] = round(fl / PEAK_FP32_VECTOR_TFLOPS, 4)
    c = flat.counts
    alg = flops_seg * seg_per_s / 1e12
    out["algorithmic_equivalent"] = {
        "flops_per_segment": flops_seg, "tflops": round(alg, 3),
        "of_fp32_peak": round(alg / PEAK_FP32_VECTOR_TFLOPS, 4),
        "note": "work of the reference's linear scan (18/sphere + 14/plane + 18/cube + 45/triangle + 80 per "
                "segment, mul/add counted singly) x segments / kernel time; a box tree (scenes with a run of >= 32 "
                "spheres or >= 64 triangles) skips most of it, so this is not a utilisation there",
        "tree": kernel.startswith("render_tiles<2") or kernel.startswith("render_tiles<3"),
    }
    return out


def flat_scan_probe(r, wl, flops_seg):
    """The same frame through the list-order scan (RT_HIP_FLAT_SCAN=1: every sphere of the
    list is tested for every segment, eight per step through the packed-FMA rejection
    test, exactly the work the reference's linear scan does): 8 spp, timed with HIP events."""
    from raytracing_amd import make_params
    w, h, spp = wl["width"], wl["height"], wl["spp"]
    os.environ["RT_HIP_FLAT_SCAN"] = "1"
    try:
        r.set_scene(wl["flat"])
    finally:
        del os.environ["RT_HIP_FLAT_SCAN"]
    r.begin_frame(w, h)
    r.render(make_params(w, h, spp=spp, sample_begin=0, sample_count=1))   # warm-up
    r.sync()
    seg0 = r.ray_count()
    r.timer_start()
    r.render(make_params(w, h, spp=spp, sample_begin=1, sample_count=8))
    ms = r.timer_stop()
    seg = r.ray_count() - seg0
    kernel = r.kernel_variant()
    r.set_scene(wl["flat"])
    achieved = flops_seg * seg / (ms * 1e-3) / 1e12
    return {"kernel": kernel, "kernel_ms": round(ms, 3), "mrays_per_s": round(seg / ms / 1e3, 3),
            "algorithmic_tflops": round(achieved, 3),
            "of_fp32_peak": round(achieved / PEAK_FP32_VECTOR_TFLOPS, 4),
            "sample": "8 spp of the full frame, %d segments; every sphere of the list tested for every segment "
                      "(7 FMAs per sphere in the rejection test)" % seg}


def other_workload(name, tables, launches=3):
    """A short single-GPU measurement of another configuration: `launches` launches of up to 512 spp of
    the full frame after one warm-up, kernel time from HIP events.  A workload that is a run of several
    frames (c1: the reference's ten) is rendered the way the library offers for that, all frames in one
    rt_render_frames launch; the time of its first frame alone in a launch of its own is given beside it."""
    from raytracing_amd import Renderer, make_params, workloads
    wl = workloads.workload(name)
    w, h, spp = wl["width"], wl["height"], wl["spp"]
    sps = min(512, spp)
    cams = wl.get("cameras")
    n_frames = len(cams) if cams else 1
    flops_seg = workloads.flops_per_segment(wl["flat"])
    with Renderer(0, tables) as r:
        r.set_scene(wl["flat"]); r.set_camera(wl["camera"]); r.begin_frames(w, h, n_frames)
        kernel = r.kernel_variant()

        def launch(begin):
            p = make_params(w, h, spp=spp, sample_begin=begin, sample_count=sps)
            if cams:
                r.render_frames(p, cams)
            else:
                r.render(p)

        launch(0)
        r.sync()
        r.begin_frames(w, h, n_frames)
        seg0 = r.ray_count()
        ms = []
        for k in range(launches):
            r.timer_start()
            launch((k * sps) % max(1, spp - sps + 1))
            ms.append(r.timer_stop())
        seg = r.ray_count() - seg0
        alone = None
        if cams:
            one = []
            for k in range(launches + 1):
                r.timer_start()
                r.render(make_params(w, h, spp=spp, sample_begin=0, sample_count=sps))
                one.append(r.timer_stop())
            alone = {"kernel_ms": round(float(np.mean(one[1:])), 4),
                     "mrays_per_s": round(seg / launches / n_frames / float(np.mean(one[1:])) / 1e3, 3)}
    kms = float(np.mean(ms))
    roof = valu_roofline(name, kernel, seg / launches, kms, sps, flops_seg, wl["flat"])
    out = {"workload": "%s: %s" % (name, wl["description"]), "spp_per_launch": sps, "kernel": kernel,
           "frames_per_launch": n_frames, "kernel_ms": round(kms, 4 if kms < 1 else 3),
           "mrays_per_s": round(seg / launches / kms / 1e3, 3),
           "ms_per_frame": round(kms * spp / sps / n_frames, 4 if kms < 1 else 2),
           "roofline_frac": roof["frac"], "valu_lane_tops": roof["achieved"],
           "frac_null_because": roof.get("frac_null_because"),
           "algorithmic_tflops": roof["algorithmic_equivalent"]["tflops"],
           "algorithmic_of_fp32_peak": roof["algorithmic_equivalent"]["of_fp32_peak"],
           "hbm_bytes_per_launch": roof["traffic"], "hbm_gbps": roof["hbm_gbps"],
           "hbm_frac_of_8TBps": roof["hbm_frac_of_8TBps"]}
    if alone is not None:
        out["one_frame_per_launch"] = alone
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--spp-per-step", type=int, default=0,
                    help="samples of every pixel one step renders (0 = the frame's spp: a step is a whole frame)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-flat-probe", action="store_true")
    ap.add_argument("--others", default="c1,c2,c3,c5",
                    help="other configurations measured briefly on rank 0 of a single-GPU run ('' = none)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        launch_ranks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: librt_hip.so has no CPU path")
    rehearsal = os.environ.get("RT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 and rehearsal:
        dist.init_process_group("gloo")
    elif world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from raytracing_amd import Renderer, load_tables, make_params
    from raytracing_amd import multi_gpu, workloads

    tables = load_tables()
    wl = workloads.workload(args.workload)
    w, h, spp = wl["width"], wl["height"], wl["spp"]
    sps = max(1, min(args.spp_per_step or spp, spp))
    launches_per_step = -(-sps // 512)       # rt_render cuts a sample range into launches of at most 512 samples
    total_bands = (h + 7) // 8
    max_bands = len(range(0, total_bands, world))
    flops_seg = workloads.flops_per_segment(wl["flat"])

    cams = wl.get("cameras")                 # a run of several frames: one rt_render_frames launch per step
    n_frames = len(cams) if cams else 1
    r = Renderer(local_rank, tables)
    r.set_scene(wl["flat"])
    r.set_camera(wl["camera"])
    r.begin_frames(w, h, n_frames)
    kernel = r.kernel_variant()
    block = torch.zeros((max_bands * 8, w), dtype=torch.int32, device="cuda")
    gathered = torch.zeros((world * max_bands * 8, w), dtype=torch.int32, device="cuda") if world > 1 else None
    frame = [None]

    def step(k):
        """samples [k*sps, (k+1)*sps) modulo the frame's spp, so any number of steps is valid; returns the
        kernel time (HIP events on the library's stream around the launches) and the time of the fan-in
        (quantise this rank's bands, all-gather, put the picture together), both in ms"""
        begin = (k % max(1, spp // sps)) * sps
        p = make_params(w, h, spp=spp, sample_begin=begin, sample_count=sps, band_begin=rank,
                        band_stride=world)
        r.timer_start()
        if cams:
            r.render_frames(p, cams)
        else:
            r.render(p)
        ms = r.timer_stop()
        t1 = time.perf_counter()
        r.export_packed_bands(block.data_ptr(), block.numel(), rank, world)
        r.sync()
        if world > 1 and rehearsal:
            parts = [torch.empty(block.shape, dtype=block.dtype) for _ in range(world)]
            dist.all_gather(parts, block.cpu())
            gathered.copy_(torch.cat(parts))
        elif world > 1:
            dist.all_gather_into_tensor(gathered, block)
        if world > 1:
            frame[0] = multi_gpu.compose(gathered.view(world, max_bands * 8, w), h)   # display order, every rank
            torch.cuda.current_stream().synchronize()    # the gather has read `block` before the next step writes it
        return ms, (time.perf_counter() - t1) * 1e3

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    r.begin_frames(w, h, n_frames)   # timed steps start from zeroed sums
    barrier()
    seg0 = r.ray_count()
    kernel_ms, fanin_ms = [], []
    t0 = time.perf_counter()
    for k in range(args.steps):
        a, b = step(k)
        kernel_ms.append(a); fanin_ms.append(b)
    t_own = time.perf_counter() - t0       # this rank's steps, before it waits for the others
    barrier()
    dt = time.perf_counter() - t0
    seg = r.ray_count() - seg0

    ranks = None
    if world > 1:
        # per rank: wall time of its steps, segments, kernel time and fan-in time per step - what tells an
        # uneven split (segments, kernel time) from launch tails (kernel time against its share of the
        # one-GPU time) and from the gather (fan-in time) when the scaling falls short
        mine = torch.tensor([dt, float(seg), float(np.mean(kernel_ms)), float(np.max(kernel_ms)),
                             float(np.mean(fanin_ms)), t_own], dtype=torch.float64,
                            device="cpu" if rehearsal else "cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        every = torch.stack(every).cpu().numpy()
        dt = float(every[:, 0].max()); seg_total = int(every[:, 1].sum())
        spread = lambda col: {"min": round(float(every[:, col].min()), 3), "mean": round(float(every[:, col].mean()), 3),
                              "max": round(float(every[:, col].max()), 3)}
        ranks = {"kernel_ms_per_step": spread(2), "kernel_ms_slowest_step": spread(3),
                 "fan_in_ms_per_step": spread(4), "own_steps_s": spread(5),
                 "segments_per_step": {"min": int(every[:, 1].min() / args.steps), "mean": int(every[:, 1].mean() / args.steps),
                                       "max": int(every[:, 1].max() / args.steps)},
                 "fan_in": "quantise own bands + all_gather of packed pixels + compose, wall clock per step"}
    else:
        seg_total = seg

    if rehearsal and world > 1 and rank == 0:
        # the same samples on one renderer, all bands: the gathered frame must be identical
        alone = Renderer(0, tables)
        alone.set_scene(wl["flat"]); alone.set_camera(wl["camera"]); alone.begin_frame(w, h)
        for k in range(args.steps):
            begin = (k % max(1, spp // sps)) * sps
            alone.render(make_params(w, h, spp=spp, sample_begin=begin, sample_count=sps))
        want = alone.read_packed()
        got = frame[0].cpu().numpy().view(np.uint32)
        same = bool(np.array_equal(got.reshape(want.shape), want))
        alone_segments = alone.ray_count()
        alone.close()
        print("rehearsal: gathered frame of %d ranks %s the single-renderer frame; segments %d vs %d"
              % (world, "equals" if same else "DIFFERS FROM", seg_total, alone_segments), file=sys.stderr, flush=True)
        if not same or seg_total != alone_segments:
            sys.exit(1)

    flat = None
    if rank == 0 and world == 1 and not args.no_flat_probe and (wl["flat"].counts["n_spheres"] >= 64):
        flat = flat_scan_probe(r, wl, flops_seg)

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        paths = w * h * sps * args.steps * n_frames
        # roofline of the dominant kernel (render_tiles) on THIS rank, from its mean launch
        # duration (HIP events) and its segments per launch
        n_launch = args.steps * launches_per_step
        roof = valu_roofline(args.workload, kernel, seg / n_launch, float(np.sum(kernel_ms)) / n_launch,
                             -(-sps // launches_per_step), flops_seg, wl["flat"])
        out = {
            "metric": METRIC,
            "value": round(seg_total / dt / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %s" % (args.workload, wl["description"]),
                "spp_per_step": sps, "steps_per_frame": round(spp / sps, 3), "launches_per_step": launches_per_step,
                "frames_per_step": n_frames,
                "ms_per_frame": round(ms_per_step * (spp / sps) / n_frames, 4 if ms_per_step < 10 else 1),
                "mpaths_per_s": round(paths / dt / 1e6, 3),
                "segments_per_path": round(seg_total / paths, 4),
                "segments_per_step": int(round(seg_total / args.steps)),
                "partition": "interleaved 8-row bands, rank r renders bands r, r+N, ...; "
                             "RCCL all_gather of packed pixels per step" if world > 1 else "single GPU",
                "sampling": "deterministic schedule (RT_SAMPLING_DET), seed 0",
                "scan": "implicit 8-ary box tree over each long sphere/triangle run (results identical to the "
                        "reference's list-order scan)",
            },
            "roofline": roof,
        }
        if ranks is not None:
            out["config"]["ranks"] = ranks
        if flat is not None:
            out["roofline"]["flat_scan"] = flat
        if world == 1 and args.others:
            r.close()
            others = []
            for name in [x.strip() for x in args.others.split(",") if x.strip() and x.strip() != args.workload]:
                others.append(other_workload(name, tables))
            out["config"]["others"] = others
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, tables)
        print(json.dumps(out), flush=True)

    r.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
