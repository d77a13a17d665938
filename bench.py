#!/usr/bin/env python3
"""Benchmark of the per-pixel trace loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one complete render of the workload: every pixel x every sample through the HIP kernels,
the per-GPU radiance sums added and divided by the sample count.  The scene and the output buffer
are resident in HBM before the timed region starts.

The headline workload is the same for every N: c4, the reference scene at 3840x2160, 1024 spp,
the frame dealt out to the N GPUs in small blocks of pixels, round-robin (fixed total work: "scaling":
"strong"; every pixel's samples are traced and added on one GPU, so the image has the bits of a one-GPU
render).  Two drivers of the split:

  * one process, several GPUs (the plain command above, no launcher): the library's own multi-GPU
    path -- rtx_opts.n_devices = N, one worker and stream per GPU inside librtx.so, the sums added by
    the peer-access slice reduce (--reduce p2p, default) or RCCL (--reduce rccl), the image left in a
    device buffer of the first GPU (rtx_render_device).  --devices 0,0,0 names the ordinals (an
    ordinal may repeat: a rehearsal of N workers on one GPU).
  * one process per GPU (under `python -m torch.distributed.run --nproc-per-node N`; WORLD_SIZE is
    set): rank g takes share g of N (rtx_set_share), rtx_render_samples, one all_reduce(SUM) over RCCL of
    buffers that are zero outside their share, rtx_divide.

With no --workload and N = 1 the line also carries, under "workloads", the other configurations
(timed the same way): c2 with its full roofline and counters, c3 and c5 with both kernels, c2 with
the wavefront kernel.  All GPU work runs in one block before the CPU baselines.

Workloads:
  c4  reference scene at 3840x2160, 1024 spp, depth 6                (headline, every N)
  c2  reference scene at 1920x1080, 256 spp, depth 6, seed 1
  c3  synthetic 10k-object scene, 1080p, 64 spp, BVH
  c5  dielectric / metallic scene, depth 16, 1080p, 256 spp
  c1  reference scene as shipped: 512x512, 10 spp
  c2lit  c2 with a coloured light: the scene class the two-pixel kernel is not chosen for
Rank 0 prints ONE JSON line.
"""
import argparse
import dataclasses
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def build_workload(name, spp_override=None, size_override=None):
    from raytracing_amd import reference_scene, dielectric_scene, synthetic_scene
    if name == "c2":
        sc, desc = reference_scene(1920, 1080, 256, 6), "reference scene (8 objects, 1 light), 1920x1080, 256 spp, depth 6"
    elif name == "c4":
        sc, desc = reference_scene(3840, 2160, 1024, 6), "reference scene (8 objects, 1 light), 3840x2160, 1024 spp, depth 6, the frame dealt out to the GPUs"
    elif name == "c3":
        sc, desc = synthetic_scene(10000, 12345, 1920, 1080, 64, 6), "synthetic 10k objects (8 lights), 1920x1080, 64 spp, depth 6, BVH"
    elif name == "c5":
        sc, desc = dielectric_scene(1920, 1080, 256, 16), "dielectric/metallic scene (33 objects), 1920x1080, 256 spp, depth 16"
    elif name == "c2lit":
        # not a configuration of the baseline: c2 with a COLOR line on its light (albedo 0.78).  Paths no
        # longer end at the light, so the library picks its general one-pixel kernel instead of the
        # two-pixel one that the reference's own scene (black light) gets
        sc = reference_scene(1920, 1080, 256, 6)
        prims = list(sc.prims)
        prims[5] = dataclasses.replace(prims[5], color=(0.78, 0.78, 0.78))
        sc, desc = sc.replace(prims=prims), "reference scene with a COLOR 0.78 line on its light, 1920x1080, 256 spp, depth 6"
    elif name == "c1":
        sc, desc = reference_scene(512, 512, 10, 6), "reference scene at its native 512x512, 10 spp, depth 6"
    elif name[0] == "s" and name[1:].isdigit():
        # tuning aid, not a configuration of the baseline: C3's generator at another object count
        sc, desc = synthetic_scene(int(name[1:]), 12345, 1920, 1080, 64, 6), f"synthetic {name[1:]} objects, 1920x1080, 64 spp, depth 6"
    else:
        raise SystemExit(f"unknown workload {name}")
    if size_override:
        sc = sc.replace(width=size_override[0], height=size_override[1])
        desc += f" [size overridden to {size_override[0]}x{size_override[1]}]"
    if spp_override:
        sc = sc.replace(n_samples=spp_override)
        desc += f" [spp overridden to {spp_override}]"
    return sc, desc


# ---------------------------------------------------------------------------------- CPU baseline
CPU_NOTE = ("C restatement of the reference (oracle/rt_oracle.c, not the Rust binary), gcc -O2 -ffp-contract=off, "
            "1 thread, reference evaluation order, full paths (every path to the depth limit as src/trace.rs does: "
            "4.24 rays per sample on the reference scene, where the GPU ends zero-throughput paths at 2.83)")


def _time_oracle(scene, threads=1):
    from oracle import pyoracle
    t0 = time.perf_counter()
    pyoracle.render(scene, seed=1, order=pyoracle.ORDER_REFERENCE, threads=threads)
    dt = time.perf_counter() - t0
    return scene.width * scene.height * scene.n_samples, dt


def cpu_baseline(scene, seconds_target=12.0):
    """The oracle (1 thread, the reference's evaluation order and full paths) timed on this host on
    bounded samples: the headline's scene at 480x270, C1 exactly as the reference ships it, and
    reduced C3 and C5."""
    small = scene.replace(width=240, height=135, n_samples=4)
    n, dt = _time_oracle(small)
    spp = max(4, min(256, int(seconds_target * (n / dt) / (480 * 270))))
    sample = scene.replace(width=480, height=270, n_samples=spp)
    n, dt = _time_oracle(sample)
    # for orientation only: the same sample with the image's columns spread over this job's
    # host threads (the reference itself is single-threaded, src/main.rs:13-31)
    threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), len(os.sched_getaffinity(0))))
    n_all, dt_all = _time_oracle(sample, threads)
    out = {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": f"the reference scene (c2 / c4) at 480x270, {spp} spp ({n} samples, {dt:.1f} s); " + CPU_NOTE,
           "host_cpus": os.cpu_count(),
           "all_threads": {"value": n_all / dt_all / 1e6, "unit": "Msamples/s", "cores": threads}}
    for key, (wl, size, spp_small, what) in {
            "c1": ("c1", None, None, "assets/scene.txt as shipped: 512x512, 10 spp, depth 6"),
            "c3": ("c3", (128, 72), 1, "the 10 000-object scene at 128x72, 1 spp (a linear scan over all objects per ray, src/trace.rs:109-120)"),
            "c5": ("c5", (480, 270), 8, "the glass / metal scene at 480x270, 8 spp, depth 16")}.items():
        sc, _ = build_workload(wl, spp_small, size)
        n, dt = _time_oracle(sc)
        out[key] = {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
                    "sample": f"{what} ({n} samples, {dt:.1f} s), 1 thread, reference order, full paths"}
    return out


# ------------------------------------------------------------------------- committed rocprofv3 runs
def profile_files(workload, mode):
    """The committed rocprofv3 summaries of this workload (thread-active fraction, HBM bytes, instruction
    classes), newest round last."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", f"{workload}_{mode}_pmc.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    out = {"source": os.path.relpath(files[-1], ROOT), "thread_active_fraction": d.get("thread_active_fraction"),
           "hbm_bytes_per_step": d.get("hbm_bytes_per_launch"), "valu_instructions_per_step": d.get("SQ_INSTS_VALU")}
    # registers and occupancy of the step's dominant kernel from the compiler's resource report
    res = os.path.join(os.path.dirname(files[-1]), "resource_usage.txt")
    names = [re.sub(r"\([^()]*\)$", "", k.replace("(anonymous namespace)::", "")).replace("void ", "").replace("rtx::", "")
             for k in d.get("kernels", [])]
    # the megakernel's probe launch (last template argument true) is not the dominant kernel
    kernels = [k for k in names if not (k.startswith("k_megakernel") and k.endswith("true>"))] or names
    if kernels:
        out["kernel"] = kernels[-1]
    if kernels and kernels[-1] and os.path.exists(res):
        want = out["kernel"].replace(", ", ",")
        for line in open(res):
            if line.replace(", ", ",").startswith(want):
                f = line[72:].split()
                try:
                    out["vgprs"], out["spilled_vgprs"], out["waves_per_simd"] = int(f[1]), int(f[4]), int(f[6])
                except (IndexError, ValueError):
                    pass
    return out, d


def measured_traffic(pmc):
    """HBM bytes per step from the committed rocprofv3 --pmc summary (FETCH_SIZE and WRITE_SIZE in KiB,
    FETCH_SIZE doubled as the gfx950 counter reports half of a streamed read).  PMC passes cannot run
    inside the timed bench, so this is read back, not measured live."""
    if not pmc or "FETCH_SIZE" not in pmc or "WRITE_SIZE" not in pmc:
        return None
    return (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0


# ------------------------------------------------------------------------------------ one workload
def measure(name, mode_name, accel_name, steps, warmup, seed, run, want_stats=True, spp=None, size=None, depth=None,
            stats_spp=None, host_steps=0):
    """Times `steps` steps of one workload (scene and buffers resident in HBM before the timed region,
    barrier + synchronize on both sides, max over ranks) and counts its path events in an untimed launch.
    run: how the GPUs are driven (see main).  stats_spp: count the events on that many samples per pixel
    only and scale (for workloads whose counting launch would take seconds).  host_steps: also time that
    many rtx_render calls that deliver the image to host memory (in-library driver only)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from raytracing_amd import _abi
    from raytracing_amd.renderer import Renderer
    from raytracing_amd.flops import (algorithmic_flops, elided_flops, executed_flops, PEAK_FP32_VECTOR_TFLOPS,
                                      PEAK_FP32_NONPACKED_TFLOPS)

    scene, desc = build_workload(name, spp, size)
    if depth is not None:
        scene = scene.replace(ray_depth=depth)
        desc += f" [depth overridden to {depth}]"
    mode = {"megakernel": _abi.MODE_MEGAKERNEL, "wavefront": _abi.MODE_WAVEFRONT}[mode_name]
    accel = {"auto": _abi.ACCEL_AUTO, "lds": _abi.ACCEL_BRUTE_LDS, "scalar": _abi.ACCEL_BRUTE_SCALAR,
             "bvh": _abi.ACCEL_BVH}[accel_name]
    ranks = run["driver"] == "ranks"
    world, rank = run["world"], run["rank"]
    n_floats = scene.width * scene.height * 3
    buf = torch.zeros(n_floats, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()

    def context(collect_stats=False):
        if ranks:
            return Renderer(seed=seed, mode=mode, accel=accel, device=run["device"], collect_stats=collect_stats)
        if len(run["devices"]) == 1 and run["reduce"] == _abi.REDUCE_P2P:
            return Renderer(seed=seed, mode=mode, accel=accel, device=run["devices"][0], collect_stats=collect_stats)
        return Renderer(seed=seed, mode=mode, accel=accel, devices=run["devices"], reduce=run["reduce"],
                        collect_stats=collect_stats)

    r = context()
    if ranks:
        r.set_share(rank, world)   # this rank's share of the frame's work units, every sample of them
    r.set_scene(scene)
    first, count = 0, scene.n_samples
    kernel_times = []

    def step(timed=False):
        if ranks:
            r.render_samples(first, count, buf.data_ptr(), stream)
            if run["use_dist"]:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            r.divide(buf.data_ptr(), n_floats, float(scene.n_samples), stream)
        else:
            # split, render, add, divide: one call, all of the context's GPUs (returns when the image is complete)
            r.render_device(scene.n_samples, buf.data_ptr())
            if timed:
                kernel_times.append(r.last_kernel_ms())   # HIP events of the library on each worker's stream: the slowest worker

    for _ in range(warmup):
        step()
    if run["use_dist"]:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(True)
    torch.cuda.synchronize()
    if run["use_dist"]:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if ranks:
        # the render kernels of this rank's last step (events on the launch stream, recorded by the library)
        kernel_times.append(r.last_kernel_ms())
    if run["use_dist"]:
        tt = torch.tensor([elapsed, kernel_times[-1]], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, kernel_times = float(tt[0].item()), [float(tt[1].item())]
    kernel_ms = sum(kernel_times) / max(1, len(kernel_times))
    checksum = float(buf.double().sum().item())

    total_samples = scene.width * scene.height * scene.n_samples
    res = {"workload": f"{name}: {desc}", "mode": mode_name, "accel": accel_name, "ms_per_step": elapsed / steps * 1e3,
           "msamples_per_s": total_samples * steps / elapsed / 1e6, "kernel_ms": kernel_ms, "steps": steps,
           "samples_per_step": total_samples, "checksum": checksum, "elapsed": elapsed}

    if host_steps and not ranks:
        # the reference's host calls render(&mut scene) and gets the image in host memory (src/main.rs:38):
        # rtx_render end to end, device -> host delivery included (never the headline value)
        img = r.render(scene)
        t1 = time.perf_counter()
        for _ in range(host_steps):
            img = r.render(scene)
        dt = (time.perf_counter() - t1) / host_steps
        res["rtx_render_to_host"] = {"ms_per_call": dt * 1e3, "msamples_per_s": total_samples / dt / 1e6,
                                     "kernel_ms": r.last_kernel_ms(), "calls": host_steps,
                                     "checksum": float(np.asarray(img, dtype=np.float64).sum())}
    r.close()

    if want_stats:
        # event counters: an untimed launch of the counting kernels on the same inputs (paths are
        # deterministic, so the events are those of a timed step)
        rs = context(collect_stats=True)
        rs.set_scene(scene)
        n_counted = count if not stats_spp else min(count, stats_spp)
        if ranks:
            rs.render_samples(first, n_counted, buf.data_ptr(), stream)
            torch.cuda.synchronize()
        else:
            rs.render_device(n_counted, buf.data_ptr())
        stats = rs.stats()
        rs.close()
        scale = count / max(1, n_counted)
        bvh = stats.get("bvh_node_tests", 0) > 0
        # this process's share; the in-library driver counts all of its GPUs' events together
        flops = algorithmic_flops(stats, stats.get("bvh_node_tests", 0)) * scale
        elided = elided_flops(stats, scene, bvh) * scale
        rays = torch.tensor([stats["rays"] * scale, flops, elided], dtype=torch.float64, device="cuda")
        if run["use_dist"]:
            dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        job_rays, job_flops, job_elided = (float(x) for x in rays.tolist())
        n_gpus = world if ranks else len(run["devices"])
        # the dominant kernel runs on every GPU side by side: per-GPU rate = the GPU's share / its kernel time
        achieved = job_flops / n_gpus / (kernel_ms * 1e-3) / 1e12
        res.update({"algorithmic_tflops": achieved, "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
                    "frac_of_nonpacked_fma_peak": achieved / PEAK_FP32_NONPACKED_TFLOPS,
                    "algorithmic_flops_per_launch": job_flops / n_gpus,
                    "algorithmic_flops_elided": job_elided / n_gpus,
                    "frac_without_elided": (job_flops - job_elided) / n_gpus / (kernel_ms * 1e-3) / 1e12 / PEAK_FP32_VECTOR_TFLOPS,
                    "flops_per_sample": job_flops / max(1, total_samples),
                    "rays_per_sample": stats["rays"] / max(1, stats["samples"]),
                    "mrays_per_s": job_rays * steps / elapsed / 1e6,
                    "events_counted_on_spp": n_counted})
    prof, pmc = profile_files(name, mode_name)
    if prof and (ranks is False and len(run["devices"]) == 1 or ranks and world == 1) and not spp and not size and depth is None:
        res["profile"] = prof
        ex = executed_flops(pmc)
        if ex is not None:
            # counters of one profiled step of this workload (committed) over this run's kernel time
            res["executed_tflops"] = ex / (kernel_ms * 1e-3) / 1e12
            res["frac_executed"] = res["executed_tflops"] / PEAK_FP32_VECTOR_TFLOPS
        res["traffic"] = measured_traffic(pmc)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--mode", default="megakernel", choices=["megakernel", "wavefront"])
    ap.add_argument("--accel", default="auto", choices=["auto", "lds", "scalar", "bvh"])
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--size", type=int, nargs=2, default=None)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--depth", type=int, default=None, help="override RAY_DEPTH (experiments only)")
    ap.add_argument("--devices", default=None, help="in-library driver: comma-separated HIP ordinals, one per worker (default 0..N-1; an ordinal may repeat)")
    ap.add_argument("--reduce", default="p2p", choices=["p2p", "rccl"], help="in-library driver: how the GPUs' sums are added")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stats", action="store_true")
    ap.add_argument("--no-host", action="store_true", help="skip the timing of rtx_render into host memory (profiling runs: only the timed steps launch kernels)")
    ap.add_argument("--no-extra", action="store_true", help="N = 1: only the headline workload, not the other configurations")
    args = ap.parse_args()

    # Rank 0 prints ONE line on stdout.  Libraries write there too (RCCL prints a version banner
    # when the process group comes up), so everything but the result line goes to stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from raytracing_amd import _abi
    from raytracing_amd.flops import PEAK_FP32_VECTOR_TFLOPS

    if "WORLD_SIZE" in os.environ:
        # one process per GPU, started by torch.distributed.run
        world, rank, local_rank = int(os.environ["WORLD_SIZE"]), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
        torch.cuda.set_device(local_rank)
        # the process group is always created, so the RCCL path is the one exercised even with a single rank
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        run = {"driver": "ranks", "world": world, "rank": rank, "device": local_rank, "use_dist": True}
        parallelism = f"one process per GPU x{world}: the frame's work units dealt out round-robin (rtx_set_share) + RCCL all-reduce (torch.distributed)" if world > 1 else "one process, one GPU (RCCL all-reduce of one rank)"
        visible = torch.cuda.device_count()
    else:
        # one process: the library's own multi-GPU path
        visible = torch.cuda.device_count()
        if args.devices is not None:
            devices = [int(x) for x in args.devices.split(",") if x != ""]
            if len(devices) != args.gpus:
                raise SystemExit(f"--devices names {len(devices)} workers, --gpus says {args.gpus}")
        else:
            if args.gpus > visible:
                raise SystemExit(f"--gpus {args.gpus}: only {visible} GPU(s) visible (use --devices to place several workers on one GPU)")
            devices = list(range(args.gpus))
        torch.cuda.set_device(devices[0])
        reduce = {"p2p": _abi.REDUCE_P2P, "rccl": _abi.REDUCE_RCCL}[args.reduce]
        run = {"driver": "library", "world": 1, "rank": 0, "devices": devices, "reduce": reduce, "use_dist": False}
        world, rank = args.gpus, 0
        parallelism = (f"one process, rtx_opts.n_devices = {world}: the frame's work units dealt out round-robin over {len(set(devices))} GPU(s), {args.reduce} reduce in librtx"
                       if world > 1 or args.reduce == "rccl" else "single GPU")

    name = args.workload or "c4"
    common = dict(seed=args.seed, run=run)
    single = world == 1
    head = measure(name, args.mode, args.accel, args.steps, args.warmup, want_stats=not args.no_stats, spp=args.spp,
                   size=args.size, depth=args.depth, stats_spp=64 if name == "c4" and not args.spp else None,
                   host_steps=2 if run["driver"] == "library" and not args.no_host else 0, **common)

    # N = 1, default workload: the other configurations in the same run, all GPU work in one block
    # before the CPU baselines
    extra = {}
    if single and args.workload is None and not args.no_extra and run["driver"] == "library" and args.reduce == "p2p":
        for key, wl, mode, steps in (("c2", "c2", "megakernel", 5), ("c3", "c3", "megakernel", 3), ("c3_wavefront", "c3", "wavefront", 3),
                                     ("c5", "c5", "megakernel", 3), ("c5_wavefront", "c5", "wavefront", 3),
                                     ("c2_wavefront", "c2", "wavefront", 3), ("c2_coloured_light", "c2lit", "megakernel", 3)):
            m = measure(wl, mode, "auto", steps, 1, want_stats=not args.no_stats, host_steps=2 if key == "c2" else 0, **common)
            m.pop("elapsed")
            extra[key] = m

    if rank == 0:
        total_samples = head["samples_per_step"]
        value = total_samples * args.steps / head["elapsed"] / 1e6
        out = {
            "metric": "Msamples/sec (whole node)", "value": value, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if name in ("c3", "c5") else "the reference's assets/scene.txt, restated",
            "config": {"workload": head["workload"], "mode": args.mode, "accel": args.accel, "seed": args.seed,
                       "samples_per_step": total_samples, "parallelism": parallelism, "driver": run["driver"],
                       "gpus_visible": visible},
            "mrays_per_s": head.get("mrays_per_s"),
            "checksum": head["checksum"],
        }
        if run["driver"] == "library":
            out["config"]["devices"] = run["devices"]
            out["config"]["distinct_gpus"] = len(set(run["devices"]))
            out["config"]["reduce"] = args.reduce
            # what the reduce really ran over: peer-enabled GPUs (p2p) or RCCL communicator ranks
            out["config"]["peer_enabled_gpus" if args.reduce == "p2p" else "rccl_ranks"] = len(set(run["devices"]))
        else:
            out["config"]["rccl_ranks"] = world
        if "rtx_render_to_host" in head:
            out["rtx_render_to_host"] = head["rtx_render_to_host"]
        if "frac" in head:
            prof = head.get("profile")
            out["roofline"] = {"bound": "valu", "achieved": head["algorithmic_tflops"], "peak": PEAK_FP32_VECTOR_TFLOPS,
                               "unit": "TFLOP/s", "frac": head["frac"], "traffic": head.get("traffic"),
                               "traffic_source": prof["source"] if prof else None,
                               "per": "GPU (the kernel runs on every GPU side by side: a GPU's share of the FLOPs / the slowest GPU's kernel time)",
                               "frac_of_nonpacked_fma_peak": head["frac_of_nonpacked_fma_peak"],
                               "kernel": (("k_megakernel_scan2 (two pixels per lane; units drawn from a work queue, the last 8 192 in chunks of the sample range)"
                                           if name in ("c2", "c4", "c1") else
                                           "k_megakernel (units longest first: the scene's tile costs are measured once, by an 8-sample probe launch "
                                           "of the first render after rtx_set_scene -- here in the warm-up -- and reused by the timed renders)")
                                          if args.mode == "megakernel" else "k_wavefront"),
                               "kernel_ms": head["kernel_ms"], "algorithmic_flops_per_launch": head["algorithmic_flops_per_launch"],
                               "algorithmic_flops_elided": head["algorithmic_flops_elided"],
                               "frac_without_elided": head["frac_without_elided"],
                               "executed_tflops": head.get("executed_tflops"), "frac_executed": head.get("frac_executed"),
                               "executed_note": "executed = (ADD + MUL + TRANS + 2 FMA wave-instructions of one profiled step) x 64 x thread-active fraction "
                                                "/ this run's kernel time; a packed instruction counts once, so a lower bound",
                               "flops_per_sample": head["flops_per_sample"], "rays_per_sample": head["rays_per_sample"],
                               "events_counted_on_spp": head["events_counted_on_spp"],
                               "counters": prof}
        if extra:
            out["workloads"] = extra
        if single and not args.no_cpu_baseline:
            scene, _ = build_workload(name, args.spp, args.size)
            cb = cpu_baseline(scene)
            out["cpu_baseline"] = cb
            speed = {name: value / cb["value"]}
            for key, m in extra.items():
                base = cb["value"] if key.startswith("c2") else cb.get(key[:2], {}).get("value")
                if base:
                    speed[key] = m["msamples_per_s"] / base
            out["speedup_vs_cpu_1thread"] = speed[name]
            out["speedup_vs_cpu_1thread_by_workload"] = speed
            # C1's GPU rate is not timed here (2.6 M samples are 0.4 ms of kernel time); its CPU time is the
            # reference's own native job
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if run["use_dist"]:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
