#!/usr/bin/env python3
"""Benchmark of the per-pixel trace loop on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one complete render of the workload: every pixel x every sample through the
HIP kernels, the per-GPU radiance sums combined (RCCL all-reduce when N > 1) and
divided by the sample count.  The scene and the output buffer are resident in HBM
before the timed region starts.

With no --workload and N = 1 the line also carries, under "workloads", the other configurations
(3 steps each, timed the same way): c4 on one GPU -- the base the multi-GPU runs of c4 compare
with --, c3 and c5 with both kernels, c2 with the wavefront kernel.  All GPU work runs in one
block before the CPU baseline.

Workloads:
  c2  reference scene at 1920x1080, 256 spp, depth 6, seed 1      (default for N = 1)
  c4  reference scene at 3840x2160, 1024 spp, samples split over the N GPUs
      (default for N > 1; fixed total work, so scaling = "strong")
  c3  synthetic 10k-object scene, 1080p, 64 spp, BVH
  c5  dielectric / metallic scene, depth 16, 1080p, 256 spp
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def build_workload(name, spp_override=None, size_override=None):
    from raytracing_amd import reference_scene, dielectric_scene, synthetic_scene
    if name == "c2":
        sc, desc = reference_scene(1920, 1080, 256, 6), "reference scene (8 objects, 1 light), 1920x1080, 256 spp, depth 6"
    elif name == "c4":
        sc, desc = reference_scene(3840, 2160, 1024, 6), "reference scene, 3840x2160, 1024 spp, depth 6, spp split over GPUs"
    elif name == "c3":
        sc, desc = synthetic_scene(10000, 12345, 1920, 1080, 64, 6), "synthetic 10k objects (8 lights), 1920x1080, 64 spp, depth 6, BVH"
    elif name == "c5":
        sc, desc = dielectric_scene(1920, 1080, 256, 16), "dielectric/metallic scene (33 objects), 1920x1080, 256 spp, depth 16"
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


def cpu_baseline(scene, seconds_target=15.0):
    """The oracle (C restatement of the reference, 1 thread, reference evaluation order)
    timed on this host on a bounded sample of the same workload."""
    from oracle import pyoracle
    small = scene.replace(width=240, height=135, n_samples=4)
    t0 = time.perf_counter()
    pyoracle.render(small, seed=1, order=pyoracle.ORDER_REFERENCE)
    dt = time.perf_counter() - t0
    rate = small.width * small.height * small.n_samples / dt
    spp = max(4, min(256, int(seconds_target * rate / (480 * 270))))
    sample = scene.replace(width=480, height=270, n_samples=spp)
    t0 = time.perf_counter()
    pyoracle.render(sample, seed=1, order=pyoracle.ORDER_REFERENCE)
    dt = time.perf_counter() - t0
    n = sample.width * sample.height * sample.n_samples
    # for orientation only: the same sample with the image's columns spread over this job's
    # host threads (the reference itself is single-threaded, src/main.rs:13-31)
    threads = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), len(os.sched_getaffinity(0))))
    t0 = time.perf_counter()
    pyoracle.render(sample, seed=1, order=pyoracle.ORDER_REFERENCE, threads=threads)
    dt_all = time.perf_counter() - t0
    return {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"same scene at 480x270, {spp} spp ({n} samples, {dt:.1f} s), "
                      "C restatement of the reference (not the Rust binary), gcc -O2 -ffp-contract=off, 1 thread",
            "host_cpus": os.cpu_count(),
            "all_threads": {"value": n / dt_all / 1e6, "unit": "Msamples/s", "cores": threads}}


def measured_traffic(workload, mode, n_gpus):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc summary
    of this workload (profiles/<round>/<workload>_<mode>_pmc.json: FETCH_SIZE and WRITE_SIZE in
    KiB, FETCH_SIZE doubled as the gfx950 counter reports half of a streamed read).  PMC
    passes cannot run inside the timed bench, so this is read back, not measured live."""
    import glob
    if n_gpus != 1:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", f"{workload}_{mode}_pmc.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
        return None, None
    return (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0, os.path.relpath(files[-1], ROOT)


def measure(name, mode_name, accel_name, steps, warmup, seed, rank, world, local_rank, use_dist, want_stats=True,
            spp=None, size=None, depth=None, stats_spp=None):
    """Times `steps` steps of one workload on this rank's GPU (scene and buffers resident in HBM before
    the timed region, barrier + synchronize on both sides, max over ranks) and counts its path events
    in an untimed launch.  stats_spp: count the events on that many samples per pixel only and scale
    (for workloads whose counting launch would take seconds)."""
    import torch
    import torch.distributed as dist
    from raytracing_amd import _abi
    from raytracing_amd.renderer import Renderer
    from raytracing_amd.flops import (algorithmic_flops, PEAK_FP32_VECTOR_TFLOPS, PEAK_FP32_NONPACKED_TFLOPS)
    from raytracing_amd.parallel import sample_range

    scene, desc = build_workload(name, spp, size)
    if depth is not None:
        scene = scene.replace(ray_depth=depth)
        desc += f" [depth overridden to {depth}]"
    mode = {"megakernel": _abi.MODE_MEGAKERNEL, "wavefront": _abi.MODE_WAVEFRONT}[mode_name]
    accel = {"auto": _abi.ACCEL_AUTO, "lds": _abi.ACCEL_BRUTE_LDS, "scalar": _abi.ACCEL_BRUTE_SCALAR,
             "bvh": _abi.ACCEL_BVH}[accel_name]
    first, count = sample_range(scene.n_samples, rank, world)
    n_floats = scene.width * scene.height * 3
    buf = torch.zeros(n_floats, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    r = Renderer(seed=seed, mode=mode, accel=accel, device=local_rank)
    r.set_scene(scene)

    def step(ev=None):
        if ev:
            ev[0].record()
        r.render_samples(first, count, buf.data_ptr(), stream)
        if ev:
            ev[1].record()
        if use_dist:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        r.divide(buf.data_ptr(), n_floats, float(scene.n_samples), stream)

    for _ in range(warmup):
        step()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(events[k])
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernel_ms = sum(a.elapsed_time(b) for a, b in events) / max(1, steps)
    checksum = float(buf.double().sum().item())
    r.close()

    # event counters of this rank's share: an untimed launch of the counting kernels on
    # the same inputs (paths are deterministic, so the events are those of a timed step)
    total_samples = scene.width * scene.height * scene.n_samples
    res = {"workload": f"{name}: {desc}", "mode": mode_name, "accel": accel_name, "ms_per_step": elapsed / steps * 1e3,
           "msamples_per_s": total_samples * steps / elapsed / 1e6, "kernel_ms": kernel_ms, "steps": steps,
           "samples_per_step": total_samples, "checksum": checksum, "elapsed": elapsed}
    if want_stats:
        rs = Renderer(seed=seed, mode=mode, accel=accel, device=local_rank, collect_stats=True)
        rs.set_scene(scene)
        n_counted = count if not stats_spp else min(count, stats_spp)
        rs.render_samples(first, n_counted, buf.data_ptr(), stream)
        torch.cuda.synchronize()
        stats = rs.stats()
        rs.close()
        scale = count / max(1, n_counted)
        flops_rank = algorithmic_flops(stats, stats.get("bvh_node_tests", 0)) * scale
        achieved = flops_rank / (kernel_ms * 1e-3) / 1e12
        rays = torch.tensor([stats["rays"] * scale], dtype=torch.float64, device="cuda")
        if use_dist:
            dist.all_reduce(rays, op=dist.ReduceOp.SUM)
        res.update({"algorithmic_tflops": achieved, "frac": achieved / PEAK_FP32_VECTOR_TFLOPS,
                    "frac_of_nonpacked_fma_peak": achieved / PEAK_FP32_NONPACKED_TFLOPS,
                    "algorithmic_flops_per_launch": flops_rank,
                    "flops_per_sample": flops_rank / max(1, stats["samples"] * scale),
                    "rays_per_sample": stats["rays"] / max(1, stats["samples"]),
                    "mrays_per_s": float(rays.item()) * steps / elapsed / 1e6,
                    "events_counted_on_spp": n_counted})
    return res


def profile_files(workload, mode):
    """The committed rocprofv3 summaries of this workload (thread-active fraction, HBM bytes)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", f"{workload}_{mode}_pmc.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        d = json.load(f)
    out = {"source": os.path.relpath(files[-1], ROOT), "thread_active_fraction": d.get("thread_active_fraction"),
           "hbm_bytes_per_step": d.get("hbm_bytes_per_launch"), "valu_instructions_per_step": d.get("SQ_INSTS_VALU")}
    # registers and occupancy of the step's dominant kernel from the compiler's resource report
    res = os.path.join(os.path.dirname(files[-1]), "resource_usage.txt")
    want = {("c2", "megakernel"): "k_megakernel_scan2<0, false, false, false>", ("c4", "megakernel"): "k_megakernel_scan2<0, false, false, false>",
            ("c3", "megakernel"): "k_megakernel<2, false, true, false>", ("c5", "megakernel"): "k_megakernel<2, false, false, false>",
            ("c2", "wavefront"): "k_wavefront<0, false, false, WfShapeT<512u, 16u, true>", ("c3", "wavefront"): "k_wavefront<2, false, true, WfShapeT<64u, 4u, false>",
            ("c5", "wavefront"): "k_wavefront<2, false, false, WfShapeT<64u, 2u, true>"}.get((workload, mode))
    if want and os.path.exists(res):
        for line in open(res):
            if line.startswith(want):
                f = line[72:].split()
                out["kernel"] = want + ">" if want.count("<") > want.count(">") else want
                out["vgprs"], out["spilled_vgprs"], out["waves_per_simd"] = int(f[1]), int(f[4]), int(f[6])
    return out


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stats", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="N = 1: only the headline workload, not the other configurations")
    args = ap.parse_args()

    # Rank 0 prints ONE line on stdout.  Libraries write there too (RCCL prints a version banner
    # when the process group comes up), so everything but the result line goes to stderr.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from raytracing_amd.flops import PEAK_FP32_VECTOR_TFLOPS

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    torch.cuda.set_device(local_rank)
    # under torch.distributed.run (RANK set) the process group is always created, so the
    # RCCL path is the one exercised even with a single rank
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    name = args.workload or ("c2" if world == 1 else "c4")
    common = dict(seed=args.seed, rank=rank, world=world, local_rank=local_rank, use_dist=use_dist)
    head = measure(name, args.mode, args.accel, args.steps, args.warmup, want_stats=not args.no_stats, spp=args.spp,
                   size=args.size, depth=args.depth, stats_spp=64 if name == "c4" else None, **common)

    # N = 1, default workload: the other configurations in the same run, all GPU work in one block
    # before the CPU baseline.  c4 on one GPU is the base the multi-GPU runs of c4 compare with.
    extra = {}
    if world == 1 and args.workload is None and not args.no_extra:
        for key, wl, mode, steps, sspp in (("c4_single_gpu", "c4", "megakernel", 3, 64), ("c3", "c3", "megakernel", 3, None),
                                           ("c3_wavefront", "c3", "wavefront", 3, None), ("c5", "c5", "megakernel", 3, None),
                                           ("c5_wavefront", "c5", "wavefront", 3, None), ("c2_wavefront", "c2", "wavefront", 3, None)):
            m = measure(wl, mode, "auto", steps, 1, want_stats=not args.no_stats, stats_spp=sspp, **common)
            m.pop("elapsed")
            m["profile"] = profile_files(wl, mode)
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
                       "samples_per_step": total_samples,
                       "parallelism": f"spp-split x{world} + all-reduce" if world > 1 else "single GPU"},
            "mrays_per_s": head.get("mrays_per_s"),
            "checksum": head["checksum"],
        }
        if "frac" in head:
            traffic, traffic_src = measured_traffic(name, args.mode, world)
            out["roofline"] = {"bound": "valu", "achieved": head["algorithmic_tflops"], "peak": PEAK_FP32_VECTOR_TFLOPS,
                               "unit": "TFLOP/s", "frac": head["frac"], "traffic": traffic, "traffic_source": traffic_src,
                               "frac_of_nonpacked_fma_peak": head["frac_of_nonpacked_fma_peak"],
                               "kernel": ("k_megakernel (probe launch of 8 samples, k_tile_order, main launch in cost order)"
                                          if args.mode == "megakernel" else "k_wavefront"),
                               "kernel_ms": head["kernel_ms"], "algorithmic_flops_per_launch": head["algorithmic_flops_per_launch"],
                               "flops_per_sample": head["flops_per_sample"], "rays_per_sample": head["rays_per_sample"],
                               "events_counted_on_spp": head["events_counted_on_spp"],
                               "counters": profile_files(name, args.mode) if world == 1 else None}
        if extra:
            out["workloads"] = extra
            out["c4_single_gpu"] = {k: extra["c4_single_gpu"][k] for k in ("ms_per_step", "msamples_per_s", "frac", "mrays_per_s")
                                    if k in extra["c4_single_gpu"]}
        if world == 1 and not args.no_cpu_baseline:
            scene, _ = build_workload(name, args.spp, args.size)
            cb = cpu_baseline(scene)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_1thread"] = value / cb["value"]
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
