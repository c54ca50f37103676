#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

  python bench.py [--gpus N --steps K --warmup W] [--config c2|c1|c3|c4|c5] [--accel lbvh|linear]

A step is one pass of the hot path over one batch of synthetic input:
  c2 (default)  one 1920x1080 frame of the 487-sphere scene at 256 spp, depth 50  -> Mpaths/s
  c1            the reference's own default render (320x180, 8 spp, xorshift RNG)   -> Mpaths/s
  c3            the Perlin-textured scene, 1920x1080 at 1024 spp                   -> Mpaths/s
  c4            2^20 spheres, a 10^8-ray closest-hit batch (LBVH)                  -> Mrays/s
  c5            the 487-sphere scene at 3840x2160, 4096 spp in total                -> Mpaths/s
Scene and rays are resident in HBM before the timed region starts.

The default command (one GPU, c2) also times, after the headline's timed region and each between barriers of its
own, three more legs that land in the same JSON line: `linear_scan` (the reference's algorithm on the c2 frame),
`c3` and `c4` — each with its own value, ms_per_step and roofline (`--no-extras` skips them).

With N > 1 (one process per GPU, started by torch.distributed.run) the SAME job is divided among the ranks (strong
scaling: `value` at N ranks over `value` at one rank is the speed-up).  c2: every rank renders every N-th strip of
eight rows of the frame at the full sample count, and since strips are disjoint every rank sends rank 0 only the
rows it rendered (gather), queued behind the render on the device and inside the timed step.  c5: interleaved strip
sets x sample ranges (from four ranks on: N/2 strip sets x 2 sample ranges), the per-rank accumulators summed onto
rank 0 with an RCCL reduce inside the timed step.  c4 splits the ray batch by index range and needs no collective.
`--scaling weak` gives every rank its own 256-sample range of the whole frame instead (N x the work); with N > 1 the
line also carries that figure as `weak_scaling`, measured after the timed steps of the headline.
The N > 1 paths have run on hardware with one rank only (the build's GPU box has one GPU) and, with two and three
ranks, through gloo on the CPU and on ranks sharing one GPU (tests/test_distributed.py).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X: 256 CU x 2.4 GHz x 256 flop/clk/CU
HBM_PEAK_GBS = 8000.0

CONFIGS = {
    "c1": dict(scene="default", width=320, height=180, spp=8, rng="xorshift", workload="default scene 320x180x8spp depth50 (reference's built-in render)"),
    "c2": dict(scene="random_spheres", width=1920, height=1080, spp=256, rng="counter", workload="487-sphere scene 1920x1080x256spp depth50"),
    "c3": dict(scene="perlin", width=1920, height=1080, spp=1024, rng="counter", workload="Perlin-textured scene 1920x1080x1024spp depth50"),
    "c5": dict(scene="random_spheres", width=3840, height=2160, spp=4096, rng="counter", strong=True,
               workload="487-sphere scene 3840x2160x4096spp depth50"),
    "c4": dict(scene="cloud", n_spheres=1 << 20, n_rays=100_000_000, workload="2^20 spheres, 1e8-ray closest-hit batch"),
}


# Read requests per second that L2 gets answered by the fabric behind it, measured with kernels that do nothing else
# (tools/microbench/fetch_calib, profiles/r2_fetch_calibration.txt): 54.6 - 55.7 G/s for 64-byte gathers at random, one
# or four in flight per lane, over 1 GiB (HBM) as over 64 MiB (Infinity Cache); 48 G/s of 128-byte requests for
# streamed reads.  The ceiling of a batch of incoherent tree walks, whose every L2 miss is such a request.
FABRIC_REQUEST_CEILING_GPS = 55.7
# Vector instructions a SIMD issues per second at the render kernels' four waves per SIMD when multiplies and adds are
# separate instructions, as the numerical contract makes them (tools/microbench/fma_rate.hip,
# profiles/r1_microbench_fma_rate.txt: 47.1 T lane-instructions/s on the whole chip = one wave64 instruction per 3.34
# cycles per SIMD; v_fma_f32 reaches one per 3.0 there)
VALU_LANE_INSTS_PER_S_AT_4_WAVES = 47.1e12

TRAFFIC_JSON = os.path.join(ROOT, "profiles", "pmc_traffic.json")
ISSUE_JSON = os.path.join(ROOT, "profiles", "c2_issue.json")


def replayed(path, key):
    """A record of a committed counter profile (counters cannot be read from inside the timed run), or (None, why):
    the profile is replayed only while the kernel sources still hash to what it was taken at."""
    from raytrace_amd.build import kernel_sources_sha
    try:
        with open(path) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return None, f"{os.path.relpath(path, ROOT)} is missing"
    meta = doc.get("_meta", {})
    now = kernel_sources_sha()
    if meta.get("kernel_sources_sha") != now:
        return None, (f"{os.path.relpath(path, ROOT)} was taken at commit {meta.get('commit', '?')} (kernel sources "
                      f"{meta.get('kernel_sources_sha', 'unrecorded')}); the sources now hash to {now}: not replayed")
    rec = doc.get(key)
    if rec is None:
        return None, f"{os.path.relpath(path, ROOT)} holds no passes of this kernel on this workload"
    rec = dict(rec)
    rec["_meta"] = meta
    return rec, None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))  # c5 at N=1 takes about 3 s per step
    ap.add_argument("--accel", default="lbvh", choices=["lbvh", "linear"])
    ap.add_argument("--width", type=int)
    ap.add_argument("--height", type=int)
    ap.add_argument("--spp", type=int)
    ap.add_argument("--rays", type=int)
    ap.add_argument("--spheres", type=int)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=16)
    # "gloo" rehearses the N > 1 flow on a box with fewer GPUs than ranks (ranks then share devices); never a result
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--bands", type=int, default=1, help="row bands per step whose sums overlap the next band's render (N > 1)")
    ap.add_argument("--collective", choices=["gather", "reduce"], default="gather",
                    help="strip shares (N > 1): gather each rank's rows onto rank 0, or sum whole accumulators")
    ap.add_argument("--sample-groups", type=int, default=0,
                    help="c5 with N > 1: sample ranges per strip set (0: one — strips only, rows gathered)")
    ap.add_argument("--no-extras", action="store_true", help="skip the linear-scan, c3, c4 and weak-scaling legs")
    return ap.parse_args()


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_render(rt, scene, cam, cfg, threads):
    """The CPU restatement of the reference shader, timed on a bounded sample of the same workload."""
    import oracle
    # Mpaths/s does not depend on spp; pick the spp that gives roughly 10-30 s of CPU work.
    w, h = cfg["width"], cfg["height"]
    probe = rt.make_params(cam, w, h, 1, rng_mode=rt.RNG_COUNTER if cfg["rng"] == "counter" else rt.RNG_XORSHIFT_REF)
    probe.tile_y1 = max(1, h // 16)
    t0 = time.perf_counter()
    _, c = oracle.render(scene, probe, threads=threads)
    dt = time.perf_counter() - t0
    rate = c["paths"] / max(dt, 1e-6)  # paths/s on `threads` threads
    spp = int(max(1, min(cfg["spp"], round(rate * 8.0 / (w * h)))))
    p = rt.make_params(cam, w, h, spp, rng_mode=probe.rng_mode)
    t0 = time.perf_counter()
    _, c = oracle.render(scene, p, threads=threads)
    dt = time.perf_counter() - t0
    out = {"value": c["paths"] / dt / 1e6, "unit": "Mpaths/s", "cores": threads, "host_cores": os.cpu_count(), "cpu_model": cpu_model(),
           "kind": "port",
           "sample": f"same scene and camera, {w}x{h} at {spp} spp ({c['paths']} paths, {dt:.1f} s wall), "
                     f"CPU restatement of the reference shader (linear scan), {threads} threads"}
    # one thread, on a band of rows sized for about 5 s
    one = rt.make_params(cam, w, h, 1, rng_mode=probe.rng_mode)
    one.tile_y1 = int(max(1, min(h, round(h * (rate / max(threads, 1)) * 5.0 / (w * h)))))
    t0 = time.perf_counter()
    _, c1 = oracle.render(scene, one, threads=1)
    dt1 = time.perf_counter() - t0
    out["single_thread"] = {"value": c1["paths"] / dt1 / 1e6, "unit": "Mpaths/s",
                            "sample": f"rows 0..{one.tile_y1 - 1} at 1 spp ({c1['paths']} paths, {dt1:.1f} s wall)"}
    return out


def cpu_baseline_trace(rt, scene, threads, seed):
    import oracle
    n = 64
    o, d = rt.gen_rays(seed, 0, n)
    t0 = time.perf_counter()
    oracle.trace(scene, o, d, threads=threads)
    dt = time.perf_counter() - t0
    n2 = int(max(n, min(20000, n * 10.0 / max(dt, 1e-6))))
    o, d = rt.gen_rays(seed, 0, n2)
    t0 = time.perf_counter()
    oracle.trace(scene, o, d, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": n2 / dt / 1e6, "unit": "Mrays/s", "cores": threads, "host_cores": os.cpu_count(), "cpu_model": cpu_model(), "kind": "port",
            "sample": f"first {n2} rays of the batch against all {scene.n_spheres} spheres ({dt:.1f} s wall), "
                      f"CPU restatement of hit_any (linear scan), {threads} threads"}


class Workload:
    """One configuration set up on this rank: `step(counted=False)` queues one pass of the hot path."""
    step = None
    counters = None
    units_per_step = 0     # whole job, all ranks
    unit = metric = ""
    config = None
    scaling = "strong"
    algo_bytes = None      # this rank's compulsory bytes per step (closest-hit batches)
    n_local = 0            # this rank's rays per step
    build_ms = None
    node_visits_per_ray = None
    keep = None            # device tensors the step reads and writes


def trace_workload(rt, torch, r, dev, args, cfg, rank, world, accel):
    wl = Workload()
    n_sph = args.spheres or cfg["n_spheres"]
    n_rays_total = args.rays or cfg["n_rays"]
    scene = rt.build_scene(rt.SCENE_CLOUD, n=n_sph, seed=0xC4C4C4C4)
    r.upload(scene, accel)
    per = (n_rays_total + world - 1) // world
    first = rank * per
    n = max(0, min(per, n_rays_total - first))
    o_h, d_h = rt.gen_rays(0xC4C4C4C4, first, n)
    o = torch.from_numpy(o_h).to(dev)
    d = torch.from_numpy(d_h).to(dev)
    del o_h, d_h
    ids = torch.empty(n, dtype=torch.int32, device=dev)
    t = torch.empty(n, dtype=torch.float32, device=dev)
    nrm = torch.empty((n, 3), dtype=torch.float32, device=dev)
    wl.keep = (o, d, ids, t, nrm)
    wl.scene = scene

    def step(counted=False):
        r.trace_device(o.data_ptr(), d.data_ptr(), n, ids.data_ptr(), t.data_ptr(), nrm.data_ptr(), counted)

    wl.step = step
    step(counted=True)
    wl.counters = r.counters()
    if wl.counters["segments"]:
        wl.node_visits_per_ray = wl.counters["trav_lane_steps"] / wl.counters["segments"]
    wl.units_per_step = n_rays_total
    wl.n_local = n
    wl.unit, wl.metric = "Mrays/s", "closest-hit rays per second (Mrays/s)"
    wl.algo_bytes = 44.0 * n  # 24 B in + 20 B out per ray
    info = r.scene_info()
    wl.config = {"workload": cfg["workload"] if not (args.spheres or args.rays) else f"{n_sph} spheres, {n_rays_total}-ray closest-hit batch",
                 "accel": "lbvh" if accel == rt.ACCEL_LBVH else "linear", "builder": info["builder"], "tree_nodes": info["n_nodes"],
                 "node_bytes": info["node_bytes"], "spheres": n_sph, "rays": n_rays_total, "lbvh_depth": r.lbvh_depth(), "ranks": world,
                 "parallelism": f"ray index ranges x{world}"}
    wl.scaling = "strong"  # the batch is fixed; ranks take index ranges of it
    if accel == rt.ACCEL_LBVH and info["builder"] != "none":
        wl.build_ms = r.last_build_ms()
    return wl


def c5_sample_groups(args, world, pixels, lanes):
    """Sample ranges per strip set of a strong-scaling frame with very many samples: one, unless --sample-groups asks
    for more.  Strips alone need no sum (every rank's rows are gathered), and the library schedules a share of any
    size well: up to six pixels per lane in chunks of samples with every sample stored, six to twenty-four as several
    such parts, beyond that with whole pixels as jobs (tools/share_rate.py --config c5: 0.99 to 1.02 of the whole job / N for
    N = 2, 4, 8)."""
    return args.sample_groups if args.sample_groups > 0 else 1


def render_workload(rt, torch, dist, r, dev, args, cfg, rank, world, accel, sid, overrides=True):
    from raytrace_amd import distributed as rtd
    wl = Workload()
    scene = rt.build_scene(sid)
    cam = rt.scene_camera(sid)
    rng_mode = rt.RNG_COUNTER if cfg["rng"] == "counter" else rt.RNG_XORSHIFT_REF
    if world > 1 and rng_mode != rt.RNG_COUNTER:
        raise SystemExit("sample-range splitting needs the counter RNG")
    w, h, spp = cfg["width"], cfg["height"], cfg["spp"]
    r.upload(scene, accel)
    accum = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    r.set_accumulator(accum.data_ptr(), w, h)
    wl.keep = (accum,)
    wl.scene, wl.cam, wl.cfg = scene, cam, cfg
    strong = args.scaling == "strong" or bool(cfg.get("strong"))
    lanes = torch.cuda.get_device_properties(dev).multi_processor_count * 1024 if dev.type == "cuda" else 256 * 1024
    groups = c5_sample_groups(args, world, w * h, lanes) if cfg.get("strong") else 1

    def shares_of(strongly):
        total = spp if strongly else spp * world
        if not strongly:      # weak: rank r renders samples [r*spp, (r+1)*spp) of every pixel
            return rtd.split_frame(h, total, world), total
        # interleaved strips of eight rows (disjoint pixels, balanced cost: contiguous bands of this frame differ
        # three-fold); c5 adds sample ranges: world / groups strip sets x groups sample ranges
        return rtd.split_interleaved(h, total, world, sample_groups=groups), total

    def share_params(strongly):
        shares, total = shares_of(strongly)
        return rtd.apply_share(rt.make_params(cam, w, h, total, rng_mode=rng_mode), shares[rank]), total

    p, total_spp = share_params(strong)
    wl.params = p
    wl.share_params = share_params
    wl.strong = strong

    def make_step(params, gather=None):
        def step(counted=False):
            r.clear()  # memset + render are ordered on the library's own stream
            if world > 1 and not counted and gather:
                # disjoint strips: every rank sends rank 0 the rows it rendered, queued behind the render on the device
                r.render(params)
                rtd.gather_rows(accum, gather, rank, dst=0, after=r)
            elif world > 1 and not counted:
                # render and RCCL sum onto rank 0, the sum queued behind the render on the device
                rtd.render_and_combine(r, params, accum, dst=0, bands=args.bands)
            else:
                r.render(params, counted=counted)
        return step

    wl.make_step = make_step
    # strips over all ranks and nothing else (one sample range): disjoint rows, nothing to sum — rows are gathered
    strips_only = world > 1 and strong and groups == 1 and args.collective == "gather"
    wl.step = make_step(p, shares_of(True)[0] if strips_only else None)
    wl.step(counted=True)
    wl.counters = r.counters()
    wl.units_per_step = w * h * total_spp
    wl.unit, wl.metric = "Mpaths/s", "path-tracing throughput at %dx%d (Mpaths/s)" % (w, h)
    info = r.scene_info()
    custom = overrides and (args.width or args.height or args.spp)
    wl.config = {"workload": f"{cfg['workload'].split(' ')[0]} scene {w}x{h}x{spp}spp depth50" if custom else cfg["workload"],
                 # (the library scans scenes of up to 16 spheres even when a tree is asked for: the scan is faster there)
                 "accel": ("lbvh" if accel == rt.ACCEL_LBVH else "linear") if info["builder"] != "none" or accel != rt.ACCEL_LBVH
                 else "lbvh asked for, linear scan taken (few spheres)",
                 "builder": info["builder"], "tree_nodes": info["n_nodes"], "tree_in_lds": bool(info["tree_in_lds"]),
                 "spheres": scene.n_spheres, "rng": cfg["rng"],
                 "lbvh_depth": r.lbvh_depth(), "global_samples_per_pixel": total_spp,
                 "ranks": world, "collective": (None if world == 1 else f"{args.backend} gather of each rank's rows onto rank 0" if strips_only else
                                                f"{args.backend} reduce to rank 0, {args.bands} band(s) per step"),
                 "parallelism": ("single GPU" if world == 1 else
                                 f"sample ranges x{world}" if not strong else
                                 f"{world // groups} interleaved sets of 8-row strips x {groups} sample range(s)")}
    wl.scaling = "strong" if strong else "weak"
    if accel == rt.ACCEL_LBVH and info["builder"] != "none":
        wl.build_ms = r.last_build_ms()
    return wl


def roofline_of(rt, wl, avg_kernel_ms, kname, world):
    """The dominant kernel against its bound: algorithmic flops (render) or compulsory bytes (closest-hit batch) of one
    launch over the launch's device time, measured with HIP events on the library's stream."""
    flops = rt.algorithmic_flops(wl.counters)
    achieved_tflops = flops / (avg_kernel_ms * 1e-3) / 1e12
    traffic_rec, traffic_why = replayed(TRAFFIC_JSON, f"{kname}|{wl.config['workload']}")
    traffic = traffic_rec["hbm_bytes_per_step"] if traffic_rec else None
    if wl.algo_bytes is not None:
        gbs = wl.algo_bytes / (avg_kernel_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "traffic": traffic, "kernel": kname, "kernel_ms": avg_kernel_ms,
                "valu_tflops": achieved_tflops, "valu_frac": achieved_tflops / FP32_VECTOR_PEAK_TFLOPS}
        if traffic_rec and traffic_rec.get("fabric_read_requests_per_step"):
            # what bounds an incoherent batch: every L2 miss of a node or leaf read is one read request to the fabric
            n = wl.n_local
            req = traffic_rec["fabric_read_requests_per_step"] * (n / max(traffic_rec.get("rays_per_step", n), 1))
            rate = req / (avg_kernel_ms * 1e-3) / 1e9
            roof["fabric_requests"] = {
                "per_ray": req / max(n, 1), "achieved": rate, "ceiling": FABRIC_REQUEST_CEILING_GPS, "unit": "G requests/s",
                "frac": rate / FABRIC_REQUEST_CEILING_GPS,
                "source": f"requests: TCC_EA0_RDREQ pass in profiles/pmc_traffic.json, taken at commit {traffic_rec['_meta'].get('commit', '?')} "
                          "(not measured in this run); ceiling: random 64-byte gathers of tools/microbench/fetch_calib, "
                          "profiles/r2_fetch_calibration.txt"}
    else:
        roof = {"bound": "valu_fp32", "achieved": achieved_tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tflops / FP32_VECTOR_PEAK_TFLOPS, "traffic": traffic, "kernel": kname, "kernel_ms": avg_kernel_ms,
                "algorithmic_gflop_per_launch": flops / 1e9}
        if world == 1:  # what bounds the render kernels: the vector pipe's issue rate (counter passes of the same frame, committed)
            issue, _ = replayed(ISSUE_JSON, f"{kname}|{wl.config['workload']}")
            if issue:
                rate = issue["valu_insts_per_step"] * 64.0 / (avg_kernel_ms * 1e-3)
                roof["valu_issue"] = {
                    "wave_insts_per_step": issue["valu_insts_per_step"], "achieved": rate / 1e12, "limit": VALU_LANE_INSTS_PER_S_AT_4_WAVES / 1e12,
                    "unit": "T lane-instructions/s", "frac": rate / VALU_LANE_INSTS_PER_S_AT_4_WAVES,
                    "active_lanes_per_inst": issue["active_lanes_per_valu_inst"],
                    "source": f"instructions: profiles/c2_issue.json (SQ_INSTS_VALU passes at commit {issue['_meta'].get('commit', '?')}, not measured "
                              "in this run); limit: mul + add microbenchmark at 4 waves per SIMD, profiles/r1_microbench_fma_rate.txt"}
    if traffic:  # fabric bytes (Infinity Cache + HBM reads behind L2, and writes) per second of kernel time
        meta = traffic_rec["_meta"]
        roof["traffic_gbs"] = traffic / (avg_kernel_ms * 1e-3) / 1e9
        roof["traffic_source"] = (
            "profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, per kernel "
            "(factor*FETCH_SIZE + WRITE_SIZE)*1024 B with the factor calibrated on known-byte kernels (2 for streamed reads, which the "
            "counter halves; 1 for the closest-hit kernel's 64-byte node gathers, which it counts exactly), "
            f"collected {meta.get('date', '?')} at commit {meta.get('commit', '?')} on kernel sources {meta.get('kernel_sources_sha', '?')} "
            "(unchanged since); not measured in this run")
    else:
        roof["traffic_unavailable"] = traffic_why
    return roof


def main():
    args = parse_args()
    if os.environ.get("RT_BENCH_WATCHDOG"):  # seconds after which every thread's stack is dumped and the process ends
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["RT_BENCH_WATCHDOG"]), exit=True)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch  # device memory, streams and torch.distributed only
    import torch.distributed as dist
    import raytrace_amd as rt

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL
        else:
            dist.init_process_group("gloo")

    cfg = dict(CONFIGS[args.config])
    for k in ("width", "height", "spp"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    accel = rt.ACCEL_LBVH if args.accel == "lbvh" else rt.ACCEL_LINEAR
    r = rt.Renderer(local_rank)
    scene_ids = {"default": rt.SCENE_DEFAULT, "random_spheres": rt.SCENE_RANDOM_SPHERES, "perlin": rt.SCENE_PERLIN,
                 "cloud": rt.SCENE_CLOUD}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(stepfn, n, warmup=1, per_launch=None):
        """Max over ranks of the wall time of n steps between two barrier + synchronize pairs.  Each launch is bracketed by
        HIP events recorded on the library's stream (rt_last_kernel_ms); reading them waits for that launch, which at one
        launch per step costs nothing the step would not wait for anyway."""
        for _ in range(warmup):
            stepfn()
        barrier()
        t_ = time.perf_counter()
        for _ in range(n):
            stepfn()
            if per_launch is not None:
                per_launch.append(r.last_kernel_ms())
        barrier()
        dt_ = time.perf_counter() - t_
        if world > 1:
            tm = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt_ = float(tm.item())
        return dt_

    if args.config == "c4":
        wl = trace_workload(rt, torch, r, dev, args, cfg, rank, world, accel)
    else:
        wl = render_workload(rt, torch, dist, r, dev, args, cfg, rank, world, accel, scene_ids[cfg["scene"]])

    # Timed region of the headline: W untimed steps, then exactly K steps between two barrier + synchronize pairs.
    launches = []
    elapsed = timed(wl.step, args.steps, warmup=args.warmup, per_launch=launches)
    avg_kernel_ms = sum(ms for ms, _ in launches) / len(launches)
    kname = launches[-1][1]

    extras = {}
    if args.config != "c4" and not args.no_extras:
        if world > 1 and not wl.cfg.get("strong"):   # the other scaling mode, after the headline's timed steps
            other, other_total = wl.share_params(not wl.strong)
            dt_ = timed(wl.make_step(other), args.steps)
            extras["weak_scaling" if wl.strong else "strong_scaling"] = {
                "value": cfg["width"] * cfg["height"] * other_total * args.steps / dt_ / 1e6, "unit": "Mpaths/s", "ms_per_step": dt_ / args.steps * 1e3,
                "global_samples_per_pixel": other_total}
        if world == 1 and args.accel == "lbvh" and args.config in ("c2", "c1"):   # the reference's algorithm on the same frame
            r.upload(wl.scene, rt.ACCEL_LINEAR)
            n_lin = max(1, min(args.steps, 2))
            dt_ = timed(wl.step, n_lin)
            ms_lin, k_lin = r.last_kernel_ms()
            r.clear()
            r.render(wl.params, counted=True)
            c_lin = r.counters()
            lin_flops = rt.algorithmic_flops(c_lin)
            extras["linear_scan"] = {"value": wl.units_per_step * n_lin / dt_ / 1e6, "unit": wl.unit, "ms_per_step": dt_ / n_lin * 1e3,
                                     "steps": n_lin, "kernel": k_lin, "kernel_ms": ms_lin,
                                     "roofline_frac": lin_flops / (ms_lin * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                                     "algorithmic_gflop_per_launch": lin_flops / 1e9}
            r.upload(wl.scene, accel)
        if world == 1 and args.accel == "lbvh" and args.config == "c2" and not (args.width or args.height or args.spp):
            # The other single-GPU configurations, timed the same way in the same run: W untimed steps, K steps between
            # barriers, the launches bracketed by HIP events.  Each builds its own scene and device buffers first.
            for name in ("c3", "c4"):
                cfg_x = dict(CONFIGS[name])
                if name == "c4":
                    leg = trace_workload(rt, torch, r, dev, argparse.Namespace(spheres=None, rays=None), cfg_x, 0, 1, rt.ACCEL_LBVH)
                else:
                    leg = render_workload(rt, torch, dist, r, dev, args, cfg_x, 0, 1, rt.ACCEL_LBVH, scene_ids[cfg_x["scene"]], overrides=False)
                n_leg = max(1, min(args.steps, 3))
                leg_launches = []
                dt_ = timed(leg.step, n_leg, warmup=1, per_launch=leg_launches)
                ms_leg = sum(ms for ms, _ in leg_launches) / len(leg_launches)
                extras[name] = {"metric": leg.metric, "value": leg.units_per_step * n_leg / dt_ / 1e6, "unit": leg.unit, "steps": n_leg,
                                "warmup": 1, "ms_per_step": dt_ / n_leg * 1e3, "config": leg.config,
                                "roofline": roofline_of(rt, leg, ms_leg, leg_launches[-1][1], 1)}
                if leg.build_ms is not None:
                    extras[name]["build_ms"] = leg.build_ms
                if leg.node_visits_per_ray is not None:
                    extras[name]["node_visits_per_ray"] = leg.node_visits_per_ray
                del leg
                torch.cuda.empty_cache()
            r.set_accumulator(wl.keep[0].data_ptr(), cfg["width"], cfg["height"])
            r.upload(wl.scene, accel)

    if rank == 0:
        result = {"config": wl.config}
        if wl.build_ms is not None:
            result["build_ms"] = wl.build_ms
        if wl.node_visits_per_ray is not None:
            result["node_visits_per_ray"] = wl.node_visits_per_ray
        result.update({
            "metric": wl.metric, "value": wl.units_per_step * args.steps / elapsed / 1e6, "unit": wl.unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": wl.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        })
        if world > 1:
            result["multi_gpu_note"] = ("the exchange between ranks has run on hardware with one rank only before this run (two and three "
                                        "ranks: gloo on the CPU and ranks sharing one GPU)")
        if world > 1 and args.backend != "nccl":
            result["rehearsal"] = f"{args.backend} backend, ranks sharing {torch.cuda.device_count()} GPU(s): not a measurement"
        result["roofline"] = roofline_of(rt, wl, avg_kernel_ms, kname, world)
        result.update(extras)
        result["counters_per_launch"] = wl.counters
        if not args.no_cpu_baseline and world == 1:
            threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
            if args.config == "c4":
                result["cpu_baseline"] = cpu_baseline_trace(rt, wl.scene, threads, 0xC4C4C4C4)
            else:
                result["cpu_baseline"] = cpu_baseline_render(rt, wl.scene, wl.cam, cfg, threads)
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
