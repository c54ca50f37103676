#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

  python bench.py [--gpus N --steps K --warmup W] [--config c2|c1|c3|c4] [--accel lbvh|linear]

A step is one pass of the hot path over one batch of synthetic input:
  c2 (default)  one 1920x1080 frame of the 487-sphere scene at 256 spp, depth 50  -> Mpaths/s
  c1            the reference's own default render (320x180, 8 spp, xorshift RNG)   -> Mpaths/s
  c3            the Perlin-textured scene, 1920x1080 at 1024 spp                   -> Mpaths/s
  c4            2^20 spheres, a 10^8-ray closest-hit batch (LBVH)                  -> Mrays/s
  c5            the 487-sphere scene at 3840x2160, 4096 spp in total                -> Mpaths/s
Scene and rays are resident in HBM before the timed region starts.

With N > 1 (one process per GPU, started by torch.distributed.run) the SAME
job is divided among the ranks (strong scaling: `value` at N ranks over `value`
at one rank is the speed-up): every rank renders every N-th strip of eight
rows of the frame at the full sample count into its own accumulator, and the
accumulators are summed onto rank 0 with an RCCL reduce inside the timed step,
queued behind the render on the device.  c5 divides its 4096 samples per pixel
by row band and sample range (two bands from four ranks on).  c4 splits the ray
batch by index range and needs no collective.  `--scaling weak` gives every
rank its own 256-sample range of the whole frame instead (N x the work); with
N > 1 the line also carries that figure as `weak_scaling`, measured after the
timed steps of the headline.

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


def traffic_record(kernel, workload):
    """The committed counter passes of this command (profiles/pmc_traffic.json, written by tools/pmc_traffic.py from
    `rocprofv3 --pmc` runs), or None when that workload has not been profiled.  Counters cannot be read from inside
    the timed run."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(f"{kernel}|{workload}")
    except (OSError, ValueError):
        return None


def measured_traffic(kernel, workload):
    """HBM bytes per step from those passes, or None."""
    rec = traffic_record(kernel, workload)
    return rec["hbm_bytes_per_step"] if rec else None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))  # c5 at N=1 takes about 6 s per step
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
    ap.add_argument("--no-extras", action="store_true", help="skip the linear-scan and weak-scaling legs")
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


def traffic_source():
    """Where roofline.traffic comes from: the committed counter passes, with their date and commit."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            meta = json.load(f).get("_meta", {})
    except (OSError, ValueError):
        return None
    return ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, per kernel "
            "(factor*FETCH_SIZE + WRITE_SIZE)*1024 B with the factor calibrated on known-byte kernels (2 for streamed reads, which the "
            "counter halves; 1 for the closest-hit kernel's 64-byte node gathers, which it counts exactly), "
            f"collected {meta.get('date', '?')} at commit {meta.get('commit', '?')}; not measured in this run")


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
    sid = scene_ids[cfg["scene"]]
    result = {}

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.config == "c4":
        n_sph = args.spheres or cfg["n_spheres"]
        n_rays_total = args.rays or cfg["n_rays"]
        scene = rt.build_scene(sid, n=n_sph, seed=0xC4C4C4C4)
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

        def step(counted=False):
            r.trace_device(o.data_ptr(), d.data_ptr(), n, ids.data_ptr(), t.data_ptr(), nrm.data_ptr(), counted)

        step(counted=True)
        counters = r.counters()
        if counters["segments"]:
            result["node_visits_per_ray"] = counters["trav_lane_steps"] / counters["segments"]
        units_per_step = n_rays_total
        unit, metric = "Mrays/s", "closest-hit rays per second (Mrays/s)"
        algo_bytes = 44.0 * n  # 24 B in + 20 B out per ray
        info = r.scene_info()
        result["config"] = {"workload": cfg["workload"] if not (args.spheres or args.rays) else f"{n_sph} spheres, {n_rays_total}-ray closest-hit batch",
                            "accel": args.accel, "builder": info["builder"], "tree_nodes": info["n_nodes"], "node_bytes": info["node_bytes"],
                            "spheres": n_sph, "rays": n_rays_total, "lbvh_depth": r.lbvh_depth(), "ranks": world,
                            "parallelism": f"ray index ranges x{world}"}
        result["_scaling"] = "strong"  # the batch is fixed; ranks take index ranges of it
        if accel == rt.ACCEL_LBVH and info["builder"] != "none":
            result["build_ms"] = r.last_build_ms()
    else:
        scene = rt.build_scene(sid)
        cam = rt.scene_camera(sid)
        rng_mode = rt.RNG_COUNTER if cfg["rng"] == "counter" else rt.RNG_XORSHIFT_REF
        if world > 1 and rng_mode != rt.RNG_COUNTER:
            raise SystemExit("sample-range splitting needs the counter RNG")
        w, h, spp = cfg["width"], cfg["height"], cfg["spp"]
        r.upload(scene, accel)
        accum = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
        r.set_accumulator(accum.data_ptr(), w, h)
        from raytrace_amd import distributed as rtd
        strong = args.scaling == "strong" or bool(cfg.get("strong"))

        def share_params(strongly):
            total = spp if strongly else spp * world
            if not strongly:      # weak: rank r renders samples [r*spp, (r+1)*spp) of every pixel
                share = rtd.split_frame(h, total, world)[rank]
            elif cfg.get("strong"):  # c5: row bands x sample ranges of the fixed job
                share = rtd.split_frame(h, total, world, row_groups=2 if world >= 4 and world % 2 == 0 else 1)[rank]
            else:                 # every world-th strip of eight rows, all samples: disjoint pixels, balanced cost
                share = rtd.split_interleaved(h, total, world)[rank]
            return rtd.apply_share(rt.make_params(cam, w, h, total, rng_mode=rng_mode), share), total

        p, total_spp = share_params(strong)

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

        # strips over all ranks (the default job): disjoint rows, nothing to sum — rows are gathered, not reduced
        strips_only = world > 1 and strong and not cfg.get("strong") and args.collective == "gather"
        step = make_step(p, rtd.split_interleaved(h, total_spp, world) if strips_only else None)
        step(counted=True)
        counters = r.counters()
        units_per_step = w * h * total_spp
        unit, metric = "Mpaths/s", "path-tracing throughput at %dx%d (Mpaths/s)" % (w, h)
        algo_bytes = None
        info = r.scene_info()
        result["config"] = {"workload": f"{cfg['workload'].split(' ')[0]} scene {w}x{h}x{spp}spp depth50" if (args.width or args.height or args.spp) else cfg["workload"],
                            # (the library scans scenes of up to 16 spheres even when a tree is asked for: the scan is faster there)
                            "accel": args.accel if info["builder"] != "none" or accel != rt.ACCEL_LBVH else "lbvh asked for, linear scan taken (few spheres)",
                            "builder": info["builder"], "tree_nodes": info["n_nodes"], "tree_in_lds": bool(info["tree_in_lds"]),
                            "spheres": scene.n_spheres, "rng": cfg["rng"],
                            "lbvh_depth": r.lbvh_depth(), "global_samples_per_pixel": total_spp,
                            "ranks": world, "collective": (None if world == 1 else f"{args.backend} gather of each rank's rows onto rank 0" if strips_only else
                                                           f"{args.backend} reduce to rank 0, {args.bands} band(s) per step"),
                            "parallelism": ("single GPU" if world == 1 else
                                            f"row bands x sample ranges over {world} ranks" if cfg.get("strong") else
                                            f"{world} interleaved sets of 8-row strips" if strong else f"sample ranges x{world}")}
        result["_scaling"] = "strong" if strong else "weak"
        if accel == rt.ACCEL_LBVH and info["builder"] != "none":
            result["build_ms"] = r.last_build_ms()

    for _ in range(args.warmup):
        step()
    barrier()
    # Timed region: exactly K steps between two barrier + synchronize pairs.
    # Each launch is bracketed by HIP events recorded on the library's stream
    # (rt_last_kernel_ms); reading them waits for that launch, which at one
    # launch per step costs nothing the step would not wait for anyway.
    per_launch = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        ms, kname = r.last_kernel_ms()
        per_launch.append(ms)
    barrier()
    elapsed = time.perf_counter() - t0
    avg_kernel_ms = sum(per_launch) / len(per_launch)

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    def timed(stepfn, n):
        """Max over ranks of the wall time of n steps between two barriers."""
        stepfn()
        barrier()
        t_ = time.perf_counter()
        for _ in range(n):
            stepfn()
        barrier()
        dt_ = time.perf_counter() - t_
        if world > 1:
            tm = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dt_ = float(tm.item())
        return dt_

    extras = {}
    if args.config != "c4" and not args.no_extras:
        if world > 1 and not cfg.get("strong"):   # the other scaling mode, after the headline's timed steps
            other, other_total = share_params(not strong)
            dt_ = timed(make_step(other), args.steps)
            extras["weak_scaling" if strong else "strong_scaling"] = {
                "value": w * h * other_total * args.steps / dt_ / 1e6, "unit": "Mpaths/s", "ms_per_step": dt_ / args.steps * 1e3,
                "global_samples_per_pixel": other_total}
        if world == 1 and args.accel == "lbvh" and args.config in ("c2", "c1"):   # the reference's algorithm on the same frame
            r.upload(scene, rt.ACCEL_LINEAR)
            n_lin = max(1, min(args.steps, 2))
            dt_ = timed(step, n_lin)
            ms_lin, k_lin = r.last_kernel_ms()
            r.clear()
            r.render(p, counted=True)
            c_lin = r.counters()
            lin_flops = rt.algorithmic_flops(c_lin)
            extras["linear_scan"] = {"value": units_per_step * n_lin / dt_ / 1e6, "unit": unit, "kernel": k_lin, "kernel_ms": ms_lin,
                                     "roofline_frac": lin_flops / (ms_lin * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
                                     "algorithmic_gflop_per_launch": lin_flops / 1e9}
            r.upload(scene, accel)

    if rank == 0:
        flops = rt.algorithmic_flops(counters)
        value = units_per_step * args.steps / elapsed / 1e6
        achieved_tflops = flops / (avg_kernel_ms * 1e-3) / 1e12
        result.update({
            "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": result.pop("_scaling", "weak"), "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        })
        if world > 1 and args.backend != "nccl":
            result["rehearsal"] = f"{args.backend} backend, ranks sharing {torch.cuda.device_count()} GPU(s): not a measurement"
        traffic = measured_traffic(kname, result["config"]["workload"])
        if args.config == "c4":
            result["roofline"] = {"bound": "hbm", "achieved": algo_bytes / (avg_kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": algo_bytes / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "traffic": traffic, "kernel": kname, "kernel_ms": avg_kernel_ms,
                                  "valu_tflops": achieved_tflops, "valu_frac": achieved_tflops / FP32_VECTOR_PEAK_TFLOPS}
        else:
            result["roofline"] = {"bound": "valu_fp32", "achieved": achieved_tflops, "peak": FP32_VECTOR_PEAK_TFLOPS,
                                  "unit": "TFLOP/s", "frac": achieved_tflops / FP32_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                                  "kernel": kname, "kernel_ms": avg_kernel_ms,
                                  "algorithmic_gflop_per_launch": flops / 1e9}
        rec = traffic_record(kname, result["config"]["workload"])
        if args.config == "c4" and rec and rec.get("fabric_read_requests_per_step"):
            # what bounds an incoherent batch: every L2 miss of a node or leaf read is one read request to the fabric
            req = rec["fabric_read_requests_per_step"] * (n / max(rec.get("rays_per_step", n), 1))
            rate = req / (avg_kernel_ms * 1e-3) / 1e9
            result["roofline"]["fabric_requests"] = {
                "per_ray": req / max(n, 1), "achieved": rate, "ceiling": FABRIC_REQUEST_CEILING_GPS, "unit": "G requests/s",
                "frac": rate / FABRIC_REQUEST_CEILING_GPS,
                "source": "requests: TCC_EA0_RDREQ pass in profiles/pmc_traffic.json (not measured in this run); ceiling: random "
                          "64-byte gathers of tools/microbench/fetch_calib, profiles/r2_fetch_calibration.txt"}
        try:  # what bounds the render kernels: the vector pipe's issue rate (counter passes of the same frame, committed)
            with open(os.path.join(ROOT, "profiles", "r2_c2_issue.json")) as f:
                issue = json.load(f).get(f"{kname}|{result['config']['workload']}")
        except (OSError, ValueError):
            issue = None
        if issue and world == 1:
            rate = issue["valu_insts_per_step"] * 64.0 / (avg_kernel_ms * 1e-3)
            result["roofline"]["valu_issue"] = {
                "wave_insts_per_step": issue["valu_insts_per_step"], "achieved": rate / 1e12, "limit": VALU_LANE_INSTS_PER_S_AT_4_WAVES / 1e12,
                "unit": "T lane-instructions/s", "frac": rate / VALU_LANE_INSTS_PER_S_AT_4_WAVES,
                "active_lanes_per_inst": issue["active_lanes_per_valu_inst"],
                "source": "instructions: profiles/r2_c2_issue.json (SQ_INSTS_VALU passes, not measured in this run); limit: mul + add "
                          "microbenchmark at 4 waves per SIMD, profiles/r1_microbench_fma_rate.txt"}
        if traffic:  # fabric bytes (Infinity Cache + HBM reads behind L2, and writes) per second of kernel time
            result["roofline"]["traffic_gbs"] = traffic / (avg_kernel_ms * 1e-3) / 1e9
            result["roofline"]["traffic_source"] = traffic_source()
        result.update(extras)
        result["counters_per_launch"] = counters
        if not args.no_cpu_baseline and world == 1:
            threads = max(1, min(args.cpu_threads, os.cpu_count() or 1))
            if args.config == "c4":
                result["cpu_baseline"] = cpu_baseline_trace(rt, scene, threads, 0xC4C4C4C4)
            else:
                result["cpu_baseline"] = cpu_baseline_render(rt, scene, cam, cfg, threads)
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
