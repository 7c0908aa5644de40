"""Headline benchmark: path-tracing throughput on the 1080p triangle-mesh Cornell box.

A step = one full render of BASELINE.json's config 2 (1920x1080, 1024 spp, 8 bounces) on each GPU: rank r
renders sample indices [r*1024, (r+1)*1024) of the frame (spp-slice sharding, weak scaling) and the
accumulation buffers are summed onto rank 0 with one RCCL reduce per step. `value` is the whole-job
Msamples/s (1 sample = one full path of one pixel) with the scene and film resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--spp S] [--depth D] [--no-cpu] [--config C] [--scaling weak|strong]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

--scaling strong keeps the image fixed instead: --spp samples per pixel in total, split over the ranks
(`time_to_1024spp_s` is then the metric BASELINE.json names, at every N).
--config 3|4|5 runs the other GPU configurations of BASELINE.json through the same harness.

Rank 0 prints ONE JSON line. At N=1 it also carries
  * `roofline`: the render kernel's executed work (its own traversal counters, one extra counting launch outside the
    timed region) over its launch time (events on the launch stream), against the fp32 vector peak for scenes that
    are traversed from LDS and against HBM bandwidth for the 1 M-triangle scene;
  * `cpu_baseline`: the reference's own sources (oracle/_ref/ref_oracle, its ThreadPool and 128 row tasks) timed on
    this box's host cores on a bounded sample of the same scene; the plain-C restatement when that binary is absent.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "path-tracer_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X fp32 vector peak: 256 CU x 128 lanes x 2 flop x 2.4 GHz
HBM_PEAK_GBS = 8000.0
# flops per unit of work (one ray-box slab test 27, one ray-triangle test 47, one sphere test 25, one oriented-box
# test 60, one shaded hit 250) and bytes per unit on the global-memory traversal (node visit 64, or 80 for an eight-child node; triangle record 48,
# closest-hit reconstruction 112 shade + 48 material)
F_SLAB, F_TRI, F_SPHERE, F_BOX, F_SHADE = 27.0, 47.0, 25.0, 60.0, 250.0
B_NODE, B_WIDE_NODE, B_TRI, B_HIT = 64.0, 80.0, 48.0, 160.0


def host_cores():
    """(CPUs this process may run on, cgroup CPU quota in CPUs or None)."""
    avail = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        pass
    return avail, quota


def reference_flops(cnt):
    """Flops of the reference's algorithm (median-split tree, every child box tested twice) for the work the CPU
    oracle counted on the same scene."""
    return (F_SLAB * cnt["nodes_visited"] + F_TRI * cnt["tris_tested"] + F_SPHERE * cnt["sphere_tests"]
            + F_BOX * cnt["box_tests"] + F_SHADE * cnt["shades"])


def cpu_baseline(scene, depth, budget_s=20.0):
    """CPU baseline on a bounded sample of the same workload: the same scene and camera on a smaller film, spp chosen
    so that all legs together take about budget_s. Thread counts tried: every CPU this process may use (capped at 512
    software threads), and the reference's own default of 32 (main.cpp:352) when that is fewer; the faster leg is the
    baseline. Returns (baseline dict, work counters of the plain-C restatement on the sample)."""
    from oracle import pyoracle as po
    W, H = scene["film"]
    avail, quota = host_cores()
    div = 2 if avail > 32 else 4  # 128 row tasks need at least 128 rows to keep many cores busy
    w, h = W // div, H // div
    small = dict(scene, film=(w, h))
    small.pop("skybox", None)  # the reference has no sky map (PathTracer.cpp:59-60): its miss colour is the constant
    S = po.Scene(small)
    _, _, cnt = S.render(2, depth, threads=min(avail, 64), counters=True)
    legs = sorted({min(avail, 512), min(avail, 32)})
    what = "%dx%d (same scene and camera, 1/%d of the pixels), %d spp, depth %d, %.1f s"
    tried = []
    for threads in legs:
        if po.have_ref():
            try:
                _, _, _, info = po.ref_render(small, 1, depth, threads=threads, tasks=128)
                spp = int(max(2, min(1024, budget_s / len(legs) / max(info["seconds"], 1e-3))))
                _, _, _, info = po.ref_render(small, spp, depth, threads=threads, tasks=128)
                leg = {"value": info["msamples_per_s"], "unit": "Msamples/s", "cores": threads, "kind": "reference",
                       "sample": what % (w, h, div * div, spp, depth, info["seconds"])
                       + "; the reference's own sources with its ThreadPool (%d threads, 128 row tasks)" % threads}
            except Exception as e:  # a box without the binary's runtime: time the port instead, and say so
                print("bench: reference binary unusable (%s), timing the port instead" % e, file=sys.stderr)
                leg = None
        else:
            leg = None
        if leg is None:
            t0 = time.time()
            S.render(1, depth, threads=threads)
            spp = int(max(2, min(1024, budget_s / len(legs) / max(time.time() - t0, 1e-3))))
            t0 = time.time()
            S.render(spp, depth, threads=threads)
            dt = time.time() - t0
            leg = {"value": w * h * spp / dt * 1e-6, "unit": "Msamples/s", "cores": threads, "kind": "port",
                   "sample": what % (w, h, div * div, spp, depth, dt) + "; plain-C restatement, %d threads" % threads}
        tried.append(leg)
    best = max(tried, key=lambda l: l["value"])
    best["legs"] = [{"threads": l["cores"], "value": l["value"], "kind": l["kind"]} for l in tried]
    best["cores_available"] = avail
    best["cpu_quota"] = quota
    return best, cnt


def measured_traffic(config, spp, depth):
    """HBM bytes per launch of the render kernel from the memory-side counters (rocprofv3 FETCH_SIZE / WRITE_SIZE in
    separate passes, tools/profile_bench.sh, summaries committed under profiles/). Only valid for the workload it
    was measured on; None otherwise."""
    import glob
    import re
    tag = "" if config == 2 else "_config%d" % config
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic%s.json" % tag))]
    files.sort(key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])  # r1_v9 before r1_v10 before r2_v1
    for f in reversed(files):
        with open(f) as fh:
            d = json.load(fh)
        if d.get("spp", 1024) == spp and d.get("depth", 8) == depth:
            return float(d["bytes_per_launch"]), os.path.basename(f)
    return None, None


def build_config(args, tmpdir):
    """(scene dict, workload name, sharding mode, Russian roulette, default spp per GPU, depth, sky faces or None)."""
    from scenes import dragon, presets
    c = args.config
    if c == 2:
        scene = presets.mesh_cornell("tests/golden/cornell_mesh.glb", 1920, 1080)
        return scene, "triangle-mesh Cornell box (108 triangles, BVH) 1920x1080", "spp", False, 1024, 8, None
    if c == 3:
        glb = os.path.join(tmpdir, "dragon_field.glb")
        n = dragon.write_dragon_field(glb)
        scene = dragon.config3_scene(glb, 1920, 1080)
        m = 64  # a 64^2 gradient sky for the paths that leave through the room's open front
        yy = np.linspace(1.0, 0.0, m, dtype=np.float32)[:, None] * np.ones((1, m), np.float32)

        def face(level):
            v = np.clip(level, 0, 1)
            return np.stack([(120 + 100 * v), (150 + 90 * v), 255 * np.ones_like(v), 255 * np.ones_like(v)], -1).astype(np.uint8)
        sky = np.ascontiguousarray(np.stack([face(yy), face(yy), face(np.ones((m, m), np.float32)), face(np.zeros((m, m), np.float32)), face(yy), face(yy)]))
        return (scene, "dragon field: %d triangles (ten baked instances of the reference's chinese_dragon.glb, scenes/dragon.py) in the closed "
                "Cornell room, 64^2 sky cubemap behind the open front, 1920x1080" % n, "spp", False, 512, 8, sky)
    if c == 4:
        scene = presets.mixed("tests/golden/blob.glb", 3840, 2160, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0))
        return scene, "4K mixed spheres + boxes + 4 k-triangle mesh 3840x2160", "tile", False, 256, 8, None
    scene = presets.glass_mirror(1920, 1080)
    return scene, "glass/mirror spheres 1920x1080, Russian roulette", "spp", True, 256, 32, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel per GPU (weak) or in total (strong); default per config")
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json configuration: 2 (default, the headline) mesh Cornell; 3 1 M-triangle dragon field + sky, "
                         "spp-sharded; 4 4K mixed scene, tile-sharded; 5 glass/mirror, 32 bounces, Russian roulette")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="weak: --spp per GPU (default for spp-sharded configs); strong: --spp in total, split over the ranks")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    os.chdir(ROOT)
    torch.cuda.set_device(local_rank)
    distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # launched by torch.distributed.run, world 1 included
    if distributed:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import ptamd
    import shard
    ptamd.init(local_rank)
    stream = torch.cuda.current_stream()
    ptamd.set_stream(stream.cuda_stream)  # launches, torch events and the collective share one stream
    tmpdir = tempfile.mkdtemp(prefix="ptbench_")
    scene, name, mode, rr, spp_default, depth_default, sky = build_config(args, tmpdir)
    W, H = scene["film"]
    depth = args.depth if args.depth is not None else depth_default
    spp_arg = args.spp if args.spp is not None else spp_default
    scaling = args.scaling or ("weak" if mode == "spp" else "strong")
    if mode == "tile" and scaling == "weak":
        raise SystemExit("the tile-sharded configuration has a fixed film: strong scaling only")
    pt, cam = ptamd.build_scene(scene)
    if sky is not None:
        pt.SetSkybox(sky, 1.5)
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    samples = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    film = ptamd.Film(W, H, external=(accum.data_ptr(), samples.data_ptr()))
    if mode == "tile":
        spp_total, spp_begin, spp_count = spp_arg, 0, spp_arg
        pixels_rank = None
    elif scaling == "weak":
        spp_total = spp_arg * world
        spp_begin, spp_count = shard.spp_slice(rank, world, spp_total)
        assert spp_count == spp_arg
    else:
        spp_total = spp_arg
        spp_begin, spp_count = shard.spp_slice(rank, world, spp_total)

    def launch(count=False):
        if mode == "tile":
            ptamd.Render(cam, pt, film, spp_count, depth, spp_begin=spp_begin, russian_roulette=rr, tiles=(32, rank, world), count=count)
        elif spp_count:
            ptamd.Render(cam, pt, film, spp_count, depth, spp_begin=spp_begin, russian_roulette=rr, count=count)

    def barrier():
        if distributed and world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        accum.zero_(); samples.zero_()
        launch()
        shard.reduce_film(accum, dst=0)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        accum.zero_(); samples.zero_()
        ev[i][0].record(stream)
        launch()
        ev[i][1].record(stream)
        shard.reduce_film(accum, dst=0)  # one RCCL reduce over xGMI per render (sample counts are known: spp_total)
    barrier()
    elapsed = time.perf_counter() - t0
    if distributed and world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # executed work of this rank's launch (one extra counting launch, outside the timed region)
    accum.zero_(); samples.zero_()
    launch(count=True)
    torch.cuda.synchronize()
    stats = ptamd.RenderStats()

    if rank == 0:
        samples_per_step = float(W) * H * spp_total
        value = samples_per_step * args.steps / elapsed * 1e-6
        rays_per_sample = stats["rays"] / max(1, stats["samples"])
        sharding = ("interleaved 32x32 tiles per rank, one launch (ptamd_render_tiles)" if mode == "tile" else "spp-slice per rank") + \
                   " + one RCCL reduce of the %.1f MB accumulation buffer per step" % (W * H * 12 / 1e6)
        out = {"metric": "Msamples/s", "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
               "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s, %d spp %s, %d bounces" % (name, spp_arg, "per GPU" if scaling == "weak" else "in total", depth),
                          "baseline_config": args.config, "sharding": sharding, "seed": "0x5EED"},
               "mrays_per_s": value * rays_per_sample, "rays_per_sample": rays_per_sample,
               "time_to_1024spp_s": (elapsed / args.steps) * (1024.0 / spp_total) if (args.config == 2 and (scaling == "strong" or world == 1)) else None,
               "kernel_ms_per_launch": kernel_ms}
        if world == 1:
            n_sph = sum(1 for o in scene["objects"] if o["type"] == "sphere")
            n_box = sum(1 for o in scene["objects"] if o["type"] == "box")
            # a visit of a two-child node tests two boxes and reads 64 B; a visit of an eight-child node tests eight
            # quantised boxes and reads 80 B (meshes of 200 000 faces and more in scenes that do not fit LDS)
            wide = pt.Info()["wide_nodes"] > 0
            slabs, node_bytes = (8, B_WIDE_NODE) if wide else (2, B_NODE)
            flops = (slabs * F_SLAB * stats["nodes"] + F_TRI * stats["tris"] + (F_SPHERE * n_sph + F_BOX * n_box) * stats["rays"] + F_SHADE * stats["shade_lanes"])
            lds_resident = stats["steal_rounds"] == 0  # the work-sharing loop only runs on the global-memory traversal
            launched = float(W) * H * spp_count
            scale = launched / max(1, stats["samples"])
            traffic, traffic_file = measured_traffic(args.config, spp_count, depth)
            if args.config == 3:
                algo_bytes = (node_bytes * stats["nodes"] + B_TRI * stats["tris"] + B_HIT * stats["shade_lanes"]) * scale + 32.0 * W * H
                achieved = algo_bytes / (kernel_ms * 1e-3) * 1e-9
                roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "algorithmic_bytes_per_launch": algo_bytes,
                        "bytes_per_ray": algo_bytes / max(1.0, stats["rays"] * scale),
                        "note": "gather bound: %d B per node visit, 48 B per triangle test, 160 B per reconstructed hit, as executed "
                                "by this kernel's own walk (counting launch), plus the film read and written once; most of these "
                                "requests are served by L2 and the Infinity Cache, so `traffic` (memory side) is below" % node_bytes,
                        "node_width": slabs}
            else:
                achieved = flops * scale / (kernel_ms * 1e-3) * 1e-12
                roof = {"bound": "valu", "achieved": achieved, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / FP32_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                        "note": "fp32 vector ALU bound (no MFMA: not a dense contraction; scene geometry lives in %s, film traffic is 32 B "
                                "per pixel and launch): `achieved` prices the slab / triangle / sphere / box tests and shaded hits this "
                                "kernel executed (its own counters)" % ("LDS" if lds_resident else "L2")}
            roof.update({"kernel": "render_kernel", "traffic_unit": "bytes per launch (memory-side counters, PMC passes committed under profiles/)",
                         "traffic_source": traffic_file, "executed_flops_per_sample": flops / max(1, stats["samples"]),
                         "executed_per_ray": {"nodes": stats["nodes"] / max(1, stats["rays"]), "tris": stats["tris"] / max(1, stats["rays"])},
                         "lane_utilisation": {"node_loop": stats["nodes"] / max(1.0, 64.0 * stats["node_trips"]),
                                              "triangle_loop": stats["tris"] / max(1.0, 64.0 * stats["tri_trips"]),
                                              "shading": stats["shade_lanes"] / max(1.0, 64.0 * stats["shade_trips"])}})
            if traffic is not None:
                roof["hbm"] = {"achieved": traffic / (kernel_ms * 1e-3) * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": traffic / (kernel_ms * 1e-3) * 1e-9 / HBM_PEAK_GBS}
            out["roofline"] = roof
            if not args.no_cpu:
                base, cnt = cpu_baseline(scene, depth)
                ref_fps = reference_flops(cnt) / max(1, cnt["samples"])
                out["roofline"]["reference_equivalent"] = {
                    "flops_per_sample": ref_fps, "tflops": ref_fps * launched / (kernel_ms * 1e-3) * 1e-12,
                    "note": "the same sample rate priced with the flops the REFERENCE's algorithm spends per sample on this scene "
                            "(CPU oracle counters: median-split tree, every child box tested twice); not a utilisation figure"}
                out["cpu_baseline"] = base
                out["speedup_vs_cpu_baseline"] = value / base["value"]
            else:
                out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
