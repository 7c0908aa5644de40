"""Headline benchmark: path-tracing throughput on the 1080p triangle-mesh Cornell box.

A step = one full render of BASELINE.json's config 2 (1920x1080, 1024 spp, 8 bounces). With N GPUs the 1024 samples of
every pixel are split over the ranks (spp-slice sharding: rank r renders sample indices [r*1024/N, (r+1)*1024/N) of the
whole frame) and the accumulation buffers are summed onto rank 0 with one RCCL reduce per step, inside the timed region.
`value` is the whole-job Msamples/s (1 sample = one full path of one pixel) with the scene and film resident in HBM;
`time_to_1024spp_s` is BASELINE.json's other metric, filled at every N.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--spp S] [--depth D] [--no-cpu] [--config C] [--scaling weak|strong]

`--gpus N` with N > 1 starts its own N ranks (one process per GPU through torch.distributed.run, before this process has
touched a GPU) unless it already runs as one of them (RANK set by a launcher), and relays rank 0's JSON line.
--scaling weak gives every rank --spp samples per pixel instead (image = N x spp).
--config 3|4|5 runs one of the other GPU configurations of BASELINE.json through the same harness; without --config,
at N = 1, the headline line also carries short runs of all three under "other_configs".

Rank 0 prints ONE JSON line. At N=1 it also carries
  * `roofline`: the render kernel's executed work (its own traversal counters, one extra counting launch outside the
    timed region) over its launch time (events on the launch stream) against the fp32 vector peak, the bound of every
    configuration here (vector-instruction issue; no MFMA, no HBM stream); memory-side bytes beside it;
  * `cpu_baseline`: the reference's own sources (oracle/_ref/ref_oracle, its ThreadPool and 128 row tasks) timed on
    this box's host cores on a bounded sample of the same scene; the plain-C restatement when that binary is absent.

Test hooks (environment): PTAMD_BENCH_BACKEND=gloo joins the ranks over gloo instead of RCCL, PTAMD_BENCH_DEVICE=k puts
every rank on GPU k (two ranks on a one-GPU box), PTAMD_BENCH_DUMP=path makes rank 0 save the last step's summed film.
"""
import argparse
import glob
import json
import os
import re
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "path-tracer_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X fp32 vector peak: 256 CU x 128 lanes x 2 flop x 2.4 GHz
HBM_PEAK_GBS = 8000.0
# flops per unit of work (one ray-box slab test 27, one ray-triangle test 47, one sphere test 25, one oriented-box
# test 60, one shaded hit 250) and bytes requested per unit on the global-memory traversal (node visit 64, or 80 for an
# eight-child node; triangle record 48, closest-hit reconstruction 112 shade + 48 material)
F_SLAB, F_TRI, F_SPHERE, F_BOX, F_SHADE = 27.0, 47.0, 25.0, 60.0, 250.0
B_NODE, B_WIDE_NODE, B_TRI, B_HIT = 64.0, 80.0, 48.0, 160.0


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """Starts n fresh rank processes of this file (torch.distributed.run, one per GPU, rendezvous on 127.0.0.1), relays
    the last JSON line they print and returns the launcher's exit code. Called before anything here touches a GPU: a
    process that has initialised the GPU must not exec or fork workers."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    lines = p.stdout.splitlines()
    js = [l for l in lines if l.startswith("{")]
    for l in lines:
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if js:
        print(js[-1], flush=True)
    return p.returncode if p.returncode != 0 else (0 if js else 1)


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
    baseline. The reference has neither a sky map nor Russian roulette (PathTracer.cpp:35-36, 59-60): it renders the
    constant miss colour and full-length paths. Returns (baseline dict, work counters of the plain-C restatement)."""
    from oracle import pyoracle as po
    W, H = scene["film"]
    avail, quota = host_cores()
    div = 2 if avail > 32 else 4  # 128 row tasks need at least 128 rows to keep many cores busy
    w, h = W // div, H // div
    small = dict(scene, film=(w, h))
    small.pop("skybox", None)
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
    was measured on; (None, None) otherwise."""
    tag = "" if config == 2 else "_config%d" % config
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic%s.json" % tag))]
    files.sort(key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])  # r1_v9 before r1_v10 before r2_v1
    for f in reversed(files):
        with open(f) as fh:
            d = json.load(fh)
        if d.get("spp", 1024) == spp and d.get("depth", 8) == depth:
            return float(d["bytes_per_launch"]), os.path.basename(f)
    return None, None


def load_sky_faces():
    """The six 1024^2 faces of the reference's own (unused) cubemap, assets/skyboxes/sky/{p,n}{x,y,z}.png, kept as a data
    fixture under scenes/sky/, as float radiance: texel / 255 decoded from sRGB and scaled so that the brightest
    texels reach 4 (an HDR map: values above 1 do occur). (6, N, N, 3) float32, faces +x -x +y -y +z -z, row 0 on top."""
    from scenes import skybox
    return skybox.load_hdr_faces(os.path.join(ROOT, "scenes", "sky"))


def build_config(config, tmpdir):
    """(scene dict, workload name, sharding mode, Russian roulette, default spp, depth, float sky faces or None)."""
    from scenes import dragon, presets
    c = config
    if c == 2:
        scene = presets.mesh_cornell("tests/golden/cornell_mesh.glb", 1920, 1080)
        return scene, "triangle-mesh Cornell box (108 triangles, BVH) 1920x1080", "spp", False, 1024, 8, None
    if c == 3:
        glb = os.path.join(tmpdir, "dragon_field.glb")
        n = dragon.write_dragon_field(glb)
        scene = dragon.config3_scene(glb, 1920, 1080)
        return (scene, "dragon field: %d triangles (ten baked instances of the reference's chinese_dragon.glb, scenes/dragon.py) in the closed "
                "Cornell room, the reference's assets/skyboxes/sky cubemap (1024^2 faces, float radiance) behind the open front, 1920x1080 "
                "(BASELINE.json: 4096 spp over 8 GPUs; one GPU runs its 512-spp share)" % n,
                "spp", False, 512, 8, load_sky_faces())
    if c == 4:
        scene = presets.mixed("tests/golden/blob.glb", 3840, 2160, mesh_pos=(2.6, 1.3, 2.4), mesh_rot=(10.0, 30.0, 5.0), mesh_scale=(1.1, 0.9, 1.0))
        return (scene, "4K mixed spheres + boxes + 7680-triangle mesh 3840x2160 (BASELINE.json names 16 384 spp: the passes are identical, "
                "so the rate does not depend on their number)", "tile", False, 256, 8, None)
    scene = presets.glass_mirror(1920, 1080)
    return scene, "glass/mirror spheres 1920x1080", "spp", True, 256, 32, None


class Ranks:
    """This process's place in the job: rank, world size, device, and the collectives of the timing protocol."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if args.gpus != self.world:
            raise SystemExit("--gpus %d but the launcher started %d ranks" % (args.gpus, self.world))
        self.backend = os.environ.get("PTAMD_BENCH_BACKEND", "nccl")
        self.device = int(os.environ.get("PTAMD_BENCH_DEVICE", self.local_rank))
        torch.cuda.set_device(self.device)
        self.distributed = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # started by a launcher, a world of one included
        if self.distributed:
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.device))
            else:
                dist.init_process_group(self.backend)
        self.ranks_seen = dist.get_world_size() if self.distributed else 1

    def barrier(self):
        if self.distributed and self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if not (self.distributed and self.world > 1):
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.distributed:
            self.dist.destroy_process_group()


def run_workload(rk, config, steps, warmup, spp_arg=None, depth_arg=None, scaling="strong", want_cpu=True, cpu_budget=20.0,
                 roulette=None, dump=None):
    """Builds one configuration, times `steps` renders of it after `warmup` untimed ones, and returns rank 0's result
    dict (None on other ranks). roulette overrides the configuration's Russian-roulette setting."""
    import ptamd
    import shard
    tmpdir = tempfile.mkdtemp(prefix="ptbench_")  # generated model files: the GPU library and the CPU legs load them from here
    try:
        return _run_workload(rk, ptamd, shard, tmpdir, config, steps, warmup, spp_arg, depth_arg, scaling, want_cpu, cpu_budget, roulette, dump)
    finally:
        shutil.rmtree(tmpdir, ignore_errors=True)


def _run_workload(rk, ptamd, shard, tmpdir, config, steps, warmup, spp_arg, depth_arg, scaling, want_cpu, cpu_budget, roulette, dump):
    torch = rk.torch
    scene, name, mode, rr, spp_default, depth_default, sky = build_config(config, tmpdir)
    if roulette is not None:
        rr = roulette
    W, H = scene["film"]
    depth = depth_arg if depth_arg is not None else depth_default
    spp_in = spp_arg if spp_arg is not None else spp_default
    if config == 3 and spp_arg is None and rk.world > 1 and scaling == "strong":
        spp_in = 4096  # BASELINE.json's figure for this configuration, split over the ranks (one GPU alone runs an eighth of it)
    if mode == "tile" and scaling == "weak":
        raise SystemExit("the tile-sharded configuration has a fixed film: strong scaling only")
    pt, cam = ptamd.build_scene(scene)
    if sky is not None:
        pt.SetSkyboxFloat(sky)
    stream = torch.cuda.current_stream()
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    samples = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    film = ptamd.Film(W, H, external=(accum.data_ptr(), samples.data_ptr()))
    rank, world = rk.rank, rk.world
    if mode == "tile":
        spp_total, spp_begin, spp_count = spp_in, 0, spp_in
    elif scaling == "weak":
        spp_total = spp_in * world
        spp_begin, spp_count = shard.spp_slice(rank, world, spp_total)
        assert spp_count == spp_in
    else:
        spp_total = spp_in
        spp_begin, spp_count = shard.spp_slice(rank, world, spp_total)

    def launch(count=False):
        if mode == "tile":
            ptamd.Render(cam, pt, film, spp_count, depth, spp_begin=spp_begin, russian_roulette=rr, tiles=(32, rank, world), count=count)
        elif spp_count:
            ptamd.Render(cam, pt, film, spp_count, depth, spp_begin=spp_begin, russian_roulette=rr, count=count)

    # Scene set-up, like the commit: the library chooses the schedule of a scene's mesh walks by a measurement at its first
    # launch of 16 samples and more (ptamd_scene_info.walk_cut), so one such launch is made before anything is timed, also
    # when the run has no warm-up steps. (Nothing to measure: a scene without a mesh, or a schedule fixed through the
    # environment, as tools/profile_bench.sh does so that the profiler's kernel statistics hold frame launches only.)
    fixed = os.environ.get("PTAMD_WALK_BUDGET") is not None or os.environ.get("PTAMD_WALK_DEFER") is not None
    if spp_count >= 16 and pt.Info()["meshes"] > 0 and not fixed:
        if mode == "tile":
            ptamd.Render(cam, pt, film, 64, depth, russian_roulette=rr, tiles=(32, rank, world))
        else:
            ptamd.Render(cam, pt, film, 64, depth, russian_roulette=rr)
    for _ in range(warmup):
        accum.zero_(); samples.zero_()
        launch()
        shard.reduce_film(accum, dst=0)
    rk.barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        accum.zero_(); samples.zero_()
        ev[i][0].record(stream)
        launch()
        ev[i][1].record(stream)
        shard.reduce_film(accum, dst=0)  # one RCCL reduce over xGMI per render (sample counts are known: spp_total)
    rk.barrier()
    elapsed = rk.max_over_ranks(time.perf_counter() - t0)
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if steps else 0.0
    if dump and rank == 0:
        np.save(dump, accum.cpu().numpy())

    # executed work of this rank's launch (one extra counting launch, outside the timed region)
    accum.zero_(); samples.zero_()
    launch(count=True)
    torch.cuda.synchronize()
    stats = ptamd.RenderStats()
    info = pt.Info()
    out = None
    if rank == 0:
        samples_per_step = float(W) * H * spp_total
        value = samples_per_step * steps / elapsed * 1e-6
        rays_per_sample = stats["rays"] / max(1, stats["samples"])
        sharding = ("interleaved 32x32 tiles per rank, one launch (ptamd_render_tiles)" if mode == "tile" else "spp-slice per rank") + \
                   " + one %s reduce of the %.1f MB accumulation buffer per step" % ("RCCL" if rk.backend == "nccl" else rk.backend, W * H * 12 / 1e6)
        out = {"metric": "Msamples/s", "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": steps,
               "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
               "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s, %d spp %s, %d bounces%s" % (name, spp_in, "per GPU" if scaling == "weak" else "in total", depth,
                                                                       ", Russian roulette" if rr else ""),
                          "baseline_config": config, "sharding": sharding, "seed": "0x5EED"},
               "ranks_seen": rk.ranks_seen, "backend": rk.backend if rk.distributed else None,
               "mrays_per_s": value * rays_per_sample, "rays_per_sample": rays_per_sample,
               "time_to_1024spp_s": (elapsed / steps) * (1024.0 / spp_total) if config == 2 else None,
               "kernel_ms_per_launch": kernel_ms}
        if world == 1:
            n_sph = sum(1 for o in scene["objects"] if o["type"] == "sphere")
            n_box = sum(1 for o in scene["objects"] if o["type"] == "box")
            # a visit of a two-child node tests two boxes and reads 64 B; a visit of an eight-child node tests eight
            # quantised boxes and reads 80 B (meshes of 200 000 faces and more in scenes that do not fit LDS)
            wide = info["wide_nodes"] > 0
            slabs, node_bytes = (8, B_WIDE_NODE) if wide else (2, B_NODE)
            flops = (slabs * F_SLAB * stats["nodes"] + F_TRI * stats["tris"] + (F_SPHERE * n_sph + F_BOX * n_box) * stats["rays"] + F_SHADE * stats["shade_lanes"])
            lds_resident = info["lds_resident"] != 0
            launched = float(W) * H * spp_count
            scale = launched / max(1, stats["samples"])
            traffic, traffic_file = measured_traffic(config, spp_count, depth)
            achieved = flops * scale / (kernel_ms * 1e-3) * 1e-12
            roof = {"bound": "valu", "achieved": achieved, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / FP32_VECTOR_PEAK_TFLOPS, "traffic": traffic,
                    "note": "fp32 vector ALU bound (no MFMA: not a dense contraction; scene geometry is read from %s, film traffic is 32 B "
                            "per pixel and launch): `achieved` prices the slab / triangle / sphere / box tests and shaded hits this "
                            "kernel executed (its own counters: 27 flop per box test, 47 per triangle test, 25 per sphere, 60 per "
                            "oriented box, 250 per shaded hit) over the launch time" % ("LDS" if lds_resident else "L2 / Infinity Cache / HBM"),
                    "kernel": ("render_samples_kernel (sample-major: 16 spp and more per launch)"
                               if 16 <= spp_count <= (1 << 20) and depth >= 2 and not os.environ.get("PTAMD_NO_SAMPLE_MAJOR")
                               else "render_kernel (pixel-major)"), "traffic_unit": "bytes per launch (memory-side counters, PMC passes committed under profiles/)",
                    "traffic_source": traffic_file, "executed_flops_per_sample": flops / max(1, stats["samples"]),
                    "executed_per_ray": {"nodes": stats["nodes"] / max(1, stats["rays"]), "tris": stats["tris"] / max(1, stats["rays"])},
                    "node_width": slabs,
                    "lane_utilisation": {"node_loop": stats["nodes"] / max(1.0, 64.0 * stats["node_trips"]),
                                         "triangle_loop": stats["tris"] / max(1.0, 64.0 * stats["tri_trips"]),
                                         "shading": stats["shade_lanes"] / max(1.0, 64.0 * stats["shade_trips"])}}
            if not lds_resident:
                # what the walk asks the memory hierarchy for; most of it is served by L2 and the Infinity Cache
                req = (node_bytes * stats["nodes"] + B_TRI * stats["tris"] + B_HIT * stats["shade_lanes"]) * scale + 32.0 * W * H
                roof["l2_request_bytes"] = req
                roof["l2_request_gbs"] = req / (kernel_ms * 1e-3) * 1e-9
                roof["l2_request_bytes_per_ray"] = req / max(1.0, stats["rays"] * scale)
            if traffic is not None:
                roof["hbm"] = {"achieved": traffic / (kernel_ms * 1e-3) * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": traffic / (kernel_ms * 1e-3) * 1e-9 / HBM_PEAK_GBS}
            out["roofline"] = roof
            base = cnt = None
            if want_cpu:
                try:
                    base, cnt = cpu_baseline(scene, depth, cpu_budget)
                except Exception as e:  # the GPU measurement stands without its CPU leg
                    print("bench: CPU baseline failed: %r" % (e,), file=sys.stderr)
                    out["cpu_baseline"] = {"error": repr(e)}
            if base is not None:
                ref_fps = reference_flops(cnt) / max(1, cnt["samples"])
                roof["reference_equivalent"] = {
                    "flops_per_sample": ref_fps, "tflops": ref_fps * launched / (kernel_ms * 1e-3) * 1e-12,
                    "note": "the same sample rate priced with the flops the REFERENCE's algorithm spends per sample on this scene "
                            "(CPU oracle counters: median-split tree, every child box tested twice); not a utilisation figure"}
                out["cpu_baseline"] = base
                if not rr:
                    out["speedup_vs_cpu_baseline"] = value / base["value"]
            elif not want_cpu:
                out["cpu_baseline"] = None
    film.close()
    pt.close()
    del accum, samples
    return out


def run_config(rk, config, steps, warmup, spp, depth, scaling, want_cpu, cpu_budget, dump=None):
    """One configuration's result. Configuration 5 (Russian roulette) is measured twice: with roulette (its `value`), and
    without, which is the workload the CPU reference runs (it has no roulette): the ratio divides like by like."""
    out = run_workload(rk, config, steps, warmup, spp, depth, scaling, want_cpu=want_cpu and config != 5, cpu_budget=cpu_budget, dump=dump)
    if config == 5 and rk.world == 1:
        plain = run_workload(rk, 5, max(1, min(steps, 2)), 1, spp, depth, scaling, want_cpu=want_cpu, cpu_budget=cpu_budget, roulette=False)
        if out is not None:
            out["no_roulette"] = {k: plain[k] for k in ("value", "unit", "mrays_per_s", "rays_per_sample", "kernel_ms_per_launch", "steps")}
            out["no_roulette"]["roofline"] = {k: plain["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "lane_utilisation")}
            out["cpu_baseline"] = plain.get("cpu_baseline")
            if (plain.get("cpu_baseline") or {}).get("value"):
                out["speedup_vs_cpu_baseline"] = plain["value"] / plain["cpu_baseline"]["value"]
                out["speedup_note"] = "no-roulette GPU rate over the CPU reference (which has no roulette): like by like"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=None, help="samples per pixel in total (strong, the default) or per GPU (weak); default per config")
    ap.add_argument("--depth", type=int, default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--config", type=int, default=None, choices=[2, 3, 4, 5],
                    help="BASELINE.json configuration: 2 (the headline) mesh Cornell; 3 1 M-triangle dragon field + sky, spp-sharded; "
                         "4 4K mixed scene, tile-sharded; 5 glass/mirror, 32 bounces, Russian roulette. Default: 2, plus short runs "
                         "of 3, 4 and 5 nested under other_configs when N = 1")
    ap.add_argument("--no-others", action="store_true", help="headline only: no other_configs")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default): --spp in total, split over the ranks, so every N reports time_to_1024spp_s; weak: --spp per GPU")
    args = ap.parse_args()
    if args.steps < 1:
        raise SystemExit("--steps must be at least 1")

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    os.chdir(ROOT)
    rk = Ranks(args)
    import ptamd
    ptamd.init(rk.device)
    ptamd.set_stream(rk.torch.cuda.current_stream().cuda_stream)  # launches, torch events and the collective share one stream
    config = args.config if args.config is not None else 2
    out = run_config(rk, config, args.steps, args.warmup, args.spp, args.depth, args.scaling, not args.no_cpu, 20.0,
                     dump=os.environ.get("PTAMD_BENCH_DUMP"))
    if rk.rank == 0:
        if args.config is None and rk.world == 1 and not args.no_others and args.spp is None and args.depth is None:
            others = {}
            for c in (3, 4, 5):
                try:
                    r = run_config(rk, c, 2, 1, None, None, "strong", not args.no_cpu, 8.0)
                    keep = ("value", "unit", "ms_per_step", "steps", "warmup", "mrays_per_s", "rays_per_sample", "kernel_ms_per_launch", "config",
                            "roofline", "cpu_baseline", "speedup_vs_cpu_baseline", "speedup_note", "no_roulette")
                    others["config%d" % c] = {k: r[k] for k in keep if k in r}
                except Exception as e:  # the headline line must come out whatever happens to a side measurement
                    print("bench: configuration %d failed: %r" % (c, e), file=sys.stderr)
                    others["config%d" % c] = {"error": repr(e)}
            out["other_configs"] = others
        print(json.dumps(out), flush=True)
    rk.close()


if __name__ == "__main__":
    main()
