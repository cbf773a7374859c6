#!/usr/bin/env python3
"""bench.py -- headline metric of BASELINE.json: Mrays/s at 1920x1080, 8 spp, 4 bounces.

One "step" = one full render of the workload (accumulation reset, `spp` frames of the path-tracing
kernels + accumulation, and -- for N > 1 -- the RCCL gather of every rank's finished float4 rows to rank 0
and their assembly).  The K timed steps are queued with jpt_render_async and bracketed by barrier +
torch.cuda.synchronize() on both sides; the library overlaps the launches of consecutive queued renders, so
ms_per_step is the queued rate (the device time of one render on its own is reported as roofline.render_ms).
Scene, camera and RNG streams are synthetic and deterministic (gdpathtracing_amd/scenes.py); the scene is
resident in HBM before the timed region starts.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = ray segments actually traced by all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), dealt in the
# order the streams are first used; streams that share a queue run in submission order.  The library keeps four
# renders in flight on four streams plus the context's stream, so it wants more than four queues, and it wants to
# stay off the LAST queue of the pool (a render stream that lands there costs 23 %: 1.58 instead of 1.28 ms on C3,
# whatever the pool size -- DESIGN.md section 4).  Sixteen leaves room for the streams RCCL and torch use first.
# Read at runtime start-up: set before torch is imported.
if os.environ.get("JPT_BENCH_BACKEND", "nccl") == "nccl":   # (the gloo rehearsal puts several ranks on ONE GPU: their queues add up)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# bytes one event touches in the flattened layout (DESIGN.md "Algorithmic bytes")
WIDE_BYTES = dict(blas_expand=64, tri_tests=48, tlas_expand=64, inst_visits=64, rays=32 + 20)
# native builder: four-child 128-byte records
WIDE4_BYTES = dict(blas_expand=128, tri_tests=48, tlas_expand=128, inst_visits=64, rays=32 + 20)
# the same events priced in the reference layout
REF_BYTES = dict(blas_expand=96, tri_tests=48, tlas_expand=64, inst_visits=224, shaded_hits=320)


def algorithmic_bytes(stats, table):
    """Bytes the traversal kernel asks for, from exact event counts: one 64-B record per BLAS/TLAS expansion,
    48 B per triangle test, 64 B per instance visit, a 32-B ray in and a 20-B hit out per ray."""
    return sum(stats[k] * v for k, v in table.items())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=8)
    ap.add_argument("--bounces", type=int, default=4)
    ap.add_argument("--tris", type=int, default=51200)
    ap.add_argument("--scene", default="demo", choices=["demo", "cornell", "inst"])
    ap.add_argument("--builder", default="sah", choices=["sah", "exact", "watertight"])
    ap.add_argument("--accum", default="ldr8", choices=["ldr8", "hdr"])
    ap.add_argument("--kernel", default="wavefront", choices=["wavefront", "ref"])
    ap.add_argument("--camera", default="demo", choices=["demo", "closeup"],
                    help="demo = demo.tscn's camera (the box covers ~1/6 of the frame); closeup = camera at the box opening "
                         "(every pixel sees geometry; not the headline config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region, rank 0 re-renders the whole image alone and checks the assembled one bit for bit")
    ap.add_argument("--gather", choices=["ldr", "accum"], default="accum",
                    help="N > 1: what crosses the links each render -- the float4 accumulation rows (16 B/pixel: the exchange "
                         "BASELINE.json's north_star names, the default) or only the rgba8 display rows (4 B/pixel, what the "
                         "reference reads back)")
    ap.add_argument("--cpu-sample", default="auto")
    ap.add_argument("--pmc-json", default=os.path.join(ROOT, "profiles", "current_pmc.json"),
                    help="per-launch HBM bytes of each kernel from the committed rocprofv3 --pmc passes (tools/pmc.sh)")
    args = ap.parse_args()

    import torch
    from gdpathtracing_amd import capi, host, partition, scenes

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # JPT_BENCH_BACKEND=gloo is a functional rehearsal of the N > 1 flow on a box with fewer GPUs than ranks
    # (ranks share devices, the gather is staged through host memory); real runs use RCCL ("nccl").
    backend = os.environ.get("JPT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    if args.scene == "demo":
        sc = scenes.demo_scene(args.tris)
    elif args.scene == "cornell":
        sc = scenes.cornell_scene()
    else:
        sc = scenes.instanced_scene()
    if args.camera == "closeup":
        sc.camera = scenes.CameraDesc(scenes.transform12(None, (0.0, 0.0, 4.2)), fov_deg=75.0)
    W, H, spp, bounces = args.width, args.height, args.spp, args.bounces
    cam = scenes.camera_block(sc.camera, W, H)
    accum_mode = capi.ACCUM_REF_LDR8 if args.accum == "ldr8" else capi.ACCUM_HDR_F32

    ctx = host.Context(local_rank)
    t0 = time.time()
    ctx.build_scene(sc, {"sah": capi.BUILD_SAH, "exact": capi.BUILD_REFERENCE_EXACT, "watertight": capi.BUILD_SAH_WATERTIGHT}[args.builder])
    build_s = time.time() - t0
    ctx.set_partition(rank, world)
    ctx.set_params(W, H, bounces, accum_mode)
    ctx.set_camera(cam)
    ctx.set_kernel({"wavefront": capi.KERNEL_WAVEFRONT, "ref": capi.KERNEL_REFERENCE_LAYOUT}[args.kernel])
    # One explicit (non-null) stream carries the renders, the timing events and the collective: the context's own,
    # made torch's current stream.  (Not a stream from torch's pool: the pool is 32 streams created at once, the HIP
    # runtime deals streams to its hardware queues round-robin in creation order, and the library's slot streams,
    # created next, would land on the very queue the pool stream sits on -- one frame per render 0.40 instead of 0.31 ms.)
    stream = torch.cuda.ExternalStream(ctx.get_stream(), device=torch.device("cuda", local_rank))
    torch.cuda.set_stream(stream)
    ctx.set_kernel_timing(False)   # per-launch HIP events serialise the frame groups: they are collected after the timed region

    # gather plumbing (N > 1): the local piece (rgba8 display rows, or float4 sums) viewed as a torch tensor, no copy
    piece = gathered = None
    if world > 1:
        ptr, nbytes = ctx.device_ldr() if args.gather == "ldr" else ctx.device_accum()
        typestr, tdtype = ("<i4", torch.int32) if args.gather == "ldr" else ("<f4", torch.float32)

        class _View:
            __cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": typestr, "data": (ptr, False), "version": 2}

        piece = torch.as_tensor(_View(), device=torch.device("cuda", local_rank))
        if rank == 0:
            gathered = torch.empty((world, nbytes // 4), dtype=tdtype, device=piece.device)

    def exchange():
        """the one exchange of a render: every rank's float4 piece to rank 0, then assembly on rank 0"""
        if backend == "nccl":
            # direct point-to-point gather: every peer sends its piece over its own xGMI link
            partition.gather_to_rank0(piece, dist, rank, world, gathered)
        else:
            stream.synchronize()
            g = partition.gather_to_rank0(piece.cpu(), dist, rank, world)
            if rank == 0:
                gathered.copy_(g)
        if rank == 0:
            if args.gather == "ldr":
                ctx.assemble_ldr_from_ranks(gathered.data_ptr(), world)
            else:
                ctx.assemble_from_ranks(gathered.data_ptr(), world)

    def step():
        ctx.accum_reset()
        ctx.render(spp, 1, asynchronous=True)
        if world > 1:
            exchange()

    # exact event counts of one step (deterministic), outside the timed region
    ctx.accum_reset()
    ctx.render(spp, 1, counted=True)
    st = ctx.stats()
    counts = torch.tensor([st[k] for k in ("rays", "blas_expand", "tri_tests", "tlas_expand", "inst_visits", "shaded_hits")],
                          dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(counts)
    rays, blas_expand, tri_tests, tlas_expand, inst_visits, shaded_hits = [int(x) for x in counts.tolist()]
    total = dict(rays=rays, blas_expand=blas_expand, tri_tests=tri_tests, tlas_expand=tlas_expand,
                 inst_visits=inst_visits, shaded_hits=shaded_hits)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, like the allocations above: the library creates its pipeline slots (a stream and a workspace each) at
    # the first queued render -- ~25 ms once per context, which is not a property of a step (--warmup 0 is honoured
    # as "no warm-up steps", not as "time the resource creation").
    step()
    barrier()
    for _ in range(args.warmup):
        step()
    barrier()
    # Timed region.  Steps are queued back to back and the closing barrier waits for all of them: the library runs
    # the path kernels of consecutive asynchronous renders on its pipeline slots' streams (four workspaces), so one
    # render's launch tails overlap the next renders' kernels; the accumulation kernels run in call order, chained
    # through the context's stream.  (A queue of K renders takes K x rate + the latency of the last one, ~3 ms: small
    # K reads lower.)
    t0 = time.perf_counter()
    for i in range(args.steps):
        ctx.accum_reset()
        ctx.render(spp, 1, asynchronous=True)
        if world > 1:
            exchange()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    verified = None
    if args.verify and rank == 0:
        got = ctx.read_ldr() if (world > 1 and args.gather == "ldr") else ctx.read_accum()
        solo = host.Context(local_rank)
        solo.build_scene(sc, {"sah": capi.BUILD_SAH, "exact": capi.BUILD_REFERENCE_EXACT, "watertight": capi.BUILD_SAH_WATERTIGHT}[args.builder])
        solo.set_params(W, H, bounces, accum_mode)
        solo.set_camera(cam)
        solo.render(spp, 1)
        verified = bool(np.array_equal(got, solo.read_ldr() if (world > 1 and args.gather == "ldr") else solo.read_accum()))
        solo.close()

    # Duration of the dominant kernel, measured live with HIP events on the stream it is launched on: the library
    # brackets every traversal launch when kernel timing is on, which also makes it run the launches one after
    # another (in the timed region above, the frame groups' launches overlap, so a per-launch duration is not
    # defined there).  Three untimed renders, after the timed region.
    ctx.set_kernel_timing(True)
    trace_ms, render_ms = [], []
    for _ in range(3):
        ctx.accum_reset()
        ctx.render(spp, 1)
        st = ctx.stats()
        trace_ms.append(st["last_trace_ms"])
        render_ms.append(st["last_render_ms"])   # device time of ONE render on its own (no overlap with a neighbour)
    ctx.set_kernel_timing(False)
    n_trace_launches = (bounces + 1) if args.kernel != "ref" else spp
    kernel_ms = float(np.mean(trace_ms)) / n_trace_launches   # average duration of ONE launch of the dominant kernel

    if rank == 0:
        n_pixels = W * H
        ms_per_step = elapsed / args.steps * 1e3
        mrays = rays * args.steps / elapsed / 1e6
        # roofline of the dominant kernel (this rank's share of the events ~ total / world)
        table = WIDE4_BYTES if (args.builder != "exact" and args.kernel == "wavefront") else WIDE_BYTES
        alg = algorithmic_bytes(total, table) / world / n_trace_launches               # per launch
        alg_ref = (algorithmic_bytes(total, REF_BYTES) + n_pixels * spp * 48) / world / n_trace_launches
        achieved = alg / (kernel_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process, so the
        # per-launch figure comes from the committed rocprofv3 passes of this same command (profiles/)
        traffic, traffic_src = None, None
        dom = "wf2_trace" if args.kernel == "wavefront" else "ref_frame_kernel"
        default_run = (W, H, spp, bounces, args.tris, args.scene, args.builder, world, args.camera) == (1920, 1080, 8, 4, 51200, "demo", "sah", 1, "demo")
        if default_run and os.path.exists(args.pmc_json):
            try:
                pj = json.load(open(args.pmc_json))
                doms = ["wf2_primary", "wf2_trace"] if args.kernel == "wavefront" else [dom]
                traffic = int(sum(pj[k]["hbm_bytes_per_launch"] * pj[k]["launches"] for k in doms) /
                              sum(pj[k]["launches"] for k in doms))
                traffic_src = os.path.relpath(args.pmc_json, ROOT) + ": (2*FETCH_SIZE + WRITE_SIZE) KiB per launch, separate --pmc passes"
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/sec at 1920x1080, 8 spp, 4 bounces",
            "value": round(mrays, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "C3: S-demo (open Cornell cube + light + 2 instances of a %d-tri procedural mesh), %dx%d, %d spp, "
                            "%d bounces, accum=%s, builder=%s, kernel=%s" % (args.tris, W, H, spp, bounces, args.accum, args.builder, args.kernel)
                if (args.scene == "demo" and args.camera == "demo") else "%s (camera %s) %dx%d %d spp %d bounces" % (sc.name, args.camera, W, H, spp, bounces),
                "unique_tris": sc.n_unique_tris, "instances": len(sc.instances),
                "rays_per_step": rays, "nominal_rays_per_step": n_pixels * spp * (bounces + 1),
                "parallelism": "screen strips x%d" % world + ("" if world == 1 else ", gather of %s rows to rank 0" % ("rgba8 display" if args.gather == "ldr" else "float4 accumulation")),
                "scene_build_s": round(build_s, 4),
            },
            "roofline": {
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "wf2_primary + wf2_trace (one launch per bounce, %d per render)" % n_trace_launches if args.kernel != "ref"
                else "ref_frame_kernel (one launch per frame)",
                "kernel_ms": round(kernel_ms, 4),
                "render_ms": round(float(np.mean(render_ms)), 4),
                "render_ms_note": "one render alone, launches serialised (kernel timing on); ms_per_step is the pipelined rate",
                "algorithmic_bytes": int(alg),
                # what a render must move through HBM whatever the caches do -- the framebuffers it leaves
                # behind (float4 sums, rgba8 screen, r32f depth), per render, not per launch
                "compulsory_framebuffer_bytes_per_render": int(n_pixels * (16 + 4 + 4)),
                "ref_layout_algorithmic_bytes": int(alg_ref),
                "ref_layout_achieved": round(alg_ref / (kernel_ms * 1e-3) / 1e9, 2),
                "note": "scene is L2/Infinity-Cache resident; bytes are cache-served requests, not HBM traffic. The kernels are "
                        "VALU-issue-bound: 7.6e8 wave-level VALU instructions per C3 render x 4 cycles / (1024 SIMDs x 2.4 GHz) = "
                        "1.24 ms, against which ms_per_step (renders queued, four in flight) is to be read; lane utilisation of "
                        "wf2_trace is 30 % (rocprofv3 SQ counters, profiles/r01/r01h_sq_counters.txt, DESIGN.md section 4)",
            },
            "counters": total,
        }
        if verified is not None:
            out["verified_bit_identical_to_one_context"] = verified
        if world == 1 and not args.no_cpu_baseline:
            from oracle import binding as ob
            t0 = time.perf_counter()
            ref = ob.build_scene(sc)   # the reference's BVH / TLAS builder restated (src/bvh/bvh.cpp), one thread
            cpu_build_s = time.perf_counter() - t0
            # probe the host's rate on a small image, then size the sample for ~5-20 s of wall time
            # (the whole workload if the host is fast enough)
            pw, ph = 240, 135
            t0 = time.perf_counter()
            _, _, _, pc, used = ob.render(ref, scenes.camera_block(sc.camera, pw, ph), pw, ph, bounces, spp, 1, accum_mode)
            probe_rate = pc["rays"] / max(time.perf_counter() - t0, 1e-6)
            if args.cpu_sample != "auto":
                sw, sh = (int(x) for x in args.cpu_sample.split("x"))
            else:
                scale = min(1.0, (probe_rate * 15.0 / max(rays, 1)) ** 0.5)
                sw, sh = max(16, int(W * scale) // 16 * 16), max(9, int(H * scale) // 9 * 9)
                if scale >= 1.0:
                    sw, sh = W, H
            scam = scenes.camera_block(sc.camera, sw, sh)
            t0 = time.perf_counter()
            _, _, _, cnt, used = ob.render(ref, scam, sw, sh, bounces, spp, 1, accum_mode)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {
                "value": round(cnt["rays"] / dt / 1e6, 3), "unit": "Mrays/s", "cores": used, "kind": "port",
                "bvh_build_s": round(cpu_build_s, 4), "bvh_build_cores": 1,
                "sample": "same scene/camera/seeds at %dx%d, %d spp, %d bounces (%.1f s, %d rays); oracle = C restatement "
                          "of main.glsl over the reference-layout BVH, pthreads" % (sw, sh, spp, bounces, dt, cnt["rays"]),
            }
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
