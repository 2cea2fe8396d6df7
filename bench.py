#!/usr/bin/env python3
"""Benchmark of the wavefront integrator, one JSON line per run (contract in the task description).

    python bench.py [--config cornell|mesh82k|atrium|mesh328k|mixed|spheres] [--gpus N] [--steps K] [--warmup W] [--spp S]

Default: BASELINE.json configs[1], the configuration the metric is quoted on — the Cornell-class scene at 1920x1080, 256 spp,
depth 8; one "step" = one complete render of the frame with the scene already resident in HBM.  The other names are the other
BASELINE.json configurations (parity-test cases, selectable here so that every number in DESIGN.md has a one-line command):
    mesh82k   configs[2]  81 932 triangles (displaced icosphere in the Cornell room), 1920x1080, 512 spp, depth 8
    atrium    configs[3]  the instanced atrium SURVEY 8(d) fixes for it: 293 BLASes, ~590 TLAS leaves, ~250 k instanced triangles, 1920x1080, 1024 spp, depth 8
    mesh328k  configs[3]  (class stand-in of rounds 1-3) 327 692 triangles in ONE displaced icosphere, 1920x1080, 1024 spp, depth 8
    mixed     configs[4]  diffuse + dielectric + GGX metal Cornell boxes, 4096x4096, 4096 spp, depth 16 (hundreds of batches on two pipelines)
    spheres   configs[4]  three 5 120-triangle spheres + an instanced mirror sphere, 4096x4096, depth 16; 1024 of the 4096 spp by default (~60 s rule)
N > 1 is launched by the driver through torch.distributed.run, one rank per GPU: rows are dealt to ranks in strips (no
data-path collective while rendering) and each step ends with one RCCL gather of the framebuffer to rank 0.

`value` = rays that ENTER a traversal kernel per second (whole job).  The reference's call-site count (integrator.rs:179,56,100,103;
SURVEY 8d: what the CPU baseline counts) is `config.cast_Mray_per_s`; the difference is camera rays answered by the projection of the
world's root box and BSDF-sampled NEE rays answered by the lights' root-box test in the shading pass (`config.rays_by_class`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_CLOSEST_RAY = 48      # 32 B ray in + 16 B hit out; + 32 B per node visited + 48 B per triangle tested when the BVH is not LDS-resident
PROFILE_ROUND = "r04"           # profiles/<round>_<config>_traffic.json: the committed rocprofv3 PMC passes the counter-derived fields come from

# name -> scene function, keyword arguments, width, height, the configuration's own spp, spp of a default run, depth, workload description,
#         spp of the CPU baseline's FIXED sample (the whole frame at that many samples: the same work on every box and in every round)
CONFIGS = {
    "cornell": ("cornell_box", {}, 1920, 1080, 256, 256, 8, "Cornell box (36 triangles, 6 BLAS)", 64),
    "mesh82k": ("cornell_mesh", {"level": 6}, 1920, 1080, 512, 512, 8, "Cornell room + 81 920-triangle displaced icosphere (BASELINE configs[2] class)", 16),
    "atrium": ("atrium", {}, 1920, 1080, 1024, 1024, 8, "instanced atrium: 293 BLASes, ~590 TLAS leaves, ~250 k instanced triangles (BASELINE configs[3] as SURVEY 8d defines it)", 8),
    "mesh328k": ("cornell_mesh", {"level": 7}, 1920, 1080, 1024, 1024, 8, "Cornell room + 327 680-triangle displaced icosphere (BASELINE configs[3] class)", 8),
    "mixed": ("cornell_mixed", {}, 4096, 4096, 4096, 4096, 16, "Cornell box, tall box GGX metal, short box dielectric (BASELINE configs[4])", 8),
    "spheres": ("cornell_spheres", {}, 4096, 4096, 4096, 1024, 16, "Cornell room + diffuse / glass / GGX spheres + instanced mirror sphere (BASELINE configs[4] class)", 4),
}


def make_scene(cfg):
    from path_tracer_amd import scenes
    fn, kw, w, h = cfg[0], cfg[1], cfg[2], cfg[3]
    return getattr(scenes, fn)(w, h, **kw)


def cpu_baseline(cfg, depth, n_sobol):
    """The CPU restatement of the reference algorithm (oracle/, kind "port") timed on this host's cores on a FIXED, bounded sample of the
    same workload: the whole frame at CONFIGS[..][8] samples per pixel (Cornell: 64 of the 256 spp, ~10 s on the GPU box's host) — the
    same rays in every round and on every box, so the number is comparable (rounds 1-3 filled a time budget: 96 / 107 / 170 spp).  Its
    counters 6 and 7 (BVH nodes visited / triangles tested by the world closest-hit casts, on rays statistically identical to the timed
    ones) give SURVEY 8(d)'s algorithmic bytes per ray."""
    from oracle import oracle as O
    w, h, spp_full, spp = cfg[2], cfg[3], cfg[4], cfg[8]
    o = O.Oracle(make_scene(cfg))
    threads = max(1, (os.cpu_count() or 2) - 1)          # num_cpus::get() - 1, src/main.rs:72
    o.render(w, h, 1, first_sample=spp, max_bounces=depth, n_sobol=n_sobol, threads=threads)   # page the scene in, start the pool; not timed
    t0 = time.perf_counter()
    _, _, _, ctr = o.render(w, h, spp, max_bounces=depth, n_sobol=n_sobol, threads=threads)
    dt = time.perf_counter() - t0
    rays = int(ctr[0] + ctr[1] + ctr[2])
    closest = max(int(ctr[0]), 1)
    return {"value": rays / dt / 1e6, "unit": "Mray/s", "cores": threads, "kind": "port",
            "sample": f"{cfg[7]} {w}x{h}, samples 0..{spp - 1} of {spp_full} (fixed), depth {depth}, {threads} threads, {dt:.1f} s; rays = casts at the reference's call sites "
                      f"(compare with config.cast_Mray_per_s)",
            "nodes_visited_per_closest_ray": int(ctr[6]) / closest, "triangles_tested_per_closest_ray": int(ctr[7]) / closest}


def _profile(config):
    """Counter-derived fields from the committed rocprofv3 PMC passes of THIS round for this configuration (explicit file name, not
    "the newest file").  bench.py cannot collect PMC counters on itself: per-ray figures of the same kernels on the same scene are
    scaled by this run's counts, and the provenance is stated in the record."""
    try:
        name = f"{PROFILE_ROUND}_{config}_traffic.json"
        return name, json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(CONFIGS), default="cornell")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--batch-spp", type=int, default=0)
    ap.add_argument("--pipelines", type=int, default=0, help="wavefront pipelines (0: the library's choice; 1: one batch after another, what the profile passes use so that kernel durations do not overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-ms", action="store_true", help="skip the extra untimed step that times every kernel category (profiling runs: counters then cover the timed steps only)")
    ap.add_argument("--time-all-kernels", action="store_true", help="HIP events around every launch INSIDE the timed region (diagnostic, slower)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    width, height, spp_own, spp_default, depth, what = cfg[2], cfg[3], cfg[4], cfg[5], cfg[6], cfg[7]
    spp = args.spp if args.spp else spp_default
    big = width * height * spp > 1920 * 1080 * 2048       # many seconds per step: one timed step, no warm-up, unless asked otherwise
    steps = args.steps if args.steps is not None else (1 if big else 2)
    warmup = args.warmup if args.warmup is not None else (0 if big else 1)
    n_sobol = 512 if spp <= 512 else 1 << (2 * spp - 1).bit_length()   # SobolSampler<N> table (main.rs:48): any N works, >= 2 x spp as in the parity tests

    import torch
    import torch.distributed as dist
    from path_tracer_amd import api
    from path_tracer_amd import dist as ptdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PT_BENCH_REHEARSAL=1: every rank on cuda:0 and gloo for the collectives (framebuffer staged through the host) — the N > 1 control
    # flow of this file on a one-GPU box; RCCL refuses two ranks on one device.  Not a measurement.
    rehearsal = world > 1 and os.environ.get("PT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
        dist.init_process_group(backend="gloo")
    elif world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)
    red_dev = torch.device("cpu") if rehearsal else dev   # where the scalars of the final reductions live

    strip = 4
    r = api.Renderer(make_scene(cfg), width, height, max_bounces=depth, n_sobol=n_sobol, rank=rank, world_size=world, strip_rows=strip,
                     batch_spp=args.batch_spp, pipelines=args.pipelines, device=dev.index, flags=api.FLAG_TIMING | (api.FLAG_TIMING_ALL if args.time_all_kernels else 0))
    stream = torch.cuda.current_stream(dev)
    r.set_stream(stream.cuda_stream)
    n_rows = len(r.local_rows())

    def step():
        r.reset_accumulation()
        r.render_device(0, spp)
        if world > 1:
            ptr, _ = r.accum_device_ptr()
            fb = ptdist.wrap_device_framebuffer(ptr, n_rows, width, dev)
            if rehearsal:
                torch.cuda.synchronize(dev)
                fb = fb.cpu()
            ptdist.gather_framebuffer(fb, height, width, rank, world, strip, dst=0)

    if warmup == 0 and big:
        # the wavefront buffers are sized and allocated by the first render (hipMalloc of ~240 GiB takes seconds): do that outside the timed
        # region with a request large enough to be cut into the same full-size batches on two pipelines (256 spp of a frame this large), not
        # with a whole step
        r.render_device(0, min(spp, 256))
        torch.cuda.synchronize(dev)
    for _ in range(warmup):
        step()
    r.reset_stats()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    st = r.stats()
    traced_closest_local = st.rays_closest - st.rays_primary_culled       # camera rays answered by the projection never reach the kernel
    traversed_local = traced_closest_local + st.rays_any + st.rays_light_closest_traced
    vals = torch.tensor([dt, float(st.rays), float(st.paths), float(traversed_local)], dtype=torch.float64, device=red_dev)
    if world > 1:
        tmax = vals[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        sums = vals[1:].clone()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dt, rays, paths, traversed = float(tmax[0]), float(sums[0]), float(sums[1]), float(sums[2])
    else:
        rays, paths, traversed = float(st.rays), float(st.paths), float(traversed_local)

    kernel_ms = None
    s1 = None
    if rank == 0 and world == 1 and not args.no_kernel_ms:
        spp_k = spp if not big else max(1, spp // 16)
        # (a) the dominant kernel's launch durations: one extra, untimed render of the SAME launch structure as the timed steps (the fused
        # world + NEE launch on LDS scenes) with its batches one after another on ONE pipeline — with two pipelines an event pair on one
        # launch stream also spans the other pipeline's kernels.  This is what `rocprofv3 --kernel-trace` of `bench.py --pipelines 1` sees.
        r.set_config(flags=api.FLAG_TIMING, pipelines=1)
        r.reset_stats()
        r.reset_accumulation()
        r.render_device(0, spp_k)
        torch.cuda.synchronize(dev)
        s1 = r.stats()
        # (b) every kernel category timed, in another untimed render (HIP events around every launch serialise the side stream and add
        # idle time, and the world and NEE launches of a bounce are then separate launches); long workloads time a 1/16 sample of the spp
        r.set_config(flags=api.FLAG_TIMING_ALL, pipelines=1)
        r.reset_stats()
        r.reset_accumulation()
        r.render_device(0, spp_k)
        torch.cuda.synchronize(dev)
        s2 = r.stats()
        f = spp / spp_k
        kernel_ms = {"trace_closest": s2.ms_trace_closest * f, "trace_any": s2.ms_trace_any * f, "trace_light": s2.ms_trace_light * f, "shade": s2.ms_shade * f,
                     "generate": s2.ms_generate * f, "accumulate": s2.ms_accumulate * f,
                     "source": "one extra untimed render with HIP events around every launch (PT_FLAG_TIMING_ALL; world and NEE launches separate), batches one after another on "
                               "one pipeline (the timed steps overlap the batches of two pipelines, so these can add up to more than ms_per_step); per step"
                               + ("" if spp_k == spp else f", measured on {spp_k} spp and scaled to {spp}")}

    if rank == 0:
        launches = max(1, st.launches_trace_closest)
        avg_ms_timed = st.ms_trace_closest / launches
        # Launch duration of the dominant kernel: HIP events on its launch stream in the timed region.  When the timed steps run their batches on
        # TWO pipelines (requests that do not fit at once; BVHs in global memory), an event pair also spans the other pipeline's kernels, so the
        # one-pipeline render behind kernel_ms (events around every launch, nothing else on the device) is the clean measurement and is used
        # when it exists; on one pipeline (the headline) the two agree.
        if s1 is not None and s1.launches_trace_closest:
            avg_ms = s1.ms_trace_closest / s1.launches_trace_closest
            rays_per_launch_meas = (s1.rays_closest - s1.rays_primary_culled) / s1.launches_trace_closest
            launches_meas = int(s1.launches_trace_closest)
            launch_src = "HIP events on the launch stream around the world closest-hit launches of one extra render on ONE pipeline (same launch structure as the timed steps)"
        else:
            avg_ms = avg_ms_timed
            rays_per_launch_meas = None
            launches_meas = int(launches)
            launch_src = "HIP events on the launch stream in the timed region"
        lds_scene = bool(st.lds_scene)
        out = {
            "metric": f"Mray/s at {width}x{height}, {spp} spp; achieved HBM GB/s in traversal kernel",
            "value": traversed / dt / 1e6, "unit": "Mray/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU, gloo; not a measurement)",
            "config": {"workload": f"{what} {width}x{height}, {spp} spp, depth {depth}, NEE+MIS", "name": args.config,
                       "spp_of_configuration": spp_own,
                       "parallelism": f"rows/{world}", "mpaths_per_s": paths / dt / 1e6, "rays_per_path": rays / max(paths, 1.0),
                       # `value` counts rays that enter a traversal kernel.  The reference's call sites (integrator.rs:179,56,100,103)
                       # cast more: camera rays outside the projection of the world's root box and BSDF-sampled NEE rays that miss the lights' root
                       # box are answered without a traversal.  Per step, rank 0:
                       "cast_Mray_per_s": rays / dt / 1e6,
                       # whole job (all ranks), per step: exact integers, independent of the GPU count (the RNG is keyed by the global pixel)
                       "job_per_step": {"casts": int(rays) // steps, "traversed": int(traversed) // steps, "paths": int(paths) // steps},
                       "rays_by_class": {"closest_traversed": traced_closest_local // steps,
                                         "closest_camera_rays_culled_by_projection": st.rays_primary_culled // steps,
                                         "any_traversed": st.rays_any // steps,
                                         "light_closest_traversed": st.rays_light_closest_traced // steps,
                                         "light_closest_culled_in_shade": (st.rays_light_closest - st.rays_light_closest_traced) // steps},
                       "bvh_in_lds": lds_scene, "scene_bytes": int(st.scene_bytes), "state_GiB": st.state_bytes / 2 ** 30},
            "kernel_ms": kernel_ms,
        }
        cb = None
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(cfg, depth, n_sobol)
            out["cpu_baseline"] = cb
        # ---- roofline of the dominant kernel (world closest hit), SURVEY 8(d): algorithmic bytes per launch / launch duration
        pname, prof = _profile(args.config)
        kc = (prof or {}).get("k_closest_world") or {}
        if lds_scene:
            alg_per_ray, alg_src = float(BYTES_PER_CLOSEST_RAY), "48 B per closest-hit ray (32 in + 16 out; the BVH never leaves the chip)"
        elif cb is not None:
            # the oracle's averages are per CAST; the casts answered by the root-box projection visit no node, so per ray that enters the kernel:
            per_cast = 32.0 * cb["nodes_visited_per_closest_ray"] + 48.0 * cb["triangles_tested_per_closest_ray"]
            alg_per_ray = BYTES_PER_CLOSEST_RAY + per_cast * st.rays_closest / max(traced_closest_local, 1)
            alg_src = ("48 + (32 per node visited + 48 per triangle tested) per ray that enters the kernel; nodes / triangles per cast from the oracle's counters 6, 7 on this "
                       "run's cpu_baseline sample, x casts / traversed rays")
        elif kc.get("algorithmic_bytes_per_ray"):
            alg_per_ray, alg_src = float(kc["algorithmic_bytes_per_ray"]), f"profiles/{pname} (oracle counters 6, 7 at profiling time)"
        else:
            alg_per_ray, alg_src = float(BYTES_PER_CLOSEST_RAY), "48 B per ray only: no oracle counters in this run (--no-cpu-baseline) and no committed profile"
        rays_per_launch = rays_per_launch_meas if rays_per_launch_meas is not None else traced_closest_local / launches
        achieved = alg_per_ray * rays_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        waves = kc.get("waves_per_simd")
        va, lanes = kc.get("valu_active_frac"), kc.get("lanes_per_valu_instr")
        valu_busy = va * waves if (va is not None and waves) else None
        binding = "valu-issue" if lds_scene else "valu-issue at low lane agreement, then L2 latency"
        roof = {"bound": binding, "axis": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": ("k_closest<LDS, PRIMARY> (bounce 0) + k_trace_fused<LDS> (later bounces: the world closest-hit rays, then the few BSDF-sampled NEE rays of the bounce before)"
                           if lds_scene else "k_closest<global BVH, PRIMARY> (bounce 0) + k_closest<global BVH, WORLD> (later bounces)"),
                "avg_launch_ms": avg_ms, "launches": launches_meas, "launch_time_source": launch_src,
                "avg_launch_ms_timed_region": avg_ms_timed, "launches_timed_region": int(launches),
                "algorithmic_bytes_per_ray": alg_per_ray, "algorithmic_bytes_source": alg_src, "rays_per_launch": rays_per_launch,
                "closest_Mray_per_s_in_kernel": rays_per_launch / max(avg_ms, 1e-9) / 1e3,
                # what actually binds the kernel (profiles/r03_*_summary.md): VALU issue.  valu_busy = share of the SIMD's issue cycles that carry a VALU
                # instruction (VALU-active share of a wave's lifetime x resident waves per SIMD); effective_valu_frac = valu_busy x lanes / 64
                "binding_resource": binding,
                "valu_active_frac": va, "waves_per_simd": waves, "lanes_per_valu_instr": lanes, "valu_busy": valu_busy,
                "effective_valu_frac": (valu_busy * lanes / 64.0) if (valu_busy is not None and lanes) else None,
                "note": "`bound` names what binds the kernel (counters: profiles/); `achieved`/`peak`/`frac` are SURVEY 8(d)'s ALGORITHMIC bytes per launch "
                        "(`algorithmic_bytes_per_ray` x `rays_per_launch`) over `avg_launch_ms` on the HBM axis (`axis`), not bandwidth in use; `launches` x `rays_per_launch` "
                        "are the extra render's own; counter traffic is `traffic` per launch, `counter_GBps` per second, `counter_over_algorithmic` their ratio"}
        if kc.get("hbm_bytes_per_ray") is not None:
            roof["traffic"] = kc["hbm_bytes_per_ray"] * rays_per_launch
            roof["counter_GBps"] = roof["traffic"] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else None
            roof["counter_over_algorithmic"] = kc["hbm_bytes_per_ray"] / alg_per_ray
            roof["counters_source"] = (f"profiles/{pname} (rocprofv3 --pmc passes of `bench.py --config {args.config} --steps 1 --spp {kc.get('spp')}`, committed with this round): "
                                       f"{kc['hbm_bytes_per_ray']:.1f} B/ray over {kc.get('rays', 0) / 1e6:.0f} M rays, scaled by this run's rays per launch")
        out["roofline"] = roof
        # ---- the memory-bound kernel: counter bytes of the shading pass over its measured time
        ksh = (prof or {}).get("k_shade_surface")
        if ksh and kernel_ms and kernel_ms["shade"] > 0 and ksh.get("hbm_bytes_per_spp"):
            gb = ksh["hbm_bytes_per_spp"] * spp / 1e9
            out["roofline_shade"] = {"kernel": "k_shade_surface (all launches of a step" + ("; the Lambertian launches of this LDS-resident scene include their inline shadow walk, which moves no bytes: the pass is VALU-bound, this is its byte rate, not its limit)" if lds_scene else ")"), "counter_GB_per_step": gb, "ms_per_step": kernel_ms["shade"],
                                     "counter_GBps": gb / (kernel_ms["shade"] * 1e-3), "frac_of_8000": gb / (kernel_ms["shade"] * 1e-3) / HBM_PEAK_GBS,
                                     "source": f"profiles/{pname}: FETCH_SIZE (doubled, gfx950) + WRITE_SIZE of the shading launches per spp, x this run's spp, over this run's kernel_ms.shade"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
