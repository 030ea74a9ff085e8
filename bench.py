#!/usr/bin/env python3
"""Benchmark of the rendering hot path on MI355X.

One "step" is one full frame: every (pixel, sample) of the configuration is generated, traced and
shaded on the GPU(s), partial sums are reduced to rank 0 over RCCL when N > 1, and the mean image
is resolved.  The scene, textures, environment map and acceleration structure are resident in HBM
before the timed region starts.

Default workload = the reference's README benchmark (README.md:31-33): full PBR, BVH, 800x600,
300 spp, 6 bounces, environment farmland_overcast_1k.hdr — on the seeded procedural stand-in for
example/scene.glb, which is not available.  `value` is whole-job Mrays/s (rays = intersect_ray
calls, counted by the scheduler from its queue sizes); ms_per_step is the time to render the frame.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W          (starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Ranks.  With WORLD_SIZE in the environment (torchrun) the process is one rank of that many, and `--gpus` must say the
same number.  Without it, `--gpus N` > 1 makes this process a launcher: it starts N children of itself with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, before it has touched torch or the GPU, passes rank 0's
line through and exits non-zero if any child does.  The line carries `n_ranks_seen`, the all-reduced count of ranks that
took part, so a run that silently used fewer ranks than asked for cannot pass as N.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

CONFIGS = {
    # name: (program, width, height, spp, bounces, newrays, scene, env)
    "readme_pbr": dict(program=2, width=800, height=600, spp=300, bounces=6, newrays=0, scene="readme", env="farmland"),
    "ao_1080p": dict(program=1, width=1920, height=1080, spp=64, bounces=1, newrays=8, scene="readme", env="white"),
    "raycast_640": dict(program=0, width=640, height=360, spp=1, bounces=4, newrays=0, scene="readme", env="white"),
    "stress_4k": dict(program=2, width=3840, height=2160, spp=256, bounces=8, newrays=0, scene="stress", env="white"),
    # 12 bounces with every lobe followed is ~1000 rays per sample on this scene; the single-path estimator (one ray per
    # hit, same expectation) is what makes the configuration renderable
    "material_4k": dict(program=2, width=3840, height=2160, spp=4096, bounces=12, newrays=0, scene="material", env="farmland",
                        single=True),
}


def build_scene(kind: str, detail: float, stress_tris: int):
    from pathtracer_amd import scenes
    if kind == "stress":
        return scenes.stress_mesh(stress_tris)
    if kind == "material":
        return scenes.material_stress(detail=detail)
    return scenes.readme_standin(detail=detail)


def load_env(kind: str):
    if kind == "white":
        return None
    from pathtracer_amd import image_io
    return image_io.load_hdr(os.path.join(ROOT, "assets", "farmland_overcast_1k.hdr"))


def cpu_baseline(model, env, cfg, seconds_target: float = 8.0):
    """Time the CPU oracle (a port of the reference renderer, libm math) on a bounded sample of the same workload: a band
    of image rows around the centre at reduced spp.  Two builds of the same source are timed: the one compiled on this
    machine with -O3 -march=native -ffp-contract=off (the headline `value`; needs g++ at run time) and the parity build
    the tests use (-O2 -march=x86-64-v3)."""
    import tempfile
    from oracle import binding as ob
    from pathtracer_amd import abi
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    w, h = cfg["width"], cfg["height"]
    cam = model.cameras[0]
    flags = abi.FLAG_SINGLE_PATH if cfg.get("single") else 0

    def settings_for(spp):
        return abi.Settings(spp, cfg["bounces"], cfg["newrays"], 8, 0.0, cfg["program"], 1, 1234, 0, 1, flags)

    rows = (max(0, h // 2 - 8), min(h, h // 2 + 8))
    one_rows = (rows[0], min(rows[1], rows[0] + 4))
    probe_spp = 1 if cfg["spp"] == 1 else 2

    def measure(library, label):
        s = ob.OracleScene(model, env, settings_for(probe_spp), library=library)
        s.update_camera_transform_state(cam.position, cam.direction, cam.up, cam.yfov, w / h, cam.znear, cam.zfar)
        s.set_math_mode(ob.MATH_LIBM)
        _, st = s.render_frame(w, h, n_threads=threads, rows=rows)
        spp = int(max(probe_spp, min(cfg["spp"], seconds_target / max(st.seconds / probe_spp, 1e-9))))
        if spp > probe_spp:
            s.set_settings(settings_for(spp))
            _, st = s.render_frame(w, h, n_threads=threads, rows=rows)
        # the same build on one thread, on a quarter of the band at the probe's sample count (a few seconds)
        s.set_settings(settings_for(probe_spp))
        _, st1 = s.render_frame(w, h, n_threads=1, rows=one_rows)
        s.close()
        return {"value": round(st.rays / max(st.seconds, 1e-9) / 1e6, 4), "unit": "Mrays/s", "cores": threads, "build": label,
                "sample": f"rows {rows[0]}-{rows[1]} of {h} x {w} px at {spp} spp, {cfg['bounces']} bounces: "
                          f"{st.rays} rays in {st.seconds:.2f} s (oracle/pt_oracle.cc, libm math)",
                "one_thread": {"value": round(st1.rays / max(st1.seconds, 1e-9) / 1e6, 4), "unit": "Mrays/s",
                               "sample": f"rows {one_rows[0]}-{one_rows[1]} at {probe_spp} spp: {st1.rays} rays in {st1.seconds:.2f} s"}}

    parity = measure(None, "g++ -O2 -march=x86-64-v3 -ffp-contract=off (the build the parity tests use)")
    native = None
    try:
        path = ob.build_native(os.path.join(tempfile.gettempdir(), f"pt_oracle_native_{os.getuid()}"))
        native = measure(ob.load_library(path), "g++ -O3 -march=native -ffp-contract=off, compiled on this machine")
    except Exception as e:  # no compiler on this machine: the parity build's figure stands
        parity["native_build_error"] = f"{type(e).__name__}: {str(e)[:200]}"
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    out = dict(native or parity)
    out["kind"] = "port"
    out["cpu_model"] = cpu_model
    out["hardware_threads"] = os.cpu_count()
    if native:
        out["parity_build"] = {k: parity[k] for k in ("value", "unit", "build", "sample", "one_thread")}
    return out


PROFILE_TAG = "r03"          # profiles/<tag>_hbm_traffic.json, profiles/<tag>_counters_<name>.txt
COUNTER_FILES = {"readme_pbr": "readme", "ao_1080p": "ao", "raycast_640": "raycast", "stress_4k": "stress", "material_4k": "material"}
# Ceilings of the paths that bind this kernel, measured on this part with tools/micro/gather_bench.hip: what the vector L1s
# sustain for divergent 64-byte records read as four 16-byte loads (GB/s over all CUs), and what the memory fabric
# delivers for dependent random 64-byte sectors that miss the L2.
L1_GATHER_CEILING_GBPS = 14200.0
FABRIC_SECTOR_CEILING_GBPS = 3700.0
HBM_PEAK_GBPS = 8000.0


def roofline(args, cfg, model, r, settings, st, tot, w, h):
    """The `roofline` object of the line, for the dominant kernel (the persistent traversal kernel).

    Algorithmic bytes: per ray the kernel reads origin + direction (32 B) and its candidate list (8 B per entered instance
    plus the end word), a 128 B transform per entered instance, 64 B per visited node (binary: two children's boxes;
    four-wide: four children's quantised boxes) and 48 B per tested triangle — or 64 B per leaf record where the scene's
    leaves are stored that way — and writes a 20 B hit record; all counted on the same scene and camera at reduced spp by
    the counting form of the kernel.  Measured bytes: FETCH_SIZE + WRITE_SIZE of the committed counter passes of this
    configuration (bytes that crossed the L2 <-> fabric boundary), per ray, scaled to the rays of a launch here.
    Which resource binds is read off the committed utilisation counters of the configuration: the headline `frac` is
    against THAT resource's measured ceiling; the fraction of the HBM peak is kept beside it under `hbm`."""
    import re
    from pathtracer_amd.renderer import AccelerationStructureType, RayProgramMode, RenderSettings
    count_settings = RenderSettings(min(cfg["spp"], 8), cfg["bounces"], cfg["newrays"], 8, 0.0,
                                    RayProgramMode(cfg["program"]), AccelerationStructureType.BVH, seed=1234,
                                    singlePath=bool(cfg.get("single")), wideNodes=not args.binary_nodes)
    r.set_render_settings(count_settings)
    _, cst = r.render_counted(w, h)
    r.set_render_settings(settings)
    V = cst.nodes_visited / cst.rays
    T = cst.triangles_tested / cst.rays
    LR = getattr(cst, "leaf_records_read", 0) / cst.rays
    I = cst.instances_tested / cst.rays
    E = cst.instances_entered / cst.rays
    leaf_bytes = LR * 64 if LR > 0 else T * 48
    bytes_per_ray = V * 64 + leaf_bytes + E * 128 + (E + 1) * 8 + 32 + 20
    tlas_bytes_per_ray = I * 64 + 32 + (E + 1) * 8
    launches = max(tot["extend_launches"], 1)
    avg_launch_s = tot["traverse_ms"] / 1e3 / launches
    rays_per_launch = tot["rays"] / launches
    algorithmic = bytes_per_ray * rays_per_launch
    lds_visits = getattr(cst, "nodes_from_lds", 0) / max(cst.rays, 1)
    l1_bytes = algorithmic - lds_visits * 64 * rays_per_launch
    l1_gbps = l1_bytes / avg_launch_s / 1e9

    # ---- measured fabric traffic: the committed FETCH_SIZE / WRITE_SIZE passes of this configuration ----
    measured = source = kernel_key = None
    tpath = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_hbm_traffic.json")
    key = f"{args.config}/{'binary' if args.binary_nodes else 'wide'}"
    scene_text = f"{model.instanced_triangles()} triangles, {len(model.objects)} objects"
    if os.path.exists(tpath):
        entry = json.load(open(tpath)).get("configs", {}).get(key) or {}
        # the traversal kernel of the profile: the non-counting traverse_kernel* entry with the largest share of the GPU time
        names = [k for k in entry.get("kernels", {}) if k.startswith("traverse_kernel") and "<false" in k and "fetch_bytes_per_ray" in entry["kernels"][k]]
        if names and entry.get("scene") == scene_text and entry.get("bounces") == cfg["bounces"] and entry.get("resolution") == [w, h]:
            kernel_key = max(names, key=lambda k: entry["kernels"][k].get("share_of_gpu_time_pct", 0.0))
            k = entry["kernels"][kernel_key]
            measured = (k["fetch_bytes_per_ray"] + k["write_bytes_per_ray"]) * rays_per_launch
            source = (f"profiles/{PROFILE_TAG}_hbm_traffic.json[{key}][{kernel_key}] ({entry.get('profile', '')}): "
                      f"{k['fetch_bytes_per_ray']:.1f} B fetched + {k['write_bytes_per_ray']:.1f} B written per ray")
    fabric_gbps = (measured / avg_launch_s / 1e9) if measured else None

    # ---- utilisation of the kernel's units: the committed counter passes of this configuration ----
    util = None
    cname = COUNTER_FILES.get(args.config)
    cpath = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_counters_{cname}.txt") if cname and not args.binary_nodes else ""
    if cpath and os.path.isfile(cpath):
        text = open(cpath).read()
        blocks = re.findall(r"^== (traverse_kernel[^\n]*<false[^\n]*) ==\n((?:  [^\n]*\n)+)", text, flags=re.M)
        if blocks:
            def grab(block, pattern):
                m = re.search(pattern, block)
                return float(m.group(1)) if m else None
            # (the traversal kernel with the most busy clocks, should the pass hold more than one form)
            name, block = max(blocks, key=lambda nb: grab(nb[1], r"busy clocks per XCD\s+([0-9]+)") or 0.0)
            util = {"kernel": name,
                    "valu_utilisation": grab(block, r"VALU utilisation \([^)]*\)\s+([0-9.]+)"),
                    "waves_per_simd": grab(block, r"mean waves per SIMD \([^)]*\)\s+([0-9.]+)"),
                    "wave_time_waiting": grab(block, r"wave time: waiting ([0-9.]+)"),
                    "wave_time_ready_not_issued": grab(block, r"issue-stalled ([0-9.]+)"),
                    "address_unit_busy": grab(block, r"TA busy ([0-9.]+)"),
                    "l2_hit_rate": grab(block, r"L2: hit rate ([0-9.]+)"),
                    "source": os.path.relpath(cpath, ROOT) + " (rocprofv3 --pmc passes of this configuration at a reduced sample count)"}

    # ---- which resource binds ----
    hbm = {"achieved": round(fabric_gbps, 2) if fabric_gbps else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
           "frac": round(fabric_gbps / HBM_PEAK_GBPS, 5) if fabric_gbps else None,
           "note": "measured fabric bytes (L2 misses: served by the Infinity Cache or HBM) / kernel time against the HBM peak; "
                   "an upper bound of the HBM share"}
    if rays_per_launch < 2e6 and avg_launch_s < 1e-3:
        bound, achieved, peak = "launch-latency", l1_gbps, L1_GATHER_CEILING_GBPS
        binding = (f"a launch of {rays_per_launch:.0f} rays lasts {avg_launch_s * 1e6:.0f} us: the frame is a handful of kernel launches and a "
                   f"read-back, no unit is busy; the figure is against the L1 gather ceiling for completeness")
    elif util and fabric_gbps and (util["l2_hit_rate"] or 1.0) < 0.45 and fabric_gbps >= 0.75 * FABRIC_SECTOR_CEILING_GBPS:
        bound, achieved, peak = "fabric-sectors", fabric_gbps, FABRIC_SECTOR_CEILING_GBPS
        binding = (f"the scene is beyond the caches (L2 hit rate {util['l2_hit_rate']:.2f}): the fabric delivers {fabric_gbps:.0f} GB/s of random 64-byte "
                   f"sectors against the {FABRIC_SECTOR_CEILING_GBPS:.0f} GB/s a dependent random-sector gather reaches on this part "
                   f"(tools/micro/gather_bench.hip); address unit busy {util['address_unit_busy']:.2f}, vector ALU {util['valu_utilisation']:.2f}; "
                   f"what moves it is sectors per ray")
    elif util:
        bound, achieved, peak = "l1-gather+issue", l1_gbps, L1_GATHER_CEILING_GBPS
        binding = (f"cache-resident (L2 hit rate {util['l2_hit_rate']:.2f}): a dependent chain of gathers through the vector L1 — address unit busy "
                   f"{util['address_unit_busy']:.2f} of the time, vector ALU {util['valu_utilisation']:.2f} at {util['waves_per_simd']:.1f} waves per SIMD, "
                   f"{util['wave_time_waiting']:.2f} of wave time waiting and {util['wave_time_ready_not_issued']:.2f} ready but not issued; "
                   f"not memory bandwidth (fabric at {hbm['frac'] if hbm['frac'] is not None else float('nan'):.2f} of the HBM peak)")
    else:
        bound, achieved, peak = "l1-gather+issue", l1_gbps, L1_GATHER_CEILING_GBPS
        binding = "no utilisation counter pass is committed for this configuration and node layout: the L1 gather path is assumed"
    return {
        "kernel": kernel_key or (util or {}).get("kernel") or "traverse_kernel*",
        "bound": bound,
        "achieved": round(achieved, 2),
        "peak": peak,
        "unit": "GB/s",
        "frac": round(achieved / peak, 5),
        "traffic": round(measured) if measured else None,
        "traffic_source": source,
        "binding": binding,
        "hbm": hbm,
        "utilisation": util,
        "algorithmic_GBps": round(algorithmic / avg_launch_s / 1e9, 2),
        "algorithmic_frac_of_hbm_peak": round(algorithmic / avg_launch_s / 1e9 / HBM_PEAK_GBPS, 5),
        "algorithmic_bytes_per_launch": round(algorithmic),
        "l1_side": {"GBps": round(l1_gbps, 2), "ceiling_GBps": L1_GATHER_CEILING_GBPS, "frac": round(l1_gbps / L1_GATHER_CEILING_GBPS, 4),
                    "note": "algorithmic bytes minus the node visits served from the LDS node cache, against the rate the vector L1s "
                            "sustain for divergent 64-byte records read as four 16-byte loads back to back (tools/micro/gather_bench.hip)"},
        "note": "bound / frac: the resource the committed counters show to bind, against its measured ceiling; hbm.frac = measured fabric "
                "bytes / kernel time / 8 TB/s; algorithmic_* counts every visit as a fetch whichever cache served it (work done, not a bound)",
        "bytes_per_ray": round(bytes_per_ray, 1),
        "per_ray": {"nodes_64B": round(V, 3), "nodes_served_from_lds": round(lds_visits, 3), "triangles_tested": round(T, 3),
                    "leaf_records_64B": round(LR, 3) if LR > 0 else None, "triangle_records_48B": round(T, 3) if LR == 0 else None,
                    "instances_entered_128B_plus_8B": round(E, 3), "ray_read_B": 32, "hit_write_B": 20},
        "rays_per_launch": round(rays_per_launch),
        "avg_launch_ms": round(tot["traverse_ms"] / launches, 4),
        "launches_per_step": tot["extend_launches"] // max(args.steps, 1),
        "traverse_workgroups_per_cu": getattr(st, "traverse_workgroups_per_cu", 0),
        "traverse_share_of_step": round(tot["traverse_ms"] / (tot["elapsed"] * 1e3), 4),
        "instance_pass": {"kernel": "tlas_kernel<false>", "bytes_per_ray": round(tlas_bytes_per_ray, 1),
                          "instance_bounds_64B": round(I, 3),
                          "avg_launch_ms": round((tot["extend_ms"] - tot["traverse_ms"]) / launches, 4),
                          "share_of_step": round((tot["extend_ms"] - tot["traverse_ms"]) / (tot["elapsed"] * 1e3), 4)},
        "extend_share_of_step": round(tot["extend_ms"] / (tot["elapsed"] * 1e3), 4),
        "shade_share_of_step": round(tot["shade_ms"] / (tot["elapsed"] * 1e3), 4),
        "extend_mrays_per_s": round(tot["rays"] / (tot["extend_ms"] / 1e3) / 1e6, 2),
    }


def launch_ranks(n: int, argv: list[str]) -> int:
    """Start `n` ranks of this script (one per GPU) and wait for them.  The launcher itself never imports torch and never
    touches a GPU: the children are fresh processes, started before anything here could have initialised a device.
    Rank 0's stdout is passed through (the JSON line), the other ranks' stdout is dropped, stderr is shared.  Returns
    the first non-zero exit status of any rank, else 0; when one rank fails the others are stopped."""
    import socket
    import subprocess
    with socket.socket() as sk:  # a free port for the rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # the host tree build of every rank runs at the same time: share the cores instead of oversubscribing them n-fold
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
    build_threads = max(1, min(16, cores) // n)
    children = []
    for rank in range(n):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("PTGPU_BUILD_THREADS", str(build_threads))
        children.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, text=True,
                                         stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL))

    def relay(pipe):  # rank 0's JSON line goes to stdout; whatever else a library printed there goes to stderr
        for ln in pipe:
            print(ln, end="", file=sys.stdout if ln.lstrip().startswith("{") else sys.stderr, flush=True)

    import threading
    reader = threading.Thread(target=relay, args=(children[0].stdout,), daemon=True)
    reader.start()
    status = 0
    pending = dict(enumerate(children))
    while pending:
        for rank, ch in list(pending.items()):
            rc = ch.poll()
            if rc is None:
                continue
            del pending[rank]
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 128 - rc
                print(f"bench.py: rank {rank} of {n} exited with status {rc}; stopping the others", file=sys.stderr, flush=True)
                for other in pending.values():  # exactly the processes started above, by handle
                    other.terminate()
        time.sleep(0.05)
    reader.join(timeout=10)
    return status


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="readme_pbr", choices=sorted(CONFIGS))
    ap.add_argument("--detail", type=float, default=1.0, help="tessellation of the stand-in scene (1.0 = ~1M triangles)")
    ap.add_argument("--stress-tris", type=int, default=10_000_000)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--bounces", type=int, default=-1)
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--single-path", action="store_true", help="PBR: one spawned ray per hit (same expectation)")
    ap.add_argument("--split", choices=("samples", "tiles"), default="samples",
                    help="how N > 1 ranks share the frame: disjoint samples of every pixel, or disjoint 8x8 tiles")
    ap.add_argument("--binary-nodes", action="store_true",
                    help="traverse the binary nodes in the reference's visiting order instead of the four-wide quantised nodes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="start the ranks, form the process group, count them, print the line's rank fields and stop: "
                         "checks the launch path on a machine without GPUs (PT_BENCH_BACKEND=gloo); renders nothing")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:  # this process is the launcher; it has not imported torch and never touches a GPU
            raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher and the "
                         f"command line disagree about the number of ranks; refusing to report one as the other")

    cfg = dict(CONFIGS[args.config])
    if args.spp:
        cfg["spp"] = args.spp
    if args.bounces >= 0:
        cfg["bounces"] = args.bounces
    if args.width:
        cfg["width"] = args.width
    if args.height:
        cfg["height"] = args.height
    if args.single_path:
        cfg["single"] = True

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist
    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")  # "gloo" rehearses the N > 1 path on a one-GPU box
    if args.rendezvous_only:
        if os.environ.get("PT_BENCH_TEST_FAIL_RANK") == str(rank):  # (tests: one rank dies before the rendezvous)
            raise SystemExit(3)
        if world > 1:
            dist.init_process_group(backend)
        ones = torch.ones(1, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": world, "n_ranks_seen": int(ones.item()), "backend": backend,
                              "build_threads_per_rank": os.environ.get("PTGPU_BUILD_THREADS")}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the rendering path has no CPU fallback")
    # one rank per GPU; on a box with fewer GPUs than ranks (rehearsals only) ranks share devices
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    if world > 1 and "PTGPU_BUILD_THREADS" not in os.environ:
        # every rank of the node builds the scene's trees at the same time: share the host's cores (torchrun sets
        # LOCAL_WORLD_SIZE; this script's own launcher has set the variable already)
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
        cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or (os.cpu_count() or 1)
        os.environ["PTGPU_BUILD_THREADS"] = str(max(1, min(16, cores) // local_world))

    from pathtracer_amd.renderer import (AccelerationStructureType, Perspective, RayProgramMode, Renderer,
                                         RenderSettings)

    model = build_scene(cfg["scene"], args.detail, args.stress_tris)
    env = load_env(cfg["env"])
    w, h = cfg["width"], cfg["height"]
    from pathtracer_amd.distributed import combine_partial_sums, subset_settings, subset_tiles
    settings = (subset_tiles if args.split == "tiles" else subset_settings)(RenderSettings(cfg["spp"], cfg["bounces"], cfg["newrays"], 8, 0.0,
                                              RayProgramMode(cfg["program"]), AccelerationStructureType.BVH, seed=1234,
                                              singlePath=bool(cfg.get("single")), wideNodes=not args.binary_nodes),
                               rank, world)
    r = Renderer(device_index)
    r.set_render_settings(settings)
    t_load = time.time()
    r.load_scene(model, env)
    t_load = time.time() - t_load
    cam = model.cameras[0]
    r.update_camera_transform_state(cam.position, cam.direction, cam.up, Perspective(cam.yfov, cam.znear, w / h, cam.zfar))

    buf = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    # Render, reduce and resolve are ordered by ONE stream of torch's: the library launches on the handle it is given, and
    # an RCCL reduce makes torch's CURRENT stream (not the host) wait for it, so the stream has to be the same and has to
    # be current.  (torch's default stream has the handle 0, which the library reads as "use the context's own stream".)
    work_stream = torch.cuda.Stream()
    stream = work_stream.cuda_stream
    assert stream != 0
    torch.cuda.synchronize()  # the buffer's zero fill ran on torch's default stream, which this stream does not wait for

    def step():
        with torch.cuda.stream(work_stream):
            st = r.render_frame_sums(w, h, buf.data_ptr(), stream)
            combine_partial_sums(buf, cfg["spp"], rank, world,
                                 resolve=lambda b: r.resolve_sums(w, h, b.data_ptr(), cfg["spp"], stream))
        return st

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    rays = 0
    samples = 0
    extend_ms = 0.0
    shade_ms = 0.0
    extend_launches = 0
    traverse_ms = 0.0
    for _ in range(args.steps):
        st = step()
        rays += st.rays
        samples += st.camera_samples
        extend_ms += st.extend_ms
        traverse_ms += st.traverse_ms
        shade_ms += st.shade_ms
        extend_launches += st.extend_launches
    fence()
    elapsed = time.perf_counter() - t0

    # (the last entry counts the ranks that took part: every rank adds one)
    tot = torch.tensor([elapsed, float(rays), float(samples), 1.0], dtype=torch.float64, device="cuda")
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
    total_rays = float(tot[1])
    total_samples = float(tot[2])
    n_ranks_seen = int(round(float(tot[3])))
    if n_ranks_seen != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {n_ranks_seen} ranks took part")

    if rank == 0:
        ms_per_step = elapsed / max(args.steps, 1) * 1e3
        line = {
            "metric": "Mrays/sec (whole node) + seconds to render the README scene at 300spp 6 bounces",
            "value": round(total_rays / elapsed / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "n_ranks_seen": n_ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "seconds_per_frame": round(ms_per_step / 1e3, 4),
            "msamples_per_s": round(total_samples / elapsed / 1e6, 3),
            "rays_per_sample": round(total_rays / max(total_samples, 1.0), 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic: seeded procedural stand-in for example/scene.glb (not available); "
                    + ("assets/farmland_overcast_1k.hdr environment" if cfg["env"] == "farmland" else "white environment"),
            "config": {
                "workload": f"{args.config}: program={RayProgramMode(cfg['program']).name} BVH {w}x{h} {cfg['spp']}spp "
                            f"{cfg['bounces']} bounces{' single-path' if cfg.get('single') else ''}, "
                            f"{'binary' if args.binary_nodes else 'four-wide quantised'} nodes, "
                            f"{model.instanced_triangles()} instanced triangles, "
                            f"{len(model.objects)} objects",
                "parallelism": f"{'tile' if args.split == 'tiles' else 'sample'}-split x{world}" + ((" + RCCL" if backend == "nccl" else f" + {backend}") + " reduce(sum) to rank 0" if world > 1 else ""),
                "scene_load_s": round(t_load, 2),
            },
        }
        if not args.no_roofline:  # rank 0's own launches (every rank holds the same scene)
            line["roofline"] = roofline(args, cfg, model, r, settings, st, dict(rays=rays, elapsed=elapsed, extend_ms=extend_ms,
                                                                            traverse_ms=traverse_ms, shade_ms=shade_ms,
                                                                            extend_launches=extend_launches), w, h)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model, env, cfg)
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
