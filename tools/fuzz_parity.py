#!/usr/bin/env python3
"""usage (GPU box, repo root): python tools/fuzz_parity.py [--seeds A:B] [--seconds S]
Differential run of the library against the CPU oracle over random scenes and random settings — what the parity tests do on
a handful of fixed cases, here on as many seeds as the time allows.  Per seed: an unstructured triangle soup (1-12 objects of
50-20000 triangles under random affine transforms, sometimes far from the origin, sometimes with zero-thickness quads), then
  * closest hits of the camera rays and of random rays, and occlusion flags of random any-hit rays, on a random choice of
    acceleration structure (naive / host-built / device-built), node layout, leaf size, kernel form and leaf-record setting:
    binary nodes over the host-built tree must equal the oracle bit for bit except where two objects tie in distance, every
    other combination on >= 99.99 % of rays;
  * a small ambient-occlusion or PBR frame (random samples, bounces, branching or single-path estimator, output reservation
    through one or eight counters): ray counts and pixels (2e-5 * max(1, v)) equal up to what one or two ties can change.
The two classes of rays without a single right answer (ties between objects; rays that start on zero-thickness boxes) are
described where the checks set them aside.
Prints one line per seed, the failures again at the end, and exits non-zero if there was one."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import helpers as H  # noqa: E402
from pathtracer_amd import abi, scenes  # noqa: E402


def one_seed(seed: int) -> list[str]:
    rng = np.random.default_rng(1000 + seed)
    n_obj = int(rng.integers(1, 13))
    tris = int(rng.choice([50, 500, 4000, 20000]))
    if n_obj * tris > 120000:
        tris = 4000
    offset = float(rng.choice([0.0, 0.0, 30.0, 300.0, 5000.0]))
    flat = bool(rng.integers(0, 2))
    model = scenes.random_soup(seed, n_objects=n_obj, tris_per_object=tris, offset=offset, flat=flat)
    env = scenes.synthetic_env(seed=seed)
    leaf = int(rng.choice([1, 2, 4, 8, 16]))
    w, h = 160, 90
    fails = []
    o = H.make_oracle(model, None, H.settings(program=0, leaf=leaf), w, h)
    ref = o.render_hits(w, h)
    n = 20000
    org = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    d = rng.uniform(-1, 1, (n, 3)) - org
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    axis = rng.integers(0, n, 400)  # some axis-parallel directions: zero components take the exact box path
    d[axis] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, 400)] * rng.choice([-1.0, 1.0], (400, 1)).astype(np.float32)
    flags = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    ref_rays = o.intersect_rays(org, d, flags)
    combos = [(1, False, {})]
    for _ in range(3):
        as_type = int(rng.choice([0, 1, 2])) if n_obj * tris <= 30000 else int(rng.choice([1, 2]))
        wide = bool(rng.integers(0, 2)) and as_type != 0
        opts = {}
        if wide:
            opts["traverse_waves"] = int(rng.choice([0, 6, 7]))
            opts["flags"] = int(rng.choice([0, abi.OPT_LEAF_RECORDS, abi.OPT_NO_LEAF_RECORDS, abi.OPT_GREEDY_COLLAPSE, abi.OPT_REFERENCE_SPLITS]))
            opts["leaf_vote"] = int(rng.choice([0, 1, 4]))
        elif rng.integers(0, 4) == 0:
            opts["flags"] = abi.OPT_EXTEND_SIMPLE
        combos.append((as_type, wide, opts))
    # Two classes of rays have no single right answer, and the checks below set them aside:
    #  * ties: two triangles of different objects hit at bit-equal distance (frequent far from the origin, where the hit
    #    point is quantised coarsely before the distance is recomputed in world space) — the reference visits objects of
    #    equal entry distance in std::make_heap order, the library in table order, and the first one found wins;
    #  * zero-thickness boxes met by a ray that starts ON their plane (flat quads at coordinates where the 1e-5 offset of
    #    a spawned ray is below one unit in the last place): the reference's box test gives t1 = 0, `forward` needs
    #    t1 > 0, so whether the triangle is found depends on what else shares its leaf — the reference's own NaiveAS
    #    and BVH_AS disagree on such rays, and so do trees of different shape.
    degenerate_boxes = flat and offset >= 300.0

    def unexplained(ref_rec, got_rec, closest_only=None):
        exact, _ = H.hits_equal(ref_rec, got_rec)
        both = (ref_rec["hit"] == 1) & (got_rec["hit"] == 1)
        tie = both & (ref_rec["t"].view(np.uint32) == got_rec["t"].view(np.uint32))
        bad = ~exact & ~tie
        if closest_only is not None:
            bad = bad & closest_only
        return int(bad.sum()), int((~exact & tie).sum())

    for as_type, wide, opts in combos:
        g = H.make_gpu(model, None, H.settings(program=0, as_type=as_type, wide=wide, leaf=leaf), w, h, **opts)
        # binary nodes over the host-built tree are visited in the oracle's order: everything equal, any-hit records
        # included, except ties between objects; other trees and layouts: closest hits equal except ties
        exact_order = not wide and as_type == 1
        allowed = 0.0 if exact_order else (0.1 if degenerate_boxes else 1e-4)
        got_cam = g.render_hits(w, h)
        bad, ties = unexplained(ref.ravel(), got_cam.ravel())
        if bad > allowed * ref.size:
            fails.append(f"camera hits as={as_type} wide={wide} {opts}: {bad} differ beyond {ties} ties")
        got = g.intersect_rays(org, d, flags)
        closest = flags == 0
        bad, ties = unexplained(ref_rays, got, None if exact_order else closest)
        flag_diff = int((got["hit"] != ref_rays["hit"]).sum())
        if bad > allowed * n or (not exact_order and flag_diff > allowed * n) or (exact_order and ties > 2e-3 * n):
            fails.append(f"ray hits as={as_type} wide={wide} {opts}: {bad} differ beyond {ties} ties, {flag_diff} hit flags differ")
        del g
    # a frame
    program = int(rng.choice([1, 2, 2]))
    spp = int(rng.choice([1, 2, 5]))
    bounces = int(rng.choice([1, 2, 4, 7]))
    single = bool(rng.integers(0, 2)) and program == 2
    newrays = int(rng.choice([0, 1, 3])) if program == 1 else 0
    wide = bool(rng.integers(0, 2))
    fw, fh = (int(x) for x in rng.choice([(96, 54), (128, 72), (100, 37)]))
    st = H.settings(program=program, spp=spp, bounces=bounces, newrays=newrays, single=single, wide=wide, leaf=leaf, seed=seed)
    so_ = H.settings(program=program, spp=spp, bounces=bounces, newrays=newrays, single=single, leaf=leaf, seed=seed)
    opts = {"flags": abi.OPT_ONE_COUNTER} if rng.integers(0, 3) == 0 else {}
    if rng.integers(0, 3) == 0:
        opts.update(chunk_rays=4096, pool_gib=1)
    g = H.make_gpu(model, env, st, fw, fh, **opts)
    oo = H.make_oracle(model, env, so_, fw, fh)
    img_g, sg = H.gpu_render(g, fw, fh)
    img_o, so = oo.render_frame(fw, fh)
    bad = (np.abs(img_g[..., :3] - img_o[..., :3]) > 2e-5 * np.maximum(1.0, np.abs(img_o[..., :3]))).any(axis=-1)
    if not wide and sg.rays == so.rays and not bad.any():
        # the reference's visiting order: the traversal counters are the oracle's (a tie that falls the other way moves them
        # by a ray's worth)
        _, sc_ = g.render_counted(fw, fh)
        for name in ("nodes_visited", "triangles_tested", "instances_entered", "instances_tested"):
            a_, b_ = int(getattr(sc_, name)), int(getattr(so, name))
            if abs(a_ - b_) > max(2000, 1e-4 * b_):
                fails.append(f"counter {name}: {a_} vs {b_}")
    what = f"frame program={program} {fw}x{fh} spp={spp} b={bounces} r={newrays} single={single} wide={wide} {opts}"
    # a tie resolved the other way changes one path's subtree: a handful of rays, one pixel
    few_rays, few_pixels = max(8, 2e-5 * so.rays), max(1, 3e-4 * fw * fh)
    if not wide:
        if abs(int(sg.rays) - int(so.rays)) > few_rays or bad.sum() > few_pixels or (sg.rays == so.rays and bad.any()):
            fails.append(f"{what}: rays {sg.rays} vs {so.rays}, {int(bad.sum())} pixels off")
    elif program == 1 and bounces >= 2:
        pass  # four-wide nodes report another triangle for an any-hit ray: an occlusion chain continues from another point
    elif flat and offset >= 30.0:
        pass  # rays spawned on zero-thickness boxes, see above
    elif abs(int(sg.rays) - int(so.rays)) > few_rays or bad.sum() > few_pixels:
        fails.append(f"{what}: rays {sg.rays} vs {so.rays}, {int(bad.sum())} pixels off")
    return [f"seed {seed} ({n_obj} x {tris} triangles, offset {offset}, flat {flat}, leaf {leaf}): {f}" for f in fails]


def zoo_scene(seed: int):
    """Random instances of the procedural primitives under random affine transforms (non-uniform scale, mirrors), random
    glTF materials: textures of every kind, alpha modes, transmission with and without a volume, emission."""
    from pathtracer_amd.model import Camera, ModelBuilder, rotate, scale, translate
    rng = np.random.default_rng(50000 + seed)
    b = ModelBuilder(f"zoo_{seed}")
    n_tex = int(rng.choice([32, 32, 24, 37, 8]))  # (sizes that are not powers of two wrap by division)
    imgs = [b.add_image(scenes.checker_texture(n_tex, 4, seed=seed)), b.add_image(scenes.orm_texture(n_tex, seed=seed)),
            b.add_image(scenes.bump_normal_map(n_tex)), b.add_image(scenes.alpha_gradient_texture(n_tex)),
            b.add_image(scenes.leaf_mask_texture(n_tex, seed=seed)), b.add_image(scenes.eyes_emissive_texture(n_tex))]

    def tex(p=0.4):
        return int(rng.choice(imgs)) if rng.uniform() < p else -1

    mats = []
    for _ in range(int(rng.integers(2, 8))):
        volume = rng.uniform() < 0.3
        mats.append(b.add_material(
            base_color=tuple(rng.uniform(0.1, 1.0, 3)) + (float(rng.choice([1.0, 1.0, 0.6])),),
            emissive=tuple(rng.uniform(0, 1, 3)) if rng.uniform() < 0.25 else (0, 0, 0),
            metallic=float(rng.choice([0.0, 1.0, rng.uniform()])), roughness=float(rng.uniform(0.02, 1.0)),
            base_color_texture=tex(), metallic_roughness_texture=tex(0.3), normal_texture=(imgs[2] if rng.uniform() < 0.3 else -1),
            ior=float(rng.uniform(1.0, 2.0)), transmission=float(rng.choice([0.0, 0.0, 0.5, 1.0])), transmission_texture=tex(0.15),
            emissive_texture=tex(0.15), emissive_strength=float(rng.uniform(0.5, 20.0)), double_sided=bool(rng.integers(0, 2)),
            attenuation_color=tuple(rng.uniform(0.1, 0.9, 3)) if volume else None,
            attenuation_distance=float(rng.uniform(0.2, 2.0)) if volume else None,
            alpha_mode=str(rng.choice(["OPAQUE", "OPAQUE", "BLEND", "MASK"])), alpha_cutoff=float(rng.uniform(0.2, 0.8))))
    makers = [lambda: scenes.plane(int(rng.integers(1, 6)), int(rng.integers(1, 6)), 3.0, 3.0, uv_scale=2.0),
              lambda: scenes.quad(1.5, 1.0), lambda: scenes.uv_sphere(int(rng.integers(4, 14)), int(rng.integers(6, 20)), 1.0),
              lambda: scenes.blob(int(rng.integers(6, 24)), int(rng.integers(6, 24)), 1.0, 0.2, seed=seed),
              lambda: scenes.torus(int(rng.integers(6, 20)), int(rng.integers(4, 12)), 0.8, 0.3), lambda: scenes.box(1.0, 0.7, 0.5)]
    def shape():
        pos, nrm, idx, uv, tan = scenes._args(makers[int(rng.integers(0, len(makers)))]())
        if rng.uniform() < 0.4:  # texture coordinates far outside the unit square, mirrored, off the texel grid
            uv = np.asarray(uv, np.float64) * float(rng.choice([-2.5, 3.0, 37.7, 0.013])) + rng.uniform(-5.0, 5.0, 2)
        return pos, nrm, idx, uv, tan

    meshes = [b.add_mesh(*shape(), material=int(rng.choice(mats + [-1])) if rng.uniform() < 0.9 else None)
              for _ in range(int(rng.integers(1, 7)))]
    n_obj = int(rng.choice([1, 2, 5, 9, 14, 30, 120, 250]))
    spread = 2.5 if n_obj < 40 else 6.0
    for _ in range(n_obj):
        sc = rng.uniform(0.3, 1.4, 3) * (0.5 if n_obj >= 40 else 1.0)
        if rng.uniform() < 0.15:
            sc[int(rng.integers(0, 3))] *= -1.0  # a mirror
        if rng.uniform() < 0.05:
            sc[int(rng.integers(0, 3))] = 0.0  # scaled to nothing: no inverse, never hit
        m = translate(*rng.uniform(-spread, spread, 3)) @ rotate(rng.normal(size=3), float(rng.uniform(0, 360))) @ scale(*sc)
        b.add_object(int(rng.choice(meshes)), m)
    eye = np.array([rng.uniform(-1, 1), rng.uniform(0, 2), 7.0 if n_obj < 40 else 14.0])
    b.add_camera(Camera(position=eye, direction=-eye / np.linalg.norm(eye), up=(0, 1, 0), yfov=np.radians(45.0), znear=0.1, zfar=100.0))
    return b.build(), n_obj


def one_zoo_seed(seed: int) -> list[str]:
    rng = np.random.default_rng(70000 + seed)
    model, n_obj = zoo_scene(seed)
    env = scenes.synthetic_env(seed=seed)
    fails = []
    leaf = int(rng.choice([1, 2, 4, 8]))
    w, h = 128, 72
    o = H.make_oracle(model, None, H.settings(program=0, leaf=leaf), w, h)
    ref = o.render_hits(w, h)
    n = 15000
    org = rng.uniform(-6, 6, (n, 3)).astype(np.float32)
    d = rng.uniform(-2, 2, (n, 3)) - org
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    flags = (rng.uniform(size=n) < 0.5).astype(np.uint8)
    ref_rays = o.intersect_rays(org, d, flags)
    for wide in (False, True):
        g = H.make_gpu(model, None, H.settings(program=0, leaf=leaf, wide=wide), w, h)
        for name, a, bb, closest in (("camera", ref.ravel(), g.render_hits(w, h).ravel(), None),
                                    ("rays", ref_rays, g.intersect_rays(org, d, flags), flags == 0)):
            exact, _ = H.hits_equal(a, bb)
            both = (a["hit"] == 1) & (bb["hit"] == 1)
            tie = both & (a["t"].view(np.uint32) == bb["t"].view(np.uint32))
            bad = ~exact & ~tie
            if wide and closest is not None:
                bad &= closest
            flag_diff = int((a["hit"] != bb["hit"]).sum())
            # four-wide nodes: a ray through an edge two triangles share may get the other triangle, and about one ray in
            # 10^7 that grazes the corner of a quantised box is culled where the reference's box test lets it through
            allowed = 0 if not wide else max(1, 1e-4 * len(a))
            if bad.sum() > allowed or flag_diff > allowed:
                fails.append(f"{name} hits wide={wide}: {int(bad.sum())} differ beyond ties, {flag_diff} hit flags differ")
        del g
    # the entry point that allocates nothing, on device pointers: the records of the allocating one
    import torch
    g = H.make_gpu(model, None, H.settings(program=0, leaf=leaf, wide=bool(rng.integers(0, 2))), w, h)
    m = int(rng.integers(1, min(n, abi.STATIC_MAX_RAYS)))
    d_o, d_d, d_f = (torch.from_numpy(a[:m].copy()).cuda() for a in (org, d, flags))
    d_out = torch.zeros(m * abi.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    g.intersect_rays_static(m, d_o.data_ptr(), d_d.data_ptr(), d_f.data_ptr(), d_out.data_ptr())
    got_static = d_out.cpu().numpy().view(abi.HIT_DTYPE)
    exact, _ = H.hits_equal(g.intersect_rays(org[:m], d[:m], flags[:m]), got_static)
    if not exact.all():
        fails.append(f"static entry point: {int((~exact).sum())} of {m} records differ from ptgpu_intersect_rays")
    del g
    program = int(rng.choice([0, 1, 2, 2, 2]))
    single = bool(rng.integers(0, 2)) and program == 2
    bounces = int(rng.choice([1, 2, 3, 4] if not single else [2, 5, 9, 14]))
    spp = int(rng.choice([1, 2, 3, 7]))
    newrays = int(rng.choice([0, 1, 4])) if program == 1 else 0
    fw, fh = (int(x) for x in rng.choice([(64, 36), (96, 54), (70, 45), (33, 17)]))
    step = int(rng.choice([1, 1, 2, 3]))
    begin = int(rng.integers(0, min(step, spp)))  # (a subset without samples has no mean)
    rot = float(rng.choice([0.0, rng.uniform(0, 6.28)]))
    st = H.settings(program=program, spp=spp, bounces=bounces, newrays=newrays, single=single, leaf=leaf, seed=seed, begin=begin, step=step, env_rot=rot)
    opts = {}
    if rng.integers(0, 3) == 0:
        opts.update(chunk_rays=int(rng.choice([1024, 4096, 65536])), pool_gib=1)
    g = H.make_gpu(model, env, st, fw, fh, **opts)
    oo = H.make_oracle(model, env, st, fw, fh)
    img_g, sg = H.gpu_render(g, fw, fh)
    img_o, so = oo.render_frame(fw, fh)
    tol = 2e-5
    bad = (np.abs(img_g[..., :3] - img_o[..., :3]) > tol * np.maximum(1.0, np.abs(img_o[..., :3]))).any(axis=-1)
    nan_mismatch = np.isnan(img_g[..., :3]).any(axis=-1) != np.isnan(img_o[..., :3]).any(axis=-1)
    few_rays, few_pixels = max(8, 2e-5 * so.rays), max(1, 3e-4 * fw * fh)
    # (pass-through chains that do not advance are cut by the same rule on both sides: ptgpu_stats.rays_abandoned)
    if abs(int(sg.rays) - int(so.rays)) > few_rays or bad.sum() > few_pixels or nan_mismatch.any() or (sg.rays == so.rays and bad.any()) \
            or (sg.rays == so.rays and sg.rays_abandoned != so.chains_cut):
        fails.append(f"frame program={program} {fw}x{fh} spp={spp} b={bounces} r={newrays} single={single} samples {begin}+{step}k "
                     f"rot={rot:.2f} {opts}: rays {sg.rays} vs {so.rays}, {int(bad.sum())} pixels off, {int(nan_mismatch.sum())} NaN mismatches, "
                     f"chains cut {sg.rays_abandoned} vs {so.chains_cut}")
    # the same frame over another tree / the four-wide nodes: another visiting order, so ties and edge hits may fall the other
    # way and (ambient occlusion beyond one bounce) any-hit chains continue from other points — everything else must agree
    if not (program == 1 and bounces >= 2) and so.rays < 3_000_000:
        as_type, wide = int(rng.choice([1, 2])), bool(rng.integers(0, 2))
        if as_type == 2 or wide:
            st2 = H.settings(program=program, spp=spp, bounces=bounces, newrays=newrays, single=single, leaf=leaf, seed=seed, begin=begin,
                             step=step, env_rot=rot, as_type=as_type, wide=wide)
            g2 = H.make_gpu(model, env, st2, fw, fh)
            img_2, s2 = H.gpu_render(g2, fw, fh)
            bad2 = (np.abs(img_2[..., :3] - img_o[..., :3]) > tol * np.maximum(1.0, np.abs(img_o[..., :3]))).any(axis=-1)
            if abs(int(s2.rays) - int(so.rays)) > max(16, 1e-4 * so.rays) or bad2.sum() > max(2, 2e-3 * fw * fh):
                fails.append(f"frame as={as_type} wide={wide} program={program} {fw}x{fh} spp={spp} b={bounces} single={single}: rays {s2.rays} vs "
                             f"{so.rays}, {int(bad2.sum())} pixels off")
    # the multi-GPU split on one GPU: the partial sums of three ranks' sample subsets, or tile subsets, add up to the frame
    if step == 1 and so.rays < 1_500_000:
        from pathtracer_amd.distributed import subset_settings, subset_tiles
        split = subset_tiles if rng.integers(0, 2) else subset_settings
        world = int(rng.choice([2, 3, 5]))
        total = torch.zeros((fh, fw, 4), dtype=torch.float32, device="cuda")
        rays = 0
        for rank in range(world):
            gr = H.make_gpu(model, env, split(st, rank, world), fw, fh)
            part = torch.zeros((fh, fw, 4), dtype=torch.float32, device="cuda")
            rays += gr.render_frame_sums(fw, fh, part.data_ptr()).rays
            torch.cuda.synchronize()
            total += part
            del gr
        g.resolve_sums(fw, fh, total.data_ptr(), spp)
        torch.cuda.synchronize()
        img_s = total.cpu().numpy()
        bad3 = (np.abs(img_s[..., :3] - img_o[..., :3]) > tol * np.maximum(1.0, np.abs(img_o[..., :3]))).any(axis=-1)
        if rays != sg.rays or bad3.sum() > few_pixels:
            fails.append(f"{split.__name__} x{world}: rays {rays} vs {sg.rays}, {int(bad3.sum())} pixels off the oracle's frame")
    return [f"zoo seed {seed} ({n_obj} objects, leaf {leaf}): {f}" for f in fails]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:100000")
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--kind", choices=("soup", "zoo"), default="soup",
                    help="soup: unstructured triangle soups, every structure and layout; zoo: instanced primitives, every material feature")
    args = ap.parse_args()
    a, b = (int(x) for x in args.seeds.split(":"))
    t0 = time.time()
    failures = []
    done = 0
    for seed in range(a, b):
        if time.time() - t0 > args.seconds:
            break
        f = one_seed(seed) if args.kind == "soup" else one_zoo_seed(seed)
        failures += f
        done += 1
        print(f"seed {seed}: {'ok' if not f else 'FAIL'} ({time.time() - t0:.0f} s)", flush=True)
        for line in f:
            print("   ", line, flush=True)
    print(f"{done} seeds, {len(failures)} failures")
    for line in failures:
        print(line)
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
