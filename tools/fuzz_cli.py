#!/usr/bin/env python3
"""usage (GPU box, repo root): python tools/fuzz_cli.py [--seeds A:B] [--seconds S]
The C++ host against the Python mirror on the random scenes of tools/fuzz_parity.py (instanced primitives, every material
feature, mirrors, objects scaled to nothing): per seed the scene is written as the flat container and as a GLB file, the
PathTracer command line renders both — a hit dump of the ray caster and a small PBR frame — and
  * the container's hit records must equal the Python mirror's bit for bit (same library, same arrays);
  * the GLB's hit records must name the same triangles on >= 99.9 % of the pixels with distances within 1e-4 (its reader
    recomputes the normal matrices in binary32) and its PBR frame must agree with the container's on >= 85 % of the pixels
    (a path that differs in its last bits may end on a lamp or not)."""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import helpers as H  # noqa: E402
from fuzz_parity import zoo_scene  # noqa: E402
from pathtracer_amd import abi, gltf_export, image_io  # noqa: E402

CLI = os.path.join(ROOT, "pathtracer_amd", "bin", "PathTracer")


def run(*args):
    return subprocess.run([CLI, *map(str, args)], capture_output=True, text=True, timeout=120)


def one_seed(seed: int, tmp: str) -> list[str]:
    model, n_obj = zoo_scene(seed)
    w, h = 96, 54
    fails = []
    model.save(os.path.join(tmp, "s.ptsc"))
    gltf_export.save_glb(model, os.path.join(tmp, "s.glb"))
    hits, frames = {}, {}
    for ext in ("ptsc", "glb"):
        dump, out = os.path.join(tmp, f"{ext}.hits"), os.path.join(tmp, f"{ext}.hdr")
        r = run("-f", os.path.join(tmp, f"s.{ext}"), "-p", 1, "-a", 2, "-s", 1, "--width", w, "--height", h, "--seed", seed, "-o",
                os.path.join(tmp, "o.png"), "--dump-hits", dump)
        if r.returncode != 0:
            return [f"cli seed {seed} ({n_obj} objects): {ext}: exit code {r.returncode}: {r.stderr.strip()[-200:]}"]
        hits[ext] = np.fromfile(dump, dtype=abi.HIT_DTYPE).reshape(h, w)
        r = run("-f", os.path.join(tmp, f"s.{ext}"), "-p", 3, "-a", 2, "-s", 4, "-b", 3, "--width", w, "--height", h, "--seed", seed, "-o", out)
        if r.returncode != 0:
            return [f"cli seed {seed} ({n_obj} objects): {ext} PBR: exit code {r.returncode}: {r.stderr.strip()[-200:]}"]
        frames[ext] = image_io.load_hdr(out)[..., :3]
    g = H.make_gpu(model, None, H.settings(program=0, spp=1, bounces=4, seed=seed), w, h)
    exact, _ = H.hits_equal(g.render_hits(w, h), hits["ptsc"])
    if not exact.all():
        fails.append(f"container: {int((~exact).sum())} hit records differ from the Python mirror's")
    a, b = hits["ptsc"], hits["glb"]
    both = (a["hit"] == 1) & (b["hit"] == 1)
    same_ids = (a["hit"] == b["hit"]) & (~both | ((a["triangle_index"] == b["triangle_index"]) & (a["object_index"] == b["object_index"])))
    close = ~both | (np.abs(a["t"] - b["t"]) <= 1e-4 * np.maximum(1.0, np.abs(a["t"])))
    if same_ids.mean() < 0.999 or (same_ids & ~close).any():
        fails.append(f"GLB: {int((~same_ids).sum())} pixels name another triangle, {int((same_ids & ~close).sum())} distances off")
    # (a path that differs in its last bits after the first bounce may end on a lamp or not: the pixels that differ at all
    # are few, but those can differ a lot)
    d = np.abs(frames["glb"] - frames["ptsc"]).max(axis=-1)
    tol = 1e-2 * np.maximum(frames["ptsc"].max(axis=-1), 1e-3)  # (Radiance files keep 8 bits of the largest channel)
    if (d > tol).mean() > 0.15:
        fails.append(f"GLB PBR frame: {(d > tol).mean() * 100:.1f} % of the pixels differ, mean difference {d.mean():.3g} of mean {frames['ptsc'].mean():.3g}")
    return [f"cli seed {seed} ({n_obj} objects): {f}" for f in fails]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:100000")
    ap.add_argument("--seconds", type=float, default=180.0)
    args = ap.parse_args()
    a, b = (int(x) for x in args.seeds.split(":"))
    t0, failures, done = time.time(), [], 0
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(a, b):
            if time.time() - t0 > args.seconds:
                break
            f = one_seed(seed, tmp)
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
