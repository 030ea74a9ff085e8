"""The C++ host (Renderer / Viewer / PathTracer command line) above the C ABI.

CPU tests: the command line keeps the reference's contract for usage errors (src/main.cc:16-19, 27-30: exit
code 1) and fails loudly, not silently, when no GPU is there.  GPU tests: the command line renders the same
pixels and hit records as the Python mirror of the same interface, and its PNG / Radiance writers produce what
the Python writers produce.
"""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import helpers as H
from pathtracer_amd import abi, image_io, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "pathtracer_amd", "bin", "PathTracer")

needs_cli = pytest.mark.skipif(not os.path.exists(CLI), reason="C++ host not built (run __graft_entry__.build())")


def run(*args):
    return subprocess.run([CLI, *map(str, args)], capture_output=True, text=True, timeout=300)


def read_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(tag + body) & 0xffffffff)
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 6)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * 4)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(h, w, 4)


@needs_cli
def test_cli_usage_errors():
    r = run("-h")
    assert r.returncode == 1 and "Usage: PathTracer" in r.stdout
    r = run()  # no -f: the short message on stdout, exit code 1 (src/arguments.cc:136-139, src/main.cc:18-20)
    assert r.returncode == 1 and "Use -h or --help for usage info." in r.stdout
    r = run("-f", "/nonexistent/scene.ptsc")
    assert r.returncode == 1 and "Failed to load model from" in r.stderr
    # out-of-range values are repaired, not rejected (src/arguments.cc:14-18, 121-132): the run gets as far as loading
    r = run("-f", "/nonexistent/scene.ptsc", "-p", "7", "-a", "0", "-s", "0", "-b", "-1", "-r", "-2", "-m", "0", "--width", "0",
            "--height", "-5")
    assert r.returncode == 1 and "Failed to load model from" in r.stderr and "out of range" not in r.stderr
    r = run("-f", "x", "--no-such-flag")
    assert r.returncode == 1 and "unknown argument" in r.stderr


@needs_cli
def test_cli_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    scene = tmp_path / "tiny.ptsc"
    scenes.tiny_scene().save(str(scene))
    r = run("-f", scene, "-o", tmp_path / "o.hdr", "--width", 16, "--height", 8)
    assert r.returncode == 1 and "ptgpu_create failed" in r.stderr
    assert not (tmp_path / "o.hdr").exists()
    # an empty environment path is the default (white) environment, not a file to open (src/arguments.cc:140-144): the run
    # gets past the environment and fails at the same place
    r = run("-f", scene, "-e", "", "-o", tmp_path / "o.hdr", "--width", 16, "--height", 8)
    assert r.returncode == 1 and "ptgpu_create failed" in r.stderr and "loaded in" in r.stdout


@needs_cli
@pytest.mark.gpu
def test_cli_matches_python_mirror(tmp_path):
    model = scenes.tiny_scene()
    scene = tmp_path / "tiny.ptsc"
    model.save(str(scene))
    w, h = 64, 40
    hits_path, png, hdr = tmp_path / "hits.bin", tmp_path / "o.png", tmp_path / "o.hdr"
    # config "plumbing": ray caster, BVH, 1 sample, white environment
    r = run("-f", scene, "-p", 1, "-a", 2, "-s", 1, "--width", w, "--height", h, "--seed", 1234, "-o", png, "--dump-hits",
            hits_path, "--stats")
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "loaded in" in out and "rendered in" in out and "output saved in" in out
    g = H.make_gpu(model, None, H.settings(program=0, spp=1, bounces=4), w, h)
    img, _ = H.gpu_render(g, w, h)
    assert np.array_equal(read_png(png), image_io.float_pixels_to_srgb8(img))
    hits = np.fromfile(hits_path, dtype=abi.HIT_DTYPE).reshape(h, w)
    ref = g.render_hits(w, h)
    exact, _ = H.hits_equal(ref, hits)
    assert exact.all()
    # PBR with the HDR environment file, Radiance output
    env_path = os.path.join(ROOT, "assets", "farmland_overcast_1k.hdr")
    r = run("-f", scene, "-e", env_path, "-p", 3, "-a", 2, "-s", 4, "-b", 3, "--width", w, "--height", h, "--seed", 7, "-o", hdr)
    assert r.returncode == 0, r.stderr
    g = H.make_gpu(model, image_io.load_hdr(env_path), H.settings(program=2, spp=4, bounces=3, seed=7), w, h)
    img, _ = H.gpu_render(g, w, h)
    got = image_io.load_hdr(str(hdr))
    # RGBE keeps 8 mantissa bits of the largest channel; float atomics make the two renders differ in the last bits
    scale = np.maximum(img[..., :3].max(axis=-1, keepdims=True), 1e-6)
    assert np.abs(got[..., :3] - img[..., :3]).max() <= (scale / 100.0).max()
    assert (np.abs(got[..., :3] - img[..., :3]) <= scale / 100.0).all()
    # the two additive switches: trees built on the GPU and the single-path estimator (hit dump: ray caster records)
    r = run("-f", scene, "-p", 3, "-a", 3, "-s", 16, "-b", 5, "--single-path", "--width", w, "--height", h, "--seed", 9, "-o", hdr,
            "--dump-hits", hits_path, "--stats")
    assert r.returncode == 0, r.stderr
    g = H.make_gpu(model, None, H.settings(program=2, spp=16, bounces=5, seed=9, as_type=2, single=True), w, h)
    img, st = H.gpu_render(g, w, h)
    assert f"rays {st.rays} " in r.stdout
    got = image_io.load_hdr(str(hdr))
    scale = np.maximum(img[..., :3].max(axis=-1, keepdims=True), 1e-6)
    assert (np.abs(got[..., :3] - img[..., :3]) <= scale / 100.0).all()
    exact, _ = H.hits_equal(g.render_hits(w, h), np.fromfile(hits_path, dtype=abi.HIT_DTYPE).reshape(h, w))
    assert exact.all()


@needs_cli
@pytest.mark.gpu
def test_cli_repairs_out_of_range_values_like_the_reference(tmp_path):
    """-p 9 is the ray caster, -a 0 the naive structure, -s 0 one sample, -b -1 no bounce, -m 0 one triangle per leaf,
    a width or height below 1 is 1 (src/arguments.cc:14-18, 121-132)."""
    scene = tmp_path / "tiny.ptsc"
    scenes.tiny_scene().save(str(scene))
    a, b = tmp_path / "a.png", tmp_path / "b.png"
    r = run("-f", scene, "-p", 9, "-a", 0, "-s", 0, "-b", -1, "-r", -3, "-m", 0, "--width", 48, "--height", 32, "-o", a)
    assert r.returncode == 0, r.stderr
    r = run("-f", scene, "-p", 1, "-a", 1, "-s", 1, "-b", 0, "-r", 0, "-m", 1, "--width", 48, "--height", 32, "-o", b)
    assert r.returncode == 0, r.stderr
    assert np.array_equal(read_png(a), read_png(b))
    # -e "" is the white default environment (src/arguments.cc:140-144, arguments.h:32)
    r = run("-f", scene, "-p", 3, "-s", 2, "-b", 2, "--seed", 3, "-e", "", "--width", 48, "--height", 32, "-o", a)
    assert r.returncode == 0, r.stderr
    r = run("-f", scene, "-p", 3, "-s", 2, "-b", 2, "--seed", 3, "-e", "white", "--width", 48, "--height", 32, "-o", b)
    assert r.returncode == 0, r.stderr
    assert np.abs(read_png(a).astype(int) - read_png(b).astype(int)).max() <= 1
    r = run("-f", scene, "-p", 1, "--width", 0, "--height", -4, "-o", a)
    assert r.returncode == 0, r.stderr
    assert read_png(a).shape == (1, 1, 4)


@needs_cli
@pytest.mark.gpu
def test_cli_multi_gpu_path_with_a_communicator_of_one(tmp_path):
    """--gpus N splits every pixel's samples over N devices and adds the partial sums with an RCCL reduce.  A box with one
    GPU can still run the whole path — contexts, per-device sum buffer and stream, ncclCommInitAll, ncclReduce, the
    division on the first device — with one device (--collective); the frame must be the plain one up to the order of
    the float additions.  More devices than the node has is an error, not a silent fallback."""
    import torch
    scene = tmp_path / "tiny.ptsc"
    scenes.tiny_scene().save(str(scene))
    w, h = 64, 40
    common = ["-f", scene, "-p", 3, "-a", 2, "-s", 6, "-b", 3, "--width", w, "--height", h, "--seed", 11, "--stats"]
    a, b = tmp_path / "plain.hdr", tmp_path / "reduced.hdr"
    r1 = run(*common, "-o", a)
    assert r1.returncode == 0, r1.stderr
    r2 = run(*common, "--gpus", 1, "--collective", "-o", b)
    assert r2.returncode == 0, r2.stderr
    rays = [ln.split()[1] for ln in (r1.stdout + r2.stdout).splitlines() if ln.startswith("rays ")]
    assert len(rays) == 2 and rays[0] == rays[1]
    # the acceleration structure is built once per load_scene however many devices render (src/renderer.cc:41-64)
    assert "acceleration structure builds 1 devices 1" in r2.stdout
    x, y = image_io.load_hdr(str(a))[..., :3], image_io.load_hdr(str(b))[..., :3]
    scale = np.maximum(x.max(axis=-1, keepdims=True), 1e-6)
    assert (np.abs(x - y) <= scale / 100.0).all()      # RGBE keeps 8 bits of the largest channel
    n = torch.cuda.device_count()
    r3 = run(*common, "--gpus", n + 1, "-o", b)
    assert r3.returncode == 1 and "device index out of range" in r3.stderr
    if n >= 2:   # a node with several GPUs: the real split
        r4 = run(*common, "--gpus", 2, "-o", b)
        assert r4.returncode == 0, r4.stderr
        y = image_io.load_hdr(str(b))[..., :3]
        assert (np.abs(x - y) <= scale / 100.0).all()
        assert "acceleration structure builds 1 devices 2" in r4.stdout   # the second device copied the first one's scene


@needs_cli
@pytest.mark.gpu
def test_cli_renders_glb_like_the_scene_container(tmp_path):
    """-f scene.glb goes through the C++ glTF reader; the same scene as a flat container must give the same picture
    (normal matrices are recomputed in binary32 by the reader, hence a tolerance instead of equality)."""
    from pathtracer_amd import gltf_export
    model = scenes.tiny_scene()
    model.save(str(tmp_path / "s.ptsc"))
    gltf_export.save_glb(model, str(tmp_path / "s.glb"))
    outs = {}
    for ext in ("ptsc", "glb"):
        out = tmp_path / f"{ext}.hdr"
        r = run("-f", tmp_path / f"s.{ext}", "-p", 3, "-a", 2, "-s", 8, "-b", 3, "--width", 64, "--height", 40, "--seed", 5, "-o", out)
        assert r.returncode == 0, r.stderr
        outs[ext] = image_io.load_hdr(str(out))[..., :3]
    diff = np.abs(outs["glb"] - outs["ptsc"])
    assert diff.mean() < 2e-3 * max(1.0, outs["ptsc"].mean()) and np.median(diff) == 0.0
    # a node scaled to nothing (an exporter's way of hiding an object: no inverse, a normal matrix of infinities and NaNs):
    # the scene must load through both readers and look as if the object were not there
    import copy
    hidden = copy.deepcopy(model)
    extra = hidden.objects[:1].copy()
    extra["model_matrix"][0] = np.diag([0.0, 0.0, 0.0, 1.0]).astype(np.float32).T.reshape(-1)
    extra["normal_matrix"][0] = np.nan
    hidden.objects = np.concatenate([extra, hidden.objects])
    hidden.save(str(tmp_path / "h.ptsc"))
    gltf_export.save_glb(hidden, str(tmp_path / "h.glb"))
    for ext in ("ptsc", "glb"):
        out = tmp_path / f"h_{ext}.hdr"
        r = run("-f", tmp_path / f"h.{ext}", "-p", 3, "-a", 2, "-s", 8, "-b", 3, "--width", 64, "--height", 40, "--seed", 5, "-o", out)
        assert r.returncode == 0, r.stderr
        got = image_io.load_hdr(str(out))[..., :3]
        d = np.abs(got - outs[ext])
        assert d.mean() < 2e-3 * max(1.0, outs[ext].mean()) and np.median(d) == 0.0
