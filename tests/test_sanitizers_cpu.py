"""The CPU code under AddressSanitizer + UndefinedBehaviorSanitizer (tools/sanitize, built by __graft_entry__.build()):
the parity oracle through every program mode, the C++ host's file readers over well-formed and corrupted files, and
the library's host tree builder.  A sanitizer report ends those programs with a non-zero status and its text on stderr."""
import os
import subprocess

import pytest

from pathtracer_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "sanitize", "build")


def tool(name):
    path = os.path.join(BIN, name)
    if not os.path.exists(path):
        pytest.skip(f"{path} not built (make -C tools/sanitize)")
    return path


def run(cmd, timeout=600):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.returncode, r.stderr[-2000:])
    return r.stdout


def test_oracle_under_sanitizers(tmp_path):
    scene = tmp_path / "tiny.ptsc"
    scenes.tiny_scene().save(str(scene))
    out = run([tool("san_oracle"), str(scene)])
    lines = dict(ln.split(" ", 1) for ln in out.strip().splitlines())
    assert set(lines) == {"raycast", "raycast-naive", "ao", "pbr", "pbr-libm-minstd", "pbr-single"}
    assert lines["raycast"] == lines["raycast-naive"]   # NaiveAS and BVH_AS give the same picture
    material = tmp_path / "material.ptsc"
    scenes.material_stress(detail=0.05).save(str(material))
    run([tool("san_oracle"), str(material)])


def test_host_readers_under_sanitizers(tmp_path):
    from mutations import mutated_files
    paths = mutated_files(tmp_path)
    out = run([tool("san_host")] + paths)
    accepted, refused = (int(x) for x in out.split()[1::2])
    assert accepted >= 4 and refused >= 20 and accepted + refused == len(paths)


def test_tree_builder_under_sanitizers():
    out = run([tool("san_bvh")])
    assert "far leaf 1 form 0" in out and "flat leaf 8 form 2" in out and "empty leaf 64 form 1" in out
    # the whole scene build with four-wide nodes (collapse, quantisation, top nodes, instance tree) ran and its invariants held:
    # 24 regular builds + 8 with the leaves as 64-byte records (every record's triangles equal to the 48-byte ones, bit for
    # bit) + 6 with one object scaled to nothing (left out of the instances) + 24 on coordinates that validation refuses
    # (those must stay inside their arrays all the same)
    last = out.strip().splitlines()[-1]
    assert last.startswith("scene builds: wide nodes walked ") and last.endswith("builds 62"), last
    assert int(last.split()[5].rstrip(",")) > 10000
