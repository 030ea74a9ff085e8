"""CPU tests of the parity oracle: golden-vector regression, internal cross-checks, and the one
reference-held pin (the Fibonacci sequence of the reference's tests/fibonacci_sequence.py)."""
import os

import numpy as np
import pytest

import helpers as H
from oracle import binding as ob
from pathtracer_amd import scenes

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLD, "oracle_golden.npz"))


def test_fibonacci_matches_reference_script():
    """oracle fibonacci2D (brdf.cc:73-78, binary32, GOLDEN_RATIO = 1.618034f) against the 256 points the
    reference's own double-precision script produces; tolerance 1e-4 on a wrapped [0,1) difference:
    (i+1) * |1.618034f - phi| <= 3e-5 plus one binary32 ulp of (i+1)*phi <= 3e-5."""
    ref = np.load(os.path.join(GOLD, "fibonacci_reference_256.npy"))
    out = np.zeros((256, 2), np.float32)
    for i in range(256):
        ob.lib().orc_kat_fibonacci2d(i, np.float32(1.0 / 256), out[i].ctypes.data_as(ob.abi.c_float_p))
    assert np.allclose(out[:, 0], ref[:, 0], atol=1e-7)
    dy = np.abs(out[:, 1].astype(np.float64) - ref[:, 1])
    dy = np.minimum(dy, 1.0 - dy)
    assert dy.max() < 1e-4


def test_schlick_and_f0_against_reference_script():
    """oracle fresnel_schlick (ray_program.cc:40-45: mix(f0, f90, (1 - c)^5), binary32) and f0_dielectric
    (brdf.cc:70-72) against the 91 angles the reference's own double-precision script tests/fresnel_compare.py
    evaluates for air -> glass.  Tolerance 4e-7 absolute: the values are in [0.04, 1], binary32 keeps 6e-8 there, and
    the product x^5 is formed in four roundings (x2 = x*x, x2*x2, *x, then the mix)."""
    from pathtracer_amd.model import f0_dielectric
    ref = np.load(os.path.join(GOLD, "fresnel_reference.npy"))
    f0_ref = ref[0, 3]
    f0 = f0_dielectric(1.5, 1.0)
    assert f0.dtype == np.float32 and abs(float(f0) - f0_ref) < 4e-9
    one = np.ones(3, np.float32)
    f0v = np.full(3, f0, np.float32)
    out = np.zeros(3, np.float32)
    worst = 0.0
    for angle, schlick, exact, _ in ref:
        c = np.float32(np.cos(angle))
        ob.lib().orc_kat_schlick(f0v.ctypes.data_as(ob.abi.c_float_p), one.ctypes.data_as(ob.abi.c_float_p), c,
                                 out.ctypes.data_as(ob.abi.c_float_p))
        assert out[0] == out[1] == out[2]
        worst = max(worst, abs(float(out[0]) - schlick))
    assert worst < 4e-7, worst
    # sanity of the vector itself: Schlick is an approximation of the exact curve, equal at 0 and 90 degrees
    assert abs(ref[0, 1] - ref[0, 2]) < 1e-12 and abs(ref[-1, 1] - ref[-1, 2]) < 1e-12
    assert np.abs(ref[:, 1] - ref[:, 2]).max() < 0.05


def test_golden_kat_tables(golden):
    import sys
    sys.path.insert(0, GOLD)
    import make_golden
    now = make_golden.kat_tables()
    for k, v in now.items():
        assert np.array_equal(np.asarray(v), golden[k], equal_nan=True), k


@pytest.mark.parametrize("spp", [1, 64, 256, 300, 4096])
def test_golden_subsamples(golden, spp):
    s = ob.subsample_positions(spp)
    assert np.array_equal(s, golden[f"subsamples_{spp}"])
    assert (s >= 0).all() and (s < 1).all()
    if int(np.sqrt(spp)) ** 2 == spp:  # stratified grid (renderer.cc:141-147)
        n = int(np.sqrt(spp))
        assert np.allclose(np.sort(np.unique(s[:, 0])), (np.arange(n) + 0.5) / n)


def test_golden_tiny_scene(golden):
    tiny = scenes.tiny_scene()
    env = scenes.synthetic_env()
    w, h = 64, 36
    o = H.make_oracle(tiny, env, H.settings(program=0, spp=1), w, h)
    m, org = o.get_camera_raw()
    assert np.array_equal(m, golden["tiny_ndc2world"]) and np.array_equal(org, golden["tiny_origin"])
    hits = o.render_hits(w, h)
    g = golden["tiny_hits_64x36"]
    for f in hits.dtype.names:
        assert np.array_equal(hits[f], g[f]), f
    assert np.array_equal(o.render_frame(w, h)[0], golden["tiny_raycast_64x36"])
    ao = H.make_oracle(tiny, env, H.settings(program=1, spp=64, bounces=1, newrays=8), w, h)
    assert np.array_equal(ao.render_frame(w, h)[0], golden["tiny_ao_64spp"])
    pbr = H.make_oracle(tiny, env, H.settings(program=2, spp=64, bounces=4), w, h)
    assert np.array_equal(pbr.render_frame(w, h)[0], golden["tiny_pbr_64spp"])


def test_golden_on_hit_vectors(golden):
    import sys
    sys.path.insert(0, GOLD)
    import make_golden
    now = make_golden.on_hit_vectors(scenes.tiny_scene(), scenes.synthetic_env(), 64, 36)
    g = golden["tiny_on_hit_pbr"]
    assert now.shape == g.shape and np.array_equal(now, g, equal_nan=True)
    spawned = g[:, 5]
    assert spawned.max() >= 3 and spawned.min() == 0  # misses spawn nothing, transmissive hits spawn three rays


def test_naive_and_bvh_agree():
    """NaiveAS (brute force) and BVH_AS must return the same closest hits: the self-check that the
    restated BVH build and traversal lose no triangle."""
    model = scenes.readme_standin(detail=0.05)
    env = scenes.synthetic_env()
    w, h = 80, 45
    a = H.make_oracle(model, env, H.settings(program=0, as_type=0), w, h).render_hits(w, h)
    b = H.make_oracle(model, env, H.settings(program=0, as_type=1), w, h).render_hits(w, h)
    exact, ids = H.hits_equal(a, b)
    assert exact.all()
    rng = np.random.default_rng(5)
    o = rng.uniform(-4, 4, (4000, 3)).astype(np.float32)
    o[:, 1] = np.abs(o[:, 1]) + 0.2
    d = rng.normal(size=(4000, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    na = H.make_oracle(model, env, H.settings(program=0, as_type=0), w, h)
    bv = H.make_oracle(model, env, H.settings(program=0, as_type=1), w, h)
    exact, _ = H.hits_equal(na.intersect_rays(o, d), bv.intersect_rays(o, d))
    assert exact.all()
    any_flags = np.ones(len(o), np.uint8)
    ha, hb = na.intersect_rays(o, d, any_flags), bv.intersect_rays(o, d, any_flags)
    assert np.array_equal(ha["hit"], hb["hit"])  # any-hit: the same rays are occluded, whichever triangle is found


def test_bvh_statistics():
    model = scenes.readme_standin(detail=0.1)
    o = H.make_oracle(model, None, H.settings(program=0, leaf=8), 32, 18)
    for m in range(len(model.meshes)):
        info = o.bvh_info(m)
        assert info.n_triangles == model.meshes["n_indices"][m] // 3
        assert info.n_leaves == info.n_nodes + 1
        assert info.max_tris_in_leaf <= 8
        assert abs(info.mean_tris_in_leaf * info.n_leaves - info.n_triangles) < 1e-3 * max(1, info.n_triangles)


def test_portable_math_close_to_libm(golden):
    """The polynomial sin/cos/atan2/asin kernels shared with the GPU stay within 2e-6 of libm on the
    quantities the renderer consumes (unit directions, environment colours up to ~30)."""
    assert np.abs(golden["cos_dir"][:, 0] - golden["cos_dir"][:, 1]).max() < 2e-6
    assert np.abs(golden["ggx"][:, 0] - golden["ggx"][:, 1]).max() < 2e-6
    e = golden["env_res"]
    assert np.abs(e[:, 0] - e[:, 1]).max() < 1e-3 * max(1.0, np.abs(e).max())


def test_portable_environment_coordinates_close_to_libm():
    """sample_environment (brdf.cc:62-69) with the polynomial atan2 / asin shared with the GPU against the same function
    with libm, on the lookup coordinates themselves: the environment is a ramp (red = (x + 0.5) / width, green =
    (y + 0.5) / height), which bilinear filtering reproduces exactly, so the sampled colour IS (u, v) away from the wrap
    seam.  Bound 5e-7 = 4 binary32 ulps of a coordinate near 1 (measured: 99 % of directions within 1.2e-7)."""
    w, h = 256, 128
    env = np.zeros((h, w, 4), np.float32)
    env[..., 0] = ((np.arange(w) + 0.5) / w)[None, :]
    env[..., 1] = ((np.arange(h) + 0.5) / h)[:, None]
    rng = np.random.default_rng(3)
    d = rng.normal(size=(4000, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    fp = ob.abi.c_float_p
    out = np.zeros((2, len(d), 4), np.float32)
    for mode in (ob.MATH_LIBM, ob.MATH_PORTABLE):
        for i in range(len(d)):
            ob.lib().orc_kat_sample_environment(env.ctypes.data_as(fp), w, h, d[i].ctypes.data_as(fp), np.float32(0.0), mode,
                                                out[mode, i].ctypes.data_as(fp))
    angle = np.arctan2(-d[:, 2].astype(np.float64), -d[:, 0].astype(np.float64))
    away = (np.abs(angle) < np.pi - 0.05) & (np.abs(angle) > 0.05) & (np.abs(d[:, 1]) < 0.99)   # not at the seam, not at a pole
    assert away.mean() > 0.8
    assert np.abs(out[0, away, 0] - out[1, away, 0]).max() < 5e-7
    assert np.abs(out[0, away, 1] - out[1, away, 1]).max() < 5e-7
    # and the ramp does reproduce the coordinate: u of the libm evaluation against the formula in double precision
    u = (angle / np.pi) * 0.5 + 0.5
    got = out[0, :, 0].astype(np.float64)
    du = np.abs(got - u)
    assert np.minimum(du, np.abs(du - 1.0))[away].max() < 2e-6


def test_sums_and_mean_agree():
    model = scenes.tiny_scene()
    env = scenes.synthetic_env()
    w, h = 32, 18
    o = H.make_oracle(model, env, H.settings(program=2, spp=9, bounces=3), w, h)
    mean, st = o.render_frame(w, h)
    sums, _ = o.render_frame(w, h, sums=True)
    assert (sums[..., 3] == 9).all()
    assert np.allclose(sums[..., :3] / 9, mean[..., :3], rtol=1e-5, atol=1e-6)
    assert st.camera_samples == w * h * 9 and st.rays > st.camera_samples


def test_counter_and_minstd_samplers_converge_to_the_same_image():
    """The counter-based sampler replaces the reference's minstd_rand: both estimate the same image.
    Two independent 256-spp renders of a 24x14 frame must differ by noise only: the RMSE between them
    is below 2.5x the RMSE between two minstd renders with different seeds."""
    model = scenes.tiny_scene()
    env = scenes.synthetic_env(16, 8)
    w, h = 24, 14
    imgs = {}
    for name, rng_mode, seed in (("counter", ob.RNG_COUNTER, 1), ("minstd_a", ob.RNG_MINSTD, 2), ("minstd_b", ob.RNG_MINSTD, 3)):
        o = H.make_oracle(model, env, H.settings(program=2, spp=256, bounces=3, seed=seed), w, h,
                          math_mode=ob.MATH_LIBM, rng_mode=rng_mode)
        imgs[name] = np.minimum(o.render_frame(w, h)[0][..., :3], 4.0)

    def rmse(a, b):
        return float(np.sqrt(np.mean((a - b) ** 2)))
    noise = rmse(imgs["minstd_a"], imgs["minstd_b"])
    assert rmse(imgs["counter"], imgs["minstd_a"]) < 2.5 * noise + 1e-3
    assert abs(imgs["counter"].mean() - imgs["minstd_a"].mean()) < 0.05 * imgs["minstd_a"].mean()
