"""ptgpu_intersect_rays_static — the entry point that allocates nothing — against ptgpu_intersect_rays, through the
C ABI: bit-identical hit records on two calls in a row, the second with a smaller input (all calls on one GPU share
one built-in work array, which keeps the last call's contents)."""
from __future__ import annotations

import numpy as np
import pytest

import helpers as H
from pathtracer_amd import abi, scenes
from pathtracer_amd._lib import PtgpuError

pytestmark = pytest.mark.gpu


def random_rays(model, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = model.vertices["position"].min(axis=0) - 2.0, model.vertices["position"].max(axis=0) + 2.0
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    target = rng.uniform(lo * 0.4, hi * 0.4, (n, 3)).astype(np.float32)
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    any_hit = (rng.uniform(size=n) < 0.3).astype(np.uint8)
    return o, d.astype(np.float32), any_hit


def static_hits(gpu, o, d, any_hit):
    import torch
    n = len(o)
    t_o = torch.from_numpy(o).cuda()
    t_d = torch.from_numpy(d).cuda()
    t_a = torch.from_numpy(any_hit).cuda() if any_hit is not None else None
    out = torch.zeros(n * 8, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    gpu.intersect_rays_static(n, t_o.data_ptr(), t_d.data_ptr(), t_a.data_ptr() if t_a is not None else 0, out.data_ptr())
    return out.cpu().numpy().view(abi.HIT_DTYPE)


@pytest.mark.parametrize("wide", [False, True])
def test_static_entry_point_equals_intersect_rays(wide):
    model = scenes.readme_standin(detail=0.08)
    gpu = H.make_gpu(model, None, H.settings(program=0, wide=wide), 64, 64)
    for n, seed in ((100_000, 1), (7_321, 2), (abi.STATIC_MAX_RAYS, 3), (1, 4)):   # large, then smaller, the maximum, one ray
        o, d, any_hit = random_rays(model, n, seed)
        ref = gpu.intersect_rays(o, d, any_hit)
        got = static_hits(gpu, o, d, any_hit)
        assert ref["hit"].sum() > 0 or n == 1
        for field in abi.HIT_DTYPE.names:
            assert np.array_equal(ref[field].view(np.uint32), got[field].view(np.uint32)), (n, field)
    o, d, _ = random_rays(model, 5000, 9)   # no any-hit array: closest hits
    ref = gpu.intersect_rays(o, d)
    got = static_hits(gpu, o, d, None)
    assert ref.tobytes() == got.tobytes()
    with pytest.raises(PtgpuError, match="built-in work array holds"):
        gpu.intersect_rays_static(abi.STATIC_MAX_RAYS + 1, 8, 8, 0, 8)


def test_static_entry_point_against_the_oracle():
    """The same records against the CPU oracle (not only against the allocating entry point): closest-hit and any-hit rays,
    binary nodes, so the any-hit records are the reference's too."""
    model = scenes.readme_standin(detail=0.05)
    st = H.settings(program=0)
    gpu = H.make_gpu(model, None, st, 64, 64)
    orc = H.make_oracle(model, None, st, 64, 64)
    o, d, any_hit = random_rays(model, 40_000, 11)
    got = static_hits(gpu, o, d, any_hit)
    ref = orc.intersect_rays(o, d, any_hit)
    exact, _ = H.hits_equal(ref, got)
    assert ref["hit"].sum() > 1000 and exact.all(), f"{(~exact).sum()} of {len(o)} records differ from the oracle's"


def test_static_entry_point_on_a_caller_stream_without_waiting():
    """wait = 0 on a stream of the caller's: the call returns after queueing, the records are there once the stream has been
    waited for, and the device's free memory is the same before and after (nothing was allocated on the way)."""
    import torch
    model = scenes.readme_standin(detail=0.05)
    gpu = H.make_gpu(model, None, H.settings(program=0, wide=True), 64, 64)
    o, d, any_hit = random_rays(model, 50_000, 21)
    ref = gpu.intersect_rays(o, d, any_hit)
    n = len(o)
    t_o, t_d, t_a = torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda(), torch.from_numpy(any_hit).cuda()
    out = torch.zeros(n * 8, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    torch.cuda.synchronize()
    gpu.intersect_rays_static(n, t_o.data_ptr(), t_d.data_ptr(), t_a.data_ptr(), out.data_ptr())  # (first call on this context)
    out.zero_()
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        gpu.intersect_rays_static(n, t_o.data_ptr(), t_d.data_ptr(), t_a.data_ptr(), out.data_ptr(), stream=stream.cuda_stream, wait=False)
    stream.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after == free_before, (free_before, free_after)
    got = out.cpu().numpy().view(abi.HIT_DTYPE)
    assert ref.tobytes() == got.tobytes()
