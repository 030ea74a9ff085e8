"""Error paths of the C ABI on the GPU: a failed device allocation must come back as PTGPU_E_NOMEM and leave a
context that still renders — never a context whose recorded buffer sizes claim memory that is not there."""
from __future__ import annotations

import os

import numpy as np
import pytest

import helpers as H
from pathtracer_amd import scenes
from pathtracer_amd._lib import PtgpuError

pytestmark = pytest.mark.gpu

E_NOMEM = -5


def test_failed_allocations_leave_a_usable_context():
    model = scenes.tiny_scene()
    env = scenes.synthetic_env()
    st = H.settings(program=2, spp=2, bounces=3)
    w, h = 64, 40
    gpu = H.make_gpu(model, env, st, w, h)
    ref, ref_stats = H.gpu_render(gpu, w, h)
    failures = 0
    try:
        # every render below needs larger work buffers than the one before, so each of them allocates again: fail the
        # k-th allocation of the call for every k a render can reach (result image, three arrays per ray pool, hit
        # records, spill, cursors, candidate lists, ...)
        for k in range(1, 17):
            gpu.set_render_settings(H.settings(program=2, spp=2 + k, bounces=3))
            os.environ["PTGPU_TEST_FAIL_ALLOC"] = f"{k}:{k}"
            big_w, big_h = w + 64 * k, h + 40 * k
            try:
                H.gpu_render(gpu, big_w, big_h)
            except PtgpuError as e:
                assert e.code == E_NOMEM, e
                failures += 1
            os.environ["PTGPU_TEST_FAIL_ALLOC"] = ""
            # a smaller frame right after the failure: the sizes kept in the context must not point at freed buffers
            gpu.set_render_settings(st)
            img, stats = H.gpu_render(gpu, w, h)
            assert stats.rays == ref_stats.rays
            assert np.abs(img - ref).max() <= 2e-5 * max(1.0, float(ref.max()))
    finally:
        os.environ.pop("PTGPU_TEST_FAIL_ALLOC", None)
    assert failures >= 6, failures
