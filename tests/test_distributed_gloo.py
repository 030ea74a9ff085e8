"""World-size-2 test of the sample-split path on CPU (gloo): each rank renders its sample subset
(with the CPU oracle standing in for the GPU kernels), partial sums are reduced to rank 0 by the
package's own combine step, and the result must equal the single-process frame."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, Hh, SPP = 24, 14, 7


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers as H
    from pathtracer_amd import scenes
    from pathtracer_amd.distributed import combine_partial_sums, samples_of_rank, subset_settings
    model = scenes.tiny_scene()
    env = scenes.synthetic_env(16, 8)
    st = subset_settings(H.settings(program=2, spp=SPP, bounces=3), rank, world)
    o = H.make_oracle(model, env, st, W, Hh)
    sums, stats = o.render_frame(W, Hh, sums=True)
    assert (sums[..., 3] == samples_of_rank(SPP, rank, world)).all()
    buf = torch.from_numpy(sums.copy())
    img = combine_partial_sums(buf, SPP, rank, world)
    if rank == 0:
        np.save(out_path, img.numpy())
    else:
        assert img is None
    dist.destroy_process_group()


def test_two_rank_sample_split_equals_single_render(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    from pathtracer_amd import scenes
    out = str(tmp_path / "img.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    split = np.load(out)
    model = scenes.tiny_scene()
    env = scenes.synthetic_env(16, 8)
    o = H.make_oracle(model, env, H.settings(program=2, spp=SPP, bounces=3), W, Hh)
    full, _ = o.render_frame(W, Hh)
    assert (split[..., 3] == 1).all()
    assert np.allclose(split[..., :3], full[..., :3], rtol=2e-5, atol=1e-6)


def test_subset_settings_partition_the_samples():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    from pathtracer_amd.distributed import samples_of_rank, subset_settings
    st = H.settings(spp=300)
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            s = subset_settings(st, r, world)
            seen += list(range(s.sampleBegin, 300, s.sampleStep))
            assert samples_of_rank(300, r, world) == len(range(r, 300, world))
        assert sorted(seen) == list(range(300))


def test_subset_tiles_partition_the_tiles():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    from pathtracer_amd.distributed import subset_tiles
    st = H.settings(spp=16)
    n_tiles = ((3840 + 7) // 8) * ((2160 + 7) // 8)
    for world in (1, 2, 4, 8):
        seen = 0
        for r in range(world):
            s = subset_tiles(st, r, world)
            assert (s.sampleBegin, s.sampleStep) == (0, 1) and s.to_abi().tile_step == world
            seen += len(range(s.tileBegin, n_tiles, s.tileStep))
        assert seen == n_tiles
    with pytest.raises(ValueError):
        subset_tiles(st, 2, 2)
