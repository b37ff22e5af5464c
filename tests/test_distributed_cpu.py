"""N > 1 path on CPU: two gloo ranks shard the sample window exactly as the GPU
ranks do (goblin_amd/distributed.py), each accumulates its own full-size film,
one all-reduce(sum) rebuilds the whole film.  The per-rank compute here is the
oracle (this is a test of sharding + reduction, which have no GPU dependency)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from goblin_amd import distributed as gd
from goblin_amd import scene as gs

OV = gs.config_overrides(resolution=(40, 24), spp=4, depth=4)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, mode, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import oracle_binding as ob
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    r, lr, w = gd.init(backend="gloo")
    assert (r, w) == (rank, world)
    scene = gs.load_scene("bunny", OV)
    oracle = ob.Oracle(scene)
    part = gd.shard_for(rank, world, mode, base_seed=100)
    film = np.zeros((scene.desc.film.yres, scene.desc.film.xres, 4), np.float32)
    if part["shard"] is None:
        windows = [oracle.window()]
    else:
        windows = gd.tiles_of(oracle.window(), *part["shard"])
    paths = 0
    for win in windows:
        samples = oracle.native_samples(part["seed"], window=win)
        li, _ = oracle.li_replay(samples)
        oracle.splat(samples, li, film)
        paths += samples.shape[0]
    t = torch.from_numpy(film)
    gd.allreduce_film(t)
    total = torch.tensor([float(paths)], dtype=torch.float64)
    dist.all_reduce(total)
    gd.barrier()
    if rank == 0:
        np.save(os.path.join(out_dir, "film_%s.npy" % mode), t.numpy())
        np.save(os.path.join(out_dir, "paths_%s.npy" % mode), total.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["tiles", "samples"])
def test_two_ranks_rebuild_the_film(tmp_path, mode):
    import oracle_binding as ob
    world = 2
    mp.spawn(_rank_main, args=(world, free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    film = np.load(tmp_path / ("film_%s.npy" % mode))
    paths = float(np.load(tmp_path / ("paths_%s.npy" % mode))[0])
    scene = gs.load_scene("bunny", OV)
    oracle = ob.Oracle(scene)

    def whole(seed):
        s = oracle.native_samples(seed)
        li, _ = oracle.li_replay(s)
        return oracle.splat(s, li)

    if mode == "tiles":      # fixed job: the shards partition one frame
        assert paths == scene.num_paths()
        np.testing.assert_allclose(film, whole(100), rtol=2e-5, atol=1e-6)
    else:                    # fixed work per rank: rank r contributes the sample set seeded 100 + r
        assert paths == 2 * scene.num_paths()
        np.testing.assert_allclose(film, whole(100) + whole(101), rtol=2e-5, atol=1e-6)


def _bench_rank_main(rank, world, port, out_dir):
    """bench.py's own timed-region code (run_steps) and work split on two gloo ranks, with the oracle as the renderer."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    import bench
    import oracle_binding as ob
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    gd.init(backend="gloo")
    scene = gs.load_scene("grid", gs.config_overrides(resolution=(40, 24), spp=4, depth=4))   # configs[3]'s scene, tiny
    oracle = ob.Oracle(scene)
    part = gd.shard_for(rank, world, "tiles", base_seed=20261003)                            # bench.py's N > 1 default: strong
    film = torch.zeros((scene.desc.film.yres, scene.desc.film.xres, 4), dtype=torch.float32)
    windows = gd.tiles_of(oracle.window(), *part["shard"]) if part["shard"] else [oracle.window()]
    traced = [0]

    def render():
        for win in windows:
            samples = oracle.native_samples(part["seed"], window=win)
            li, _ = oracle.li_replay(samples)
            oracle.splat(samples, li, film.numpy())
            traced[0] += samples.shape[0]

    elapsed, per_step = bench.run_steps(render=render, zero_film=film.zero_, allreduce=lambda: gd.allreduce_film(film), barrier=gd.barrier,
                                        sync=lambda: None, steps=2, warmup=1, world=world)
    assert elapsed > 0.0 and per_step is None
    if rank == 0:
        np.save(os.path.join(out_dir, "bench_film.npy"), film.numpy())
        np.save(os.path.join(out_dir, "bench_paths.npy"), np.array([traced[0]]))
    dist.destroy_process_group()


def test_bench_n2_branch_rebuilds_the_one_gpu_film(tmp_path):
    """bench.py --gpus 2 (strong: interleaved 8x8 tiles, one all-reduce per step) leaves the film a single rank renders."""
    import oracle_binding as ob
    world = 2
    mp.spawn(_bench_rank_main, args=(world, free_port(), str(tmp_path)), nprocs=world, join=True)
    film = np.load(tmp_path / "bench_film.npy")
    scene = gs.load_scene("grid", gs.config_overrides(resolution=(40, 24), spp=4, depth=4))
    oracle = ob.Oracle(scene)
    s = oracle.native_samples(20261003)
    li, _ = oracle.li_replay(s)
    np.testing.assert_allclose(film, oracle.splat(s, li), rtol=2e-5, atol=1e-6)     # one step's film: zeroed, traced, reduced
    # rank 0 traced its half of the tiles in each of the 3 steps (1 warmup + 2 timed)
    assert 0 < int(np.load(tmp_path / "bench_paths.npy")[0]) < 3 * scene.num_paths()


def test_tile_shards_partition_the_window():
    win = (-2, 43, -2, 27)   # ragged: 45 x 29 pixels
    for world in (1, 2, 3, 8):
        seen = np.zeros((29, 45), np.int32)
        for r in range(world):
            for (x0, x1, y0, y1) in gd.tiles_of(win, r, world):
                assert 0 < x1 - x0 <= 8 and 0 < y1 - y0 <= 8
                seen[y0 + 2:y1 + 2, x0 + 2:x1 + 2] += 1
        assert (seen == 1).all()
    assert gd.shard_for(3, 8, "tiles", 7) == {"shard": (3, 8), "seed": 7}
    assert gd.shard_for(3, 8, "samples", 7) == {"shard": None, "seed": 10}
    assert gd.shard_for(0, 1, "tiles", 7) == {"shard": None, "seed": 7}


@pytest.mark.parametrize("world", [2, 4, 8])
def test_tile_split_of_the_multi_gpu_config_is_balanced(world):
    """bench.py --gpus N renders BASELINE configs[3] (grid.json, 1024x1024 film: a 1028x1028 sample window, 129x129 tiles of
    8x8, the last row / column 4 wide) under the interleaved tile split: every rank owns within +-2 % of the pixels (hence of
    the camera paths), every tile belongs to exactly one rank, and the tiles of a rank are spread over the whole image
    (Film::mergeTile's workers take whichever tile comes next, GoblinRenderer.cpp:99-126: no spatial partition there either)."""
    scene = gs.load_scene("grid", gs.config_overrides(resolution=(1024, 1024), spp=256, depth=8))
    import ctypes as C
    from goblin_amd import _abi
    w = (C.c_int32 * 4)()
    _abi.host_lib().gbl_host_sample_window(C.byref(scene.desc.film), w)
    window = tuple(w)
    assert window == (-2, 1026, -2, 1026)
    owned = np.zeros((1028, 1028), np.int32)
    counts = []
    for rank in range(world):
        px = 0
        rows = set()
        for (x0, x1, y0, y1) in gd.tiles_of(window, rank, world):
            owned[y0 + 2:y1 + 2, x0 + 2:x1 + 2] += 1
            px += (x1 - x0) * (y1 - y0)
            rows.add(y0)
        counts.append(px)
        assert len(rows) == 129          # a rank's tiles touch every tile row
    assert (owned == 1).all()
    assert sum(counts) == 1028 * 1028 and sum(counts) * scene.spp() == scene.num_paths()
    assert max(counts) <= 1.02 * min(counts), counts
