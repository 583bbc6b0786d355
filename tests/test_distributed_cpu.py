"""CPU tier: the multi-rank path (one process per GPU, rows of the stratification per rank, one sum-reduce) with
world size 2 over gloo, using the host build of the device core in place of the kernels."""
import os
import subprocess
import sys

import numpy as np
import pytest

import raytracing_rust_amd as rt
from raytracing_rust_amd import scenes, sharding
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(tmp_path, mode, port):
    out = tmp_path / f"{mode}.npy"
    env = dict(os.environ, OMP_NUM_THREADS="1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port),
                           os.path.join(ROOT, "tests", "dist_worker.py"), str(out), mode], env=env, timeout=600)
    return np.load(out)


def single(rows, spp=16):
    g = rt.Graph()
    w, l, cam = scenes.build("cornell", g, width=24, spp=spp, depth=8)
    desc = rt.flatten(g, w, l)
    accum = np.zeros((cam.image_height, cam.image_width, 3))
    for (b, e, seed) in rows:
        p, _ = H.hostsim_render(desc, cam, rt.render_opts(cam, sample_begin=b, sample_end=e, seed=seed), threads=1)
        accum += p
    return accum, cam


def test_rank_rows_partition():
    for world in (1, 2, 3, 8):
        for n in (1, 4, 31, 70):
            spans = [sharding.rank_rows(r, world, n) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_rank_samples_differ_by_at_most_one():
    """One frame's sample indices over the ranks: contiguous, complete, shares within one index of each other."""
    for world in (1, 2, 3, 5, 8):
        spans = [sharding.rank_samples(r, world, 4900) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == 4900 and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [e - b for b, e in spans]
        assert max(sizes) - min(sizes) <= 1
    assert [e - b for b, e in (sharding.rank_samples(r, 8, 4900) for r in range(8))] == [613] * 4 + [612] * 4


def test_sample_rows_are_disjoint_over_ranks_and_steps():
    seen = set()
    for rank in range(8):
        for step in range(7):
            b, e, seed = sharding.sample_row(rank, step, 7, 70, 100)
            assert e - b == 70 and b % 70 == 0
            assert (b, seed) not in seen
            seen.add((b, seed))


def test_two_ranks_whole_image_equals_one_process(tmp_path):
    two = launch(tmp_path, "whole", 29511)
    one, cam = single([(0, 25, 5)], spp=25)
    assert np.allclose(two, one, rtol=1e-13, atol=0)      # same paths, summed in another order
    assert one.sum() > 0


def test_two_ranks_bench_schedule_equals_one_process(tmp_path):
    two = launch(tmp_path, "bench", 29512)
    rows = [sharding.sample_row(rank, k, 3, 4, 5) for rank in range(2) for k in range(3)]
    one, _ = single(rows)
    assert np.allclose(two, one, rtol=1e-13, atol=0)
