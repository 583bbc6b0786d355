"""Worker of the world-size-2 CPU test: each rank renders its rows of the stratification with the host build of the
device core, the accumulators are summed with one gloo reduce, rank 0 writes the result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracing_rust_amd as rt  # noqa: E402
from raytracing_rust_amd import scenes, sharding  # noqa: E402
import helpers as H  # noqa: E402


def main():
    out_path, mode = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = rt.Graph()
    w, l, cam = scenes.build("cornell", g, width=24, spp=25 if mode == "whole" else 16, depth=8)
    desc = rt.flatten(g, w, l)
    accum = np.zeros((cam.image_height, cam.image_width, 3))
    if mode == "whole":          # one frame: its 25 sample indices split by index (13 and 12: not whole rows of 5)
        first, end = sharding.rank_samples(rank, world, cam.sqrt_spp ** 2)
        rows = [(first, end, 5)]
    else:                        # the benchmark's schedule: `steps` rows per rank
        steps = 3
        rows = [sharding.sample_row(rank, k, steps, cam.sqrt_spp, 5) for k in range(steps)]
    for (b, e, seed) in rows:
        p, _ = H.hostsim_render(desc, cam, rt.render_opts(cam, sample_begin=b, sample_end=e, seed=seed), threads=1)
        accum += p
    t = torch.from_numpy(accum)
    sharding.reduce_sum(t, dist, dst=0)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
