"""How the sample loop is split over ranks (one process per GPU).

Every camera path is independent (camera.rs:259-267 is a map-reduce over samples), and the path RNG is addressed by
(pixel, sample index), so a rank can render any subset of sample indices of every pixel into its own accumulator;
the sum over ranks is the image.  Units of work are ROWS of the sqrt_spp x sqrt_spp stratification: row r holds the
sample indices [r * sqrt_spp, (r + 1) * sqrt_spp)."""


def sample_row(rank, step, steps_per_rank, sqrt_spp, base_seed):
    """The row that `rank` renders in its `step`-th step, as (sample_begin, sample_end, seed).  Rows are disjoint over
    (rank, step); past the sqrt_spp rows of one stratification the next one starts with another seed."""
    row = rank * steps_per_rank + step
    r = row % sqrt_spp
    return r * sqrt_spp, (r + 1) * sqrt_spp, base_seed + row // sqrt_spp


def rank_rows(rank, world_size, n_rows):
    """Contiguous split of rows [0, n_rows) over ranks for a whole-image render: (first_row, end_row)."""
    per, extra = divmod(n_rows, world_size)
    first = rank * per + min(rank, extra)
    return first, first + per + (1 if rank < extra else 0)


def reduce_sum(accum, dist, dst=0):
    """The one collective of the path: sum of the per-rank accumulators on `dst` (RCCL on GPUs, gloo on CPUs)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum
