"""How the sample loop is split over ranks (one process per GPU).

Every camera path is independent (camera.rs:259-267 is a map-reduce over samples), and the path RNG is addressed by
(pixel, sample index), so a rank can render any subset of sample indices of every pixel into its own accumulator;
the sum over ranks is the image.  Units of work are ROWS of the sqrt_spp x sqrt_spp stratification: row r holds the
sample indices [r * sqrt_spp, (r + 1) * sqrt_spp)."""


def sample_batch(rank, step, steps_per_rank, sqrt_spp, base_seed, rows_per_step):
    """The batch of `rows_per_step` rows that `rank` renders in its `step`-th step, as (sample_begin, sample_end,
    seed).  Batches are disjoint over (rank, step); past the sqrt_spp rows of one stratification the next one starts
    with another seed.  rows_per_step must divide sqrt_spp."""
    if rows_per_step < 1 or sqrt_spp % rows_per_step:
        raise ValueError("rows_per_step must divide sqrt_spp")
    per_strat = sqrt_spp // rows_per_step
    batch = rank * steps_per_rank + step
    r = batch % per_strat
    begin = r * rows_per_step * sqrt_spp
    return begin, begin + rows_per_step * sqrt_spp, base_seed + batch // per_strat


def sample_row(rank, step, steps_per_rank, sqrt_spp, base_seed):
    """sample_batch with one row per step."""
    return sample_batch(rank, step, steps_per_rank, sqrt_spp, base_seed, 1)


def rank_rows(rank, world_size, n_rows):
    """Contiguous split of rows [0, n_rows) over ranks for a whole-image render: (first_row, end_row)."""
    per, extra = divmod(n_rows, world_size)
    first = rank * per + min(rank, extra)
    return first, first + per + (1 if rank < extra else 0)


def rank_samples(rank, world_size, n_samples):
    """Contiguous split of the sample indices [0, n_samples) of ONE frame over ranks: (first, end).  By index, not by
    row of the stratification, so that the shares of any two ranks differ by at most one sample of every pixel
    (4900 indices over 8 ranks: 613 or 612 each; whole rows of 70 would give 9 or 8 rows, 11 % apart)."""
    return rank_rows(rank, world_size, n_samples)


def reduce_sum(accum, dist, dst=0):
    """The one collective of the path: sum of the per-rank accumulators on `dst` (RCCL on GPUs, gloo on CPUs)."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum
