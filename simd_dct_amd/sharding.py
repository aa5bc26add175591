"""Block-row sharding across ranks: every 8x8 block is independent, so a
plane (or a batch of planes) splits into disjoint block-row ranges with no halo and no
reduction.  The reference exposes the same hook through startY/endY
(simd_dct.cpp:2245-2255); here ranges are half-open and in units of block rows."""


def shard_rows(n_block_rows, world_size, rank):
    """Contiguous, balanced, half-open [b0, b1) for `rank`; the union over ranks is
    [0, n_block_rows) and shards are in rank order, so for the Q32 / BLOCK / plane layouts
    the outputs concatenate into the single-range result (all-gather friendly)."""
    if world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError("bad rank/world_size")
    base, extra = divmod(n_block_rows, world_size)
    b0 = rank * base + min(rank, extra)
    return b0, b0 + base + (1 if rank < extra else 0)


def shard_planes(n_planes, world_size, rank):
    """Whole-plane sharding of a batch (config 4 alternative): same arithmetic on planes."""
    return shard_rows(n_planes, world_size, rank)


def equal_shards(n_block_rows, world_size):
    """True when every rank gets the same number of rows (required for a plain all_gather
    into one tensor; otherwise pad or use all_gather with a list)."""
    return n_block_rows % world_size == 0
