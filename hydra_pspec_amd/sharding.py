"""Partition of baselines over ranks / GPUs (no collective on the data path).

Same rule as the reference's MPI scatter (run-hydra-pspec.py:268-287):
contiguous blocks, the first ``rem`` ranks get ``quot + 1`` baselines."""


def split_counts(n_items, n_ranks):
    """Block sizes per rank.  Raises ValueError when there are fewer items than
    ranks (the reference aborts the MPI job in that case, :273-276)."""
    quot, rem = divmod(int(n_items), int(n_ranks))
    if quot == 0:
        raise ValueError(f"Number of baselines ({n_items}) should be >= number of ranks ({n_ranks})!")
    return [quot + 1 if r < rem else quot for r in range(n_ranks)]


def block_range(n_items, n_ranks, rank):
    """(start, stop) of rank's contiguous block."""
    counts = split_counts(n_items, n_ranks)
    start = sum(counts[:rank])
    return start, start + counts[rank]


def split_data_for_scatter(data, n_ranks):
    """List of per-rank sub-lists (reference name and semantics)."""
    counts = split_counts(len(data), n_ranks)
    out, pos = [], 0
    for c in counts:
        out.append(data[pos:pos + c])
        pos += c
    return out
