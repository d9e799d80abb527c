"""Round-robin stream sharding and the node-global VU combine.

Independent capture streams are the unit of data parallelism: global stream s lives on
rank s % N for its whole life (parameters, VU window, EQ state never migrate) and no
data-path collective exists.  The only collective is the optional node-global VU of
BASELINE config 5: one record of NODE_WORDS int64 per rank (cmhip_batch_vu_node_partial),
combined with SUM over the first half and MAX over the second half.  The exchange is
latency bound (a few hundred bytes over xGMI), so records of several blocks travel
together: gather_node_records() moves B records per rank with ONE all-gather and combines
them locally; combine_node_records() is the one-record form with two all-reduces (RCCL with
backend "nccl"; the same code runs over gloo on CPU tensors in the tests).
"""

NODE_WORDS = 34
NODE_SUM_WORDS = 17


def shard(total_streams, world, rank):
    """-> (local stream count, first global id, global id step) for this rank"""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    count = total_streams // world + (1 if rank < total_streams % world else 0)
    return count, rank, world


def global_id(local_stream, world, rank):
    return rank + local_stream * world


def owner(global_stream, world):
    """-> (rank, local stream index)"""
    return global_stream % world, global_stream // world


def combine_node_records(dist, words):
    """in-place all-reduce of one node record (a 1-D int64 tensor of NODE_WORDS)"""
    if words.numel() != NODE_WORDS:
        raise ValueError("node record must have %d words" % NODE_WORDS)
    dist.all_reduce(words[:NODE_SUM_WORDS], op=dist.ReduceOp.SUM)
    dist.all_reduce(words[NODE_SUM_WORDS:], op=dist.ReduceOp.MAX)
    return words


def gather_node_records(dist, records, scratch=None):
    """records: int64 tensor [B, NODE_WORDS] of this rank (B blocks).  One all-gather, then the
    combine on the local device: -> [B, NODE_WORDS], sums added and keys maximised over the ranks.
    (keys are below 2^63, so the signed maximum is the unsigned one.)"""
    import torch
    if records.dim() != 2 or records.shape[1] != NODE_WORDS:
        raise ValueError("node records must be [B, %d]" % NODE_WORDS)
    world = dist.get_world_size()
    if scratch is None or tuple(scratch.shape) != (world,) + tuple(records.shape):
        scratch = torch.empty((world,) + tuple(records.shape), dtype=records.dtype, device=records.device)
    if records.is_cuda:
        dist.all_gather_into_tensor(scratch, records)
    else:                                            # gloo: the list form
        parts = [scratch[r] for r in range(world)]
        dist.all_gather(parts, records)
    out = torch.empty_like(records)
    out[:, :NODE_SUM_WORDS] = scratch[:, :, :NODE_SUM_WORDS].sum(dim=0)
    out[:, NODE_SUM_WORDS:] = scratch[:, :, NODE_SUM_WORDS:].max(dim=0).values
    return out


def max_over_ranks(dist, seconds, device=None):
    """the slowest rank's time, as bench.py reports it"""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
