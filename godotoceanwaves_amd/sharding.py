"""Cascade sharding across the GPUs of a node.

Cascades are independent units (wave_generator.gd:65-85 touches nothing shared between cascades): rank r owns
the global cascades r*C .. r*C+C-1 with all their state (h0, foam, time) and runs the two frame kernels on them;
there is NO data-path collective.  The only exchange is the final gather of the finished RGBA16F layers to every
rank (one all_gather per map over RCCL/xGMI), and it is optional: consumers on the owning GPU read in place.
torch is used for device memory + the collective only.
"""


def owned_cascades(rank, world, per_rank):
    """global cascade ids of `rank`: a contiguous block (weak scaling: per-rank work is fixed)"""
    if not (0 <= rank < world) or per_rank < 1:
        raise ValueError(f"bad shard request rank={rank} world={world} per_rank={per_rank}")
    return list(range(rank * per_rank, (rank + 1) * per_rank))


def owner_of(cascade, per_rank):
    """(rank, local layer) that holds global cascade `cascade`"""
    return cascade // per_rank, cascade % per_rank


def alloc_gather_buffers(torch, world, disp, norm):
    """[world, layers, N, N, 4] receive buffers for gather_maps (allocated once, outside the timed region)"""
    return (torch.empty((world,) + tuple(disp.shape), dtype=disp.dtype, device=disp.device),
            torch.empty((world,) + tuple(norm.shape), dtype=norm.dtype, device=norm.device))


def gather_maps(dist, gathered, disp, norm):
    """all_gather of both maps; gathered[0][r, l] is the displacement layer l of rank r = global cascade r*C + l"""
    # flat views: the concatenating form of all_gather_into_tensor, accepted by both RCCL and gloo
    dist.all_gather_into_tensor(gathered[0].view(-1), disp.reshape(-1))
    dist.all_gather_into_tensor(gathered[1].view(-1), norm.reshape(-1))
    return gathered
