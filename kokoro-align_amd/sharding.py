"""Multi-GPU layout: one process per GPU, lattices sharded, no data-path collective.

Each audio file is an independent lattice (the reference loops files sequentially,
run_example.py:248-254), so ranks never exchange anything while aligning.  The only collective
is the start-up broadcast of the acoustic-model weights (2.3 MB) from rank 0 over RCCL/xGMI.
"""
import heapq


def lattice_cost(T, S, beam_size=1000):
    """Band cells of a lattice ~ T * min(beam, 2S+1): the DP's work (align.py:64-65)."""
    return int(T) * min(int(beam_size), 2 * int(S) + 1)


def lpt_partition(costs, n_ranks):
    """Longest-processing-time-first assignment of lattices to ranks.
    Returns n_ranks lists of lattice indices (each sorted ascending); deterministic."""
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    heap = [(0, r) for r in range(n_ranks)]
    parts = [[] for _ in range(n_ranks)]
    for i in order:
        load, r = heapq.heappop(heap)
        parts[r].append(i)
        heapq.heappush(heap, (load + costs[i], r))
    return [sorted(p) for p in parts]


def shard_for_rank(shapes, rank, world_size, beam_size=1000):
    """shapes: list of (T, S).  Indices of the lattices this rank aligns."""
    costs = [lattice_cost(t, s, beam_size) for t, s in shapes]
    return lpt_partition(costs, world_size)[rank]


def broadcast_model_weights(device, model=None, src=0):
    """One flat-buffer broadcast of the AudioToChar parameters from ``src`` (RCCL when the
    process group is 'nccl', gloo in the CPU tests).  Returns the model."""
    import torch
    import torch.distributed as dist
    from .model import AudioToChar
    if model is None:
        torch.manual_seed(0 if dist.get_rank() == src else 1 + dist.get_rank())
        model = AudioToChar().to(device)
    tensors = [p.data for p in model.parameters()] + [b.data for b in model.buffers()]
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.broadcast(flat, src=src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    return model


def gather_rank_stats(frames, seconds, device):
    """all_gather of (frames, seconds) for the report; optional."""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([float(frames), float(seconds)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [(float(o[0]), float(o[1])) for o in out]
