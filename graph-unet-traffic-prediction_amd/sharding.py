"""Batch sharding of independent (city, time-window) samples over the GPUs of one node.

The reference runs one sample per call in one process (train_utils.py:69-72; its batched-graph
path does not exist), so a multi-GPU run is a set of replicas: each rank owns a contiguous block
of the sample list (grouped by city, so a rank builds only the plans it needs), weights are
replicated and no data moves between ranks inside a forward.  torch.distributed (RCCL on GPUs,
gloo on CPUs) is used for two things only: gathering the predictions on one rank and summing the
metric numerators.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, world_size, rank):
    """Contiguous block [lo, hi) of rank `rank`; the first n_items % world_size ranks get one extra."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_by_city(sample_cities, world_size, rank):
    """Indices of the samples rank `rank` runs.  Samples are ordered by city first (stable), then
    cut into contiguous blocks, so one rank touches as few cities as possible."""
    order = sorted(range(len(sample_cities)), key=lambda i: sample_cities[i])
    lo, hi = shard_bounds(len(order), world_size, rank)
    return order[lo:hi]


def city_groups(sample_ids, sample_cities):
    """[(city, [sample ids])] for a rank's (city-ordered) sample ids: runs of equal city, which a
    rank forwards as one batch over that city's plan."""
    groups = []
    for i in sample_ids:
        c = sample_cities[i]
        if groups and groups[-1][0] == c:
            groups[-1][1].append(i)
        else:
            groups.append((c, [i]))
    return groups


def row_counts(n_local, device, group=None):
    """Row count of every rank's block (a list of ints): one small all-reduce.  The counts of a
    fixed sharding never change, so a caller gathers them once and passes them to gather_rows."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return [int(n_local)]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = torch.zeros(world, dtype=torch.int64, device=device)
    counts[rank] = int(n_local)
    dist.all_reduce(counts, group=group)
    return [int(c) for c in counts.tolist()]


def gather_rows(local, dst=0, group=None, counts=None, out=None):
    """Gather variable-length [n_i, ...] tensors on rank dst (concatenated in rank order; None on
    the other ranks).  Row counts may differ between ranks (cities differ in node count).
    counts: row_counts() of this sharding (fetched when absent); out: [sum(counts), ...] buffer on
    rank dst to receive into (allocated when absent).  Point-to-point: every sender writes its
    block straight into its slice of the collector's buffer, all receives posted as one batch so
    that they run on the senders' own links side by side."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if counts is None:
        counts = row_counts(local.shape[0], local.device, group)
    local = local.contiguous()
    if rank == dst:
        total = sum(counts)
        if out is None:
            out = torch.empty((total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        offs = [sum(counts[:r]) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, out[offs[r]:offs[r] + counts[r]], _global_rank(r, group), group)
               for r in range(world) if r != rank and counts[r] > 0]
        out[offs[rank]:offs[rank] + counts[rank]].copy_(local)
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return out
    if counts[rank] > 0:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, _global_rank(dst, group), group)]):
            req.wait()
    return None


def _global_rank(r, group):
    return r if group is None else dist.get_global_rank(group, r)


def gather_equal(local, dst=0, group=None, out=None):
    """Gather same-shaped tensors on rank dst as one [world, ...] tensor (None elsewhere)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local.unsqueeze(0)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if rank == dst:
        if out is None:
            out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local.contiguous(), gather_list=list(out.unbind(0)), dst=dst, group=group)
        return out
    dist.gather(local.contiguous(), gather_list=None, dst=dst, group=group)
    return None


def sum_metrics(values, group=None):
    """All-reduce (sum) a small tensor of metric numerators / counts."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(values, group=group)
    return values


class StepExchange:
    """What a multi-GPU run exchanges per step: the predictions of every rank gathered on one rank and
    a small metrics vector summed over the ranks.  Both are posted asynchronously and double-buffered
    (step i uses slot i & 1 and first waits for step i - 2 in that slot), so the transfers of one
    step overlap the forward of the next.  With `rotate` the collecting rank of step i is i % world
    instead of always rank 0: every GPU then receives (world - 1) / world of one step's predictions
    per step on average, instead of one GPU's links and memory taking all of them while the others
    wait for it at the next barrier.  With one rank nothing is posted."""

    def __init__(self, like, group=None, rotate=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        self.rotate = rotate
        self.out = [torch.empty_like(like) for _ in range(2)] if self.world > 1 else None
        self.gathered = [None, None]
        if self.world > 1 and (rotate or self.rank == 0):
            self.gathered = [torch.empty((self.world,) + tuple(like.shape), dtype=like.dtype, device=like.device)
                             for _ in range(2)]
        self.collector = [None, None]     # which rank holds the gathered predictions of each slot
        self.metrics = [None, None]
        self.pending = [None, None]

    def post(self, step, y, metrics):
        """y: this rank's predictions (same shape on every rank); metrics: 1-D float64 numerators."""
        if self.world == 1:
            return
        slot = step & 1
        self.wait(slot)
        self.out[slot].copy_(y)
        self.metrics[slot] = metrics
        dst = step % self.world if self.rotate else 0
        self.collector[slot] = dst
        works = [dist.all_reduce(metrics, group=self.group, async_op=True)]
        gather_list = list(self.gathered[slot].unbind(0)) if self.rank == dst else None
        works.append(dist.gather(self.out[slot], gather_list=gather_list, dst=dst, group=self.group, async_op=True))
        self.pending[slot] = works

    def wait(self, slot):
        if self.pending[slot] is not None:
            for w in self.pending[slot]:
                w.wait()
            self.pending[slot] = None

    def drain(self):
        self.wait(0)
        self.wait(1)
