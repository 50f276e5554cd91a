"""Multi-GPU plumbing for the hot path: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

The path shards by independent units, so there is NO data-path collective for the 7B configs:
  * single-delta models: data-parallel replicas over independent sequences;
  * multi-tenant serving: the 16-bit base is replicated, tenants (their masks/coeffs) are partitioned across ranks
    (`tenants_for_rank`), requests are routed by tenant id.
The only exchange step is the 70B tensor-parallel case (new, not in the reference): K-split o_proj / down_proj produce
partial sums that are all-reduced (`shard_mask_rows` + `all_reduce_partial`); alpha is a per-matrix scalar so it commutes
with the reduction.  `timed_region` implements the bench contract: barrier + synchronize on both sides, MAX over ranks.
"""
import os
import time

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:       # BD_DIST_BACKEND=gloo lets the multi-rank control flow be exercised on a box with fewer GPUs than ranks
            backend = os.environ.get("BD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def runtime_info():
    """What actually got initialised -- goes into every bench line, so a multi-GPU run says which transport it measured."""
    if dist.is_available() and dist.is_initialized():
        return {"n_ranks_seen": dist.get_world_size(), "backend": dist.get_backend()}
    return {"n_ranks_seen": 1, "backend": "none (single process)"}


def tenants_for_rank(n_tenants, rank, world):
    """Contiguous, balanced partition of tenant ids over ranks (earlier ranks take the remainder)."""
    q, r = divmod(n_tenants, world)
    lo = rank * q + min(rank, r)
    return list(range(lo, lo + q + (1 if rank < r else 0)))


def route(tenant_id, n_tenants, world):
    """Rank that owns `tenant_id` under tenants_for_rank."""
    q, r = divmod(n_tenants, world)
    cut = r * (q + 1)
    return tenant_id // (q + 1) if tenant_id < cut else r + (tenant_id - cut) // max(q, 1)


def shard_mask_columns(mask, rank, world):
    """N-split (q/k/v/gate/up in tensor parallel): packed words are [K/32, N] so a column slice is a plain slice."""
    N = mask.shape[-1]
    assert N % world == 0
    n = N // world
    return mask[..., rank * n:(rank + 1) * n].contiguous()


def shard_mask_rows(mask, rank, world):
    """K-split (o_proj/down_proj): rank r keeps word rows [r*K/32/world, ...); K/world must be a multiple of 32."""
    KW = mask.shape[-2]
    assert KW % world == 0, "K/world must be a multiple of 32"
    k = KW // world
    return mask[..., rank * k:(rank + 1) * k, :].contiguous()


def all_reduce_partial(y):
    """Sum the K-split partial outputs across ranks (RCCL over xGMI on the GPU box; gloo in CPU tests)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(y, op=dist.ReduceOp.SUM)
    return y


def timed_region(fn, steps, device_sync=None):
    """Time `steps` calls of fn bracketed by barrier + device synchronize on both sides; returns MAX seconds over ranks."""
    def sync():
        if device_sync is not None:
            device_sync()
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if device_sync is not None:
        device_sync()
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt
