"""One process per GPU: rank discovery, work sharding, Film reduction.

The reference's only "collective" is the mutex-guarded sum of per-thread full-film
tiles (Film::mergeTile, /root/reference/src/GoblinFilm.cpp:140-153,
GoblinThreadLocalStorage.h:69-75).  Across GPUs that is one all-reduce(sum) of the
W*H float4 accumulators over RCCL (torch.distributed backend "nccl" on ROCm); there
is no other exchange: camera samples are independent.

Two ways to split a render over N ranks, both sum-decomposable on the film:
  * "tiles"   rank r renders the 8x8 sample tiles t with t % N == r   (fixed job, strong scaling)
  * "samples" every rank renders the whole window with its own sample set (seed + rank); the
              reduced film has N x spp samples per pixel              (fixed work per GPU, weak scaling)
"""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, local_rank, world)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # GBL_DIST_BACKEND=gloo rehearses the N > 1 path with several ranks on ONE GPU (RCCL refuses that)
            backend = os.environ.get("GBL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_for(rank, world, mode="tiles", base_seed=0):
    """What rank `rank` of `world` renders: dict(shard=(index,count)|None, seed=...)."""
    if world <= 1:
        return {"shard": None, "seed": base_seed}
    if mode == "tiles":
        return {"shard": (rank, world), "seed": base_seed}
    if mode == "samples":
        return {"shard": None, "seed": base_seed + rank}
    raise ValueError("unknown sharding mode %r" % mode)


def tiles_of(window, rank, world, tile=8):
    """The (x0, x1, y0, y1) rectangles of the sample tiles rank owns under "tiles" sharding
    (row-major tile numbering over the window, as the kernel's tile_shard_* parameters)."""
    x0, x1, y0, y1 = window
    tiles_x = (x1 - x0 + tile - 1) // tile
    tiles_y = (y1 - y0 + tile - 1) // tile
    out = []
    for t in range(rank, tiles_x * tiles_y, max(1, world)):
        tx, ty = t % tiles_x, t // tiles_x
        out.append((x0 + tile * tx, min(x0 + tile * tx + tile, x1), y0 + tile * ty, min(y0 + tile * ty + tile, y1)))
    return out


def allreduce_film(accum):
    """Sum the film accumulators over all ranks (in place).  `accum` is a torch tensor on the
    rank's device (RCCL) or on the CPU (gloo)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "gloo" and accum.is_cuda:   # rehearsal mode: stage through the host
            host = accum.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            accum.copy_(host)
        else:
            dist.all_reduce(accum, op=dist.ReduceOp.SUM)
    return accum


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":   # name the device: RCCL otherwise guesses it from the rank
            import torch
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
