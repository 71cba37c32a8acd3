"""Multi-GPU orchestration for bench.py (one process per GPU, torch.distributed / RCCL)."""


def load_shard(job, pfx, rank, world, pairs):
    raise NotImplementedError


def sharded_step(job, dist, torch, rank, world):
    raise NotImplementedError
