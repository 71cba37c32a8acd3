"""Multi-GPU orchestration of the genotyper stage: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI,
"gloo" in the CPU tests).  Reads shard trivially (contiguous slices of the fragments in file order, no data-path collective
during read-end assignment and pairing).  The exchange steps are:

  * coverage      one in-place int32 all-reduce (sum) of the per-base coverage difference array -- exact
  * read groups   all-gather of each rank's coalesced group table (byte strings of t1k_job_groups_serialize); every rank absorbs
                  them in rank order, which reproduces the global first-appearance numbering of the groups
  * EM            the merged groups are split evenly over the ranks; each EMupdate all-reduces (sum, f64) the per-class expected
                  read counts, the M-step / SQUAREM control / allele selection then run identically on every rank
"""
import itertools

import numpy as np


class _CudaView:
    """zero-copy view of raw device memory for torch.as_tensor (CUDA array interface v2)"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_tensor(torch, ptr, n, typestr):
    return torch.as_tensor(_CudaView(ptr, n, typestr), device="cuda")


def load_shard(job, pfx, rank, world, pairs):
    """rank r owns fragments [r*pairs, (r+1)*pairs) of the sample, in file order"""
    def seqs(path):
        with open(path) as f:
            lines = itertools.islice(f, 4 * rank * pairs, 4 * (rank + 1) * pairs)
            return [l.rstrip("\n") for i, l in enumerate(lines) if i % 4 == 1]
    job.set_reads(seqs(pfx + "_1.fq"), seqs(pfx + "_2.fq"))


def all_gather_bytes(dist, torch, buf, device):
    """variable-length all-gather of a uint8 numpy array -> list of numpy arrays in rank order"""
    world = dist.get_world_size()
    n = torch.tensor([buf.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes)
    mine = torch.zeros(cap, dtype=torch.uint8, device=device)
    mine[:buf.size] = torch.from_numpy(buf).to(device)
    out = [torch.empty(cap, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(out, mine)
    return [o[:s].cpu().numpy() for o, s in zip(out, sizes)]


def install_allreduce(job, dist, torch):
    """EM hook: sum the device-resident per-class read counts over the ranks (called between the E and M steps)"""
    def cb(dev_ptr, n, _user):
        t = device_tensor(torch, dev_ptr, n, "<f8")
        dist.all_reduce(t)
        torch.cuda.current_stream().synchronize()
    job.set_allreduce(cb)


def sharded_step(job, dist, torch, rank, world):
    job.run_local()
    ptr, n = job.coverage_device()
    cov = device_tensor(torch, ptr, n, "<i4")
    dist.all_reduce(cov)
    torch.cuda.current_stream().synchronize()
    tables = all_gather_bytes(dist, torch, job.groups_serialize(), "cuda")
    job.groups_reset()
    for t in tables:
        job.groups_absorb(t)
    g = job.counts()["groups"]
    job.finish(rank * g // world, (rank + 1) * g // world)
