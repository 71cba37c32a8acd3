"""CPU test of the N>1 path (world_size 2, gloo).  A sharded job moves every fragment row to the rank that OWNS its allele pattern
(t1k_rowset_exchange: hash of the pattern mod nRanks); the owner coalesces the group over all its fragments in global fragment order,
and the owners' tables are gathered on every rank and merged by first fragment (t1k_job_groups_merge).  Here the two ranks run the
host restatement of that coalescing (t1k_job_coalesce_rows, CoalesceReadAssignments of Genotyper.hpp:841-908) on the fragments they
own, exchange their tables with a gloo all-gather and merge them; the merged table must EQUAL what a single process gets from
coalescing all fragments in file order: same numbering, same start, same `end` (the order-dependent rule of 893-894) and
bit-identical float weights."""
import os
import subprocess
import sys
import textwrap

import numpy as np

import util


WORKER = textwrap.dedent("""
    import os, sys, pickle
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %(root)r)
    sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import t1k_amd, util
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    data = pickle.load(open(%(data)r, "rb"))
    rows, counts, owner = data["rows"], data["counts"], data["owner"]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    mine = np.nonzero(owner %% world == rank)[0]          # fragments whose pattern this rank owns, in global order
    sel = np.concatenate([np.arange(off[f], off[f + 1]) for f in mine]) if len(mine) else np.zeros(0, np.int64)
    job = t1k_amd.Job(%(ref)r, device=-1, allele_digit_units=1, allele_delimiter=".")
    job.coalesce_rows(rows[sel], counts[mine], fragments=mine)
    buf = job.groups_serialize()
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([buf.size], dtype=torch.int64))
    cap = max(int(x.item()) for x in sizes)
    padded = torch.zeros(cap, dtype=torch.uint8)
    padded[:buf.size] = torch.from_numpy(buf)
    out = [torch.empty(cap, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(out, padded)
    tables = [o[:int(n.item())].numpy() for o, n in zip(out, sizes)]
    merged = t1k_amd.Job(%(ref)r, device=-1, allele_digit_units=1, allele_delimiter=".")
    merged.groups_merge(tables)
    np.save(%(out)r + "_%%d.npy" %% rank, merged.groups_serialize())
    dist.destroy_process_group()
""")


def parse_table(buf):
    g, n, assigned = (int(x) for x in np.frombuffer(buf[:24].tobytes(), dtype=np.uint64))
    o = 24
    ptr = np.frombuffer(buf[o:o + (g + 1) * 8].tobytes(), dtype=np.uint64)
    o += (g + 1) * 8
    first = np.frombuffer(buf[o:o + g * 4].tobytes(), dtype=np.uint32)
    o += g * 4
    ent = np.frombuffer(buf[o:o + n * 20].tobytes(), dtype=np.dtype([("allele", "<i4"), ("start", "<i4"), ("end", "<i4"), ("w", "<f4"), ("aw", "<f4")]))
    return g, assigned, ptr, first, ent


def test_sharded_group_merge_world2(built, tmp_path):
    import pickle
    import t1k_amd
    ref = util.gunzip_to(util.CYP_RNA, str(tmp_path / "ref.fa"))
    rng = np.random.default_rng(11)
    F = 600
    patterns = [np.sort(rng.choice(190, size=rng.integers(1, 30), replace=False)) for _ in range(40)]
    rows, counts = [], []
    owner = []
    for f in range(F):
        k = int(rng.integers(0, len(patterns)))
        p = patterns[k]
        hn = rng.random() < 0.1
        perm = rng.permutation(len(p))  # rows arrive in assignment order, not allele order
        for a in p[perm]:
            s0 = int(rng.integers(0, 900))       # start / end vary per fragment and per allele: the `end` rule fires all the time
            w = float(rng.choice([1.0, 0.5, 0.1, 0.01])) / (10.0 if hn else 1.0)
            rows.append((int(a), s0, s0 + int(rng.integers(60, 400)), w, 1.0, np.float32(0.25 * w)))
        counts.append(len(p))
        owner.append(hash(tuple(int(x) for x in p)) & 0x7FFFFFFF)   # any function of the pattern will do
    rows = np.array(rows, dtype=t1k_amd.ROW_DTYPE)
    counts = np.array(counts, dtype=np.uint32)
    owner = np.array(owner, dtype=np.int64)
    data = str(tmp_path / "rows.pkl")
    pickle.dump({"rows": rows, "counts": counts, "owner": owner}, open(data, "wb"))
    single = t1k_amd.Job(ref, device=-1, allele_digit_units=1, allele_delimiter=".")
    single.coalesce_rows(rows, counts)
    g1, a1, p1, f1, e1 = parse_table(single.groups_serialize())
    assert g1 == len(patterns) and len(set(owner % 2)) == 2
    script = str(tmp_path / "worker.py")
    open(script, "w").write(WORKER % dict(root=util.ROOT, data=data, ref=ref, out=str(tmp_path / "merged")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29611",
                        script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-3000:]
    for rank in (0, 1):
        g2, a2, p2, f2, e2 = parse_table(np.load(str(tmp_path / ("merged_%d.npy" % rank))))
        assert (g2, a2) == (g1, a1)
        assert np.array_equal(p1, p2) and np.array_equal(f1, f2)
        for field in ("allele", "start", "end"):
            assert np.array_equal(e1[field], e2[field]), field
        # float32 sums in fragment order: bit-identical, not merely close
        assert np.array_equal(e1["w"].view(np.uint32), e2["w"].view(np.uint32)) and np.array_equal(e1["aw"].view(np.uint32), e2["aw"].view(np.uint32))


def test_host_only_job_cannot_run(built, tmp_path):
    import t1k_amd
    ref = util.gunzip_to(util.CYP_RNA, str(tmp_path / "ref.fa"))
    job = t1k_amd.Job(ref, device=-1, allele_digit_units=1, allele_delimiter=".")
    job.set_reads(["ACGT" * 30], ["TTGA" * 30])
    try:
        job.run()
        assert False, "a host-only job must refuse to run"
    except t1k_amd.T1kError as e:
        assert "no GPU context" in str(e)
