"""GPU tests (pytest -m gpu) that need TWO OR MORE devices: the RCCL transport of a sharded job with N > 1 (ncclAllReduce of the
coverage arrays and of the EM contributions, the grouped ncclSend/ncclRecv of the row exchange, the grouped ncclBroadcast of the group
tables) -- the same job logic the single-GPU box runs with ranks that share one device (test_gpu_parity.py), here over xGMI.  They skip
themselves where fewer than two GPUs are visible, so that the first N > 1 execution is not the multi-GPU scaling run itself."""
import json
import os
import re
import subprocess
import sys

import pytest

import goldens
import util
import t1k_amd

pytestmark = pytest.mark.gpu
GENO = os.path.join(util.ROOT, "t1k_amd", "bin", "genotyper")


def need_devices(n):
    have = t1k_amd.lib().t1k_device_count()
    if have < n:
        pytest.skip("%d GPU(s) visible, the test needs %d" % (have, n))


def _golden_files_equal(c, out):
    assert open(out + "_genotype.tsv").read() == c.expected("genotype.tsv")
    assert open(out + "_allele.tsv").read() == c.expected("allele.tsv")


@pytest.mark.parametrize("name", ["hla_synth_2x150", "cyp_rna_2x100", "cyp_dna_relax_2x150"])
@pytest.mark.parametrize("own_input", [False, True])
def test_two_gpus_over_rccl_equal_the_goldens(built, tmp_path, name, own_input):
    """genotyper --gpus 2 with one rank per device: RCCL is chosen (distinct devices), every file equals the reference's"""
    need_devices(2)
    c = goldens.Case(name, str(tmp_path))
    out = os.path.join(str(tmp_path), "two")
    env = dict(os.environ, T1K_GPUS="0,1", T1K_DEBUG_PHASES="1")
    if own_input:
        env["T1K_SHARD_INPUT"] = "1"
    r = subprocess.run([GENO] + c.args() + ["-o", out], stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    _golden_files_equal(c, out)
    one = os.path.join(str(tmp_path), "one")
    r1 = subprocess.run([GENO] + c.args() + ["-o", one], stderr=subprocess.PIPE, text=True)
    assert r1.returncode == 0, r1.stderr
    for suf in (("_aligned_1.fa", "_aligned_2.fa") if c.paired else ("_aligned.fa",)):
        assert open(out + suf, "rb").read() == open(one + suf, "rb").read(), suf


def test_two_gpus_em_allreduce_over_rccl_within_tolerance(built, tmp_path):
    """T1K_EM_COLLECTIVE=allreduce over RCCL (ncclAllReduce of E doubles per EM update): same calls, abundances within 1e-4 relative"""
    need_devices(2)
    c = goldens.Case("hla_synth_2x150", str(tmp_path))
    out = os.path.join(str(tmp_path), "two")
    r = subprocess.run([GENO] + c.args() + ["-o", out], stderr=subprocess.PIPE, text=True, env=dict(os.environ, T1K_GPUS="0,1", T1K_EM_COLLECTIVE="allreduce"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    for g, w in zip(open(out + "_genotype.tsv").read().splitlines(), c.expected("genotype.tsv").splitlines()):
        for x, y in zip(g.split("\t"), w.split("\t")):
            if x != y:
                assert abs(float(x) - float(y)) <= 1e-4 * max(abs(float(x)), abs(float(y))), (g, w)


def test_communicator_is_rccl_across_devices(built):
    need_devices(2)
    import threading
    group = t1k_amd.CommGroup(2)
    ctxs = [t1k_amd.Context(device=d) for d in (0, 1)]
    comms, errs = [None, None], []

    def mk(r):
        try:
            comms[r] = t1k_amd.Comm(ctxs[r], 2, r, group=group)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=mk, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert all(c.is_rccl() for c in comms)
    for c in comms:
        c.close()
    for c in ctxs:
        c.close()
    group.close()


@pytest.mark.parametrize("ranks", [2, 4, 8])
def test_bench_under_torchrun_equals_single_gpu_text(built, tmp_path, ranks):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one process per GPU, the job's own RCCL
    communicator): the JSON line appears and the genotype text equals the single-GPU executable's on the same sample"""
    need_devices(ranks)
    wd = str(tmp_path)
    pairs = 40000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", str(29600 + ranks),
           os.path.join(util.ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--pairs", str(pairs), "--workdir", wd, "--no-cpu-baseline"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["steps"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    args = ["-f", os.path.join(wd, "hla_g24_s1.0.fa")]
    for i in range(ranks):
        args += ["-1", os.path.join(wd, "reads_g24_s1.0_p%d_seed%d_1.fq" % (pairs, 2 + i)), "-2", os.path.join(wd, "reads_g24_s1.0_p%d_seed%d_2.fq" % (pairs, 2 + i))]
    one = os.path.join(wd, "one")
    r1 = subprocess.run([GENO] + args + ["-s", "0.97", "-o", one], stderr=subprocess.PIPE, text=True)
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert open(one + "_genotype.tsv").read() == open(os.path.join(wd, "last_genotype.tsv")).read()
    assert open(one + "_allele.tsv").read() == open(os.path.join(wd, "out_allele.tsv")).read()
    for suf in ("_aligned_1.fa", "_aligned_2.fa"):
        assert open(one + suf, "rb").read() == open(os.path.join(wd, "out" + suf), "rb").read(), suf
