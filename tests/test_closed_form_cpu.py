"""CPU test of the chain's closed-form routine under sanitizers (tests/harness/closed_form_san.cpp): the text of groupFastPath and its helpers is taken
from t1k_amd/csrc/t1k_chain.hip as it is, compiled for the host with the device intrinsics shimmed, and run on random single-diagonal groups.  It checks
what a GPU cannot tell us: no read of an unset value, no local array indexed out of range, no out-of-range shift (SeqSet.hpp:1232-1556, 1697-1848)."""
import os
import shutil
import subprocess

import pytest

import util

BEGIN = "template <int MW>\n__device__ __forceinline__ void shlOr"
END = "// General group (several diagonals): restates GetOverlapsFromHits"


def _source(tmp):
    chain = open(os.path.join(util.ROOT, "t1k_amd", "csrc", "t1k_chain.hip")).read()
    a, b = chain.index(BEGIN), chain.index(END)
    harness = open(os.path.join(util.ROOT, "tests", "harness", "closed_form_san.cpp")).read()
    path = os.path.join(tmp, "closed_form_san_full.cpp")
    open(path, "w").write(harness.replace("@@ROUTINE@@", chain[a:b]))
    return path


def test_closed_form_routine_under_address_and_ub_sanitizers(tmp_path):
    src = _source(str(tmp_path))
    exe = str(tmp_path / "cf_asan")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=undefined,address", "-fno-sanitize-recover=all", "-o", exe, src], check=True, stderr=subprocess.DEVNULL)
    for seed in ("1", "2"):
        r = subprocess.run([exe, "120000", seed], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0 and r.stdout.startswith("ok"), r.stderr[-2000:]
        fin, walk = [int(x) for x in r.stdout.replace(",", "").split() if x.isdigit()]
        assert fin > 10000 and walk > 1000   # both outcomes of the pass are exercised


def test_closed_form_routine_under_memory_sanitizer(tmp_path):
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang) and not shutil.which("clang++"):
        pytest.skip("no clang++ (MemorySanitizer) here")
    clang = clang if os.path.exists(clang) else shutil.which("clang++")
    src = _source(str(tmp_path))
    exe = str(tmp_path / "cf_msan")
    c = subprocess.run([clang, "-std=c++17", "-O1", "-g", "-fsanitize=memory", "-fno-omit-frame-pointer", "-o", exe, src], stderr=subprocess.PIPE, text=True)
    if c.returncode != 0:
        pytest.skip("this clang++ has no MemorySanitizer runtime: " + c.stderr[-200:])
    r = subprocess.run([exe, "100000", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stderr[-2000:]
