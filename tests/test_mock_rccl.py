"""tests/mock_rccl — the loopback transport the multi-rank GPU tests run on (ORAMA_RCCL_LIB), checked by itself.

With MOCK_RCCL_HOST_BUFFERS=1 the library moves host buffers, so the segment / barrier / group / reduction logic runs
on a box without a GPU: world-3 jobs as three PROCESSES (ncclCommInitRank) and as one process driving three
communicators inside ncclGroupStart/End (ncclCommInitAll) — the two shapes orama_shard_group_create_rank / _create use.
"""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent
LIB = HERE / "mock_rccl" / "libmock_rccl.so"

U8, I32, I64, F64 = 1, 2, 4, 8
SUM, MAX = 0, 2


def load():
    if not LIB.exists():
        subprocess.run(["make", "-C", str(LIB.parent)], check=True, capture_output=True)
    os.environ["MOCK_RCCL_HOST_BUFFERS"] = "1"
    lib = C.CDLL(str(LIB))
    lib.ncclGetErrorString.restype = C.c_char_p
    return lib


class Uid(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def expected_gather(world, n):
    return np.concatenate([np.arange(n, dtype=np.uint8) * (r + 3) + r for r in range(world)])


def rank_job(lib, comm, rank, world, rounds=3):
    """The collectives one rank issues; returns what it ends up holding."""
    out = {}
    n = 1000 + 7
    for it in range(rounds):
        buf = np.zeros(world * n, dtype=np.uint8)
        buf[rank * n:(rank + 1) * n] = np.arange(n, dtype=np.uint8) * (rank + 3) + rank
        assert lib.ncclAllGather(C.c_void_p(buf.ctypes.data + rank * n), C.c_void_p(buf.ctypes.data), C.c_size_t(n), U8,
                                 comm, None) == 0
        out[f"gather{it}"] = buf
        df = np.array([rank + 1, 10 * (rank + 1) + it, 0, 5], dtype=np.int32)
        assert lib.ncclAllReduce(C.c_void_p(df.ctypes.data), C.c_void_p(df.ctypes.data), C.c_size_t(4), I32, SUM, comm, None) == 0
        out[f"df{it}"] = df
        mm = np.array([rank * 7 - 3, -(rank + it)], dtype=np.int64)
        assert lib.ncclAllReduce(C.c_void_p(mm.ctypes.data), C.c_void_p(mm.ctypes.data), C.c_size_t(2), I64, MAX, comm, None) == 0
        out[f"mm{it}"] = mm
    t = np.array([0.5 + rank], dtype=np.float64)
    assert lib.ncclAllReduce(C.c_void_p(t.ctypes.data), C.c_void_p(t.ctypes.data), C.c_size_t(1), F64, MAX, comm, None) == 0
    out["t"] = t
    return out


def check(out, world, rounds=3):
    n = 1007
    for it in range(rounds):
        assert np.array_equal(out[f"gather{it}"], expected_gather(world, n))
        assert out[f"df{it}"].tolist() == [sum(r + 1 for r in range(world)), sum(10 * (r + 1) + it for r in range(world)), 0,
                                           5 * world]
        assert out[f"mm{it}"].tolist() == [(world - 1) * 7 - 3, -it]
    assert out["t"][0] == 0.5 + world - 1


def test_exports_what_the_product_binds():
    lib = load()
    src = (HERE.parent / "oramacore_amd" / "csrc" / "shard_group.hip").read_text()
    import re

    syms = re.findall(r'ORAMA_RCCL_SYM\(\w+, "(\w+)"\)', src)
    assert len(syms) == 9
    for s in syms:
        assert hasattr(lib, s), s


def test_one_process_three_communicators_grouped():
    lib = load()
    world = 3
    comms = (C.c_void_p * world)()
    assert lib.ncclCommInitAll(comms, world, (C.c_int * world)(0, 0, 0)) == 0
    n = 513
    bufs = [np.zeros(world * n, dtype=np.uint8) for _ in range(world)]
    for r in range(world):
        bufs[r][r * n:(r + 1) * n] = (np.arange(n) * (r + 3) + r).astype(np.uint8)
    for _ in range(2):
        assert lib.ncclGroupStart() == 0
        for r in range(world):
            assert lib.ncclAllGather(C.c_void_p(bufs[r].ctypes.data + r * n), C.c_void_p(bufs[r].ctypes.data), C.c_size_t(n), U8,
                                     C.c_void_p(comms[r]), None) == 0
        assert lib.ncclGroupEnd() == 0
        for r in range(world):
            assert np.array_equal(bufs[r], np.concatenate([(np.arange(n) * (x + 3) + x).astype(np.uint8) for x in range(world)]))
    dfs = [np.array([r + 1, 100], dtype=np.int32) for r in range(world)]
    assert lib.ncclGroupStart() == 0
    for r in range(world):
        assert lib.ncclAllReduce(C.c_void_p(dfs[r].ctypes.data), C.c_void_p(dfs[r].ctypes.data), C.c_size_t(2), I32, SUM,
                                 C.c_void_p(comms[r]), None) == 0
    assert lib.ncclGroupEnd() == 0
    for r in range(world):
        assert dfs[r].tolist() == [6, 300]
    for r in range(world):
        assert lib.ncclCommDestroy(C.c_void_p(comms[r])) == 0


WORKER = r"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, {tests!r})
import test_mock_rccl as T
lib = T.load()
rank, world = int(sys.argv[1]), int(sys.argv[2])
uid = T.Uid()
C.memmove(C.byref(uid), bytes.fromhex(sys.argv[3]), 128)
comm = C.c_void_p()
lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, T.Uid, C.c_int]
assert lib.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
T.check(T.rank_job(lib, comm, rank, world), world)
assert lib.ncclCommDestroy(comm) == 0
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world", [2, 3])
def test_processes_as_ranks(world, tmp_path):
    lib = load()
    uid = Uid()
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(tests=str(HERE)))
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), bytes(uid).hex()], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    for r, p in enumerate(procs):
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out
        assert f"rank {r} ok" in out
