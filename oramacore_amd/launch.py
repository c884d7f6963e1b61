"""Launcher-side plumbing for the one-process-per-GPU form (bench.py under `python -m torch.distributed.run`):
rank / world from the environment, the static shard plan, and the hand-over of the 128-byte RCCL id from rank 0 to
the other ranks over a localhost TCP socket — no torch, no second runtime in the measured process.

The data path needs none of this: collectives run inside liborama_hip.so (orama_shard_*)."""
from __future__ import annotations

import os
import socket
import struct
import time
from dataclasses import dataclass

MAGIC = b"ORAMAUID"


@dataclass(frozen=True)
class RankEnv:
    rank: int
    local_rank: int
    world: int
    master_addr: str
    master_port: int

    @classmethod
    def from_env(cls) -> "RankEnv":
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
                   int(os.environ.get("WORLD_SIZE", "1")), os.environ.get("MASTER_ADDR", "127.0.0.1"),
                   int(os.environ.get("MASTER_PORT", "29500")))


@dataclass(frozen=True)
class ShardPlan:
    """Static contiguous ranges: shard g holds rows / documents [g*N/G, (g+1)*N/G)."""

    n_total: int
    world: int

    def range(self, rank: int) -> tuple[int, int]:
        return (self.n_total * rank) // self.world, (self.n_total * (rank + 1)) // self.world

    def rows(self, rank: int) -> int:
        lo, hi = self.range(rank)
        return hi - lo


def device_for(local_rank: int) -> int:
    """The GPU of a local rank: its own one.  A box with fewer GPUs than ranks can only run the job on a loopback
    transport (ORAMA_RCCL_LIB — tests/mock_rccl; real RCCL refuses two ranks on one device): ranks then share GPUs."""
    import ctypes as C

    from . import _native as N

    n = C.c_int()
    N.check(N.load().orama_device_count(C.byref(n)))
    if local_rank < n.value:
        return local_rank
    if not os.environ.get("ORAMA_RCCL_LIB"):
        raise RuntimeError(f"local rank {local_rank} needs its own GPU, this box shows {n.value} "
                           "(ranks may share a GPU only on a loopback transport: ORAMA_RCCL_LIB)")
    return local_rank % n.value


def self_launch(n_ranks: int, argv: list[str], timeout: float | None = None) -> int:
    """`python bench.py --gpus N` without an external launcher: start N copies of the calling script, one rank per GPU,
    with the RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment `torch.distributed.run` would give them.  Rank 0
    inherits stdout (its JSON line stays the last line); the other ranks' stdout goes to stderr.  Returns the worst
    exit code; a rank that dies takes the others down instead of leaving them in a collective."""
    import subprocess
    import sys

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, *argv], env=env, stdout=None if r == 0 else sys.stderr))
    deadline = None if timeout is None else time.monotonic() + timeout
    worst = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in alive:
                    q.terminate()
        if deadline is not None and time.monotonic() > deadline:
            for q in alive:
                q.kill()
            return worst or 124
        time.sleep(0.05)
    return worst


def _ports(env: RankEnv):
    # a short deterministic sequence next to the launcher's own rendezvous port (which is in use)
    base = env.master_port + 1
    return [1024 + (base + 37 * i - 1024) % (65535 - 1024) for i in range(8)]


def exchange_unique_id(env: RankEnv, make_id, timeout: float = 120.0) -> bytes:
    """Rank 0 calls `make_id()` (-> 128 bytes) and serves it; every other rank fetches it.  Returns the id."""
    if env.world == 1:
        return make_id()
    host = "127.0.0.1" if env.master_addr in ("localhost", "127.0.0.1", "") else env.master_addr
    deadline = time.monotonic() + timeout
    if env.rank == 0:
        uid = make_id()
        assert len(uid) == 128
        srv = None
        for port in _ports(env):
            try:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((host, port))
                break
            except OSError:
                srv.close()
                srv = None
        if srv is None:
            raise RuntimeError("rank 0 could not bind a bootstrap port")
        srv.listen(env.world)
        srv.settimeout(1.0)
        served = set()
        while len(served) < env.world - 1:
            if time.monotonic() > deadline:
                raise TimeoutError(f"bootstrap: only {len(served)} of {env.world - 1} ranks fetched the id")
            try:
                conn, _ = srv.accept()
            except socket.timeout:
                continue
            with conn:
                conn.settimeout(10.0)
                hello = conn.recv(len(MAGIC) + 4)
                if len(hello) == len(MAGIC) + 4 and hello[: len(MAGIC)] == MAGIC:
                    served.add(struct.unpack("<I", hello[len(MAGIC):])[0])
                    conn.sendall(MAGIC + uid)
        srv.close()
        return uid
    while True:
        for port in _ports(env):
            try:
                with socket.create_connection((host, port), timeout=2.0) as c:
                    c.settimeout(10.0)
                    c.sendall(MAGIC + struct.pack("<I", env.rank))
                    buf = b""
                    while len(buf) < len(MAGIC) + 128:
                        chunk = c.recv(len(MAGIC) + 128 - len(buf))
                        if not chunk:
                            break
                        buf += chunk
                    if len(buf) == len(MAGIC) + 128 and buf[: len(MAGIC)] == MAGIC:
                        return buf[len(MAGIC):]
            except OSError:
                pass
        if time.monotonic() > deadline:
            raise TimeoutError("bootstrap: rank 0 never served the RCCL id")
        time.sleep(0.2)
