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


def preflight(n_ranks: int, timeout: float = 120.0) -> dict:
    """What a multi-rank launch needs, checked in a throw-away child process BEFORE any rank starts (the launcher itself
    never touches the GPU runtime): the library loads, the box shows at least `n_ranks` devices (or the transport is a
    loopback one, ORAMA_RCCL_LIB, on which ranks may share GPUs), RCCL resolves and can make a communicator id.
    Returns {"ok", "devices", "rccl", "loopback", "error"}."""
    import json
    import subprocess
    import sys

    code = ("import ctypes as C, json, os\n"
            "from oramacore_amd import _native as N\n"
            "out = {'devices': 0, 'rccl': False, 'loopback': bool(os.environ.get('ORAMA_RCCL_LIB')), 'error': ''}\n"
            "try:\n"
            "    lib = N.load(); n = C.c_int(); N.check(lib.orama_device_count(C.byref(n))); out['devices'] = n.value\n"
            "    buf = C.create_string_buffer(128); N.check(lib.orama_shard_unique_id(buf)); out['rccl'] = True\n"
            "except Exception as e:\n"
            "    out['error'] = str(e)[:500]\n"
            "print('PREFLIGHT ' + json.dumps(out))\n")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    try:
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"ok": False, "devices": 0, "rccl": False, "loopback": False, "error": f"preflight timed out after {timeout:.0f} s"}
    out = None
    for line in r.stdout.splitlines():
        if line.startswith("PREFLIGHT "):
            out = json.loads(line[len("PREFLIGHT "):])
    if out is None:
        return {"ok": False, "devices": 0, "rccl": False, "loopback": False,
                "error": f"preflight child exited {r.returncode}: {(r.stderr or r.stdout)[-500:]}"}
    if not out["error"] and out["devices"] < n_ranks and not out["loopback"]:
        out["error"] = (f"{n_ranks} ranks need {n_ranks} GPUs, this box shows {out['devices']} (ranks may share a GPU only on a "
                        "loopback transport: ORAMA_RCCL_LIB)")
    out["ok"] = not out["error"]
    return out


def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n_ranks: int, argv: list[str], timeout: float | None = None, grace: float = 10.0) -> int:
    """`python bench.py --gpus N` without an external launcher: start N copies of the calling script, one rank per GPU,
    with the RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment `torch.distributed.run` would give them.  Rank 0
    inherits stdout (its JSON line stays the last line); the other ranks' stdout goes to stderr.  Every rank's stderr
    is teed into a temporary file so that a failure can be REPORTED: when a rank dies (or the timeout strikes) the other
    ranks get SIGTERM, then SIGKILL after `grace` seconds, and the launcher prints one JSON line
    {"error": ..., "failed_rank": r, "rank_stderr_tail": {...}} as the last line of stdout.  Returns the worst exit code.

    The rendezvous port is only a hint: rank 0 binds the first free port of a short sequence next to it and the others
    probe the same sequence (exchange_unique_id), so a port taken between this probe and rank 0's bind costs a retry,
    not the job."""
    import json
    import subprocess
    import sys
    import tempfile
    import threading

    port = _free_port()
    tmp = tempfile.mkdtemp(prefix="orama_launch_")
    procs, logs, tees = [], [], []

    def tee(pipe, path):
        with open(path, "wb") as f:
            for chunk in iter(lambda: pipe.readline(), b""):
                f.write(chunk)
                f.flush()
                try:
                    sys.stderr.buffer.write(chunk)
                    sys.stderr.buffer.flush()
                except Exception:  # noqa: BLE001 - stderr closed: keep the file
                    pass

    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (the host driver supports nothing else); the user's value wins
        p = subprocess.Popen([sys.executable, *argv], env=env, stdout=None if r == 0 else sys.stderr, stderr=subprocess.PIPE)
        procs.append(p)
        logs.append(os.path.join(tmp, f"rank{r}.stderr"))
        t = threading.Thread(target=tee, args=(p.stderr, logs[-1]), daemon=True)
        t.start()
        tees.append(t)

    def stop_others(alive):
        for q in alive:
            q.terminate()
        t_end = time.monotonic() + grace
        while any(q.poll() is None for q in alive) and time.monotonic() < t_end:
            time.sleep(0.05)
        for q in alive:
            if q.poll() is None:
                q.kill()

    deadline = None if timeout is None else time.monotonic() + timeout
    worst, failed_rank, reason = 0, None, ""
    alive = list(procs)
    while alive:
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0 and not worst:
                worst, failed_rank, reason = rc, procs.index(p), f"rank {procs.index(p)} exited with status {rc}"
                stop_others(alive)
        if alive and deadline is not None and time.monotonic() > deadline:
            worst, reason = worst or 124, reason or f"timeout after {timeout:.0f} s"
            stop_others(alive)
        time.sleep(0.05)
    for t in tees:
        t.join(timeout=2.0)
    if worst:
        tails = {}
        for r, path in enumerate(logs):
            try:
                with open(path, "rb") as f:
                    tails[f"rank{r}"] = f.read()[-1500:].decode(errors="replace")
            except OSError:
                tails[f"rank{r}"] = ""
        sys.stdout.flush()
        print(json.dumps({"error": reason, "failed_rank": failed_rank, "n_gpus": n_ranks, "rank_stderr_tail": tails}), flush=True)
    import shutil

    shutil.rmtree(tmp, ignore_errors=True)
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
