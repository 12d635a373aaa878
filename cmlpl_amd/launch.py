"""One process per GPU: start the ranks of a single-node job as fresh child processes.

The reference is single-process (train.py:12 pins one device); this launcher is the build's own.  The parent
never touches the GPU (no HIP call, no ``torch.cuda`` use): it only sets the rendezvous environment
(``RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT``, as ``torch.distributed.run`` would), starts
``world`` children of the same program, relays rank 0's stdout and waits for all of them.  If any child
fails, the others are terminated (by PID) and the parent reports the failure.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence, Tuple


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launched_by_rendezvous_env(env=os.environ) -> bool:
    """True when this process already is one rank of a job (torchrun or spawn_ranks set the variables)."""
    return "RANK" in env and "WORLD_SIZE" in env


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # dmabuf IPC: the host driver of this pool supports no other kind, and without it RCCL (and any CUDA-tensor sharing
    # across processes) fails with `hipIpcGetMemHandle: invalid argument`.  Only a default: an operator's own setting wins.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


DEFAULT_TIMEOUT_S = 3600.0      # a rank stuck in a collective must not hang the parent for ever
# Exit code of a rank whose rendezvous store could not bind MASTER_PORT (another process took the port between
# free_port()'s release and rank 0's bind): the ONLY failure a job is started again for.  A deterministic early failure
# (too few GPUs, bad arguments, a --check mismatch) exits with its own code and is reported as it is, once.
EXIT_PORT_TAKEN = 98


def port_taken(exc: BaseException) -> bool:
    """does this exception say that the rendezvous port was already bound? (cmlpl_amd.distributed.init_distributed
    turns it into EXIT_PORT_TAKEN)"""
    s = f"{type(exc).__name__}: {exc}".lower()
    return "address already in use" in s or "eaddrinuse" in s or "errno: 98" in s


def spawn_ranks(world: int, argv: Sequence[str], extra_env: Optional[Dict[str, str]] = None,
                timeout: Optional[float] = DEFAULT_TIMEOUT_S, poll_s: float = 0.05, retries: int = 1) -> Tuple[int, str]:
    """Run ``argv`` as ``world`` ranks; returns (exit code, rank 0's stdout).  Ranks > 0 write their stdout to
    this process's stderr; every rank's stderr is inherited.  Exit code = first non-zero child code, else 0;
    124 on timeout (the children are stopped).  The rendezvous port is picked by bind-and-release, so another
    process can take it before rank 0 binds it: a job whose rank reports exactly that (EXIT_PORT_TAKEN) is started
    again, as fresh children on a new port, at most ``retries`` times; any other failure is returned as it is."""
    rc, out = _spawn_once(world, argv, extra_env, timeout, poll_s)
    while rc == EXIT_PORT_TAKEN and retries > 0 and world > 1:
        retries -= 1
        sys.stderr.write("cmlpl_amd.launch: the rendezvous port was taken by another process; starting the job once "
                         "more on a new port\n")
        rc, out = _spawn_once(world, argv, extra_env, timeout, poll_s)
    return rc, out


def _spawn_once(world: int, argv: Sequence[str], extra_env: Optional[Dict[str, str]],
                timeout: Optional[float], poll_s: float) -> Tuple[int, str]:
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs: List[subprocess.Popen] = []
    for r in range(world):
        env = rank_env(r, world, port)
        if extra_env:
            env.update(extra_env)
        # ranks > 0: stdout -> fd 2 (this process's stderr, also when sys.stderr has been replaced by a capture)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=subprocess.PIPE if r == 0 else 2, stderr=None,
                                      text=(r == 0)))
    chunks: List[str] = []
    reader = threading.Thread(target=lambda: chunks.extend(iter(procs[0].stdout.readline, "")), daemon=True)
    reader.start()                    # rank 0's pipe is drained while we wait, whatever it prints
    t0 = time.monotonic()
    rc = 0
    pending = set(range(world))
    try:
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0 and rc == 0:
                        rc = code if code > 0 else 1
            if rc != 0:
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                rc = 124
                break
            if pending:
                time.sleep(poll_s)
    finally:
        for r in pending:                       # a rank failed or timed out: stop exactly the PIDs we started
            if procs[r].poll() is None:
                procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=10)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        reader.join(timeout=10)
    return rc, "".join(chunks)
