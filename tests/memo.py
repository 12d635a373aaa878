"""Memo of ORACLE results for the GPU tests (test infrastructure).

Several tests hold different kernel variants to the oracle on the SAME inputs (tests/test_gpu_env_paths.py walks the
CMLPL_* switches), and the oracle dominated their run time: `ntxent_loss` at B = 512 builds the reference's
[2B, 2B, D] broadcast (17 s), the eight-rank configurations replay 1024-row steps on the CPU (5 s each).  The oracle is
deterministic, so its result for given inputs is computed once: in this process (`_MEM`) and, for
results small enough to keep, in a file under the temporary directory shared with the child processes of the suite.
The key carries a hash of oracle/cmlpl_oracle.py: an edited oracle never meets a stale result."""
import hashlib
import os
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_MEM = {}
_SRC = None


def _oracle_hash():
    global _SRC
    if _SRC is None:
        _SRC = hashlib.sha256(open(os.path.join(ROOT, "oracle", "cmlpl_oracle.py"), "rb").read()).hexdigest()[:16]
    return _SRC


def tensor_key(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()[:24]


def memo(key: str, compute, disk: bool = True):
    """compute() -> anything torch.save can store; `key` must name the inputs completely"""
    key = f"{_oracle_hash()}-{key}"
    if key in _MEM:
        return _MEM[key]
    path = None
    if disk:
        d = os.path.join(tempfile.gettempdir(), f"cmlpl_test_memo_{os.getuid()}")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, hashlib.sha256(key.encode()).hexdigest()[:32] + ".pt")
        if os.path.exists(path):
            try:
                _MEM[key] = torch.load(path, weights_only=False)
                return _MEM[key]
            except Exception:       # noqa: BLE001  (a torn file of a killed run: compute again)
                pass
    val = compute()
    _MEM[key] = val
    if path is not None:
        try:
            tmp = f"{path}.{os.getpid()}.tmp"
            torch.save(val, tmp)
            os.replace(tmp, path)
        except Exception:           # noqa: BLE001  (no disk: the in-process memo still serves)
            pass
    return val
