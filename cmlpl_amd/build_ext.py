"""Build libcmlpl_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcmlpl_hip.so")
SOURCES = ["api.hip", "augment.hip", "conv0.hip", "conv3x3.hip", "dense.hip", "dist.hip", "head.hip", "loss.hip", "memobank.hip", "ntxent.hip", "optim.hip", "wgrad3x3.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def source_hash() -> str:
    """sha256 over the kernel sources and the C ABI header (sorted by name): identifies the code a recorded
    profile (profiles/pmc_traffic.json) was measured on; .git does not travel to the GPU box."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))
    files.append(os.path.join(HERE, "..", "include", "cmlpl.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


HASH_MARK = b"CMLPL_SOURCE_HASH="


def embedded_hash(lib=None):
    """The source hash a built library carries (api.hip is compiled with -DCMLPL_SOURCE_HASH, the string is also what
    `cmlpl_source_hash()` returns); None for a binary without one.  Read from the file: no dlopen."""
    try:
        blob = open(lib or LIB, "rb").read()
    except OSError:
        return None
    i = blob.find(HASH_MARK)
    return blob[i + len(HASH_MARK): i + len(HASH_MARK) + 16].decode("ascii", "replace") if i >= 0 else None


def needs_build():
    """True when the library is absent or was not built from the sources as they are now: the hash embedded in the
    binary is compared with the hash of the sources (mtimes lie after a checkout or a copy to another box)."""
    return embedded_hash() != source_hash()


def build(force=False, verbose=True, jobs=4):
    if not force and not needs_build():
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    shash = source_hash()
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [_hipcc(), *FLAGS, f'-DCMLPL_SOURCE_HASH="{shash}"', "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if len(procs) >= jobs:
            _drain(procs, verbose)
    _drain(procs, verbose)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    return LIB


def _drain(procs, verbose):
    while procs:
        src, p = procs.pop(0)
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(f"[{src}]\n{out}", file=sys.stderr)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
