"""CPU: the hot kernels must not spill.  hipcc cross-compiles conv3x3.hip and wgrad3x3.hip for gfx950 with
-Rpass-analysis=kernel-resource-usage; every conv3x3_kernel / wgrad3b_kernel / wgrad3r_kernel instantiation has to report
ScratchSize 0, and the fused per-sample kernels at least 2 waves per SIMD (they rely on two co-resident
workgroups per CU).
Guards against the register blow-ups that loop-unrolling experiments produced."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_hot_kernels_have_no_scratch(tmp_path):
    # the two heavy translation units, compiled side by side
    procs = [subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-c",
                               os.path.join(ROOT, "cmlpl_amd", "csrc", f), "-o", str(tmp_path / (f + ".o")),
                               "-Rpass-analysis=kernel-resource-usage"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for f in ("conv3x3.hip", "wgrad3x3.hip")]
    stderr = ""
    for pr in procs:
        _, err = pr.communicate(timeout=900)
        assert pr.returncode == 0, err[-2000:]
        stderr += err
    blocks = re.split(r"remark: Function Name: ", stderr)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if not any(k in name for k in ("conv3x3_kernel", "wgrad3r_kernel", "wgrad3b_kernel", "wgrad3b_pair_kernel", "conv3x3_small_kernel")):
            continue
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
        assert scratch == 0, (name, scratch)
        if "conv3x3_kernelILi2E" in name or "conv3x3_kernelILi3E" in name:   # the fused per-sample kernels
            assert occ >= 2, (name, occ)
        seen += 1
    assert seen >= 20
    # the kernels with a two-piece fp16 tap loop (and the three-piece loop beside it): the per-sample ones -- forward and
    # backward: four waves, eight waves, eight waves with eight tiles -- and the one-sample general ones -- forward and
    # data gradient: eight waves with one / two tiles per wave, four waves with the barrier-free loop
    names = [b.split()[0] for b in blocks if "conv3x3_kernel" in b.split()[0]]
    assert sum(1 for n in names if n.endswith("Lb0ELb1EEEvNS_9Conv3ArgsE")) == 10, names
    assert sum(1 for n in names if n.endswith("Lb1ELb1EEEvNS_9Conv3ArgsE")) == 2, names


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_kernel_of_the_library_uses_scratch(tmp_path):
    """Every kernel of the other translation units too: a kernel-argument struct that escapes into a generic pointer
    (e.g. `p ? p->x : a.x` selecting between two ADDRESSES) is copied to scratch whole -- round 4 doubled
    pair_exp16_kernel's time that way before this test existed."""
    files = [f for f in sorted(os.listdir(os.path.join(ROOT, "cmlpl_amd", "csrc")))
             if f.endswith(".hip") and f not in ("conv3x3.hip", "wgrad3x3.hip")]
    procs = [(f, subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-c",
                                   os.path.join(ROOT, "cmlpl_amd", "csrc", f), "-o", str(tmp_path / (f + ".o")),
                                   "-Rpass-analysis=kernel-resource-usage"], stdout=subprocess.PIPE,
                                  stderr=subprocess.PIPE, text=True)) for f in files]
    seen = 0
    for f, pr in procs:
        _, err = pr.communicate(timeout=900)
        assert pr.returncode == 0, err[-2000:]
        for b in re.split(r"remark: Function Name: ", err)[1:]:
            scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
            assert scratch == 0, (f, b.split()[0], scratch)
            seen += 1
    assert seen >= 40
