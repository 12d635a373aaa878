"""Kernel variants that the default planners do not pick at test sizes -- the wide pair_exp kernel (chosen for >= 4096
bank columns, i.e. data-parallel jobs of 4 ranks and more), the general weight-gradient fallback, the unfused
conv0 kernels on a fusable shape -- are forced through their environment switches and must pass the same parity tests.

The switches are read once per process; since round 6 the library re-reads them on request
(cmlpl_debug_reload_switches), so the variants of a GROUP run in ONE child process (tests/_env_paths_child.py: set the
environment, reload, run the pytest selection or the Philox trajectory) instead of one interpreter + library load +
device start-up per variant (35 child processes in round 5).  A test asks for its variant's result; the first test of a
group to ask runs the group's child."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

JOBS = {}          # name -> job (see _env_paths_child.py), registered at import by _job / _traj
_DONE = {}         # group -> {name: result}


def _job(group, name, env, select):
    JOBS[name] = dict(name=name, group=group, kind="pytest", env=env, select=select)
    return name


def _traj(group, name, env, *args):
    JOBS[name] = dict(name=name, group=group, kind="traj", env=env, args=list(args))
    return name


def _result(name):
    group = JOBS[name]["group"]
    if group not in _DONE:
        jobs = [j for j in JOBS.values() if j["group"] == group]
        with tempfile.TemporaryDirectory() as td:
            jp, rp = os.path.join(td, "jobs.json"), os.path.join(td, "results.json")
            json.dump(jobs, open(jp, "w"))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_env_paths_child.py"), jp, rp], cwd=ROOT,
                               capture_output=True, text=True, timeout=900)
            res = json.load(open(rp)) if os.path.exists(rp) else {}
        _DONE[group] = (res, r.returncode, r.stdout[-2000:] + r.stderr[-2000:])
    res, rc, tail = _DONE[group]
    assert name in res, f"group {group}: child ended (rc {rc}) before {name}: {tail}"
    return res[name]


def _run(name):
    """a pytest selection under the variant's switches must pass"""
    r = _result(name)
    assert r["rc"] == 0, f"{JOBS[name]['env']}: {r['tail']}"


def _lines(name):
    r = _result(name)
    assert r["rc"] == 0, f"{JOBS[name]['env']}: {r.get('tail')}"
    return r["lines"]


# (selections: the loss block in its regimes against the oracle, the sharded step at W = 2 / 4 / 8 against the single-GPU
#  engine -- the packed exchange buffers -- and configs[2] as eight ranks against the oracle: 64 + 64 local rows meet 5120
#  bank columns, the sizes these kernels are planned for)
_PAIR_SEL = ["tests/test_gpu_ops.py", "tests/test_gpu_distributed.py", "-k",
             "loss_block or sharded_step_equals or (eight_rank and B3)"]
J_TALL = _job("pair", "tall", {"CMLPL_PAIR_TALL": "1"},
              ["tests/test_gpu_ops.py", "tests/test_gpu_distributed.py", "-k", "loss_block or sharded_step_equals"])


def test_tall_pair_exp_kernel_passes_loss_parity():
    """pair_exp_tall_kernel (wide products with more than 128 local rows) forced at the test sizes"""
    _run(J_TALL)


WIDE = [("0", "0"), ("4", "1"), ("4", "2"), ("4", "3"), ("4", "4"), ("2", "2")]
# the planner's tile shape also runs the eight-rank configuration and the headline batch against the reference fixture;
# every other forced tile shape runs the loss block (plain buffers) and one or two sharded steps (packed exchange buffers)
J_WIDE = {(mb, nbw): _job("pair", f"wide-{mb}-{nbw}", {"CMLPL_PAIR_WIDE": "1", "CMLPL_PAIR_MB": mb, "CMLPL_PAIR_NBW": nbw},
                          ["tests/test_gpu_ops.py", "tests/test_gpu_distributed.py", "tests/test_gpu_step.py", "-k",
                           "loss_block or sharded_step_equals or (eight_rank and B3) or b2_b256" if (mb, nbw) == ("0", "0")
                           else "loss_block or (sharded_step_equals and (8-B2 or 2-P))" if (mb, nbw) in (("4", "1"), ("4", "2"))
                           else "loss_block or (sharded_step_equals and 8-B2)"])
          for mb, nbw in WIDE}


@pytest.mark.parametrize("mb,nbw", WIDE)
def test_wide_pair_exp_kernel_passes_loss_parity(mb, nbw):
    """pair_exp_wide_kernel (what >= 4096 bank columns take: data-parallel jobs) forced at the test sizes, with the planner's
    tile shape and with every other one forced"""
    _run(J_WIDE[(mb, nbw)])


J_DFEAT = _job("pair", "dfeat-direct", {"CMLPL_DFEAT_LDS": "0"},
               ["tests/test_gpu_ops.py", "tests/test_gpu_distributed.py", "-k", "loss_block or (sharded_step_equals and (8-B2 or 4-B2 or 2-P))"])


def test_direct_load_feature_gradient_gemms_pass_loss_parity():
    """gemm_tn_block for the two feature-gradient GEMMs (what row counts that are not multiples of four take) instead of
    loss_dfeat_lds_kernel"""
    _run(J_DFEAT)


J_PAIR32 = _job("pair", "pair32", {"CMLPL_PAIR16": "0"}, ["tests/test_gpu_ops.py", "tests/test_gpu_step.py", "-k", "loss_block or b2_64"])


def test_32_row_pair_exp_kernel_passes_loss_parity():
    """pair_exp_kernel (32 x 32 tiles, the round-2 default; still what K > 32 classes take) instead of pair_exp16_kernel"""
    _run(J_PAIR32)


J_NTX = [_job("next", "ntx-vector", {"CMLPL_NTX_MFMA": "0"}, ["tests/test_ntxent.py", "-k", "not (512-1024 or 256-1024)"])] + [
    _job("next", f"ntx-ncw{ncw}", {"CMLPL_NTX_NCW": ncw},
         ["tests/test_ntxent.py", "-k", ("512-1024 or " if ncw == "4" else "") + "300-516 or 150-300 or 32-128"])
    for ncw in ("1", "2", "4")]


def test_vector_ntxent_gradient_kernel_passes_oracle_parity():
    """ntx_grad_kernel (embedding widths that are not a multiple of 4) instead of ntx_grad_mfma_kernel; every slice width
    of ntx_grad_mfma_kernel<NCW, KC> (the planner picks one per size: 64 / 128 / 256 columns per workgroup) at every size"""
    for j in J_NTX:
        _run(j)


J_MBFAST = _job("next", "mb-general", {"CMLPL_MB_FAST": "0"}, ["tests/test_losshelper.py"])


def test_general_memobank_infonce_kernel_passes_losshelper_parity():
    """mb_infonce_all_kernel (keys re-read from L2; what D > 1024, K > 64 or more than 16 key slots per wave take) instead
    of the keys-in-registers kernel"""
    _run(J_MBFAST)


J_UNSUP_ML = _job("next", "unsup-multi", {"CMLPL_UNSUP_ONEWG": "0"}, ["tests/test_losshelper.py", "-k", "unsupervised"])


def test_multi_launch_unsupervised_loss_passes_losshelper_parity():
    """the rank-counting kernels (what more than 8192 rows take) instead of the one-workgroup radix-select kernel"""
    _run(J_UNSUP_ML)


J_UNSUP_1 = _job("next", "unsup-onewg", {"CMLPL_UNSUP_3L": "0"}, ["tests/test_losshelper.py", "-k", "unsupervised"])


def test_one_workgroup_unsupervised_loss_up_to_8192_rows():
    """the single-workgroup kernel also where the three-launch path (1024 < B <= 8192) is the default"""
    _run(J_UNSUP_1)


J_WG_GEN = _job("conv", "wgrad-general", {"CMLPL_WGRAD3_R": "0"},
                ["tests/test_gpu_ops.py", "-k", "forward_backward and not (W12 or W16 or B4 or B5)"])


def test_general_wgrad_fallback_passes_backward_parity():
    _run(J_WG_GEN)


J_WG_F32 = _job("conv", "wgrad-f32", {"CMLPL_WGRAD3_B3": "0"},
                ["tests/test_gpu_ops.py", "-k", "forward_backward and (B2 or B4 or P or W8 or W13 or W18)"])


def test_f32_mfma_row_split_wgrad_passes_backward_parity():
    """with the split-bf16 weight gradient switched off: wgrad3r_kernel (f32-input MFMA, LDS-DMA staging; kept for the
    two maps of the headline shape as the reference point) and the general wgrad3_kernel for every other window"""
    _run(J_WG_F32)


J_WG_SEP = _job("conv", "wgrad-separate", {"CMLPL_WGRAD3_PAIR": "0"}, ["tests/test_gpu_ops.py", "-k", "forward_backward and B2"])


def test_separate_weight_gradient_launches_pass_backward_parity():
    _run(J_WG_SEP)


J_UNF0 = _job("conv", "conv0-unfused", {"CMLPL_FUSE_CONV0": "0"}, ["tests/test_gpu_ops.py", "-k", "forward_backward and B2"])


def test_unfused_conv0_kernels_pass_on_a_fusable_shape():
    _run(J_UNF0)


J_UNFSPE = _job("conv", "spe-unfused", {"CMLPL_FUSE_SPE": "0"}, ["tests/test_gpu_step.py", "-k", "b2_64 or b5"])

J_GEN_R5 = _job("conv", "general-r5", {"CMLPL_CONV0A": "0", "CMLPL_CONV3_KS": "0"},
                ["tests/test_gpu_ops.py", "tests/test_gpu_step.py", "-k", "(forward_backward and (P or W8 or W16)) or p_short or non_finite"])


def test_round5_general_path_kernels_pass_parity():
    """augment_kernel + conv0_fwd_kernel (f32-input MFMA: what windows that are not a multiple of 8 pixels, or more than
    128 channels, still take) and the LDS-staged tap weights of the one-tile general 3x3 kernels, instead of round 6's
    conv0a_fwd_kernel / barrier-free tap loop, on the reference's own 20 x 20 window and on 8 x 8 / 16 x 16 ones"""
    _run(J_GEN_R5)


def test_unfused_spectral_branch_passes_step_parity():
    _run(J_UNFSPE)


J_B3ONLY = _job("conv", "three-piece-conv1", {"CMLPL_F16X2": "0"},
                ["tests/test_gpu_ops.py", "tests/test_gpu_step.py", "-k", "(forward_backward and B2) or error_bound or b2_b256"])


def test_three_piece_tap_loops_of_the_per_sample_kernels_pass_parity():
    """CMLPL_F16X2=0: conv1's tap loops of the four-tile per-sample kernels on three bf16 pieces (the loop every sample
    whose operands leave the two-piece scheme's ranges still takes: tests/test_gpu_f16x2.py) against the oracle, the
    reference fixture and the per-element error bounds"""
    _run(J_B3ONLY)


_UNFUSED = {"CMLPL_FUSE_CONV0": "0", "CMLPL_FUSE_TAIL": "0", "CMLPL_FUSE_SPE": "0"}
J_NOISE = [_traj("traj", "noise-fused", {}, "B2", 64, 64), _traj("traj", "noise-unfused", _UNFUSED, "B2", 64, 64),
           _traj("traj", "noise-fused-again", {}, "B2", 64, 64)]


def test_fused_and_unfused_conv0_form_the_same_noise():
    """In Philox mode the fused kernels augment the raw patches in LDS (forward) and regenerate the same noise for
    conv0's weight gradient (backward), and the spectral kernel augments the spectra in registers; the unfused
    fallback reads augmented copies written by the augmentation kernel.  Same counters => the same augmented values: losses of the first step agree to rounding of the different
    summation orders, and both stay deterministic."""
    outs = [[ln.split() for ln in _lines(j) if ln.startswith(("0x", "-0x"))] for j in J_NOISE]
    fused, unfused, fused2 = outs
    assert fused == fused2                                    # bit-identical rerun
    f0 = [float.fromhex(v) for v in fused[0]]
    u0 = [float.fromhex(v) for v in unfused[0]]
    for a, b in zip(f0[:9], u0[:9]):                          # first step: same inputs, same noise
        assert abs(a - b) <= 2e-5 * abs(b) + 1e-7, (f0, u0)
    assert f0[9:13] == u0[9:13]                               # mask / graph counts


KS8_CASES = [("B2", 64, 64), ("B2", 128, 128), ("B4", 24, 40), ("W8", 16, 16), ("W10", 8, 24)]
J_KS8 = {c: [_traj("traj", f"ks8-{ks8}-{c[0]}-{c[1]}-{c[2]}", {"CMLPL_KS8": ks8}, *c) for ks8 in ("0", "1")] for c in KS8_CASES}


@pytest.mark.parametrize("shape,bt,btu", KS8_CASES)
def test_eight_wave_per_sample_kernels_are_bit_identical_to_the_four_wave_ones(shape, bt, btu):
    """conv3x3_kernel<2 / 3, 1, 1, 8> (one eight-wave workgroup per sample-net: what a launch of at most one workgroup per CU
    takes, CMLPL_KS8=1 forces it at every size) against the four-wave kernels (CMLPL_KS8=0) on the same in-kernel random
    streams: every accumulator sees the same operands in the same order (wave = (pixel tile, channel half) instead of
    (pixel half, channel half); the halves meet in the same sum), so three steps must agree BIT for bit -- parameters,
    Adam moments, gradients, banks, logits, features."""
    outs = [_lines(j) for j in J_KS8[(shape, bt, btu)]]
    assert outs[0][-1].startswith("sha ") and outs[0] == outs[1], (outs[0][-2:], outs[1][-2:])


NW8_CASES = [("P", 32, 48, {}), ("W12", 20, 30, {"CMLPL_FUSE_BIG": "0", "CMLPL_CONV3_S": "2"})]
# (both on the three-piece tap loop: the four-wave general plan has no two-piece kernels, the eight-wave one takes them by
#  default -- what is compared here is the wave arrangement)
J_NW8 = {c[0]: [_traj("traj", f"nw8-{nw8}-{c[0]}", dict(c[3], CMLPL_CONV3_NW8=nw8, CMLPL_F16X2="0"), *c[:3]) for nw8 in ("0", "1")]
         for c in NW8_CASES}


@pytest.mark.parametrize("shape,bt,btu,extra", NW8_CASES)
def test_eight_wave_general_kernels_are_bit_identical_to_the_four_wave_plan(shape, bt, btu, extra):
    """conv3x3_kernel<0 / 1, MTW, 0, 8, 1> (general 3x3 kernels with eight waves: what plan_conv3 picks where one workgroup
    fills a CU and a wave would carry two or more pixel tiles -- the reference's own 20x20x60 windows) against the
    four-wave plan (CMLPL_CONV3_NW8=0): the same tiles on other waves, the odd tile of a last round of 1 or 5 tiles shared
    by two waves (one output-channel tile each: 13 tiles at 20x20, 9 at two 12x12 windows per workgroup), every
    accumulator fed in the same order -- three steps agree bit for bit (on the three-piece loop, CMLPL_F16X2=0)."""
    outs = [_lines(j) for j in J_NW8[shape]]
    assert outs[0][-1].startswith("sha ") and outs[0] == outs[1], (outs[0][-2:], outs[1][-2:])


J_KS8_PAR = _job("ks", "ks8-parity", {"CMLPL_KS8": "1"},
                 ["tests/test_gpu_step.py", "tests/test_gpu_ops.py", "tests/test_gpu_indexed_graph.py", "-k",
                  "b2_b256 or b2_peaky or b4_b256 or deadrelu or (forward_backward and (B2 or B4)) or (indexed and B2) or (graph_replay and B2)"])


def test_eight_wave_per_sample_kernels_pass_parity():
    """... and the oracle / golden parity tests with the eight-wave kernels forced at every batch size (explicit noise, explicit
    dropout masks, batches by index, graph replay)"""
    _run(J_KS8_PAR)


J_KS4_PAR = _job("ks", "ks4-parity", {"CMLPL_KS8": "0"},
                 ["tests/test_gpu_step.py", "tests/test_gpu_ops.py", "-k", "b2_64 or b4_64 or (forward_backward and B2)"])


def test_four_wave_per_sample_kernels_pass_parity_at_small_batches():
    """the four-wave kernels where the planner now picks the eight-wave ones (grids of at most one workgroup per CU)"""
    _run(J_KS4_PAR)
