"""ctypes binding of libcmlpl_hip.so (include/cmlpl.h).

The HIP library is the product: there is no CPU fallback.  ``load()`` raises
``CmlplLibraryError`` when the shared object is missing or does not export the
full C ABI; every compute entry point raises on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CMLPL_LIB") or os.path.join(HERE, "libcmlpl_hip.so")
ABI_VERSION = 4
NUM_TENSORS = 16
NUM_LIVE = 10

# state_dict keys of BaseNet2 in the order of the flat buffer (live tensors first)
TENSOR_KEYS = (
    "conv0.weight", "conv0.bias", "conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias",
    "feat_spe.weight", "feat_spe.bias", "classifier.weight", "classifier.bias",
    "feat_ss.weight", "feat_ss.bias", "feat_ss2.weight", "feat_ss2.bias", "feat_ss3.weight", "feat_ss3.bias",
)

EXPORTS = (
    "cmlpl_abi_version", "cmlpl_layout", "cmlpl_workspace_bytes", "cmlpl_pack_weights", "cmlpl_augment",
    "cmlpl_basenet2_fwd", "cmlpl_basenet2_bwd", "cmlpl_loss_fwd_bwd", "cmlpl_adam_step", "cmlpl_train_step",
    "cmlpl_debug_region", "cmlpl_timing_begin", "cmlpl_timing_end", "cmlpl_loss_phase1", "cmlpl_loss_phase2",
    "cmlpl_dist_unpack", "cmlpl_loss_workspace_bytes", "cmlpl_extract_patches", "cmlpl_ntxent_workspace_bytes", "cmlpl_ntxent_fwd_bwd",
    "cmlpl_unsup_workspace_bytes", "cmlpl_unsup_loss", "cmlpl_memobank_select", "cmlpl_memobank_proto",
    "cmlpl_memobank_enqueue", "cmlpl_memobank_push", "cmlpl_memobank_infonce", "cmlpl_memobank_sum",
    "cmlpl_forward", "cmlpl_backward", "cmlpl_loss_phase1_g", "cmlpl_loss_phase2_g", "cmlpl_memobank_loss",
    "cmlpl_source_hash", "cmlpl_dyn_adam", "cmlpl_step_graph_create", "cmlpl_step_graph_launch",
    "cmlpl_step_graph_destroy", "cmlpl_infer_workspace_bytes", "cmlpl_infer_cube", "cmlpl_dist_stage_graph_create",
    "cmlpl_debug_reload_switches", "cmlpl_forward_spectral", "cmlpl_forward_spatial", "cmlpl_backward_data",
    "cmlpl_backward_weights", "cmlpl_dist_step", "cmlpl_rccl_bind", "cmlpl_rccl_unbind", "cmlpl_debug_two_piece",
)

KERNEL_NAMES = ("augment", "conv0_fwd", "conv1_fwd", "conv2_fwd", "spe_fwd", "head_fwd", "loss", "head_bwd",
                "cls_wgrad", "spe_wgrad", "conv2_dgrad", "conv2_wgrad", "conv2_wred", "conv1_dgrad", "conv1_wgrad",
                "conv1_wred", "conv0_wgrad", "adam", "pack", "loss_graph", "loss_fin", "loss_dfeat")


class CmlplLibraryError(RuntimeError):
    pass


class CmlplError(RuntimeError):
    def __init__(self, fn, rc):
        names = {-1: "CMLPL_E_ARG", -2: "CMLPL_E_SHAPE", -3: "CMLPL_E_WORKSPACE", -4: "CMLPL_E_COMM"}
        what = names.get(rc, f"hipError_t {rc}" if rc > 0 else (f"CMLPL_E_COMM (ncclResult_t {-4 - rc})" if rc < -4 else str(rc)))
        super().__init__(f"{fn} failed: {what}")
        self.rc = rc


class Shape(C.Structure):
    _fields_ = [("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("bands", C.c_int32), ("K", C.c_int32)]


class HParams(C.Structure):
    _fields_ = [(k, C.c_float) for k in (
        "lr", "beta1", "beta2", "eps", "temperature", "alpha", "noise_sigma", "dropout_p",
        "w_contrast", "w_mutual", "pos_thr", "neg_thr")]


class Layout(C.Structure):
    _fields_ = [("param_off", C.c_int64 * NUM_TENSORS), ("param_numel", C.c_int64 * NUM_TENSORS),
                ("param_total", C.c_int64), ("param_live", C.c_int64), ("packed_total", C.c_int64),
                ("cls_in", C.c_int32), ("reserved", C.c_int32)]


class Shard(C.Structure):
    _fields_ = [(k, C.c_int32) for k in ("bt_g", "btu_g", "lab0", "nlab", "unl0", "nunl")]


class Batch(C.Structure):
    _fields_ = [("d_xpl", C.c_void_p), ("d_xl", C.c_void_p), ("d_xpu", C.c_void_p), ("d_xu", C.c_void_p),
                ("d_labels", C.c_void_p), ("noise8", C.POINTER(C.c_void_p)), ("bt", C.c_int32), ("btu", C.c_int32),
                ("d_lab_idx", C.c_void_p), ("d_unl_idx", C.c_void_p)]


class Dyn(C.Structure):
    """cmlpl_dyn: one step's scalars in device memory (graph replay); DYN_DTYPE is the same row as a numpy record"""
    _fields_ = [("step", C.c_uint64), ("adam_t", C.c_int64), ("lab_off", C.c_int64), ("unl_off", C.c_int64),
                ("ptr", C.c_int32 * 2), ("smooth", C.c_int32), ("adap_mask", C.c_float), ("hist_row", C.c_int32),
                ("adam_step_size", C.c_float), ("adam_bc2_sqrt", C.c_float), ("reserved", C.c_int32)]


DYN_DTYPE = [("step", "<u8"), ("adam_t", "<i8"), ("lab_off", "<i8"), ("unl_off", "<i8"), ("ptr", "<i4", (2,)),
             ("smooth", "<i4"), ("adap_mask", "<f4"), ("hist_row", "<i4"), ("adam_step_size", "<f4"),
             ("adam_bc2_sqrt", "<f4"), ("reserved", "<i4")]


class MemobankCall(C.Structure):
    _fields_ = ([(k, C.c_void_p) for k in ("d_rep", "d_rep_teacher", "d_prob_l", "d_prob_u", "d_label_l", "d_label_u",
                                           "d_low_mask", "d_high_mask")] +
                [(k, C.c_int32) for k in ("N", "n_labeled", "K", "D", "queries", "negatives")] +
                [("d_bank", C.c_void_p), ("d_state", C.c_void_p), ("d_capacity", C.c_void_p), ("capacity_stride", C.c_int32),
                 ("d_anchor_draw", C.c_void_p), ("d_neg_draw", C.c_void_p), ("seed", C.c_uint64), ("call", C.c_uint64),
                 ("d_momentum", C.c_void_p), ("d_momentum_on", C.c_void_p), ("ema", C.c_float), ("d_prototype", C.c_void_p),
                 ("temperature", C.c_float)] +
                [(k, C.c_void_p) for k in ("d_lists", "d_counts", "d_proto", "d_keys_log", "d_lossq", "d_ganchor", "d_arow",
                                           "d_drep", "d_total")])


class Gathered(C.Structure):
    _fields_ = [("d_recv_feat", C.c_void_p), ("d_logits_local", C.c_void_p), ("world", C.c_int32), ("bt_local", C.c_int32),
                ("btu_local", C.c_int32)]


class Banks(C.Structure):
    _fields_ = [("d_feats", C.c_void_p * 2), ("d_probs", C.c_void_p * 2), ("Q", C.c_int32),
                ("ptr", C.c_int32 * 2)]


class StepIO(C.Structure):
    _fields_ = [
        ("d_xpl", C.c_void_p), ("d_xl", C.c_void_p), ("d_labels", C.c_void_p),
        ("d_xpu", C.c_void_p), ("d_xu", C.c_void_p),
        ("noise8", C.POINTER(C.c_void_p)), ("d_dropmask", C.c_void_p),
        ("d_params", C.c_void_p), ("d_m", C.c_void_p), ("d_v", C.c_void_p), ("d_grads", C.c_void_p),
        ("d_packed", C.c_void_p),
        ("banks", Banks),
        ("d_scalars", C.c_void_p), ("d_logits", C.c_void_p), ("d_feat", C.c_void_p),
        ("d_workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("bt", C.c_int32), ("btu", C.c_int32), ("smooth", C.c_int32), ("adap_mask", C.c_float),
        ("adam_t", C.c_int64), ("seed", C.c_uint64), ("step", C.c_uint64),
        ("apply_update", C.c_int32), ("reserved", C.c_int32),
        ("d_lab_idx", C.c_void_p), ("d_unl_idx", C.c_void_p), ("d_dyn_table", C.c_void_p), ("d_dyn_cursor", C.c_void_p),
    ]


class DistIO(C.Structure):
    """cmlpl_dist_io: what the seven captured stages of the sharded step read and write"""
    _fields_ = [
        ("batch", Batch), ("shard", Shard), ("gathered", Gathered), ("banks", Banks),
        ("d_params", C.c_void_p), ("d_m", C.c_void_p), ("d_v", C.c_void_p), ("d_packed", C.c_void_p),
        ("d_grads", C.c_void_p), ("grad_stride", C.c_int64),
        ("d_logits_l", C.c_void_p), ("d_feat_l", C.c_void_p), ("d_labels_f", C.c_void_p),
        ("d_dlogits", C.c_void_p), ("d_dfeat", C.c_void_p),
        ("d_probs_l", C.c_void_p), ("d_probs_g", C.c_void_p), ("probs_shard_rows", C.c_int32), ("reserved", C.c_int32),
        ("d_scalars", C.c_void_p), ("d_dfeat_w_partial", C.c_void_p),
        ("d_workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("d_loss_workspace", C.c_void_p), ("loss_workspace_bytes", C.c_size_t),
        ("seed", C.c_uint64), ("d_dyn_table", C.c_void_p), ("d_dyn_cursor", C.c_void_p),
    ]


class Collectives(C.Structure):
    """cmlpl_collectives: a rank's communicator as three C functions + the side stream / events of the asynchronous
    exchanges (filled by cmlpl_rccl_bind)"""
    _fields_ = [("ctx", C.c_void_p), ("all_gather", C.c_void_p), ("reduce_scatter", C.c_void_p), ("all_reduce", C.c_void_p),
                ("side_stream", C.c_void_p), ("events", C.c_void_p * 4)]


class DistStepArgs(C.Structure):
    """cmlpl_dist_step_args: what changes from step to step, by value"""
    _fields_ = [("step", C.c_uint64), ("adam_t", C.c_int64), ("d_dropmask", C.c_void_p), ("smooth", C.c_int32),
                ("adap_mask", C.c_float), ("apply_update", C.c_int32), ("scalars_row", C.c_int32)]


STAGE_IDS = {"spectral": 0, "spatial": 1, "phase1": 2, "phase2": 3, "backward_data": 4, "backward_weights": 5, "update": 6}

_lib = None


def load(path: str = LIB_PATH):
    """Load the HIP library once; fail loudly if it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise CmlplLibraryError(
            f"{path} not found: build it with `python -m cmlpl_amd.build_ext` (hipcc, gfx950). "
            "cmlpl_amd has no CPU fallback.")
    # torch FIRST: it brings its own libamdhip64, and the library must bind to THAT runtime (the one that owns the
    # tensors' memory and streams).  Loaded before torch, the library pulls in /opt/rocm's copy, torch then loads its
    # own, and the process holds two HIP runtimes: every launch fails with hipErrorNoDevice (100).
    import torch  # noqa: F401
    try:
        lib = C.CDLL(path)
    except OSError as e:  # e.g. libamdhip64 missing
        raise CmlplLibraryError(f"cannot load {path}: {e}") from e
    missing = [s for s in EXPORTS if not hasattr(lib, s)]
    default_lib = os.path.abspath(path) == os.path.join(HERE, "libcmlpl_hip.so")
    if not default_lib and missing == ["cmlpl_source_hash"]:
        missing = []            # an older build named through CMLPL_LIB (A/B runs against a previous round's binary)
    if missing:
        raise CmlplLibraryError(f"{path} does not export {missing}")
    vp, i32, i64, u64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_size_t
    SP, HP = C.POINTER(Shape), C.POINTER(HParams)
    lib.cmlpl_abi_version.restype = i32
    if hasattr(lib, "cmlpl_source_hash"):
        lib.cmlpl_source_hash.restype = C.c_char_p
        lib.cmlpl_source_hash.argtypes = []
    lib.cmlpl_layout.argtypes = [SP, C.POINTER(Layout)]
    lib.cmlpl_workspace_bytes.argtypes = [SP, i32, i32, i32]
    lib.cmlpl_workspace_bytes.restype = sz
    lib.cmlpl_pack_weights.argtypes = [SP, i32, vp, i64, vp, vp]
    SH = C.POINTER(Shard)
    lib.cmlpl_augment.argtypes = [SP, i32, i32, i32, vp, vp, vp, vp, C.POINTER(vp), f32, u64, u64, SH, vp, vp, vp, vp]
    lib.cmlpl_basenet2_fwd.argtypes = [SP, i32, i32, vp, i64, vp, vp, vp, vp, vp, f32, i32, u64, u64, SH, vp, vp, vp,
                                       sz, vp]
    lib.cmlpl_basenet2_bwd.argtypes = [SP, i32, i32, vp, i64, vp, vp, vp, vp, f32, i32, vp, vp, vp, i64, vp, sz, vp]
    BP = C.POINTER(Batch)
    lib.cmlpl_forward.argtypes = [SP, HP, BP, SH, vp, vp, vp, i32, u64, u64, vp, vp, vp, vp, sz, vp]
    GP = C.POINTER(Gathered)
    lib.cmlpl_loss_phase1_g.argtypes = [SP, SH, GP, C.POINTER(Banks), i32, f32, HP, vp, vp, vp, vp, sz, vp]
    lib.cmlpl_loss_phase2_g.argtypes = [SP, SH, GP, C.POINTER(Banks), i32, f32, HP, vp, i32, vp, vp, vp, vp, sz, vp]
    lib.cmlpl_backward.argtypes = [SP, HP, BP, SH, vp, vp, vp, i32, u64, u64, vp, vp, vp, i64, vp, sz, vp]
    lib.cmlpl_loss_fwd_bwd.argtypes = [SP, i32, i32, vp, vp, vp, C.POINTER(Banks), i32, f32, HP, vp, vp, vp, vp,
                                       vp, sz, vp]
    lib.cmlpl_loss_phase1.argtypes = [SP, SH, vp, vp, vp, C.POINTER(Banks), i32, f32, HP, vp, vp, vp, vp, sz, vp]
    lib.cmlpl_loss_phase2.argtypes = [SP, SH, vp, vp, vp, C.POINTER(Banks), i32, f32, HP, vp, i32, vp, vp, vp, vp, sz,
                                      vp]
    lib.cmlpl_loss_workspace_bytes.argtypes = [SP, SH, i32]
    lib.cmlpl_loss_workspace_bytes.restype = sz
    lib.cmlpl_dist_unpack.argtypes = [SP, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.cmlpl_forward_spectral.argtypes = [SP, HP, BP, SH, vp, u64, u64, vp, vp, vp, sz, vp]
    lib.cmlpl_forward_spatial.argtypes = [SP, HP, BP, SH, vp, vp, vp, i32, u64, u64, vp, vp, sz, vp]
    lib.cmlpl_backward_data.argtypes = [SP, HP, BP, SH, vp, vp, vp, i32, u64, u64, vp, vp, sz, vp]
    lib.cmlpl_backward_weights.argtypes = [SP, HP, BP, SH, vp, vp, vp, i32, u64, u64, vp, vp, vp, i64, vp, sz, vp]
    lib.cmlpl_extract_patches.argtypes = [vp, i32, i32, i32, i32, vp, i32, vp, vp]
    lib.cmlpl_infer_workspace_bytes.argtypes = [vp, i32]
    lib.cmlpl_infer_workspace_bytes.restype = C.c_size_t
    lib.cmlpl_infer_cube.argtypes = [vp, vp, vp, vp, i32, i32, vp, C.c_int64, i32, vp, vp, vp, C.c_size_t, vp]
    lib.cmlpl_ntxent_workspace_bytes.argtypes = [i32, i32]
    lib.cmlpl_ntxent_workspace_bytes.restype = sz
    lib.cmlpl_ntxent_fwd_bwd.argtypes = [vp, vp, i32, i32, f32, vp, vp, vp, vp, sz, vp]
    lib.cmlpl_unsup_workspace_bytes.argtypes = [i32]
    lib.cmlpl_unsup_workspace_bytes.restype = sz
    lib.cmlpl_unsup_loss.argtypes = [vp, vp, vp, i32, i32, C.c_double, vp, vp, vp, sz, vp]
    lib.cmlpl_memobank_select.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]
    lib.cmlpl_memobank_proto.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp]
    lib.cmlpl_memobank_enqueue.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp]
    lib.cmlpl_memobank_push.argtypes = [vp, i32, i32, vp, vp, i32, vp]
    lib.cmlpl_memobank_infonce.argtypes = [vp, i32, i32, vp, i32, vp, vp, i64, vp, i32, i32, i32, vp, i32, i32, f32, f32,
                                           vp, vp, vp, vp]
    lib.cmlpl_memobank_sum.argtypes = [vp, i32, vp, vp]
    lib.cmlpl_memobank_loss.argtypes = [C.POINTER(MemobankCall), vp]
    lib.cmlpl_adam_step.argtypes = [SP, i32, vp, i64, vp, i64, vp, vp, i64, HP, vp, vp]
    lib.cmlpl_train_step.argtypes = [SP, HP, C.POINTER(StepIO), vp]
    lib.cmlpl_dyn_adam.argtypes = [HP, i64, C.POINTER(f32), C.POINTER(f32)]
    lib.cmlpl_step_graph_create.argtypes = [SP, HP, C.POINTER(StepIO), vp, C.POINTER(vp)]
    lib.cmlpl_dist_stage_graph_create.argtypes = [SP, HP, C.POINTER(DistIO), i32, vp, C.POINTER(vp)]
    lib.cmlpl_dist_step.argtypes = [SP, HP, C.POINTER(DistIO), C.POINTER(DistStepArgs), C.POINTER(Collectives), vp]
    lib.cmlpl_rccl_bind.argtypes = [C.c_char_p, vp, C.POINTER(Collectives)]
    lib.cmlpl_rccl_unbind.argtypes = [C.POINTER(Collectives)]
    lib.cmlpl_debug_two_piece.argtypes = [SP, i32, i32]
    lib.cmlpl_step_graph_launch.argtypes = [vp, vp]
    lib.cmlpl_step_graph_destroy.argtypes = [vp]
    lib.cmlpl_debug_region.argtypes = [SP, i32, i32, C.c_char_p, C.POINTER(sz), C.POINTER(sz)]
    lib.cmlpl_timing_begin.argtypes = [C.c_uint32, i32]
    lib.cmlpl_timing_end.argtypes = [C.POINTER(C.c_double), C.POINTER(i64)]
    for s in EXPORTS[1:]:
        if hasattr(lib, s) and s not in ("cmlpl_workspace_bytes", "cmlpl_loss_workspace_bytes", "cmlpl_ntxent_workspace_bytes",
                     "cmlpl_unsup_workspace_bytes", "cmlpl_source_hash", "cmlpl_infer_workspace_bytes"):
            getattr(lib, s).restype = i32
    if lib.cmlpl_abi_version() != ABI_VERSION:
        raise CmlplLibraryError(f"ABI version mismatch: library {lib.cmlpl_abi_version()}, binding {ABI_VERSION}")
    # A binary that was not built from the sources next to it is refused (a stale .so pushed to a GPU box would
    # otherwise be measured as if it were the code).  Another build named on purpose through CMLPL_LIB (A/B runs,
    # ablation builds) is taken as it is.
    if default_lib and not os.environ.get("CMLPL_ALLOW_STALE"):
        from .build_ext import source_hash
        try:
            now = source_hash()
        except OSError as e:        # a deployment that ships the library without csrc/ and include/: nothing to compare
            import warnings
            warnings.warn(f"cmlpl_amd: kernel sources not found next to {path} ({e}); stale-binary check skipped")
            now = None
        built = lib.cmlpl_source_hash().decode()
        if now is not None and built != now:
            raise CmlplLibraryError(f"{path} is stale: built from sources {built}, the sources here are {now}; "
                                    "rebuild with `python -m cmlpl_amd.build_ext` (or set CMLPL_ALLOW_STALE=1 to "
                                    "load it as it is)")
    _lib = lib
    return lib


def check(fn: str, rc: int):
    if rc != 0:
        raise CmlplError(fn, rc)


def layout(shape: Shape) -> Layout:
    out = Layout()
    check("cmlpl_layout", load().cmlpl_layout(C.byref(shape), C.byref(out)))
    return out
