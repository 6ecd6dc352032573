"""ctypes binding of libactmi.so (C ABI declared in include/actmi.h).

There is NO CPU fallback: if the HIP library is missing or a call fails, a RuntimeError is raised.
torch is imported before the library so that libactmi's ``libamdhip64.so.7`` dependency resolves (by SONAME)
to the one HIP runtime already loaded by PyTorch-ROCm; device pointers and streams are then shared.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be loaded first, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# ACTMI_LIB: another build of the library (A/B runs of two builds on one GPU box: tools/ab_bench.sh); default: the in-tree one
LIB_PATH = os.environ.get("ACTMI_LIB") or os.path.join(_HERE, "libactmi.so")

IMG_U8_NHWC = 0
IMG_F32_NCHW = 1


class ActmiConfig(C.Structure):
    # struct_size first: the ABI guard of include/actmi.h (actmi_create rejects a binding whose struct differs)
    _fields_ = [("struct_size", C.c_uint32)] + [(n, C.c_int32) for n in (
        "num_cams", "image_h", "image_w", "base_width", "hidden_dim", "nheads", "dim_feedforward", "enc_layers",
        "dec_layers", "num_queries", "state_dim", "action_dim", "latent_dim", "has_cvae_encoder", "max_batch",
        "enable_training")] + [("kl_weight", C.c_float)] + [(n, C.c_int32) for n in ("vq", "vq_class", "vq_dim")]


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64), ("mode", C.c_int32),
        ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32),
        ("stride", C.c_int32), ("pad", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("img_stride", C.c_int64),
        ("A_add", C.c_void_p), ("ld_add", C.c_int64), ("add_mod", C.c_int32), ("add_ncols", C.c_int32),
        ("Bw", C.c_void_p), ("ldb", C.c_int64),
        ("scale", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p), ("ldres", C.c_int64),
        ("res_mod", C.c_int32), ("relu", C.c_int32),
        ("C", C.c_void_p), ("ldc", C.c_int64), ("rowmap", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("groups", C.c_int32),
        ("gA", C.c_int64), ("gB", C.c_int64), ("gSB", C.c_int64), ("gC", C.c_int64), ("gRes", C.c_int64),
        ("ta", C.c_int32), ("tb", C.c_int32), ("a_rowmap", C.c_void_p),
        ("B_add", C.c_void_p), ("ld_badd", C.c_int64), ("badd_mod", C.c_int32), ("splitk", C.c_int32),
        ("mask", C.c_void_p), ("ldmask", C.c_int64), ("C2", C.c_void_p), ("scale2", C.c_void_p),
        ("alpha", C.c_float), ("groups_inner", C.c_int32),
        ("gA2", C.c_int64), ("gB2", C.c_int64), ("gC2", C.c_int64), ("gRes2", C.c_int64),
        ("gMask", C.c_int64), ("gC2out", C.c_int64),
        ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("stamps", C.c_void_p), ("prec", C.c_int32), ("b_split", C.c_int32), ("b_scale", C.c_float), ("a_scale", C.c_float), ("a_scale_dev", C.c_void_p), ("b_scale_dev", C.c_void_p),
        ("split_stride", C.c_int64), ("amax_out", C.c_void_p),
        ("epi", C.c_int32), ("epi_scale", C.c_float), ("epi_row", C.c_void_p), ("gRow", C.c_int64), ("gRow2", C.c_int64),
        ("epi_colkill", C.c_void_p), ("gColkill", C.c_int64), ("tile_hint", C.c_int32),
        ("finite_flag", C.c_void_p), ("finite_bit", C.c_uint32),
        ("Ax", C.c_void_p), ("kx_begin", C.c_int32), ("Hx", C.c_int32), ("Wx", C.c_int32), ("Cx", C.c_int32),
        ("stride_x", C.c_int32), ("gAx", C.c_int64),
        ("A_alt", C.c_void_p), ("alt_ncols", C.c_int32), ("k_tap_inner", C.c_int32),
    ]


class AttnBwdDesc(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p), ("d_o", C.c_void_p), ("lse", C.c_void_p),
                ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p)] + \
               [(n, C.c_int64) for n in ("q_bs", "q_rs", "k_bs", "k_rs", "v_bs", "v_rs", "dq_bs", "dq_rs", "dk_bs", "dk_rs", "dv_bs", "dv_rs")] + \
               [("kpm", C.c_void_p), ("kpm_bs", C.c_int64)] + [(n, C.c_int32) for n in ("B", "H", "Nq", "Nk", "HD")] + \
               [("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("delta_ws", C.c_void_p), ("do_scale", C.c_void_p), ("amax_out", C.c_void_p)]


class AttnDesc(C.Structure):
    _fields_ = [
        ("Q", C.c_void_p), ("q_bs", C.c_int64), ("q_rs", C.c_int64),
        ("K", C.c_void_p), ("k_bs", C.c_int64), ("k_rs", C.c_int64),
        ("V", C.c_void_p), ("v_bs", C.c_int64), ("v_rs", C.c_int64),
        ("O", C.c_void_p), ("o_bs", C.c_int64), ("o_rs", C.c_int64),
        ("kpm", C.c_void_p), ("kpm_bs", C.c_int64), ("lse", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("Nq", C.c_int32), ("Nk", C.c_int32), ("HD", C.c_int32),
        ("scale", C.c_float), ("ws", C.c_void_p), ("ws_floats", C.c_int64),
        ("drop_p", C.c_float), ("drop_seed", C.c_uint64), ("prec", C.c_int32), ("causal", C.c_int32),
    ]


_lib = None


def load():
    """Load libactmi.so; raise loudly when it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C act-plus-plus_amd/csrc). There is no CPU fallback for the ACT path.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
    sigs = {
        "actmi_version": ([], i32),
        "actmi_create": ([C.POINTER(ActmiConfig), C.POINTER(vp)], i32),
        "actmi_destroy": ([vp], i32),
        "actmi_last_error": ([vp], C.c_char_p),
        "actmi_num_params": ([vp], i32),
        "actmi_param_info": ([vp, i32, C.POINTER(C.c_char_p), C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)], i32),
        "actmi_set_param": ([vp, C.c_char_p, vp, C.POINTER(i64), i32, i32], i32),
        "actmi_get_param": ([vp, C.c_char_p, vp, i64, i32], i32),
        "actmi_param_ptr": ([vp, C.c_char_p, C.POINTER(vp), C.POINTER(i64)], i32),
        "actmi_finalize": ([vp, vp], i32),
        "actmi_set_forward_phase": ([vp, i32], i32),
        "actmi_forward_infer_vq": ([vp, vp, vp, i32, i32, vp, vp, vp], i32),
        "actmi_forward_infer": ([vp, vp, vp, i32, i32, vp, vp], i32),
        "actmi_forward_train": ([vp, vp, vp, i32, vp, vp, vp, C.c_uint64, f32, i32, vp, vp, vp, vp, vp], i32),
        "actmi_backward": ([vp, f32, vp], i32),
        "actmi_zero_grad": ([vp, vp], i32),
        "actmi_adamw_step": ([vp, f32, f32, f32, f32, f32, f32, i64, vp], i32),
        "actmi_adamw_step_range": ([vp, f32, f32, f32, f32, f32, f32, i64, i64, i64, vp], i32),
        "actmi_refresh_weights": ([vp, vp], i32),
        "actmi_param_arena": ([vp, C.POINTER(vp), C.POINTER(i64)], i32),
        "actmi_grad_ptr": ([vp, C.c_char_p, C.POINTER(vp), C.POINTER(i64)], i32),
        "actmi_grad_arena": ([vp, C.POINTER(vp), C.POINTER(i64)], i32),
        "actmi_grad_phase_range": ([vp, i32, C.POINTER(i64), C.POINTER(i64)], i32),
        "actmi_wait_grad_phase": ([vp, i32, vp], i32),
        "actmi_ensemble_step": ([vp, vp, vp, f64, vp, vp, i32, i32, i32, vp], i32),
        "actmi_op_gemm": ([C.POINTER(GemmDesc), vp], i32),
        "actmi_op_split16": ([vp, vp, C.c_int64, C.c_float, vp], i32),
        "actmi_op_permute_conv_k": ([vp, vp, C.c_int64, i32, i32, i32, vp], i32),
        "actmi_op_sample_onehot": ([vp, i32, i32, C.c_float, C.c_uint64, vp, vp, vp], i32),
        "actmi_op_pow2_scale": ([vp, C.c_int64, i32, i32, vp, vp], i32),
        "actmi_op_splitk_combine": ([vp, i32, C.c_int64, C.c_int64, i32, i32, vp, vp, vp, C.c_int64, i32, vp, C.c_int64, vp], i32),
        "actmi_op_attention": ([C.POINTER(AttnDesc), vp], i32),
        "actmi_op_attention_bwd": ([C.POINTER(AttnBwdDesc), vp], i32),
        "actmi_op_layernorm": ([vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, f32, vp], i32),
        "actmi_op_maxpool3x3s2": ([vp, vp, i32, i32, i32, i32, vp], i32),
        "actmi_op_conv1": ([vp, i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
        "actmi_op_conv1_workspace_floats": ([i32, i32], C.c_int64),
        "actmi_op_conv1_prepare": ([vp, vp, i32, i32, i32, vp], i32),
        "actmi_op_conv1_prepared": ([vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp], i32),
        "actmi_op_conv3x3_c64": ([vp, vp, C.c_float, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp], i32),
        "actmi_op_wgrad7x7s2": ([vp, vp, vp, vp, C.c_int64, vp, i32, i32, i32, i32, vp], i32),
        "actmi_op_wgrad3x3_c64": ([vp, vp, vp, vp, C.c_int64, vp, i32, i32, i32, i32, vp], i32),
        "actmi_op_groupnorm": ([vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, i32, vp, C.c_int64, vp], i32),
        "actmi_op_spatial_softmax": ([vp, vp, i32, i32, i32, i32, f32, vp], i32),
        "actmi_op_unfold1d": ([vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp], i32),
        "actmi_op_ddim_step": ([vp, vp, C.c_int64, f32, f32, f32, f32, i32, vp], i32),
        "actmi_op_mish": ([vp, vp, C.c_int64, vp], i32),
        "actmi_op_gelu": ([vp, vp, C.c_int64, vp], i32),
        "actmi_op_gelu_bwd": ([vp, vp, vp, C.c_int64, vp], i32),
        "actmi_op_dropout": ([vp, vp, C.c_int64, C.c_float, C.c_uint64, vp], i32),
        "actmi_op_small_attention": ([vp, vp, i32, i32, i32, i32, i32, C.c_float, C.c_uint64, vp], i32),
        "actmi_op_small_attention_bwd": ([vp, vp, vp, i32, i32, i32, i32, i32, C.c_float, C.c_uint64, vp], i32),
        "actmi_op_soft_ce_dim1": ([vp, vp, i32, i32, i32, vp, vp, vp, vp], i32),
        "actmi_op_argmax_l1": ([vp, vp, i32, i32, vp, vp, vp], i32),
        "actmi_op_layernorm_bwd": ([vp, vp, vp, vp, vp, vp, vp, i32, i32, C.c_float, vp, C.c_int64, vp], i32),
        "actmi_op_colsum": ([vp, C.c_int64, vp, i32, i32, vp, C.c_int64, vp], i32),
        "actmi_op_sum_batch": ([vp, C.c_int64, C.c_int64, vp, i32, i32, i32, i32, vp], i32),
        "actmi_op_adamw": ([vp, vp, vp, vp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, vp], i32),
        "actmi_op_u8_to_nhwc4": ([vp, vp, i32, i32, i32, i32, vp], i32),
        "actmi_op_last_error": ([], C.c_char_p),
        "actmi_debug_tensor": ([vp, C.c_char_p, C.POINTER(vp), C.POINTER(i64)], i32),
        "actmi_debug_stop_after": ([vp, C.c_char_p], i32),
        "actmi_set_gemm_prec": ([vp, i32], i32),
        "actmi_set_train_prec": ([vp, i32], i32),
        "actmi_get_flags": ([vp, C.POINTER(C.c_uint32), i32, vp], i32),
        "actmi_flags_ptr": ([vp, C.POINTER(vp)], i32),
        "actmi_profile_enable": ([i32], i32),
        "actmi_profile_reset": ([], i32),
        "actmi_profile_report": ([C.c_char_p, i32], i32),
    }
    for name, (args, res) in sigs.items():
        fn = getattr(lib, name)          # AttributeError here = header and library disagree
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def declared_symbols(header_path):
    """Names of every function declared in include/actmi.h (used by the no-GPU export test)."""
    import re
    txt = open(header_path).read()
    return sorted(set(re.findall(r"\b(actmi_[a-z0-9_]+)\s*\(", txt)))


def current_stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, handle=None, what=""):
    if rc != 0:
        lib = load()
        msg = lib.actmi_last_error(handle) if handle is not None else lib.actmi_op_last_error()
        raise RuntimeError(f"libactmi {what} failed (code {rc}): {msg.decode() if msg else ''}")


def profile_enable(on: bool):
    lib = load()
    lib.actmi_profile_reset() if on else None
    lib.actmi_profile_enable(1 if on else 0)


def profile_report():
    """-> list of {name,count,ms,flops,bytes}; synchronises the recorded events."""
    import json
    lib = load()
    buf = C.create_string_buffer(1 << 18)
    rc = lib.actmi_profile_report(buf, len(buf))
    if rc != 0:
        raise RuntimeError(f"actmi_profile_report failed ({rc})")
    return json.loads(buf.value.decode())
