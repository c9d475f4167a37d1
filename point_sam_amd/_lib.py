"""ctypes binding of csrc/libpointsam_hip.so (C ABI declared in include/pointsam_hip.h).

The product path has NO fallback: if the library has not been built (``python -m point_sam_amd.build``)
or a kernel launch fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PSAM_HIP_LIB: A/B tuning hook (another build of the SAME C ABI); there is still no non-HIP fallback
LIB_PATH = os.environ.get("PSAM_HIP_LIB") or os.path.join(_HERE, "csrc", "libpointsam_hip.so")

i32, i64, f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
ptr, size_t = ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/pointsam_hip.h one to one
SIGNATURES = {
    "psam_version": (i32, []),
    "psam_last_error_string": (ctypes.c_char_p, []),
    "psam_fps_workspace_bytes": (size_t, [i32, i32, i32]),
    "psam_fps_set_cooperative": (None, [i32]),
    "psam_fps_set_pruning": (None, [i32]),
    "psam_fps": (i32, [ptr, i32, i32, i32, ptr, ptr, ptr, size_t, ptr]),
    "psam_knn": (i32, [ptr, ptr, i32, i32, i32, i32, ptr, ptr]),
    "psam_knn_force_band": (None, [i32]),
    "psam_three_nn": (i32, [ptr, ptr, i32, i32, i32, f32, ptr, ptr, ptr]),
    "psam_group_gather": (i32, [ptr, ptr, ptr, ptr, i32, i32, i32, i32, i32, i32, ptr, ptr]),
    "psam_group_gather_r": (i32, [ptr, ptr, ptr, ptr, i32, i32, i32, i32, i32, i32, f32, ptr, ptr]),
    "psam_group_gather_ld": (i32, [ptr, ptr, ptr, ptr, i32, i32, i32, i32, i32, i32, f32, ptr, i64, ptr]),
    "psam_patch_l1": (i32, [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, f32, i32, i32, i32, i32, i32, i32, ptr, ptr]),
    "psam_patch_l1_r": (i32, [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, f32, i32, i32, i32, i32, i32, i32, f32, ptr, ptr]),
    "psam_patch_l1_ex": (i32, [ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, f32, i32, i32, i32, i32, i32, i32, f32, ptr, ptr, ptr]),
    "psam_error_regions": (i32, [ptr, ptr, ptr, ptr, i64, ptr]),
    "psam_border_farthest_workspace_bytes": (size_t, [i32, i32]),
    "psam_border_farthest": (i32, [ptr, ptr, i32, i32, i32, ptr, ptr, ptr, size_t, ptr]),
    "psam_group_max": (i32, [ptr, i64, ptr, i64, i64, i32, i32, ptr]),
    "psam_gemm_f32": (i32, [ptr, i64, i64, i64, ptr, i64, i64, i64, ptr, i64, i64, i64, ptr, ptr, i64, i64, i64, ptr, i64, i32,
                            i32, i32, i32, i32, i32, f32, i32, ptr]),
    "psam_gemm_bf16x6": (i32, [ptr, i64, i64, i64, ptr, i64, i64, i64, ptr, i64, i64, i64, ptr, ptr, i64, i64, i64, ptr, i64, i32,
                               i32, i32, i32, i32, i32, f32, i32, ptr]),
    "psam_linear": (i32, [ptr, i64, ptr, i64, ptr, ptr, i64, ptr, i64, i32, i32, i32, i32, ptr]),
    "psam_gemm_force_config": (None, [i32]),
    "psam_gemm_bf16x6_force_config": (None, [i32]),
    "psam_row_scale_f16": (i32, [ptr, i64, i32, i32, ptr, ptr]),
    "psam_pack_rows_f16x2_g8": (i32, [ptr, i64, ptr, i32, i32, ptr, i64, ptr]),
    "psam_gemm_f16x3p": (i32, [ptr, i64, ptr, ptr, i64, ptr, ptr, i64, ptr, ptr, i64, ptr, i64, i32, i32, i32, i32, f32, i32, ptr]),
    "psam_gemm_f16x3p_force_config": (None, [i32]),
    "psam_gemm_f16x3p_force_epilogue": (None, [i32]),
    "psam_gemm_f16x3p_force_splitk_fixup": (None, [i32]),
    "psam_gemm_f16x3p_force_continuous": (None, [i32]),
    "psam_attention_f16x3_force_keysplit": (None, [i32]),
    "psam_twoway_decoder_force_fork": (None, [i32]),
    "psam_attention_packed_force_variant": (None, [i32]),
    "psam_gemm_f16x3p_fused_row_ln": (i32, [i32]),
    "psam_gemm_f16x3p_stat_segs": (i32, [i32]),
    "psam_gemm_f16x3p_splitk": (i32, [i32, i32, i32, i32]),
    "psam_nn_group_feats": (i32, [ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32, i32, i32, ptr, i64, ptr]),
    "psam_scatter_amax": (i32, [ptr, i64, ptr, i64, i32, i64, i64, i32, ptr, i64, i32, ptr]),
    "psam_eva_block_prepared_bytes": (size_t, [i32, i32]),
    "psam_eva_block_prepare": (i32, [ptr, ptr, ptr, size_t, ptr]),
    "psam_eva_block_ws_bytes": (size_t, [i64, i32, i32]),
    "psam_eva_block": (i32, [ptr, ptr, ptr, i32, i32, ptr, size_t, ptr]),
    "psam_linear_skinny_multi": (i32, [ptr, i64, i64, i64, i32, i32, ptr]),
    "psam_linear_skinny_ln": (i32, [ptr, i64, ptr, i64, ptr, ptr, i64, ptr, ptr, f32, ptr, ptr, i64, i32, i32, i32, ptr, ptr]),
    "psam_linear_rows_multi": (i32, [ptr, i64, i64, i32, i32, i64, i64, i32, ptr]),
    "psam_linear_ln256": (i32, [ptr, i64, ptr, i64, ptr, ptr, i64, ptr, ptr, f32, ptr, i64, i64, i32, i32, ptr]),
    "psam_twoway_decoder_force_fast": (None, [i32]),
    "psam_scale_pack_rows_g8_add_dual": (i32, [ptr, i64, ptr, i64, i32, i32, i32, i32, ptr, ptr, ptr, ptr, i64, ptr]),
    "psam_mlp3_pair": (i32, [ptr, ptr, i32, ptr]),
    "psam_attention_small_force_split": (None, [i32]),
    "psam_linear_skinny_ln_tmp_floats": (ctypes.c_size_t, [i32, i32]),
    "psam_scale_pack_rows_g8_add": (i32, [ptr, i64, ptr, i64, i32, i32, i32, i32, ptr, i64, ptr, ptr]),
    "psam_eva_gelu_block_prepared_bytes": (size_t, [i32, i32]),
    "psam_eva_gelu_block_prepare": (i32, [ptr, ptr, ptr, size_t, ptr]),
    "psam_eva_gelu_block_ws_bytes": (size_t, [i64, i32, i32]),
    "psam_eva_gelu_block": (i32, [ptr, ptr, ptr, i32, i32, ptr, size_t, ptr, ptr]),
    "psam_patch_encoder_prepared_bytes": (size_t, [i32, i32, i32]),
    "psam_patch_encoder_prepare": (i32, [ptr, ptr, ptr, size_t, ptr]),
    "psam_patch_encoder_ws_bytes": (size_t, [i64, i64, i32, i32]),
    "psam_patch_encoder": (i32, [ptr, ptr, ptr, ptr, ptr, ptr, ptr, i32, i32, i32, i32, i32, i32, f32, ptr, ptr, size_t, ptr]),
    "psam_upscale_masks_prepared_bytes": (size_t, [i32]),
    "psam_upscale_masks_prepare": (i32, [ptr, ptr, ptr, size_t, ptr]),
    "psam_upscale_masks_ws_bytes": (size_t, [i64, i32, i32, i32, i32]),
    "psam_upscale_masks": (i32, [ptr, ptr, ptr, ptr, ptr, ptr, i32, i64, i32, i32, i32, ptr, ptr, size_t, ptr]),
    "psam_twoway_decoder_prepared_bytes": (size_t, [i32, i32, i32, i32]),
    "psam_twoway_decoder_prepare": (i32, [ptr, ptr, ptr, size_t, ptr]),
    "psam_twoway_decoder_ws_bytes": (size_t, [i64, i32, i32, i32, i32]),
    "psam_twoway_decoder": (i32, [ptr, ptr, ptr, ptr, ptr, i32, i64, i32, i32, ptr, ptr, size_t, ptr, ptr]),
    "psam_twoway_tokens_ws_floats": (i64, [i32]),
    "psam_twoway_tokens": (i32, [ptr, ptr]),
    "psam_gemm_f16x3p_ex": (i32, [ptr, i64, ptr, ptr, i64, ptr, ptr, i64, ptr, ptr, i64, ptr, i64, i32, i32, i32, i32, f32, i32, ptr, ptr]),
    "psam_ln_stats_finalize": (i32, [ptr, i32, i32, i32, f32, ptr, ptr, ptr]),
    "psam_scale_pack_rows_g8": (i32, [ptr, i64, i32, i32, ptr, i64, ptr, ptr]),
    "psam_layernorm": (i32, [ptr, i64, ptr, i64, ptr, ptr, ptr, i64, i64, i32, f32, i32, ptr]),
    "psam_layernorm_rs": (i32, [ptr, i64, ptr, i64, ptr, ptr, ptr, i64, i64, i32, f32, i32, ptr, ptr]),
    "psam_layernorm_ex": (i32, [ptr, i64, ptr, i64, ptr, ptr, ptr, i64, i64, i32, f32, i32, ptr, i32, ptr]),
    "psam_layernorm_ex2": (i32, [ptr, i64, ptr, i64, ptr, ptr, ptr, i64, i64, i32, f32, i32, ptr, i32, ptr, f32, f32, f32, ptr]),
    "psam_swiglu_ln": (i32, [ptr, i64, i32, ptr, ptr, ptr, i64, i64, i32, f32, ptr]),
    "psam_attention_f32": (i32, [ptr, i64, i64, ptr, i64, i64, ptr, i64, i64, ptr, i64, i64, i32, i32, i32, i32, i32, f32, ptr]),
    "psam_attention_f16x3": (i32, [ptr, i64, i64, ptr, i64, i64, ptr, i64, i64, ptr, i64, i64, i32, i32, i32, i32, i32, f32, ptr]),
    "psam_attention_f16x3_ex": (i32, [ptr, i64, i64, ptr, i64, i64, ptr, i64, i64, ptr, i64, i64, i32, i32, i32, i32, i32, f32, ptr, f32, f32, ptr, ptr]),
    "psam_attention_f16x3_ex2": (i32, [ptr, i64, i64, ptr, i64, i64, ptr, i64, i64, ptr, i64, i64, i32, i32, i32, i32, i32, f32, ptr, f32, f32, ptr, i32, ptr, size_t, ptr, ptr]),
    "psam_attention_f16x3_keysplit_ws_bytes": (size_t, [i32, i32, i32, i32, i32, i32]),
    "psam_attention_packed": (i32, [ptr, i64, ptr, ptr, i64, ptr, i32, i32, i32, i32, f32, f32, ptr]),
    "psam_attention_small": (i32, [ptr, i64, i64, ptr, i64, i64, ptr, i64, i64, ptr, i64, i64, i64, i32, i32, i32, i32, f32, ptr]),
    "psam_gemm_f16x3p_hyper_planes": (i32, [i32, i32]),
    "psam_sum_planes": (i32, [ptr, i32, i64, i64, ptr, ptr]),
    "psam_linear_skinny": (i32, [ptr, i64, ptr, i64, ptr, ptr, i64, ptr, i64, i32, i32, i32, i32, ptr]),
    "psam_mlp3": (i32, [ptr, i64, i64, ptr, ptr, ptr, ptr, ptr, ptr, ptr, i64, i64, i32, i32, i32, i32, i32, ptr]),
    "psam_pos_l1": (i32, [ptr, ptr, ptr, ptr, i64, ptr]),
    "psam_fourier_pe": (i32, [ptr, ptr, i32, ptr, ptr, ptr, ptr, i64, i32, i64, ptr, ptr]),
    "psam_add_bcast": (i32, [ptr, i64, i32, ptr, i64, i64, ptr, i64, i64, i64, i32, ptr]),
    "psam_interp3": (i32, [ptr, ptr, ptr, ptr, i32, i64, i32, i32, i32, ptr]),
    "psam_interp3_ex": (i32, [ptr, ptr, ptr, ptr, i32, i64, i32, i32, i32, ptr, ptr, ptr, f32, i32, ptr]),
}



class GemmFuse(ctypes.Structure):
    """psam_gemm_fuse_t (include/pointsam_hip.h)."""
    _fields_ = [("out_scale", ptr), ("out_k1", f32), ("out_k2", f32), ("pack_out", i32), ("stats", ptr), ("stat_cols", i32),
                ("ln_mean", ptr), ("ln_rstd", ptr), ("ln_c", ptr), ("gmax_out", ptr), ("gmax_ld", i64), ("gmax_k", i32), ("no_store", i32),
                ("row_ln_g", ptr), ("row_ln_b", ptr), ("row_ln_eps", f32), ("hyper", ptr), ("masks", ptr), ("hyper_c", i32), ("hyper_rows", i32), ("hyper_pstride", i64),
                ("splitk_ws", ptr), ("splitk_plane", i64), ("splitk", i32), ("out_bound", ptr), ("counters", ptr)]


class EvaBlockWeights(ctypes.Structure):
    """psam_eva_block_weights_t (include/pointsam_hip.h)."""
    _fields_ = ([(n, ptr) for n in ("norm1_w", "norm1_b", "q_w", "q_b", "k_w", "v_w", "v_b", "proj_w", "proj_b", "norm2_w", "norm2_b", "fc1_g_w", "fc1_g_b",
                                    "fc1_x_w", "fc1_x_b", "mlp_norm_w", "mlp_norm_b", "fc2_w", "fc2_b")] + [("dim", i32), ("heads", i32), ("hidden", i32), ("eps", f32)])


class EvaBlockPlan(ctypes.Structure):
    """psam_eva_block_plan_t (include/pointsam_hip.h)."""
    _fields_ = ([(n, i32) for n in ("dim", "heads", "hidden", "hidden_pad")] + [(n, f32) for n in ("eps", "qkv_bound", "v_bound", "u_c2", "u_c1", "u_c0")] +
                [(n, ptr) for n in ("norm1_w", "norm1_b", "norm2_w", "norm2_b", "proj_b")] +
                [(n, i64) for n in ("o_wqkv", "o_sqkv", "o_bqkv", "o_wproj", "o_sproj", "o_w1", "o_s1", "o_b1", "o_w2g", "o_s2g", "o_lnc", "o_lnd")])


class Mlp3Args(ctypes.Structure):
    """psam_mlp3_args_t (include/pointsam_hip.h)."""
    _fields_ = [(n, ptr) for n in ("x", "w1", "b1", "w2", "b2", "w3", "b3", "out")] + [(n, i64) for n in ("ldx", "sx", "ldo", "so")] + [(n, i32) for n in ("M", "din", "dh", "dout")]


class SkinnyJob(ctypes.Structure):
    """psam_skinny_job_t (include/pointsam_hip.h)."""
    _fields_ = [(n, ptr) for n in ("x", "xadd", "W", "bias", "y")] + [("ldy", i64), ("N", i32), ("act", i32)]


class SkinnyJobs(ctypes.Structure):
    """psam_skinny_jobs_t (include/pointsam_hip.h)."""
    _fields_ = [("job", SkinnyJob * 6), ("n", i32)]      # PSAM_SKINNY_MAX_JOBS


class EvaGeluBlockWeights(ctypes.Structure):
    """psam_eva_gelu_block_weights_t (include/pointsam_hip.h)."""
    _fields_ = ([(n, ptr) for n in ("norm1_w", "norm1_b", "qkv_w", "q_bias", "v_bias", "proj_w", "proj_b", "norm2_w", "norm2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")] +
                [("dim", i32), ("heads", i32), ("hidden", i32), ("precision", i32), ("eps", f32)])


class EvaGeluBlockPlan(ctypes.Structure):
    """psam_eva_gelu_block_plan_t (include/pointsam_hip.h)."""
    _fields_ = ([(n, i32) for n in ("dim", "heads", "hidden", "precision")] + [(n, f32) for n in ("eps", "vk1", "vk2", "u_c1", "u_c0")] + [("attn_keysplit", i32)] +
                [(n, ptr) for n in ("norm1_w", "norm1_b", "norm2_w", "norm2_b", "proj_b", "fc1_b", "fc2_b")] +
                [(n, i64) for n in ("o_wqkv", "o_sqkv", "o_bqkv", "o_wproj", "o_sproj", "o_w1", "o_s1", "o_w2", "o_s2")])


class PatchEncoderWeights(ctypes.Structure):
    """psam_patch_encoder_weights_t (include/pointsam_hip.h)."""
    _fields_ = ([(n, ptr) for n in ("c10_w", "c10_b", "c11_w", "c11_b", "c13_w", "c13_b", "c20_w", "c20_b", "c21_w", "c21_b", "c23_w", "c23_b")] +
                [(n, i32) for n in ("cin", "h0", "h1", "cout")] + [("eps", f32)])


class PatchEncoderPlan(ctypes.Structure):
    """psam_patch_encoder_plan_t (include/pointsam_hip.h)."""
    _fields_ = ([(n, i32) for n in ("cin", "h0", "h1", "cout")] + [(n, f32) for n in ("eps", "k1", "k2", "ln21_bound")] +
                [(n, ptr) for n in ("c10_w", "c10_b", "c11_w", "c11_b", "c13_b", "c20_w", "c20_b", "c21_w", "c21_b", "c23_b")] +
                [(n, i64) for n in ("o_w13", "o_s13", "o_w20m", "o_s20m", "o_w20x", "o_s20x", "o_w23", "o_s23")])


class UpscaleWeights(ctypes.Structure):
    """psam_upscale_weights_t (include/pointsam_hip.h)."""
    _fields_ = [(n, ptr) for n in ("u0_w", "u0_b", "u1_w", "u1_b", "u3_w", "u3_b")] + [("dim", i32), ("eps", f32)]


class UpscalePlan(ctypes.Structure):
    """psam_upscale_plan_t (include/pointsam_hip.h)."""
    _fields_ = [("dim", i32), ("eps", f32)] + [(n, ptr) for n in ("u0_w", "u0_b", "u1_w", "u1_b", "u3_b")] + [(n, i64) for n in ("o_w0", "o_s0", "o_w3", "o_s3")]


class AttnWeights(ctypes.Structure):
    """psam_attn_weights_t (include/pointsam_hip.h)."""
    _fields_ = [(n, ptr) for n in ("q_w", "q_b", "k_w", "k_b", "v_w", "v_b", "o_w", "o_b")]


class TwoWayLayerW(ctypes.Structure):
    """psam_twoway_layer_weights_t."""
    _fields_ = [("self_attn", AttnWeights), ("t2i", AttnWeights), ("i2t", AttnWeights)] + [(n, ptr) for n in (
        "n1_w", "n1_b", "n2_w", "n2_b", "n3_w", "n3_b", "n4_w", "n4_b", "m1_w", "m1_b", "m2_w", "m2_b")]


TWOWAY_MAX_DEPTH = 4


class TwoWayWeights(ctypes.Structure):
    """psam_twoway_weights_t."""
    _fields_ = [(n, i32) for n in ("depth", "dim", "heads", "mlp", "downsample")] + [("eps", f32), ("layers", ctypes.POINTER(TwoWayLayerW)),
                                                                                      ("final_attn", AttnWeights), ("nf_w", ptr), ("nf_b", ptr)]


class TwoWayPlan(ctypes.Structure):
    """psam_twoway_plan_t."""
    _fields_ = [("weights", TwoWayWeights), ("layers", TwoWayLayerW * TWOWAY_MAX_DEPTH), ("o_packed", i64 * (14 * TWOWAY_MAX_DEPTH + 4)),
                ("o_scales", i64 * (14 * TWOWAY_MAX_DEPTH + 4)), ("o_cat_packed", i64 * TWOWAY_MAX_DEPTH), ("o_cat_scales", i64 * TWOWAY_MAX_DEPTH),
                ("o_cat_bias", i64 * TWOWAY_MAX_DEPTH)]


class TwoWayTokens(ctypes.Structure):
    """psam_twoway_tokens_t (include/pointsam_hip.h)."""
    _fields_ = ([(n, i32) for n in ("Z", "T", "G", "heads", "mlp", "mode", "skip_pe", "reserved")] + [("eps", f32), ("queries", ptr), ("pe", ptr),
                ("kimg", ptr), ("ldk", i64), ("sk", i64), ("vimg", ptr), ("ldv", i64), ("sv", i64)] +
                [(n, ptr) for n in ("sq_w", "sq_b", "sk_w", "sk_b", "sv_w", "sv_b", "so_w", "so_b", "n1_g", "n1_b", "cq_w", "cq_b", "co_w", "co_b", "n2_g", "n2_b",
                                    "m1_w", "m1_b", "m2_w", "m2_b", "n3_g", "n3_b", "ik_w", "ik_b", "iv_w", "iv_b", "ktok", "vtok", "ws")] + [("ws_floats", i64)])


_lib = None
# entry points of the experiments build only (PSAM_BUILD_EXPERIMENTS=1, point_sam_amd/build.py): bound when the library exports them
EXPERIMENTAL = ("psam_twoway_decoder_force_fork", "psam_twoway_tokens_ws_floats", "psam_twoway_tokens", "psam_gemm_f16x3p_force_continuous")
_has_experiments = False


class PointSamHipError(RuntimeError):
    pass


def has_experiments() -> bool:
    """Whether the loaded library was built with PSAM_BUILD_EXPERIMENTS=1 (the measured-and-rejected paths and their entry points)."""
    load()
    return _has_experiments


def load():
    """Loads the library once.  Import torch first so that libamdhip64.so.7 resolves to the runtime torch uses
    (same soname; the loader reuses the already-mapped copy, so streams and device pointers are shared)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PointSamHipError(
                f"{LIB_PATH} is missing: build it with `python -m point_sam_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback on the product path."
            )
        lib = ctypes.CDLL(os.environ.get("PSAM_LIB_PATH", LIB_PATH))      # PSAM_LIB_PATH: an alternative build of the same library (A/B measurements)
        global _has_experiments
        _has_experiments = all(hasattr(lib, n) for n in EXPERIMENTAL)
        for name, (res, args) in SIGNATURES.items():
            if name in EXPERIMENTAL and not _has_experiments:
                continue
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status: int, what: str):
    if status != 0:
        msg = load().psam_last_error_string().decode()
        kind = "invalid argument" if status < 0 else f"hipError {status}"
        raise PointSamHipError(f"{what}: {kind} ({status}): {msg}")
