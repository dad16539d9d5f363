"""stable-diffusion.cpp_amd — ctypes binding of the MI355X denoise + VAE-decode engine.

The product is native: libsdcpp-host.so (C++ host: graph front-end, UNet/VAE graph builders, sampler,
the C ABI of include/sd-mi355x.h) and libggml-mi355x.so (the hand-written HIP ggml backend for gfx950).
This module only marshals numpy buffers across that C ABI; it performs no arithmetic.

The directory name contains a '.', so import it through `sdcpp_amd_loader.load()` (repo root) or
importlib (see __graft_entry__.py); inside Python it is registered as module `sdcpp_amd`.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
ROOT = PKG_DIR.parent
LIB_DIR = PKG_DIR / "lib"
HOST_LIB = Path(os.environ.get("SDCPP_HOST_LIB", str(LIB_DIR / "libsdcpp-host.so")))  # override: tests/test_abi.py runs the "forked enum" host build
BACKEND_LIB = Path(os.environ.get("SDCPP_BACKEND_LIB", str(LIB_DIR / "libggml-mi355x.so")))  # override: experiment builds (build.py SDCPP_BUILD_VARIANT)

# ggml_type numeric values (include/ggml-abi.h)
F32, F16, Q4_0, Q8_0, I32, BF16 = 0, 1, 2, 8, 26, 30
TYPE_SIZE = {F32: 4, F16: 2, BF16: 2, I32: 4}

# sd_model_family_t
SD15, SDXL, SD15_TINY, SDXL_TINY, SD35_LARGE, SD35_TINY, FLUX_DEV, FLUX_TINY, SD35_WIDE2, FLUX_WIDE1, SD3M_TINY, SD35_WIDE8, FLUX_WIDE8 = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12
# sd_pair_exchange_fn (include/sd-mi355x.h): (device address of the f32 eps buffer, element count, hipStream_t, user) -> ok
PAIR_EXCHANGE_FN = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)
# sdm_graph_eval_callback_t (include/sd-mi355x.h; the reference's sd_graph_eval_callback_t, include/stable-diffusion.h:442): (tensor, ask, user) -> bool
EVAL_CALLBACK_FN = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_bool, C.c_void_p)
# sdm_sample_method_t / sdm_scheduler_t (include/sd-mi355x.h): the reference's numeric values (include/stable-diffusion.h:38-83)
EULER, EULER_A, HEUN, DPM2, DPMPP2S_A, DPMPP2M, DPMPP2Mv2, IPNDM, IPNDM_V, LCM, DDIM_TRAILING = range(11)
TCD, RES_MULTISTEP, RES_2S, ER_SDE, EULER_CFG_PP, EULER_A_CFG_PP, EULER_GE, DPMPP2M_SDE, DPMPP2M_SDE_BT, LMS = range(11, 21)
SAMPLE_METHOD_DEFAULT = 21   # Euler for the DiT families, Euler-A otherwise (sd_get_default_sample_method)
SCHED_DISCRETE, SCHED_KARRAS, SCHED_EXPONENTIAL, SCHED_AYS, SCHED_GITS, SCHED_SGM_UNIFORM, SCHED_SIMPLE, SCHED_SMOOTHSTEP, SCHED_KL_OPTIMAL, SCHED_LCM = range(10)
SCHED_BONG_TANGENT, SCHED_FLUX, SCHED_BETA, SCHEDULER_DEFAULT = 10, 14, 15, 16   # DEFAULT: LCM for the LCM method, SIMPLE for DDIM trailing, FLUX for FLUX, DISCRETE otherwise (sd_get_default_scheduler)


class EngineError(RuntimeError):
    pass


class GgmlTensor(C.Structure):
    """struct ggml_tensor (include/ggml-abi.h) — read-only view for tests."""
    _fields_ = [
        ("type", C.c_int),
        ("buffer", C.c_void_p),
        ("ne", C.c_int64 * 4),
        ("nb", C.c_size_t * 4),
        ("op", C.c_int),
        ("op_params", C.c_int32 * 16),
        ("flags", C.c_int32),
        ("src", C.c_void_p * 10),
        ("view_src", C.c_void_p),
        ("view_offs", C.c_size_t),
        ("data", C.c_void_p),
        ("name", C.c_char * 160),
        ("extra", C.c_void_p),
        ("padding", C.c_char * 8),
    ]


class GgmlInitParams(C.Structure):
    _fields_ = [("mem_size", C.c_size_t), ("mem_buffer", C.c_void_p), ("no_alloc", C.c_bool)]


class SdCtxParams(C.Structure):
    _fields_ = [
        ("backend", C.c_char_p),
        ("model", C.c_int),
        ("wtype", C.c_int),
        ("diffusion_flash_attn", C.c_bool),
        ("diffusion_conv_direct", C.c_bool),
        ("vae_decode_only", C.c_bool),
        ("weight_seed", C.c_uint64),
        ("n_threads", C.c_int),
    ]


class SdSampleParams(C.Structure):
    _fields_ = [("txt_cfg", C.c_float), ("scheduler", C.c_int), ("sample_method", C.c_int),
                ("sample_steps", C.c_int), ("eta", C.c_float), ("custom_sigmas", C.c_void_p), ("custom_sigmas_count", C.c_int),
                ("slg_layers", C.c_void_p), ("slg_layer_count", C.c_int), ("slg_layer_start", C.c_float), ("slg_layer_end", C.c_float), ("slg_scale", C.c_float),
                ("apg_eta", C.c_float), ("apg_momentum", C.c_float), ("apg_norm_threshold", C.c_float), ("apg_norm_threshold_smoothing", C.c_float),
                ("shifted_timestep", C.c_int), ("flow_shift", C.c_float)]


class SdCondition(C.Structure):
    _fields_ = [("c_crossattn", C.POINTER(C.c_float)), ("ctx_dim", C.c_int64), ("n_tokens", C.c_int64),
                ("c_vector", C.POINTER(C.c_float)), ("vector_dim", C.c_int64)]


class SdImgGenParams(C.Structure):
    _fields_ = [("cond", SdCondition), ("uncond", SdCondition), ("width", C.c_int), ("height", C.c_int),
                ("sample_params", SdSampleParams), ("seed", C.c_int64), ("batch_count", C.c_int),
                ("device_batch", C.c_int), ("decode", C.c_bool), ("fuse_cfg_pair", C.c_bool), ("device_sampler", C.c_bool),
                ("init_latent", C.c_void_p), ("denoise_mask", C.c_void_p), ("strength", C.c_float)]


class SdTokenList(C.Structure):
    """sd_token_list_t (include/sd-mi355x.h)"""
    _fields_ = [("ids", C.POINTER(C.c_int32)), ("weights", C.POINTER(C.c_float)), ("n", C.c_int)]


class SdImage(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("channel", C.c_uint32), ("data", C.POINTER(C.c_uint8))]


class SdStats(C.Structure):
    _fields_ = [("last_sample_ms", C.c_double), ("last_decode_ms", C.c_double), ("unet_calls", C.c_int64),
                ("graph_nodes", C.c_int64), ("compute_buffer_bytes", C.c_size_t), ("weight_bytes", C.c_size_t),
                ("host_build_ms", C.c_double), ("host_alloc_ms", C.c_double), ("host_submit_ms", C.c_double),
                ("graph_cache_hits", C.c_int64)]


_lib = None

_PTR_FUNCS = """ggml_init ggml_new_tensor_1d ggml_new_tensor_2d ggml_new_tensor_3d ggml_new_tensor_4d ggml_dup_tensor
ggml_add ggml_add_inplace ggml_sub ggml_mul ggml_mul_inplace ggml_div ggml_scale ggml_scale_inplace ggml_silu
ggml_silu_inplace ggml_gelu ggml_gelu_inplace ggml_gelu_quick ggml_sigmoid ggml_tanh ggml_relu ggml_norm ggml_rms_norm
ggml_group_norm ggml_mul_mat ggml_cpy ggml_cast ggml_cont ggml_reshape_1d ggml_reshape_2d ggml_reshape_3d ggml_reshape_4d
ggml_view_1d ggml_view_2d ggml_view_3d ggml_view_4d ggml_permute ggml_transpose ggml_repeat ggml_concat ggml_soft_max
ggml_soft_max_inplace ggml_soft_max_ext ggml_im2col ggml_conv_2d ggml_conv_2d_direct ggml_upscale ggml_pad
ggml_timestep_embedding ggml_flash_attn_ext ggml_new_graph ggml_new_graph_custom ggml_graph_node ggml_set_name
ggml_gallocr_new ggml_backend_load ggml_backend_dev_get ggml_backend_dev_by_name ggml_backend_dev_init
ggml_backend_get_default_buffer_type ggml_backend_alloc_ctx_tensors ggml_backend_dev_buffer_type ggml_unary
ggml_unary_inplace ggml_get_rows ggml_dup sdm_new_ctx""".split()


def lib() -> C.CDLL:
    """Load libsdcpp-host.so (builds nothing; call build.build_all() first)."""
    global _lib
    if _lib is not None:
        return _lib
    if not HOST_LIB.exists():
        raise EngineError(f"{HOST_LIB} missing — run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(str(HOST_LIB), mode=C.RTLD_GLOBAL)
    for name in _PTR_FUNCS:
        getattr(L, name).restype = C.c_void_p
    L.ggml_init.argtypes = [GgmlInitParams]
    L.ggml_free.argtypes = [C.c_void_p]
    for n, k in (("ggml_new_tensor_1d", 1), ("ggml_new_tensor_2d", 2), ("ggml_new_tensor_3d", 3), ("ggml_new_tensor_4d", 4)):
        getattr(L, n).argtypes = [C.c_void_p, C.c_int] + [C.c_int64] * k
    for n, k in (("ggml_reshape_1d", 1), ("ggml_reshape_2d", 2), ("ggml_reshape_3d", 3), ("ggml_reshape_4d", 4)):
        getattr(L, n).argtypes = [C.c_void_p, C.c_void_p] + [C.c_int64] * k
    L.ggml_view_4d.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int64] * 4 + [C.c_size_t] * 4
    L.ggml_view_3d.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int64] * 3 + [C.c_size_t] * 3
    L.ggml_view_2d.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int64] * 2 + [C.c_size_t] * 2
    for n in ("ggml_add", "ggml_add_inplace", "ggml_sub", "ggml_mul", "ggml_mul_inplace", "ggml_div", "ggml_mul_mat",
              "ggml_cpy", "ggml_repeat", "ggml_get_rows"):
        getattr(L, n).argtypes = [C.c_void_p] * 3
    for n in ("ggml_silu", "ggml_silu_inplace", "ggml_gelu", "ggml_gelu_inplace", "ggml_gelu_quick", "ggml_sigmoid",
              "ggml_tanh", "ggml_relu", "ggml_cont", "ggml_transpose", "ggml_soft_max", "ggml_soft_max_inplace", "ggml_dup"):
        getattr(L, n).argtypes = [C.c_void_p] * 2
    L.ggml_scale.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
    L.ggml_scale_inplace.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
    L.ggml_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
    L.ggml_rms_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
    L.ggml_group_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
    L.ggml_cast.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.ggml_permute.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4
    L.ggml_concat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.ggml_soft_max_ext.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float]
    L.ggml_im2col.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6 + [C.c_bool, C.c_int]
    L.ggml_conv_2d.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6
    L.ggml_conv_2d_direct.argtypes = [C.c_void_p] * 3 + [C.c_int] * 6
    L.ggml_upscale.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.ggml_pad.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4
    L.ggml_timestep_embedding.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.ggml_flash_attn_ext.argtypes = [C.c_void_p] * 5 + [C.c_float] * 3
    L.ggml_flash_attn_ext_set_prec.argtypes = [C.c_void_p, C.c_int]
    L.ggml_unary.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.ggml_unary_inplace.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.ggml_new_graph.argtypes = [C.c_void_p]
    L.ggml_new_graph_custom.argtypes = [C.c_void_p, C.c_size_t, C.c_bool]
    L.ggml_build_forward_expand.argtypes = [C.c_void_p, C.c_void_p]
    L.ggml_graph_n_nodes.argtypes = [C.c_void_p]
    L.ggml_graph_node.argtypes = [C.c_void_p, C.c_int]
    L.ggml_set_name.argtypes = [C.c_void_p, C.c_char_p]
    L.ggml_set_input.argtypes = [C.c_void_p]
    L.ggml_set_output.argtypes = [C.c_void_p]
    L.ggml_nbytes.argtypes = [C.c_void_p]
    L.ggml_nbytes.restype = C.c_size_t
    L.ggml_nelements.argtypes = [C.c_void_p]
    L.ggml_nelements.restype = C.c_int64
    L.ggml_gallocr_new.argtypes = [C.c_void_p]
    L.ggml_gallocr_free.argtypes = [C.c_void_p]
    L.ggml_gallocr_alloc_graph.argtypes = [C.c_void_p, C.c_void_p]
    L.ggml_gallocr_alloc_graph.restype = C.c_bool
    L.ggml_gallocr_get_buffer_size.argtypes = [C.c_void_p, C.c_int]
    L.ggml_gallocr_get_buffer_size.restype = C.c_size_t
    L.ggml_backend_load.argtypes = [C.c_char_p]
    L.ggml_backend_dev_count.restype = C.c_size_t
    L.ggml_backend_dev_get.argtypes = [C.c_size_t]
    L.ggml_backend_dev_by_name.argtypes = [C.c_char_p]
    L.ggml_backend_dev_name.argtypes = [C.c_void_p]
    L.ggml_backend_dev_name.restype = C.c_char_p
    L.ggml_backend_dev_description.argtypes = [C.c_void_p]
    L.ggml_backend_dev_description.restype = C.c_char_p
    L.ggml_backend_dev_init.argtypes = [C.c_void_p, C.c_char_p]
    L.ggml_backend_dev_supports_op.argtypes = [C.c_void_p, C.c_void_p]
    L.ggml_backend_dev_supports_op.restype = C.c_bool
    L.ggml_backend_dev_memory.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.ggml_backend_free.argtypes = [C.c_void_p]
    L.ggml_backend_name.argtypes = [C.c_void_p]
    L.ggml_backend_name.restype = C.c_char_p
    L.ggml_backend_get_default_buffer_type.argtypes = [C.c_void_p]
    L.ggml_backend_alloc_ctx_tensors.argtypes = [C.c_void_p, C.c_void_p]
    L.ggml_backend_buffer_free.argtypes = [C.c_void_p]
    L.ggml_backend_buffer_set_usage.argtypes = [C.c_void_p, C.c_int]
    L.ggml_backend_tensor_set.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    L.ggml_backend_tensor_get.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
    L.ggml_backend_graph_compute.argtypes = [C.c_void_p, C.c_void_p]
    L.ggml_backend_graph_compute.restype = C.c_int
    L.ggml_backend_supports_op.argtypes = [C.c_void_p, C.c_void_p]
    L.ggml_backend_supports_op.restype = C.c_bool
    L.ggml_backend_synchronize.argtypes = [C.c_void_p]
    L.ggml_quantize_chunk.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
    L.ggml_quantize_chunk.restype = C.c_size_t
    L.ggml_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
    L.ggml_row_size.argtypes = [C.c_int, C.c_int64]
    L.ggml_row_size.restype = C.c_size_t
    # engine API
    L.sd_load_backend.argtypes = [C.c_char_p]
    L.sd_load_backend.restype = C.c_bool
    L.sd_device_name.argtypes = [C.c_int]
    L.sd_device_name.restype = C.c_char_p
    L.sd_device_description.argtypes = [C.c_int]
    L.sd_device_description.restype = C.c_char_p
    L.sd_last_error.restype = C.c_char_p
    L.sdm_ctx_params_init.argtypes = [C.POINTER(SdCtxParams)]
    L.sdm_sample_params_init.argtypes = [C.POINTER(SdSampleParams)]
    L.sdm_img_gen_params_init.argtypes = [C.POINTER(SdImgGenParams)]
    L.sdm_new_ctx.argtypes = [C.POINTER(SdCtxParams)]
    L.sdm_free_ctx.argtypes = [C.c_void_p]
    L.sd_tensor_count.argtypes = [C.c_void_p]
    L.sd_tensor_count.restype = C.c_int64
    L.sd_tensor_name.argtypes = [C.c_void_p, C.c_int64]
    L.sd_tensor_name.restype = C.c_char_p
    L.sd_tensor_info.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
    L.sd_tensor_info.restype = C.c_bool
    L.sd_get_tensor_f32.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    L.sd_get_tensor_f32.restype = C.c_bool
    L.sd_set_tensor_f32.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    L.sd_set_tensor_f32.restype = C.c_bool
    L.sd_unet_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    L.sd_unet_forward.restype = C.c_bool
    L.sd_vae_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.sd_vae_decode.restype = C.c_bool
    L.sd_sample_latents.argtypes = [C.c_void_p, C.POINTER(SdImgGenParams), C.c_void_p]
    L.sd_sample_latents.restype = C.c_bool
    L.sdm_generate_image.argtypes = [C.c_void_p, C.POINTER(SdImgGenParams), C.POINTER(C.POINTER(SdImage)), C.POINTER(C.c_int)]
    L.sdm_generate_image.restype = C.c_bool
    L.sdm_free_images.argtypes = [C.POINTER(SdImage), C.c_int]
    L.sd_philox_randn.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    L.sd_philox_uint32.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    L.sd_get_sigmas.argtypes = [C.c_int, C.c_void_p]
    L.sd_get_flow_sigmas.argtypes = [C.c_int, C.c_float, C.c_void_p]
    L.sd_get_flux_sigmas.argtypes = [C.c_int, C.c_int, C.c_void_p]
    L.sd_set_guidance.argtypes = [C.c_void_p, C.c_float]
    L.sd_set_guidance.restype = None
    L.sd_set_vae_conv2d_scale.argtypes = [C.c_void_p, C.c_float]
    L.sd_set_vae_conv2d_scale.restype = C.c_bool
    L.sd_set_pair_exchange.argtypes = [C.c_void_p, PAIR_EXCHANGE_FN, C.c_void_p, C.c_int]
    L.sd_set_pair_exchange.restype = None
    L.sd_rccl_get_unique_id.argtypes = [C.c_void_p]
    L.sd_rccl_get_unique_id.restype = C.c_bool
    L.sd_rccl_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.sd_rccl_comm_create.restype = C.c_void_p
    L.sd_rccl_comm_destroy.argtypes = [C.c_void_p]
    L.sd_rccl_comm_destroy.restype = None
    L.sd_set_pair_exchange_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.sd_set_pair_exchange_rccl.restype = C.c_bool
    L.sd_rccl_last_error.restype = C.c_char_p
    L.sd_gen_flux_pe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_float, C.c_void_p]
    L.sd_sigma_to_t.argtypes = [C.c_float]
    L.sd_sigma_to_t.restype = C.c_float
    L.sd_get_stats.argtypes = [C.c_void_p, C.POINTER(SdStats)]
    L.sd_load_weights_prefixed.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.sd_load_weights_prefixed.restype = C.c_int64
    L.sd_convert_tensor_name.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_size_t]
    L.sd_convert_tensor_name.restype = C.c_bool
    L.sd_text_encoders_init.argtypes = [C.c_void_p]
    L.sd_text_encoders_init.restype = C.c_bool
    L.sd_clip_forward.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_bool, C.c_int, C.c_void_p, C.c_int64]
    L.sd_clip_forward.restype = C.c_int64
    L.sd_t5_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    L.sd_t5_forward.restype = C.c_int64
    L.sd_t5_relative_position_buckets.argtypes = [C.c_int, C.c_int, C.c_void_p]
    L.sd_get_learned_condition.argtypes = [C.c_void_p, C.POINTER(SdTokenList), C.POINTER(SdTokenList), C.POINTER(SdTokenList), C.c_int, C.c_int, C.c_int,
                                           C.c_bool, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.sd_get_learned_condition.restype = C.c_bool
    L.sdm_set_backend_eval_callback.argtypes = [EVAL_CALLBACK_FN, C.c_void_p]
    L.sdm_set_backend_eval_callback.restype = None
    L.sdm_backend_graph_compute_with_eval_callback.argtypes = [C.c_void_p, C.c_void_p, EVAL_CALLBACK_FN, C.c_void_p]
    L.sdm_backend_graph_compute_with_eval_callback.restype = C.c_int
    _lib = L
    return L


_loaded_plugins: set[str] = set()


def load_backend(path: os.PathLike | str) -> None:
    """Register a ggml backend plug-in (.so exporting ggml_backend_init)."""
    p = str(Path(path).resolve())
    if p in _loaded_plugins:
        return
    if not lib().sd_load_backend(p.encode()):
        raise EngineError(f"failed to load backend plug-in {p}")
    _loaded_plugins.add(p)


def load_mi355x_backend() -> None:
    """Load the product backend.  Raises (never falls back) when the .so or a gfx950 GPU is missing."""
    if not BACKEND_LIB.exists():
        raise EngineError(f"{BACKEND_LIB} missing — the HIP backend was not built")
    load_backend(BACKEND_LIB)
    names = devices()
    if not any(n.upper().startswith("MI355X") for n in names):
        raise EngineError(f"libggml-mi355x.so loaded but exposes no MI355X device (devices: {names}); is a gfx950 GPU visible?")


_BACKEND_STAT_FIELDS = ("graphs_computed plans_built nodes_seen kernels_planned kernels_launched fused_conv fused_conv_bounced fused_linear "
                        "fused_norm fused_geglu fused_attention generic_matmul swizzled_weight_bytes graph_replays fused_linear_geglu "
                        "split_k_gemms head_major_gemms fused_modulate fused_gate fused_gelu fused_rope fused_concat_heads qgemv_linears fused_chan_add fused_proj_tokens gemm_attention fused_q16 split_k_inlaunch qgemm16_linears fgemv_linears fused_presilu fused_sibling_linears hoisted_kv_linears window_convs hoisted_emb_linears fused_rows16 fused_joint_qkv jit_images fused_cat_rows16 fused_gn_stats fused_ln_reduce redirect_fallbacks fused_concat_gn fused_conv_scale view_graphs plans_evicted hoisted_mod_linears jit_overlapped view_external_nodes qinloop_linears flash_out_alias flash_slice_images").split()


class BackendStats(C.Structure):
    """struct ggml_backend_mi355x_stats (include/ggml-mi355x.h)"""
    _fields_ = [(n, C.c_int64) for n in _BACKEND_STAT_FIELDS]


_backend_cdll = None


def _backend() -> C.CDLL:
    global _backend_cdll
    if _backend_cdll is None:
        load_mi355x_backend()
        _backend_cdll = C.CDLL(str(BACKEND_LIB))  # same handle the plug-in loader dlopen()ed
        _backend_cdll.ggml_backend_mi355x_get_stats.argtypes = [C.POINTER(BackendStats)]
        _backend_cdll.ggml_backend_mi355x_set_option.argtypes = [C.c_char_p, C.c_int]
    return _backend_cdll


def backend_stats() -> dict:
    """Process-wide planner counters of the MI355X backend (plans built, fusions taken, kernels launched)."""
    st = BackendStats()
    _backend().ggml_backend_mi355x_get_stats(C.byref(st))
    return {n: int(getattr(st, n)) for n in _BACKEND_STAT_FIELDS}


class KernelTiming(C.Structure):
    """struct ggml_backend_mi355x_kernel_timing (include/ggml-mi355x.h)"""
    _fields_ = [("kernel", C.c_char * 96), ("launches", C.c_int64), ("total_ms", C.c_double), ("total_flops", C.c_double),
                ("total_bytes", C.c_double), ("bound", C.c_int32), ("family", C.c_int32)]


KF_ALL = 0x3FFFF  # every kernel family (include/ggml-mi355x.h lists the bit positions)


def kernel_timing_enable(on) -> None:
    """HIP events on the launch stream around the dispatches of the enabled kernel families (resets the accumulators).
    True / 1: the dominant family only (256-row conv tiles — cheap enough for a timed region); an int > 1 is a family bit mask; 0 / False: off."""
    b = _backend()
    b.ggml_backend_mi355x_kernel_timing_enable_mask.argtypes = [C.c_uint32]
    mask = 1 if on is True else int(on or 0)
    b.ggml_backend_mi355x_kernel_timing_enable_mask(mask)


def _timing_dict(kt) -> dict:
    return {"kernel": kt.kernel.decode(), "family": int(kt.family), "bound": "hbm" if kt.bound else "mfma", "launches": int(kt.launches),
            "total_ms": float(kt.total_ms), "total_flops": float(kt.total_flops), "total_bytes": float(kt.total_bytes)}


def kernel_timings() -> list:
    """Per-family totals since enable / the previous call (synchronises the device and resets)."""
    arr = (KernelTiming * 32)()
    b = _backend()
    b.ggml_backend_mi355x_get_kernel_timings.argtypes = [C.POINTER(KernelTiming), C.c_int]
    b.ggml_backend_mi355x_get_kernel_timings.restype = C.c_int
    n = b.ggml_backend_mi355x_get_kernel_timings(arr, 32)
    return [_timing_dict(arr[i]) for i in range(n)]


def kernel_timing() -> dict:
    """The first timed family (the dominant kernel when only that one is enabled)."""
    t = kernel_timings()
    return t[0] if t else {"kernel": "(no timed launches)", "family": -1, "bound": "mfma", "launches": 0, "total_ms": 0.0, "total_flops": 0.0, "total_bytes": 0.0}


class Calibration(C.Structure):
    """struct ggml_backend_mi355x_calibration (include/ggml-mi355x.h)"""
    _fields_ = [("mfma_f16_tflops", C.c_float), ("mfma_clock_mhz", C.c_float), ("copy_tbs", C.c_float), ("read_tbs", C.c_float), ("compute_units", C.c_int)]


def calibrate() -> dict | None:
    """What the current device delivers (MFMA loop, float4 copy / read, clock under MFMA load): csrc/kernels/calib.hip."""
    b = _backend()
    b.ggml_backend_mi355x_calibrate.argtypes = [C.POINTER(Calibration)]
    c = Calibration()
    if b.ggml_backend_mi355x_calibrate(C.byref(c)) != 0:
        return None
    return {"mfma_f16_tflops": round(c.mfma_f16_tflops, 1), "mfma_clock_mhz": round(c.mfma_clock_mhz, 1), "copy_tbs": round(c.copy_tbs, 3), "read_tbs": round(c.read_tbs, 3),
            "compute_units": c.compute_units}


def backend_set_option(key: str, value: int) -> None:
    """ggml_backend_mi355x_set_option: fusion / mfma_gemm / hip_graph / flash_pattern / gemm16 / gemm16_variant"""
    _backend().ggml_backend_mi355x_set_option(key.encode(), int(value))


def devices() -> list[str]:
    L = lib()
    return [L.sd_device_name(i).decode() for i in range(L.sd_device_count())]


def _fptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class Engine:
    """RAII wrapper over sdm_ctx_t.  Array layouts are ggml's: ne0 fastest == numpy C-order reversed, i.e. a latent
    batch is a numpy array of shape [N, C, H, W]."""

    def __init__(self, model: int = SD15, backend: str | None = None, wtype: int = F16, flash_attn: bool = False,
                 conv_direct: bool = False, weight_seed: int = 1234):
        L = lib()
        p = SdCtxParams()
        L.sdm_ctx_params_init(C.byref(p))
        if backend is None:
            load_mi355x_backend()
            backend = "MI355X0"
        self._backend_name = backend.encode()
        p.backend = self._backend_name
        p.model = model
        p.wtype = wtype
        p.diffusion_flash_attn = flash_attn
        p.diffusion_conv_direct = conv_direct
        p.weight_seed = weight_seed
        self._ctx = L.sdm_new_ctx(C.byref(p))
        if not self._ctx:
            raise EngineError("sdm_new_ctx failed: " + L.sd_last_error().decode())
        self.model = model

    def close(self):
        if getattr(self, "_ctx", None):
            lib().sdm_free_ctx(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- text encoders / conditioner ----
    def text_encoders_init(self) -> None:
        if not lib().sd_text_encoders_init(self._ctx):
            raise EngineError("sd_text_encoders_init failed: " + lib().sd_last_error().decode())

    def clip_forward(self, which: int, ids, max_token_idx: int = 0, return_pooled: bool = False, clip_skip: int = -1) -> np.ndarray:
        """One CLIP text tower (0 = ViT-L, 1 = bigG) on token ids [n_tokens] -> hidden states [n_tokens, hidden] or the pooled vector."""
        ids = np.ascontiguousarray(ids, dtype=np.int32).ravel()
        self.text_encoders_init()
        cap = int(ids.size) * 4096 + 4096
        out = np.empty(cap, dtype=np.float32)
        n = lib().sd_clip_forward(self._ctx, which, ids.ctypes.data_as(C.c_void_p), int(ids.size), int(max_token_idx), bool(return_pooled), int(clip_skip),
                                  _fptr(out), cap)
        if n < 0:
            raise EngineError("sd_clip_forward failed: " + lib().sd_last_error().decode())
        out = out[:n].copy()
        return out if return_pooled else out.reshape(ids.size, -1)

    def t5_forward(self, ids) -> np.ndarray:
        ids = np.ascontiguousarray(ids, dtype=np.int32).ravel()
        self.text_encoders_init()
        cap = int(ids.size) * 4096
        out = np.empty(cap, dtype=np.float32)
        n = lib().sd_t5_forward(self._ctx, ids.ctypes.data_as(C.c_void_p), int(ids.size), _fptr(out), cap)
        if n < 0:
            raise EngineError("sd_t5_forward failed: " + lib().sd_last_error().decode())
        return out[:n].copy().reshape(ids.size, -1)

    def get_learned_condition(self, clip_l=None, clip_g=None, t5=None, *, clip_skip: int = -1, width: int = 512, height: int = 512,
                              zero_out_masked: bool = False):
        """Token ids (+ weights) -> (c_crossattn [1, n_tokens, ctx_dim], c_vector [1, dim] or None).  Each argument is an id array or an
        (ids, weights) pair; see sd_get_learned_condition."""
        keep = []

        def tl(x):
            if x is None:
                return None
            ids, w = x if isinstance(x, tuple) else (x, None)
            ids = np.ascontiguousarray(ids, dtype=np.int32).ravel()
            t = SdTokenList()
            t.ids = ids.ctypes.data_as(C.POINTER(C.c_int32))
            t.n = int(ids.size)
            keep.append(ids)
            if w is not None:
                w = np.ascontiguousarray(w, dtype=np.float32).ravel()
                assert w.size == ids.size
                t.weights = w.ctypes.data_as(C.POINTER(C.c_float))
                keep.append(w)
            return t

        lists = [tl(clip_l), tl(clip_g), tl(t5)]
        ptrs = [C.byref(t) if t is not None else None for t in lists]
        ne = (C.c_int64 * 2)()
        vn = C.c_int64()
        L = lib()
        args = (self._ctx, ptrs[0], ptrs[1], ptrs[2], int(clip_skip), int(width), int(height), bool(zero_out_masked))
        # sizes are a function of the token counts only: ask first (runs the encoders), then fetch
        cross = np.empty(max(1, sum(t.n for t in lists if t is not None)) * 2 * 4096, dtype=np.float32)
        vec = np.empty(8192, dtype=np.float32)
        if not L.sd_get_learned_condition(*args, _fptr(cross), cross.size, ne, _fptr(vec), vec.size, C.byref(vn)):
            raise EngineError("sd_get_learned_condition failed: " + L.sd_last_error().decode())
        c = cross[: ne[0] * ne[1]].copy().reshape(1, ne[1], ne[0])
        y = vec[: vn.value].copy().reshape(1, -1) if vn.value > 0 else None
        return c, y

    # ---- weights ----
    def tensor_names(self) -> list[str]:
        L = lib()
        return [L.sd_tensor_name(self._ctx, i).decode() for i in range(L.sd_tensor_count(self._ctx))]

    def tensor_info(self, name: str):
        ne = (C.c_int64 * 4)()
        ty = C.c_int()
        nb = C.c_size_t()
        if not lib().sd_tensor_info(self._ctx, name.encode(), ne, C.byref(ty), C.byref(nb)):
            raise KeyError(name)
        return list(ne), ty.value, nb.value

    def get_tensor(self, name: str) -> np.ndarray:
        """Dequantised f32 copy, numpy shape = reversed ggml ne (trailing 1s dropped)."""
        ne, _, _ = self.tensor_info(name)
        n = int(np.prod(ne))
        out = np.empty(n, dtype=np.float32)
        if not lib().sd_get_tensor_f32(self._ctx, name.encode(), _fptr(out), n):
            raise EngineError("sd_get_tensor_f32 failed")
        shape = [d for d in reversed(ne)]
        while len(shape) > 1 and shape[0] == 1:
            shape = shape[1:]
        return out.reshape(shape)

    def load_weights(self, path, prefix: str | None = None) -> dict:
        """Load a safetensors / GGUF checkpoint into the declared parameters.  File names are converted to the engine's canonical
        dialect first (diffusers UNet / VAE, OpenCLIP, component aliases); `prefix` is prepended to every file name before that
        (diffusers keeps one un-prefixed file per sub-model: "unet.", "vae.", "text_encoder.", "text_encoder_2.")."""
        miss, unused = C.c_int64(0), C.c_int64(0)
        L = lib()
        n = L.sd_load_weights_prefixed(self._ctx, str(path).encode(), prefix.encode() if prefix else None, C.byref(miss), C.byref(unused))
        if n < 0:
            raise EngineError("sd_load_weights failed: " + L.sd_last_error().decode())
        return {"loaded": int(n), "missing": int(miss.value), "unused": int(unused.value)}

    def convert_tensor_name(self, name: str) -> str:
        """A checkpoint tensor name in the engine's canonical dialect (convert_tensor_name, src/name_conversion.cpp:1346)."""
        buf = C.create_string_buffer(512)
        if not lib().sd_convert_tensor_name(self._ctx, name.encode(), buf, 512):
            raise EngineError("sd_convert_tensor_name: name too long")
        return buf.value.decode()

    def set_vae_conv2d_scale(self, scale: float) -> None:
        """AutoEncoderKL::set_conv2d_scale: conv(x * s) / s + b on every VAE conv (SDXL engines start with 1/32, like the reference without --vae)."""
        if not lib().sd_set_vae_conv2d_scale(self._ctx, float(scale)):
            raise EngineError(lib().sd_last_error().decode())

    def set_guidance(self, guidance: float) -> None:
        """FLUX distilled-guidance input (default 3.5)"""
        lib().sd_set_guidance(self._ctx, float(guidance))

    def set_pair_exchange(self, fn, branch: int = 0) -> None:
        """CFG-pair split (sd_set_pair_exchange): `fn(device_ptr, count, stream) -> bool` sums this rank's weighted eps with its partner's in
        place, in device memory, once per step of the device-resident trajectory; branch 0 = cond, 1 = uncond.  fn = None removes it."""
        if fn is None:
            self._pair_cb = PAIR_EXCHANGE_FN(0)
        else:
            def cb(ptr, count, stream, _user, _fn=fn):
                try:
                    return bool(_fn(ptr, int(count), stream))
                except Exception:  # never unwind through the C frames
                    import traceback
                    traceback.print_exc()
                    return False
            self._pair_cb = PAIR_EXCHANGE_FN(cb)   # kept alive as long as it is installed
        lib().sd_set_pair_exchange(self._ctx, self._pair_cb, None, int(branch))

    def set_pair_exchange_rccl(self, comm, branch: int = 0) -> None:
        """The native form (csrc/host/rccl_exchange.cpp): one in-place ncclAllReduce per step on the backend stream, issued from C++ through a
        communicator made by rccl_comm_create; comm = None removes it."""
        if not lib().sd_set_pair_exchange_rccl(self._ctx, comm, int(branch)):
            raise EngineError("sd_set_pair_exchange_rccl failed: " + lib().sd_rccl_last_error().decode())

    def set_tensor(self, name: str, value: np.ndarray) -> None:
        v = _f32(value).ravel()
        if not lib().sd_set_tensor_f32(self._ctx, name.encode(), _fptr(v), v.size):
            raise EngineError(f"sd_set_tensor_f32({name}) failed")

    # ---- hot path ----
    def unet_forward(self, x: np.ndarray, timesteps: np.ndarray, context: np.ndarray, y: np.ndarray | None = None) -> np.ndarray:
        """x [N,C,H,W]; timesteps [N]; context [Nc,77,ctx_dim] (Nc in {1,N}); y [Ny,adm] or None."""
        x = _f32(x)
        t = _f32(timesteps)
        ctxt = _f32(context)
        n, c, h, w = x.shape
        out = np.empty_like(x)
        yy = None if y is None else _f32(y)
        ok = lib().sd_unet_forward(self._ctx, _fptr(x), w, h, c, n, _fptr(t), _fptr(ctxt), ctxt.shape[2], ctxt.shape[1],
                                   ctxt.shape[0], _fptr(yy), 0 if yy is None else yy.shape[1], 0 if yy is None else yy.shape[0],
                                   _fptr(out))
        if not ok:
            raise EngineError("sd_unet_forward failed: " + lib().sd_last_error().decode())
        return out

    def unet_forward_skip_layers(self, x, timesteps, context, y, skip_layers) -> np.ndarray:
        """sd_unet_forward_skip_layers: the MMDiT forward without the listed joint blocks (skip-layer guidance's extra evaluation)."""
        x, t, ctxt, yy = _f32(x), _f32(timesteps), _f32(context), (None if y is None else _f32(y))
        n, c, h, w = x.shape
        out = np.empty_like(x)
        sk = np.ascontiguousarray(skip_layers, dtype=np.int32)
        L = lib()
        L.sd_unet_forward_skip_layers.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                                  C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
        L.sd_unet_forward_skip_layers.restype = C.c_bool
        if not L.sd_unet_forward_skip_layers(self._ctx, _fptr(x), w, h, c, n, _fptr(t), _fptr(ctxt), ctxt.shape[2], ctxt.shape[1], ctxt.shape[0], _fptr(yy),
                                             0 if yy is None else yy.shape[1], 0 if yy is None else yy.shape[0], sk.ctypes.data_as(C.c_void_p), sk.size, _fptr(out)):
            raise EngineError("sd_unet_forward_skip_layers failed: " + L.sd_last_error().decode())
        return out

    def vae_decode(self, latents: np.ndarray) -> np.ndarray:
        """latents [N,C,h,w] (diffusion scale) -> rgb [N,3,8h,8w] in [0,1]."""
        z = _f32(latents)
        n, c, h, w = z.shape
        out = np.empty((n, 3, h * 8, w * 8), dtype=np.float32)
        if not lib().sd_vae_decode(self._ctx, _fptr(z), w, h, c, n, _fptr(out)):
            raise EngineError("sd_vae_decode failed: " + lib().sd_last_error().decode())
        return out

    def vae_encode(self, rgb: np.ndarray, seed: int = 42, return_moments: bool = False):
        """sd_vae_encode: rgb [N,3,H,W] in [0,1] -> diffusion latents [N,zc,H/8,W/8] sampled with Philox(seed) (+ the moments [N,2*zc,H/8,W/8] the graph produced)."""
        x = _f32(rgb)
        n, c, h, w = x.shape
        assert c == 3
        zc = 16 if self.model in (SD35_LARGE, SD35_TINY, FLUX_DEV, FLUX_TINY, SD35_WIDE2, FLUX_WIDE1, SD3M_TINY, SD35_WIDE8, FLUX_WIDE8) else 4
        out = np.empty((n, zc, h // 8, w // 8), dtype=np.float32)
        mom = np.empty((n, 2 * zc, h // 8, w // 8), dtype=np.float32)
        L = lib()
        L.sd_vae_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
        L.sd_vae_encode.restype = C.c_bool
        if not L.sd_vae_encode(self._ctx, _fptr(x), w, h, n, seed, _fptr(out), _fptr(mom)):
            raise EngineError("sd_vae_encode failed: " + L.sd_last_error().decode())
        return (out, mom) if return_moments else out

    def tae_decode(self, latents: np.ndarray) -> np.ndarray:
        """sd_tae_decode — TAESD (src/model/vae/tae.hpp): latents [N,C,h,w] (diffusion scale, unscaled) -> rgb [N,3,8h,8w], not clamped."""
        z = _f32(latents)
        n, c, h, w = z.shape
        out = np.empty((n, 3, h * 8, w * 8), dtype=np.float32)
        L = lib()
        L.sd_tae_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.sd_tae_decode.restype = C.c_bool
        if not L.sd_tae_decode(self._ctx, _fptr(z), w, h, c, n, _fptr(out)):
            raise EngineError("sd_tae_decode failed: " + L.sd_last_error().decode())
        return out

    def set_prediction(self, prediction: int):
        """sd_set_prediction: 0 = eps (default), 1 = v-prediction (CompVisVDenoiser's scalings)."""
        L = lib()
        L.sd_set_prediction.argtypes = [C.c_void_p, C.c_int]
        L.sd_set_prediction.restype = C.c_bool
        if not L.sd_set_prediction(self._ctx, int(prediction)):
            raise EngineError("sd_set_prediction failed: " + L.sd_last_error().decode())

    def use_tae(self, on: bool = True):
        """sd_use_tae: generate_image decodes with TAESD instead of the KL-VAE (the reference's --taesd without --taesd-preview-only)."""
        L = lib()
        L.sd_use_tae.argtypes = [C.c_void_p, C.c_bool]
        L.sd_use_tae.restype = C.c_bool
        if not L.sd_use_tae(self._ctx, bool(on)):
            raise EngineError("sd_use_tae failed: " + L.sd_last_error().decode())

    def _gen_params(self, cond, uncond, width, height, steps, cfg, seed, batch, device_batch, method, eta, cond_y=None, uncond_y=None,
                    fuse_cfg=False, device_sampler=False, scheduler=SCHEDULER_DEFAULT, init_latent=None, strength=0.75, custom_sigmas=None, flow_shift=None, denoise_mask=None, slg=None, shifted_timestep=0, apg=None):
        p = SdImgGenParams()
        lib().sdm_img_gen_params_init(C.byref(p))
        keep = []

        def fill(dst, ctx_arr, vec):
            a = _f32(ctx_arr)
            keep.append(a)
            dst.c_crossattn = a.ctypes.data_as(C.POINTER(C.c_float))
            dst.n_tokens, dst.ctx_dim = a.shape[-2], a.shape[-1]
            if vec is not None:
                v = _f32(vec)
                keep.append(v)
                dst.c_vector = v.ctypes.data_as(C.POINTER(C.c_float))
                dst.vector_dim = v.shape[-1]

        fill(p.cond, cond, cond_y)
        if uncond is not None:
            fill(p.uncond, uncond, uncond_y)
        p.width, p.height = width, height
        p.sample_params.txt_cfg = cfg
        p.sample_params.sample_steps = steps
        p.sample_params.sample_method = method
        p.sample_params.scheduler = scheduler
        p.sample_params.eta = eta
        p.seed = seed
        p.batch_count = batch
        p.device_batch = device_batch
        p.fuse_cfg_pair = fuse_cfg
        p.device_sampler = device_sampler
        if custom_sigmas is not None:
            cs = _f32(custom_sigmas)
            keep.append(cs)
            p.sample_params.custom_sigmas = cs.ctypes.data_as(C.c_void_p)
            p.sample_params.custom_sigmas_count = cs.size
        p.sample_params.shifted_timestep = shifted_timestep
        if apg is not None:   # (eta, momentum, norm_threshold, norm_threshold_smoothing)
            p.sample_params.apg_eta, p.sample_params.apg_momentum, p.sample_params.apg_norm_threshold, p.sample_params.apg_norm_threshold_smoothing = apg
        if flow_shift is not None:
            p.sample_params.flow_shift = flow_shift
        if slg is not None:   # (layers, scale[, start, end])
            ly = np.ascontiguousarray(slg[0], dtype=np.int32)
            keep.append(ly)
            p.sample_params.slg_layers = ly.ctypes.data_as(C.c_void_p)
            p.sample_params.slg_layer_count = ly.size
            p.sample_params.slg_scale = slg[1]
            if len(slg) > 2:
                p.sample_params.slg_layer_start, p.sample_params.slg_layer_end = slg[2], slg[3]
        if init_latent is not None:
            il = _f32(init_latent)
            keep.append(il)
            p.init_latent = il.ctypes.data_as(C.c_void_p)
            p.strength = strength
            if denoise_mask is not None:
                dm = _f32(denoise_mask)
                assert dm.shape == il.shape[-2:]
                keep.append(dm)
                p.denoise_mask = dm.ctypes.data_as(C.c_void_p)
        return p, keep

    def sample_latents(self, cond, uncond=None, width=512, height=512, steps=20, cfg=7.0, seed=42, batch=1, device_batch=0,
                       method=SAMPLE_METHOD_DEFAULT, eta=float("inf"), cond_y=None, uncond_y=None, fuse_cfg=False, device_sampler=False,
                       scheduler=SCHEDULER_DEFAULT, init_latent=None, strength=0.75, custom_sigmas=None, flow_shift=None, denoise_mask=None, slg=None, shifted_timestep=0, apg=None) -> np.ndarray:
        """init_latent [C,h/8,w/8] (+ strength): img2img — the trajectory starts from the noised init latent, (int)(steps * strength) steps before the end of the ladder;
        custom_sigmas: the ladder to use instead of the scheduler's; flow_shift: the flow families' time shift."""
        p, keep = self._gen_params(cond, uncond, width, height, steps, cfg, seed, batch, device_batch, method, eta, cond_y, uncond_y, fuse_cfg,
                                   device_sampler, scheduler, init_latent, strength, custom_sigmas, flow_shift, denoise_mask, slg, shifted_timestep, apg)
        ch = 16 if self.model in (SD35_LARGE, SD35_TINY, FLUX_DEV, FLUX_TINY, SD35_WIDE2, FLUX_WIDE1, SD3M_TINY, SD35_WIDE8, FLUX_WIDE8) else 4
        out = np.empty((batch, ch, height // 8, width // 8), dtype=np.float32)
        if not lib().sd_sample_latents(self._ctx, C.byref(p), _fptr(out)):
            raise EngineError("sd_sample_latents failed: " + lib().sd_last_error().decode())
        return out

    def generate_image(self, cond, uncond=None, width=512, height=512, steps=20, cfg=7.0, seed=42, batch=1, device_batch=0,
                       method=SAMPLE_METHOD_DEFAULT, eta=float("inf"), cond_y=None, uncond_y=None, fuse_cfg=False, device_sampler=False,
                       scheduler=SCHEDULER_DEFAULT, init_latent=None, strength=0.75) -> np.ndarray:
        """-> uint8 [batch, H, W, 3]"""
        p, keep = self._gen_params(cond, uncond, width, height, steps, cfg, seed, batch, device_batch, method, eta, cond_y, uncond_y, fuse_cfg,
                                   device_sampler, scheduler, init_latent, strength)
        imgs = C.POINTER(SdImage)()
        n = C.c_int()
        if not lib().sdm_generate_image(self._ctx, C.byref(p), C.byref(imgs), C.byref(n)):
            raise EngineError("sdm_generate_image failed: " + lib().sd_last_error().decode())
        out = np.empty((n.value, height, width, 3), dtype=np.uint8)
        for i in range(n.value):
            out[i] = np.ctypeslib.as_array(imgs[i].data, shape=(height, width, 3))
        lib().sdm_free_images(imgs, n.value)
        return out

    def stats(self) -> dict:
        s = SdStats()
        lib().sd_get_stats(self._ctx, C.byref(s))
        return {f[0]: getattr(s, f[0]) for f in SdStats._fields_}


def op_number(name: str) -> int:
    """The host's numeric value of ggml op `name` (resolved through ggml_op_name like the plug-in does at init)."""
    L = lib()
    L.ggml_op_name.argtypes = [C.c_int]
    L.ggml_op_name.restype = C.c_char_p
    for i in range(200):
        nm = L.ggml_op_name(i)
        if nm is None:
            break
        if nm.decode() == name:
            return i
    raise EngineError(f"host has no ggml op named {name}")


class EvalTrace:
    """Node-by-node evaluation of every graph the engines compute while the context manager is active (sdm_set_backend_eval_callback: the reference's
    sd_set_backend_eval_callback / imatrix pattern, src/runtime/imatrix.cpp:39-100).  `want(index, tensor_struct)` decides which nodes the graph is cut
    behind; each wanted node is downloaded after its slice ran (f32 / f16 tensors; `with_src1` also downloads src[1] of a MUL_MAT, what the imatrix
    collector reads).  records: list of (node index, op, name, value, src1 value or None) in evaluation order; `stop_after` makes the callback return false
    after that many records (-> GGML_STATUS_ABORTED)."""

    def __init__(self, want, with_src1: bool = True, stop_after: int | None = None, mul_mat_op: int | None = None):
        self.want, self.with_src1, self.stop_after, self.mul_mat_op = want, with_src1, stop_after, mul_mat_op
        self.records, self.asked, self.graphs = [], 0, 0
        self._index, self._reset = -1, True
        if self.mul_mat_op is None:
            self.mul_mat_op = op_number("MUL_MAT")
        self._cb = EVAL_CALLBACK_FN(self._call)

    def _fetch(self, tptr):
        """f32 / f16 tensor -> float32 numpy [ne3, ne2, ne1, ne0]; a strided view is read out of the (contiguous) tensor it aliases."""
        L = lib()
        ts = C.cast(tptr, C.POINTER(GgmlTensor)).contents
        if ts.type not in (F32, F16) or not ts.data:
            return None
        dt = np.float32 if ts.type == F32 else np.float16
        esz = TYPE_SIZE[ts.type]
        shape = [int(ts.ne[i]) for i in range(3, -1, -1)]
        strides = [int(ts.nb[i]) for i in range(3, -1, -1)]

        def contiguous(t):
            return t.nb[0] == TYPE_SIZE.get(t.type, 0) and all(t.nb[i] == t.nb[i - 1] * t.ne[i - 1] for i in range(1, 4))

        if contiguous(ts):
            out = np.empty(shape, dtype=dt)
            L.ggml_backend_tensor_get(tptr, out.ctypes.data_as(C.c_void_p), 0, out.nbytes)
            return out.astype(np.float32)
        if not ts.view_src:
            return None
        root = C.cast(ts.view_src, C.POINTER(GgmlTensor)).contents
        if not contiguous(root) or not root.data:
            return None
        raw = np.empty(int(L.ggml_nbytes(ts.view_src)), dtype=np.uint8)
        L.ggml_backend_tensor_get(ts.view_src, raw.ctypes.data_as(C.c_void_p), 0, raw.nbytes)
        off = int(ts.data) - int(root.data)
        last = off + sum((n - 1) * st for n, st in zip(shape, strides)) + esz
        if off < 0 or last > raw.nbytes:
            return None
        return np.lib.stride_tricks.as_strided(raw[off:].view(dt) if (raw.nbytes - off) % esz == 0 else raw[off:off + (raw.nbytes - off) // esz * esz].view(dt),
                                               shape=shape, strides=strides).astype(np.float32)

    def _call(self, tptr, ask, _user):
        ts = C.cast(tptr, C.POINTER(GgmlTensor)).contents
        if ask:
            if self._reset:  # the node after a graph's final result starts the next graph
                self._index, self._reset = -1, False
                self.graphs += 1
            self._index += 1
            self.asked += 1
            self._reset = ts.name == b"ggml_runner_final_result_tensor"
            return bool(self.want(self._index, ts))
        src1 = None
        if self.with_src1 and self.mul_mat_op is not None and ts.op == self.mul_mat_op and ts.src[1]:
            src1 = self._fetch(ts.src[1])
        self.records.append((self._index, int(ts.op), ts.name.decode(errors="replace"), self._fetch(tptr), src1))
        return not (self.stop_after is not None and len(self.records) >= self.stop_after)

    def __enter__(self):
        lib().sdm_set_backend_eval_callback(self._cb, None)
        return self

    def __exit__(self, *a):
        lib().sdm_set_backend_eval_callback(EVAL_CALLBACK_FN(), None)


def t5_relative_position_buckets(q_len: int, k_len: int) -> np.ndarray:
    """T5 bidirectional relative-position buckets [q_len, k_len] (t5.hpp:463-530)."""
    out = np.empty(q_len * k_len, dtype=np.int32)
    lib().sd_t5_relative_position_buckets(q_len, k_len, out.ctypes.data_as(C.c_void_p))
    return out.reshape(q_len, k_len)


def rccl_unique_id() -> bytes:
    """ncclGetUniqueId through the host library (librccl.so loaded with dlopen): the 128 bytes rank 0 ships to its partner(s)."""
    buf = C.create_string_buffer(128)
    if not lib().sd_rccl_get_unique_id(buf):
        raise EngineError("sd_rccl_get_unique_id failed: " + lib().sd_rccl_last_error().decode())
    return buf.raw


def rccl_comm_create(device: int, nranks: int, rank: int, unique_id: bytes):
    """ncclCommInitRank on HIP device `device`; returns the opaque communicator for Engine.set_pair_exchange_rccl."""
    assert len(unique_id) == 128
    comm = lib().sd_rccl_comm_create(int(device), int(nranks), int(rank), C.create_string_buffer(unique_id, 128))
    if not comm:
        raise EngineError("sd_rccl_comm_create failed: " + lib().sd_rccl_last_error().decode())
    return comm


def rccl_comm_destroy(comm) -> None:
    lib().sd_rccl_comm_destroy(comm)


def cfg_combine(cond: np.ndarray, uncond: np.ndarray, scale: float) -> np.ndarray:
    """The host sampler's CFG combine (sd_cfg_combine): uncond + scale * (cond - uncond), each operation rounded to f32."""
    c, u = _f32(cond).ravel(), _f32(uncond).ravel()
    assert c.size == u.size
    out = np.empty_like(c)
    L = lib()
    L.sd_cfg_combine.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]
    L.sd_cfg_combine.restype = None
    L.sd_cfg_combine(c.ctypes.data, u.ctypes.data, c.size, float(scale), out.ctypes.data)
    return out.reshape(np.shape(cond))


def philox_randn(seed: int, offset: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.float32)
    lib().sd_philox_randn(seed, offset, n, _fptr(out))
    return out


def philox_uint32(seed: int, offset: int, n: int) -> np.ndarray:
    """The four Philox4x32-10 output words of counters (offset, 0, i, 0), i < n -> uint32 [n, 4]."""
    out = np.empty((n, 4), dtype=np.uint32)
    lib().sd_philox_uint32(seed, offset, n, out.ctypes.data_as(C.c_void_p))
    return out


def planar_rgb_to_u8(chw: np.ndarray) -> np.ndarray:
    """The pixel stage of sdm_generate_image on caller memory (sd_planar_rgb_to_u8): [3, H, W] floats -> [H, W, 3] bytes."""
    x = np.ascontiguousarray(chw, dtype=np.float32)
    _, h, w = x.shape
    out = np.empty((h, w, 3), np.uint8)
    L = lib()
    L.sd_planar_rgb_to_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.sd_planar_rgb_to_u8.restype = None
    L.sd_planar_rgb_to_u8(x.ctypes.data, w, h, out.ctypes.data)
    return out


def sample_synthetic(family: int, steps: int, n: int, seed: int, method: int = EULER_A, eta: float = float("inf"), image_seq_len: int = 0):
    """The host sampler loop on one image of n floats with a synthetic model (sd_sample_synthetic): returns (latents, aux[steps, 5] = c_skip, c_out, c_in, t, sigma)."""
    L = lib()
    L.sd_sample_synthetic.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_uint64, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
    L.sd_sample_synthetic.restype = C.c_int
    out, aux = np.empty(n, np.float32), np.zeros((steps, 5), np.float32)
    k = L.sd_sample_synthetic(family, steps, image_seq_len, n, seed, method, eta, out.ctypes.data, aux.ctypes.data)
    if k != steps + 1:
        raise EngineError(f"sd_sample_synthetic returned {k}")
    return out, aux


def sample_synthetic2(family: int, steps: int, n: int, seed: int, method: int, scheduler: int = SCHEDULER_DEFAULT, eta: float = float("inf"), image_seq_len: int = 0):
    """sd_sample_synthetic2: any implemented method / scheduler; returns (latents, aux[calls, 5] = c_skip, c_out, c_in, t, sigma per model call in call order)."""
    L = lib()
    L.sd_sample_synthetic2.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_uint64, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int]
    L.sd_sample_synthetic2.restype = C.c_int
    out, aux = np.empty(n, np.float32), np.zeros((2 * steps, 5), np.float32)
    k = L.sd_sample_synthetic2(family, steps, image_seq_len, n, seed, method, scheduler, eta, out.ctypes.data, aux.ctypes.data, 2 * steps)
    if k < 0:   # (0 model calls: DPM++ 2M SDE BT on a one-sigma ladder returns the noise as it is, like the reference)
        raise EngineError(f"sd_sample_synthetic2 returned {k}")
    return out, aux[:k]


def get_sigmas_sched(family: int, scheduler: int, steps: int, image_seq_len: int = 0, shift: float = 0.0) -> np.ndarray:
    """sd_get_sigmas_sched: the ladder of a denoiser family (0 CompVis SD1.x, 1 discrete flow, 2 FLUX flow, 3 CompVis + SDXL AYS table) under sdm_scheduler_t `scheduler`."""
    L = lib()
    L.sd_get_sigmas_sched.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
    L.sd_get_sigmas_sched.restype = C.c_int
    out = np.empty(steps + 2, np.float32)
    k = L.sd_get_sigmas_sched(family, scheduler, steps, image_seq_len, shift, out.ctypes.data)
    if k < 0:
        raise EngineError(f"sd_get_sigmas_sched: scheduler {scheduler} is not implemented")
    return out[:k].copy()


def apg_sequence(cond: np.ndarray, uncond: np.ndarray, scale: float, eta: float, momentum: float, norm_threshold: float, smoothing: float) -> np.ndarray:
    """sd_apg_sequence: adaptive projected guidance over successive calls ([steps, n] arrays) of one image."""
    c, u = _f32(cond), _f32(uncond)
    out = np.empty_like(c)
    L = lib()
    L.sd_apg_sequence.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.sd_apg_sequence.restype = None
    L.sd_apg_sequence(c.ctypes.data, u.ctypes.data, c.shape[1], c.shape[0], scale, eta, momentum, norm_threshold, smoothing, out.ctypes.data)
    return out


def get_sigmas(steps: int) -> np.ndarray:
    out = np.empty(steps + 1, dtype=np.float32)
    lib().sd_get_sigmas(steps, _fptr(out))
    return out


def get_flux_sigmas(steps: int, image_seq_len: int) -> np.ndarray:
    """FluxScheduler sigma ladder, src/runtime/denoiser.hpp:721-782"""
    out = np.empty(steps + 1, dtype=np.float32)
    lib().sd_get_flux_sigmas(steps, image_seq_len, _fptr(out))
    return out


def gen_flux_pe(h: int, w: int, patch_size: int, context_len: int, axes_dim, theta: float = 10000.0) -> np.ndarray:
    """Rope::gen_flux_pe (src/model/common/rope.hpp:424): -> [L, d_head/2, 2, 2] rotation matrices, text tokens first"""
    ax = (C.c_int * len(axes_dim))(*axes_dim)
    half = sum(a // 2 for a in axes_dim)
    L = context_len + ((h + patch_size // 2) // patch_size) * ((w + patch_size // 2) // patch_size)
    out = np.empty((L, half, 2, 2), dtype=np.float32)
    n = lib().sd_gen_flux_pe(h, w, patch_size, context_len, ax, len(axes_dim), theta, _fptr(out))
    assert n == out.size
    return out


def get_flow_sigmas(steps: int, shift: float = 3.0) -> np.ndarray:
    """DiscreteFlowDenoiser sigma ladder (SD3 / SD3.5), src/runtime/denoiser.hpp:1239-1283"""
    out = np.empty(steps + 1, dtype=np.float32)
    lib().sd_get_flow_sigmas(steps, shift, _fptr(out))
    return out
