"""ctypes binding of ``libnsr_hip.so`` (the C ABI declared in ``include/nsr_hip.h``).

PyTorch is plumbing here: it owns device memory and the HIP stream; every compute kernel of the hot
path is in the shared library.  There is NO CPU fallback: importing this package without the built
library, or calling an op with a non-GPU tensor, raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NSR_HIP_LIB: developer switch for A/B runs of differently tuned builds (csrc/build.sh <outdir> with NSR_EXTRA_FLAGS)
LIB_PATH = os.environ.get("NSR_HIP_LIB") or os.path.join(_HERE, "libnsr_hip.so")

NSR_MAX_LEVELS = 32
ABI_VERSION = 1


class NsrGridDesc(ctypes.Structure):
    _fields_ = [
        ("n_levels", ctypes.c_uint32),
        ("n_features", ctypes.c_uint32),
        ("log2_hashmap_size", ctypes.c_uint32),
        ("base_resolution", ctypes.c_uint32),
        ("per_level_scale", ctypes.c_float),
        ("n_entries", ctypes.c_uint32),
        ("scale", ctypes.c_float * NSR_MAX_LEVELS),
        ("resolution", ctypes.c_uint32 * NSR_MAX_LEVELS),
        ("size", ctypes.c_uint32 * NSR_MAX_LEVELS),
        ("offset", ctypes.c_uint32 * (NSR_MAX_LEVELS + 1)),
    ]


class NsrMlpDesc(ctypes.Structure):
    _fields_ = [
        ("n_in", ctypes.c_uint32),
        ("in_pad", ctypes.c_uint32),
        ("n_out", ctypes.c_uint32),
        ("out_pad", ctypes.c_uint32),
        ("n_hidden", ctypes.c_uint32),
        ("output_activation", ctypes.c_uint32),
    ]


class NsrTableAdam(ctypes.Structure):
    """include/nsr_hip.h: AdamW applied to the hash table inside the owner-computes table backward"""
    _fields_ = [("params", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("shadow", ctypes.c_void_p), ("step", ctypes.c_void_p), ("hyper", ctypes.c_void_p),
                ("base_lr", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double),
                ("gamma", ctypes.c_double), ("milestone0", ctypes.c_int32), ("milestone1", ctypes.c_int32),
                ("milestone2", ctypes.c_int32), ("eps", ctypes.c_float), ("weight_decay", ctypes.c_float)]


class NsrTableExchange(ctypes.Structure):
    """include/nsr_hip.h: the ray-sharded main pass hands the table gradient to the exchange as bf16, in level groups"""
    _fields_ = [("grad_bf16", ctypes.c_void_p), ("grad_bf16_elems", ctypes.c_uint64), ("n_groups", ctypes.c_uint32),
                ("level_begin", ctypes.c_uint32 * 4), ("level_end", ctypes.c_uint32 * 4),
                ("event_group", ctypes.c_void_p * 4), ("event_small", ctypes.c_void_p)]


class NsrRenderGrads(ctypes.Structure):
    """include/nsr_hip.h: upstream gradients of nsr_nerf_render_backward (NULL = zero)"""
    _fields_ = [("comp_rgb", ctypes.c_void_p), ("opacity", ctypes.c_void_p), ("depth", ctypes.c_void_p),
                ("weights", ctypes.c_void_p)]


class NsrNeusUpstream(ctypes.Structure):
    """include/nsr_hip.h: gradients of a caller-owned loss for the NeuS backward kernels (NULL = zero)"""
    _fields_ = [(k, ctypes.c_void_p) for k in ("comp_rgb_full", "comp_rgb", "opacity", "depth", "weights", "sdf_samples",
                                               "sdf_grad_samples", "sdf_laplace_samples")]


class NsrVanillaLayer(ctypes.Structure):
    """include/nsr_hip.h: the nn.Linear tensors of one VanillaMLP layer (weight_g NULL: plain weight)"""
    _fields_ = [("weight_v", ctypes.c_void_p), ("weight_g", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("grad_v", ctypes.c_void_p), ("grad_g", ctypes.c_void_p), ("grad_bias", ctypes.c_void_p),
                ("n_out", ctypes.c_uint32), ("n_in", ctypes.c_uint32)]


class NsrAdamSegment(ctypes.Structure):
    _fields_ = [("params", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("n", ctypes.c_uint64), ("lr", ctypes.c_float)]


class NsrVmlpDesc(ctypes.Structure):
    _fields_ = [("n_in", ctypes.c_uint32), ("in_pad", ctypes.c_uint32), ("n_out", ctypes.c_uint32),
                ("n_hidden", ctypes.c_uint32), ("activation", ctypes.c_uint32)]


class NsrNerfStepDesc(ctypes.Structure):
    _fields_ = [("grid", NsrGridDesc), ("mlp_density", NsrMlpDesc), ("mlp_color", NsrMlpDesc),
                ("radius", ctypes.c_float), ("contraction", ctypes.c_int), ("density_bias", ctypes.c_float),
                ("early_stop_eps", ctypes.c_float), ("grad_scale", ctypes.c_float), ("loss_scale", ctypes.c_float)]


class NsrNerfPruneLayout(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint64) for k in ("x01", "enc", "out1", "acts1", "total_bytes")]


class NsrNerfMainLayout(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint64) for k in (
        "ray_indices", "t_starts", "t_ends", "weights", "comp_rgb", "opacity", "depth", "loss_acc", "trans", "x01",
        "dirs", "enc", "out1", "acts1", "tex_in", "out2", "acts2", "g_comp", "d_rgb", "d_logit", "d_tex", "d_enc",
        "partials", "grid_ws", "total_bytes")]


_P, _I, _U, _F, _U64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_float, ctypes.c_uint64
_D = ctypes.c_double
_GD, _MD = ctypes.POINTER(NsrGridDesc), ctypes.POINTER(NsrMlpDesc)
_SD = ctypes.POINTER(NsrNerfStepDesc)
_VD = ctypes.POINTER(NsrVmlpDesc)

# name -> argtypes  (restype is int unless listed in _RESTYPES); mirrors include/nsr_hip.h one to one
SIGNATURES = {
    "nsr_last_error": [],
    "nsr_abi_version": [],
    "nsr_hashgrid_make_desc": [_GD, _U, _U, _U, _U, _F],
    "nsr_hashgrid_forward": [_P, _P, _P, _U, _U, _U, _GD, _P],
    "nsr_hashgrid_forward_ex": [_P, _P, _P, _U, _U, _I, _U, _GD, _P, _P],
    "nsr_hashgrid_backward_params": [_P, _P, _I, _U, _P, _U, _U, _F, _GD, _P],
    "nsr_hashgrid_backward_params_workspace_floats": [_GD, _U],
    "nsr_hashgrid_backward_params_owner": [_P, _P, _I, _U, _P, _P, _U, _U, _F, _I, _GD, _P, _P],
    "nsr_hashgrid_backward_params_owner_bin": [_P, _P, _U, _U, _GD, _P, _P],
    "nsr_hashgrid_backward_params_owner_bin_second_order": [_P, _P, _U, _U, _GD, _P, _P],
    "nsr_hashgrid_backward_params_owner_accumulate": [_P, _P, _I, _U, _P, _P, _U, _U, _F, _I, _GD, _P, _P],
    "nsr_hashgrid_backward_input": [_P, _P, _P, _I, _U, _P, _U, _U, _GD, _P],
    "nsr_hashgrid_backward_backward_input": [_P, _P, _P, _I, _U, _P, _P, _U, _P, _P, _U, _U, _GD, _P],
    "nsr_hashgrid_backward_backward_input_ws": [_P, _P, _P, _I, _U, _P, _P, _U, _P, _P, _P, _U, _U, _GD, _P],
    "nsr_hashgrid_forward_variant": [_I, _I],
    "nsr_hashgrid_forward_jac": [_P, _P, _P, _U, _U, _I, _U, _GD, _P, _P, _P],
    "nsr_hashgrid_jac_apply": [_P, _U, _GD, _P, _U, _P, _P, _P, _U, _P, _P],
    "nsr_hashgrid_jac_apply_ex": [_P, _U, _GD, _P, _U, _P, _P, _P, _U, _P, _P, _P],
    "nsr_hashgrid_forward_taps": [_P, _P, _P, _U, _U, _I, _U, _GD, _P, _P],
    "nsr_hashgrid_forward_taps_masks": [_P, _P, _P, _U, _U, _I, _U, _GD, _P, _P],
    "nsr_hashgrid_backward_params_owner_with_second_order": [_P, _P, _P, _U, _P, _P, _P, _U, _U, _I, _I, _GD, _P],
    "nsr_hashgrid_backward_params_owner_with_second_order_adam": [_P, _P, _P, _U, _P, _P, _U, _U, _I, _GD, _P, _P],
    "nsr_hashgrid_backward_params_owner_accumulate_taps_adam": [_P, _P, _P, _P, _U, _U, _GD, _P, _P],
    "nsr_hashgrid_backward_params_owner_with_second_order_bf16": [_P, _P, _P, _U, _P, _P, _P, _U, _U, _I, _GD, _P],
    "nsr_hashgrid_backward_params_owner_accumulate_taps_bf16": [_P, _P, _P, _P, _P, _U, _U, _GD, _P],
    "nsr_sh4_forward": [_P, _P, _U, _U, _P],
    "nsr_mlp_forward": [_P, _I, _U, _P, _P, _P, _U, _MD, _P],
    "nsr_mlp_forward_ex": [_P, _I, _U, _U, _P, _P, _P, _U, _MD, _P, _P],
    "nsr_grid_mlp_supported": [_GD, _MD],
    "nsr_grid_mlp_forward": [_P, _P, _P, _P, _P, _P, _U, _I, _U, _U, _GD, _MD, _P, _P],
    "nsr_grid_mlp_backward_workspace_floats": [_GD, _MD, _U],
    "nsr_grid_mlp_backward": [_P, _I, _U, _P, _P, _P, _U, _I, _P, _P, _P, _P, _P, _U, _U, _F, _GD, _MD, _P],
    "nsr_mlp_backward_workspace_floats": [_MD, _U],
    "nsr_mlp_backward": [_P, _I, _U, _P, _P, _I, _U, _P, _P, _P, _P, _U, _P, _U, _F, _MD, _P],
    "nsr_mlp_backward_ex": [_P, _I, _U, _P, _P, _P, _I, _U, _U, _P, _P, _P, _P, _U, _U, _P, _U, _F, _MD, _P, _P],
    "nsr_mlp_backward_phases": [_P, _I, _U, _P, _P, _P, _I, _U, _U, _P, _P, _P, _P, _U, _U, _P, _U, _F, _MD, _P, _P, _I],
    "nsr_sample_positions_unit": [_P, _P, _P, _P, _P, _F, _I, _P, _P, _U, _P, _P],
    "nsr_visibility_prefix": [_P, _U, _F, _P, _P, _P, _F, _P, _U, _P],
    "nsr_copy_ray_prefix_rows": [_P, _P, _U, _P, _P, _P, _P, _P, _P, _U, _P],
    "nsr_nerf_copy_kept_rows": [_P] * 14 + [_U, _U, _U, _U, _P, _P, _P, _U, _P],
    "nsr_texture_input": [_P, _U, _P, _P, _U, _P, _P],
    "nsr_composite_forward": [_P, _U, _F, _P, _P, _P, _U, _P, _P, _P, _P, _P, _P, _P, _U, _P],
    "nsr_composite_backward": [_P, _U, _F, _P, _P, _P, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U, _P],
    "nsr_composite_backward_smooth_l1": [_P, _U, _F, _P, _P, _P, _U, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _U, _P],
    "nsr_composite_l1_partials_floats": [_U],
    "nsr_composite_forward_smooth_l1": [_P, _U, _F, _P, _P, _P, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U, _P],
    "nsr_composite_backward_smooth_l1_partials": [_P, _U, _F, _P, _P, _P, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _U, _P],
    "nsr_smooth_l1_valid": [_P, _P, _P, _P, _U, _P],
    "nsr_smooth_l1_valid_set": [_P, _P, _P, _P, _U, _P],
    "nsr_smooth_l1_valid_backward": [_P, _P, _P, _P, _F, _P, _U, _P],
    "nsr_gather_train_rays": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _U, _P],
    "nsr_prepare_train_rays": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _F, _P, _P, _P, _P, _P, _P, _P, _U, _P, _P],
    "nsr_update_ray_count": [_P, _P, _I, _I, _P, _P],
    "nsr_ray_aabb_intersect": [_P, _P, _P, _P, _P, _U, _P],
    "nsr_ray_march_count": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _U, _P],
    "nsr_ray_march_write": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P, _U, _P],
    "nsr_grid_bricks_words64": [_I, _I, _I],
    "nsr_grid_pack_bricks": [_P, _I, _I, _I, _P, _P],
    "nsr_ray_march_capacity": [ctypes.POINTER(ctypes.c_float), _F],
    "nsr_ray_march_bricks_count": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P, _U, _U, _P],
    "nsr_ray_march_bricks_write": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P, _U, _P, _P, _P, _U, _P],
    "nsr_occupancy_select_cells": [_P, _I, _I, _I, _P, _P, _P, _U, _I, _U, _P, _P, _P, _P, _P, _P, _P],
    "nsr_occupancy_update": [_P, _U, _F, _F, _F, _F, _P, _P, _P, _P, _P, _U, _U, _P, _P],
    "nsr_distortion_loss_forward": [_P, _P, _P, _P, _P, _U, _P],
    "nsr_distortion_loss_backward": [_P, _P, _P, _P, _P, _U, _P],
    "nsr_pack_from_counts": [_P, _P, _P, _U, _P],
    "nsr_pack_from_counts_capped": [_P, _P, _P, _U, _U, _P, _P, _P],
    "nsr_pack_info": [_P, _P, _U, _U, _P],
    "nsr_contract": [_P, _P, _I, _P, _U, _P],
    "nsr_contract_inv": [_P, _P, _I, _P, _U, _P],
    "nsr_grid_query_u8": [_P, _P, _P, _I, _I, _I, _I, _P, _U, _P],
    "nsr_sample_positions": [_P, _P, _P, _P, _P, _P, _P, _U, _P],
    "nsr_transmittance_from_sigma_forward": [_P, _P, _P, _P, _P, _U, _P],
    "nsr_transmittance_from_sigma_backward": [_P, _P, _P, _P, _P, _P, _U, _P],
    "nsr_transmittance_from_alpha_forward": [_P, _P, _P, _U, _P],
    "nsr_transmittance_from_alpha_backward": [_P, _P, _P, _P, _P, _U, _P],
    "nsr_accumulate_along_rays_forward": [_P, _P, _P, _U, _P, _U, _P],
    "nsr_accumulate_along_rays_backward": [_P, _P, _P, _U, _P, _P, _P, _U, _P],
    "nsr_compact_samples": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _U, _P],
    "nsr_contract_to_unisphere": [_P, _F, _I, _P, _U, _P],
    "nsr_density_activation_forward": [_P, _U, _U, _F, _P, _P, _U, _P],
    "nsr_neus_alpha_forward": [_P, _P, _P, _P, _P, _F, _P, _U, _P],
    "nsr_neus_alpha_backward": [_P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _U, _P],
    "nsr_nerf_prune_layout": [_SD, _U, ctypes.POINTER(NsrNerfPruneLayout)],
    "nsr_nerf_prune_pass": [_SD, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U, _U, _P, _U, _P, _P, _P],
    "nsr_nerf_main_layout": [_SD, _U, _U, ctypes.POINTER(NsrNerfMainLayout)],
    "nsr_nerf_main_pass": [_SD, _P, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U, _U, _I, _P, _P, _P, _P],
    "nsr_nerf_render_forward": [_SD, _P, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U, _U, _I, _P, _P, _P],
    "nsr_nerf_render_backward": [_SD, _P, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U, _U, _P, _P],
    "nsr_nerf_main_pass_exchange": [_SD, _P, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U, _U, _P, _P, _P, _P],
    "nsr_hashgrid_owner_large_from": [_U],
    "nsr_ray_march_rays_per_wave": [_U],
    "nsr_ray_march_wave_mode": [ctypes.c_int],
    "nsr_nerf_helper_stream": [],
    "nsr_nerf_wait_kept_rows": [_P],
    "nsr_nerf_defer_wgrad_join": [ctypes.c_int],
    "nsr_hashgrid_owner_tune": [ctypes.c_int, ctypes.c_float],
    "nsr_hashgrid_backward_params_owner_accumulate_range": [_P, _P, _P, _P, _P, _U, _U, _F, _U, _U, _GD, _P, _P],
    "nsr_hashgrid_backward_params_taps_workspace_floats": [_GD, _U],
    "nsr_hashgrid_backward_params_owner_bin_taps": [_P, _P, _P, _U, _U, _GD, _P],
    "nsr_hashgrid_backward_params_owner_bin_taps_masked": [_P, _P, _P, _U, _U, _GD, _P],
    "nsr_hashgrid_backward_params_owner_accumulate_taps": [_P, _P, _P, _P, _P, _U, _U, _I, _GD, _P],
    "nsr_hashgrid_backward_params_owner_accumulate_adam": [_P, _P, _I, _U, _P, _U, _U, _F, _GD, _P, _P, _P],
    "nsr_mlp_wgrad_max_blocks": [_U],
    "nsr_mlp_dgrad_pair_supported": [_MD, _MD],
    "nsr_mlp_dgrad_pair": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U, _F, _MD, _MD, _P, _P],
    "nsr_visibility_prefix_sums": [_P, _U, _F, _P, _P, _P, _F, _P, _P, _U, _P],
    "nsr_nerf_copy_kept_rows_scan": [_P] * 18 + [_U, _U, _U, _U, _P, _P, _P, _U, _P],
    "nsr_composite_forward_samples": [_P, _U, _F, _P, _P, _P, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U, _U, _P, _P],
    "nsr_composite_backward_samples": [_P, _U, _F, _P, _P, _P, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F,
                                       _P, _P, _U, _U, _P, _P],
    "nsr_nerf_step_variant": [_I, _I],
    "nsr_nerf_wait_before_mlp": [_P, _P],
    "nsr_nerf_prune_pass_deferred": [_SD, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U, _U, _P, _U, _P, _P, _P],
    "nsr_profile_enable": [_I],
    "nsr_profile_collect": [_I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64),
                            ctypes.POINTER(ctypes.c_uint64)],
    "nsr_adamw_step": [_P, _P, _P, _P, _P, _U64, _F, _F, _F, _F, _F, _F, _F, _F, _I, _P, _U64, _P],
    "nsr_scale_to_half": [_P, _P, _U64, _F, _P],
    "nsr_scale_from_half": [_P, _P, _U64, _F, _P],
    "nsr_adam_tick": [_P, _P, _D, _D, _D, _D, _I, _I, _I, _P],
    "nsr_adamw_step_scheduled": [_P, _P, _P, _P, _P, _U64, _U64, _P, _P, _P, _P, _P, _U64, _P, _P, _D, _D, _D, _D, _I, _I,
                                 _I, _F, _F, _F, _I, _P],
    "nsr_adamw_step_scheduled_to": [_P, _P, _P, _P, _P, _U64, _U64, _P, _P, _P, _P, _P, _U64, _P, _P, _P, _P, _D, _D, _D, _D,
                                    _I, _I, _I, _F, _F, _F, _I, _P],
    "nsr_overflow_guard": [_P, _F],
    "nsr_vmlp_blob_floats": [_VD],
    "nsr_vmlp_backward_workspace_floats": [_VD, _U],
    "nsr_vmlp_forward": [_VD, _P, _P, _U, _P, _U, _P, _P, _P, _U, _U, _P, _P],
    "nsr_vmlp_backward": [_VD, _P, _P, _U, _P, _U, _P, _P, _P, _P, _U, _U, _U, _U, _P, _I, _P, _U, _U, _P, _P],
    "nsr_neus_points": [_P, _P, _P, _P, _P, _F, _F, _I, _P, _P, _U, _P, _P],
    "nsr_neus_shade_forward": [_P, _P, _U, _P, _P, _F, _F, _P, _P, _P, _P, _F, _U, _F, _P, _P, _P, _P, _P, _I, _P, _U,
                               _P, _P],
    "nsr_neus_composite_forward": [_P, _P, _P, _I, _P, _P, _P, _P, _U, _P, _P, _P, _P, _P, _P, _P, _U, _P],
    "nsr_neus_loss_rays": [_P, _P, _P, _P, _P, _P, _U, _P, _P],
    "nsr_masked_loss_out_floats": [],
    "nsr_masked_loss_forward": [_P, _P, _P, _U, _U, _I, _F, _P, _P],
    "nsr_masked_loss_backward": [_P, _P, _P, _U, _U, _I, _F, _P, _P, _P, _P],
    "nsr_vmlp_fold": [_VD, _P, _U, _P, _P],
    "nsr_vmlp_unfold_gradient": [_VD, _P, _U, _P, _I, _P],
    "nsr_adamw_multi": [_P, _U, _F, _F, _F, _F, _F, _F, _I, _P, _P, _P],
    "nsr_neus_inv_s": [_P, _P, _P],
    "nsr_neus_occupancy_values": [_P, _P, _F, _P, _U, _P, _P],
    "nsr_occupancy_update_values": [_P, _F, _F, _P, _P, _P, _P, _P, _U, _U, _P, _P],
    "nsr_occupancy_density_values_sphere": [_P, _P, _F, _F, _P, _U, _P, _P],
    "nsr_neus_variance_gradient": [_P, _P, _P, _I, _P],
    "nsr_bg_visibility_prefix": [_P, _F, _P, _P, _P, _F, _P, _U, _P],
    "nsr_bg_texture_input": [_P, _U, _P, _P, _P, _U, _U, _P, _P],
    "nsr_bg_composite_forward": [_P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U, _P],
    "nsr_bg_composite_backward": [_P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _U, _P],
    "nsr_bg_join_gradients": [_P, _P, _U, _U, _P, _U, _P, _P],
    "nsr_neus_composite_backward": [_P, _P, _P, _I, _P, _P, _P, _U, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _U, _P,
                                    _P],
    "nsr_neus_composite_backward_ex": [_P, _P, _P, _I, _P, _P, _P, _U, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _U, _P,
                                       _P, _P, _P, _P],
    "nsr_neus_shade_backward_ex": [_P, _P, _P, _P, _P, _P, _P, _F, _P, _F, _F, _P, _P, _U, _P, _F, _F, _P, _P, _P, _U, _P,
                                   _P, _U, _P, _P, _P],
    "nsr_neus_shade_backward": [_P, _P, _P, _P, _P, _P, _P, _F, _P, _F, _F, _P, _P, _U, _P, _F, _F, _P, _P, _P, _U, _P,
                                _P, _U, _P, _P],
}
_RESTYPES = {"nsr_last_error": ctypes.c_char_p, "nsr_ray_march_rays_per_wave": ctypes.c_uint32, "nsr_nerf_helper_stream": ctypes.c_void_p, "nsr_hashgrid_owner_large_from": ctypes.c_uint32, "nsr_hashgrid_owner_tune": ctypes.c_float, "nsr_mlp_wgrad_max_blocks": ctypes.c_uint32, "nsr_masked_loss_out_floats": ctypes.c_uint32,
             "nsr_composite_l1_partials_floats": ctypes.c_uint64,
             "nsr_grid_mlp_backward_workspace_floats": ctypes.c_uint64, "nsr_mlp_backward_workspace_floats": ctypes.c_uint64,
             "nsr_vmlp_blob_floats": ctypes.c_uint64, "nsr_vmlp_backward_workspace_floats": ctypes.c_uint64,
             "nsr_hashgrid_backward_params_workspace_floats": ctypes.c_uint64,
             "nsr_hashgrid_backward_params_taps_workspace_floats": ctypes.c_uint64,
             "nsr_profile_enable": None, "nsr_grid_bricks_words64": ctypes.c_uint64, "nsr_ray_march_capacity": ctypes.c_uint32}


class NsrError(RuntimeError):
    pass


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `bash instant-nsr-pl_amd/csrc/build.sh` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`).  There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, ctypes.c_int)
    if lib.nsr_abi_version() != ABI_VERSION:
        raise ImportError(f"libnsr_hip.so ABI {lib.nsr_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    return lib


lib = load_library()
# NSR_OWN_TUNE="key=value,key=value": developer switch for A/B runs of the table backward's decomposition
# (nsr_hashgrid_owner_tune; e.g. "0=0" = the round-3 placement of the work units)
for _kv in filter(None, os.environ.get("NSR_OWN_TUNE", "").split(",")):
    lib.nsr_hashgrid_owner_tune(int(_kv.split("=")[0]), float(_kv.split("=")[1]))


def check(rc, what=""):
    if rc != 0:
        raise NsrError(f"{what}: rc={rc}: {lib.nsr_last_error().decode()}")


def stream_ptr():
    """the current HIP stream of the current device as a void* (the raw accessors: ``torch.cuda.current_stream()`` costs the
    host ~8 us per call -- device-index normalisation, an availability check that reads the environment, a Stream object)"""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def device_guard(dev):
    """``torch.cuda.device(dev)`` only when ``dev`` is not already the current device (the context manager costs the host
    ~10 us; a training process runs with its one device current)"""
    idx = dev.index if isinstance(dev, torch.device) else int(dev)
    if idx is None or idx == torch._C._cuda_getDevice():
        return _NO_GUARD
    return torch.cuda.device(idx)


_SHARED_STREAMS = {}


def shared_stream(device, role):
    """ONE torch stream per (device, role) for the whole process ("side": the trainers' marching stream, "helper": the NeuS
    runner's helper stream, "comm": the gradient exchange's).  HIP maps streams to a small pool of hardware queues (4 by
    default) in the order they are first used; every trainer object used to create its own side stream, so the SECOND trainer of
    a process (bench.py: the measured model behind its burn-in model) could land on the hardware queue of the step's own stream
    and serialise with it -- measured as two modes of the same step, 8 % apart, depending on how many trainers a process had
    built before (tools/forms_regime_ab.py).  With process-wide streams the mapping is the same for every trainer."""
    dev = torch.device(device) if not isinstance(device, torch.device) else device
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), role)
    s = _SHARED_STREAMS.get(key)
    if s is None:
        s = _SHARED_STREAMS[key] = torch.cuda.Stream(device=dev)
    return s


def ptr(t):
    """device pointer of a contiguous GPU tensor (None -> NULL)"""
    if t is None:
        return None
    if not t.is_cuda:
        raise NsrError("nsr_hip ops need GPU tensors (there is no CPU path in the product library)")
    if not t.is_contiguous():
        raise NsrError("nsr_hip ops need contiguous tensors")
    return ctypes.c_void_p(t.data_ptr())


def make_grid_desc(n_levels, n_features, log2_hashmap_size, base_resolution, per_level_scale):
    d = NsrGridDesc()
    check(lib.nsr_hashgrid_make_desc(ctypes.byref(d), int(n_levels), int(n_features), int(log2_hashmap_size),
                                     int(base_resolution), float(per_level_scale)), "nsr_hashgrid_make_desc")
    return d


def make_mlp_desc(n_in, n_out, n_hidden, output_activation):
    pad = lambda v: (v + 15) // 16 * 16  # noqa: E731
    act = {"none": 0, "sigmoid": 1}[str(output_activation).lower()]
    return NsrMlpDesc(int(n_in), pad(int(n_in)), int(n_out), pad(int(n_out)), int(n_hidden), act)
