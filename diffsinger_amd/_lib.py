"""ctypes binding of libdsdenoise.so (include/dsd.h).  There is NO fallback: if the library is missing or
does not load, importing the engine fails loudly - the product path never routes through a CPU/oracle
implementation."""
from __future__ import annotations

import ctypes as C
import os

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libdsdenoise.so')

DSD_ABI_VERSION = 6

# every symbol include/dsd.h declares (tests check the .so exports exactly these)
SYMBOLS = [
    'dsd_abi_version', 'dsd_last_error', 'dsd_build_id', 'dsd_create', 'dsd_destroy', 'dsd_load_weights', 'dsd_set_schedule',
    'dsd_get_schedule_table', 'dsd_set_spec_range', 'dsd_prepare', 'dsd_denoise', 'dsd_q_sample',
    'dsd_sample_ddpm', 'dsd_p_sample', 'dsd_sample_plms', 'dsd_norm_spec', 'dsd_denorm_spec',
    'dsd_set_use_graph', 'dsd_set_layer_tile', 'dsd_time_layer_kernel', 'dsd_debug_layer_timeline', 'dsd_device_bytes',
    'dsd_get_layer_tile', 'dsd_set_loop_mode', 'dsd_get_loop_mode', 'dsd_loop_timeouts', 'dsd_debug_loop_timeline', 'dsd_set_noise_seed', 'dsd_philox_normal',
    'dsd_set_split_mode', 'dsd_get_split_mode', 'dsd_debug_layer', 'dsd_loop_launches', 'dsd_p_sample_ex', 'dsd_set_lat_split', 'dsd_get_lat_split', 'dsd_set_conv_mode', 'dsd_get_conv_mode',
    'dsd_check', 'dsd_loop_parked', 'dsd_debug_hold_cus', 'dsd_debug_condproj_groups',
]
# every symbol include/dsf.h declares (the FastSpeech2 conditioner ops)
SYMBOLS_FS2 = ['dsf_padded_frames', 'dsf_packed_floats', 'dsf_pack_weight', 'dsf_conv1d', 'dsf_layer_norm', 'dsf_attention',
               'dsf_ln_bwd_workspace_floats', 'dsf_layer_norm_bwd', 'dsf_attention_bwd_workspace_floats', 'dsf_attention_bwd',
               'dsf_linear_rows_workspace_floats', 'dsf_linear_rows', 'dsf_linear_rows_bwd',
               'dsf_to_channel_major', 'dsf_from_channel_major', 'dsf_positions', 'dsf_input_cm', 'dsf_gather_frames', 'dsf_sum_embed', 'dsf_token_masks', 'dsf_pitch_coarse', 'dsf_q_sample_rows', 'dsf_l1_workspace_floats', 'dsf_l1_mean', 'dsf_l1_mean_bwd', 'dsf_p_sample', 'dsf_denorm_spec',
               'dsf_conv1d_dilated', 'dsf_set_conv_split', 'dsf_wgrad_workspace_floats', 'dsf_conv1d_wgrad', 'dsf_bias_grad',
               'dsf_train_add_step', 'dsf_train_rowsum', 'dsf_train_gate', 'dsf_train_gate_bwd', 'dsf_train_res_skip', 'dsf_train_res_skip_bwd',
               'dsf_channel_affine', 'dsf_group_norm', 'dsf_adamw_step',
               'dsf_stack_workspace_floats', 'dsf_set_stack_mode', 'dsf_set_stack_conv', 'dsf_get_stack_conv', 'dsf_set_wgrad_dual', 'dsf_debug_trb_timeline', 'dsf_stack_offsets', 'dsf_stack_forward', 'dsf_stack_backward', 'dsf_wgrad2_workspace_floats', 'dsf_conv1d_wgrad2', 'dsf_wgrad_probe', 'dsf_wgrad_probe_read']

# every symbol include/dsv.h declares (the HiFi-GAN / NSF-HiFi-GAN generator ops)
SYMBOLS_VOC = ['dsv_padded_samples', 'dsv_packed_floats', 'dsv_pack_weight', 'dsv_pad_rows', 'dsv_conv1d', 'dsv_conv1d_multi', 'dsv_set_lean', 'dsv_noise_conv', 'dsv_sine_source',
               'dsv_fold_factor', 'dsv_set_fold', 'dsv_conv1d_folded', 'dsv_chain_fold', 'dsv_chain_supported', 'dsv_resblock_chain', 'dsv_resblock_chain_multi', 'dsv_resblock_chain_sum', 'dsv_set_chain_variant', 'dsv_debug_chain_timeline',
               'dsv_pwg_first', 'dsv_pwg_upsample', 'dsv_pwg_layer']

_fp = C.POINTER(C.c_float)
_fpp = C.POINTER(C.c_void_p)


class DsdConfig(C.Structure):
    _fields_ = [('mel_bins', C.c_int32), ('residual_channels', C.c_int32), ('encoder_hidden', C.c_int32),
                ('residual_layers', C.c_int32), ('dilation_cycle_length', C.c_int32)]


class DsdWeights(C.Structure):
    _fields_ = [
        ('input_projection_w', C.c_void_p), ('input_projection_b', C.c_void_p),
        ('mlp0_w', C.c_void_p), ('mlp0_b', C.c_void_p), ('mlp2_w', C.c_void_p), ('mlp2_b', C.c_void_p),
        ('dilated_conv_w', _fpp), ('dilated_conv_b', _fpp),
        ('diffusion_projection_w', _fpp), ('diffusion_projection_b', _fpp),
        ('conditioner_projection_w', _fpp), ('conditioner_projection_b', _fpp),
        ('output_projection_w', _fpp), ('output_projection_b', _fpp),
        ('skip_projection_w', C.c_void_p), ('skip_projection_b', C.c_void_p),
        ('final_projection_w', C.c_void_p), ('final_projection_b', C.c_void_p),
    ]


_lib = None


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load (once) and type the library.  Raises RuntimeError when it is absent - build it with
    `python -m diffsinger_amd.build` (hipcc, gfx950)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(_LIB_PATH):
        raise RuntimeError(f'{_LIB_PATH} not found: the HIP denoiser library is not built '
                           f'(run `python -m diffsinger_amd.build`); there is no CPU fallback')
    # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).  One process must
    # have ONE runtime: import torch first so that our library binds to the runtime torch's tensors live in.
    import torch  # noqa: F401
    lib = C.CDLL(_LIB_PATH)
    h = C.c_void_p
    lib.dsd_abi_version.restype = C.c_int
    lib.dsd_last_error.restype = C.c_char_p
    lib.dsd_build_id.restype = C.c_char_p
    lib.dsd_create.argtypes = [C.POINTER(DsdConfig), C.c_int, C.POINTER(h)]
    lib.dsd_destroy.argtypes = [h]
    lib.dsd_destroy.restype = None
    lib.dsd_load_weights.argtypes = [h, C.POINTER(DsdWeights), C.c_void_p]
    lib.dsd_set_schedule.argtypes = [h, C.POINTER(C.c_double), C.c_int32]
    lib.dsd_get_schedule_table.argtypes = [h, C.c_int32, _fp, C.c_int32]
    lib.dsd_set_spec_range.argtypes = [h, _fp, _fp]
    lib.dsd_prepare.argtypes = [h, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
    lib.dsd_denoise.argtypes = [h, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]
    lib.dsd_q_sample.argtypes = [h, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.dsd_sample_ddpm.argtypes = [h, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.dsd_p_sample.argtypes = [h, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.dsd_sample_plms.argtypes = [h, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.dsd_norm_spec.argtypes = [h, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dsd_denorm_spec.argtypes = [h, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dsd_set_use_graph.argtypes = [h, C.c_int32]
    lib.dsd_set_layer_tile.argtypes = [h, C.c_int32]
    lib.dsd_time_layer_kernel.argtypes = [h, C.c_int32, C.c_int32, C.c_int32, _fp, C.c_void_p]
    lib.dsd_debug_layer_timeline.argtypes = [h, C.c_int32, C.c_int32, C.POINTER(C.c_uint64), C.c_int32, C.POINTER(C.c_int32), C.c_void_p]
    lib.dsd_device_bytes.argtypes = [h]
    lib.dsd_device_bytes.restype = C.c_int64
    lib.dsd_get_layer_tile.argtypes = [h]
    lib.dsd_set_loop_mode.argtypes = [h, C.c_int32]
    lib.dsd_get_loop_mode.argtypes = [h]
    lib.dsd_loop_timeouts.argtypes = [h, C.c_void_p]
    lib.dsd_loop_launches.argtypes = [h]
    lib.dsd_check.argtypes = [h]
    lib.dsd_loop_parked.argtypes = [h]
    lib.dsd_debug_hold_cus.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.dsd_debug_condproj_groups.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
    lib.dsd_set_lat_split.argtypes = [h, C.c_int32]
    lib.dsd_get_lat_split.argtypes = [h]
    lib.dsd_set_conv_mode.argtypes = [h, C.c_int32, C.c_int32]
    lib.dsd_get_conv_mode.argtypes = [h]
    lib.dsd_p_sample_ex.argtypes = [h, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_int32, C.c_void_p]
    lib.dsd_set_split_mode.argtypes = [h, C.c_int32, C.c_void_p]
    lib.dsd_get_split_mode.argtypes = [h]
    lib.dsd_debug_layer.argtypes = [h, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dsd_set_noise_seed.argtypes = [h, C.c_uint64]
    lib.dsd_philox_normal.argtypes = [h, C.c_uint64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
    lib.dsd_debug_loop_timeline.argtypes = [h, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint64), C.c_int32, C.POINTER(C.c_int32), C.c_void_p]
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.dsf_padded_frames.argtypes = [i32]
    lib.dsf_padded_frames.restype = i32
    lib.dsf_packed_floats.argtypes = [i32, i32, i32]
    lib.dsf_packed_floats.restype = i64
    lib.dsf_pack_weight.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.dsf_conv1d.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp, vp, vp]
    lib.dsf_layer_norm.argtypes = [vp, vp, vp, vp, i32, i32, i32, f32, i32, vp, vp]
    lib.dsf_attention.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    lib.dsf_ln_bwd_workspace_floats.argtypes = [i32, i32]
    lib.dsf_ln_bwd_workspace_floats.restype = i64
    lib.dsf_layer_norm_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp]
    lib.dsf_attention_bwd_workspace_floats.argtypes = [i32, i32, i32]
    lib.dsf_attention_bwd_workspace_floats.restype = i64
    lib.dsf_attention_bwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.dsf_set_conv_split.argtypes = [i32]
    lib.dsf_linear_rows_workspace_floats.argtypes = [i32, i32, i32]
    lib.dsf_linear_rows_workspace_floats.restype = i64
    lib.dsf_linear_rows.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.dsf_linear_rows_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.dsf_to_channel_major.argtypes = [vp, i64, i64, i64, vp, i32, i32, i32, vp]
    lib.dsf_from_channel_major.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.dsf_positions.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    lib.dsf_input_cm.argtypes = [vp, vp, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.dsf_gather_frames.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.dsf_sum_embed.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.dsf_token_masks.argtypes = [vp, vp, vp, vp, i64, vp]
    lib.dsf_q_sample_rows.argtypes = [vp, vp, vp, vp, vp, i32, vp, i32, i64, vp]
    lib.dsf_l1_workspace_floats.argtypes = []
    lib.dsf_l1_workspace_floats.restype = i64
    lib.dsf_l1_mean.argtypes = [vp, vp, vp, vp, i64, vp]
    lib.dsf_l1_mean_bwd.argtypes = [vp, vp, vp, vp, i64, vp]
    lib.dsf_pitch_coarse.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, C.c_double, C.c_double, i32, i32, i32, vp]
    lib.dsf_p_sample.argtypes = [vp, vp, vp, i64, f32, f32, f32, f32, f32, vp]
    lib.dsf_denorm_spec.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.dsf_conv1d_dilated.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]
    lib.dsf_wgrad_workspace_floats.argtypes = [i32, i32, i32]
    lib.dsf_wgrad_workspace_floats.restype = i64
    lib.dsf_conv1d_wgrad.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.dsf_bias_grad.argtypes = [vp, vp, i32, i32, i32, i32, vp]
    lib.dsf_train_add_step.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    lib.dsf_train_rowsum.argtypes = [vp, vp, i32, i32, vp]
    lib.dsf_train_gate.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.dsf_train_gate_bwd.argtypes = [vp, vp, vp, i32, i32, i32, vp]
    lib.dsf_train_res_skip.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.dsf_train_res_skip_bwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    lib.dsf_adamw_step.argtypes = [vp, vp, vp, vp, i64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, i64, vp, vp]
    lib.dsf_channel_affine.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.dsf_group_norm.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, vp]
    lib.dsv_padded_samples.argtypes = [i32]
    lib.dsv_padded_samples.restype = i32
    lib.dsv_packed_floats.argtypes = [i32, i32, i32]
    lib.dsv_packed_floats.restype = i64
    lib.dsv_pack_weight.argtypes = [vp, i32, i32, i32, vp, vp]
    lib.dsv_pad_rows.argtypes = [vp, vp, i64, i32, vp]
    lib.dsv_conv1d.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, f32, i32, vp]
    lib.dsv_fold_factor.argtypes = [i32, i32, i32, i32]
    lib.dsv_fold_factor.restype = i32
    lib.dsv_set_fold.argtypes = [i32]
    lib.dsv_conv1d_folded.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, f32, i32, vp]
    lib.dsv_chain_fold.argtypes = [i32]
    lib.dsv_chain_fold.restype = i32
    lib.dsv_chain_supported.argtypes = [i32, i32, i32, vp]
    lib.dsv_chain_supported.restype = i32
    lib.dsv_resblock_chain.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, f32, f32, vp]
    lib.dsv_conv1d_multi.argtypes = [i32, vp, i32, i32, i32, i32, i32, f32, vp]
    lib.dsv_set_lean.argtypes = [i32]
    lib.dsv_resblock_chain_multi.argtypes = [vp, vp, vp, C.POINTER(vp), i32, i32, i32, i32, i32, vp, f32, vp]
    lib.dsv_resblock_chain_sum.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, f32, f32, vp]
    lib.dsv_set_chain_variant.argtypes = [i32, i32, i32]
    lib.dsv_debug_chain_timeline.argtypes = [vp]
    lib.dsv_noise_conv.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.dsv_sine_source.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, f32, f32, vp]
    lib.dsv_pwg_first.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    lib.dsv_pwg_upsample.argtypes = [vp, vp, vp, i64, i32, i32, vp]
    lib.dsv_pwg_layer.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name not in ('dsd_abi_version',):
            fn.restype = C.c_int
    if lib.dsd_abi_version() != DSD_ABI_VERSION:
        raise RuntimeError(f'libdsdenoise ABI {lib.dsd_abi_version()} != binding {DSD_ABI_VERSION}: rebuild')
    _lib = lib
    return lib


def build_id() -> str:
    """sha256 of the sources the loaded library was compiled from (dsd_build_id, include/dsd.h)."""
    return load().dsd_build_id().decode()


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = load().dsd_last_error()
        raise RuntimeError(f'{what or "libdsdenoise"} failed ({rc}): {msg.decode() if msg else "?"}')
