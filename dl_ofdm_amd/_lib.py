"""ctypes binding of libdccn.so (include/dccn.h).

The library is the product: every op of this package calls it, and there is NO CPU or
PyTorch fallback -- a missing library or a missing GPU raises immediately.
"""
from __future__ import annotations

import ctypes as C
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCCN_LIB_PATH") or os.path.join(_HERE, "lib", "libdccn.so")     # override: kernel experiments


class DccnError(RuntimeError):
    pass


class Metrics(C.Structure):
    """dccn_metrics"""
    _fields_ = [("ce_sum", c_double), ("conf", c_longlong * 4), ("count", c_longlong),
                ("ce_mean", c_float), ("berlin", c_float), ("log_ber", c_float), ("reserved", c_float)]


class AdamState(C.Structure):
    """dccn_adam_state"""
    _fields_ = [("global_step", c_float), ("beta1_power", c_float), ("beta2_power", c_float), ("alpha", c_float)]


class AdamHParams(C.Structure):
    """dccn_adam_hparams (ofdmreceiver_np.py:185-189 defaults)"""
    _fields_ = [("lr0", c_float), ("decay_steps", c_float), ("decay_rate", c_float),
                ("beta1", c_float), ("beta2", c_float), ("eps", c_float)]

    @classmethod
    def default(cls, lr0: float = 1e-3):
        return cls(lr0, 500.0, 0.98, 0.9, 0.999, 1e-8)


class RxShape(C.Structure):
    """dccn_rx_shape"""
    _fields_ = [("batch", c_int), ("S", c_int), ("kin", c_int), ("F", c_int), ("D", c_int), ("nbits", c_int)]


class RxBuffers(C.Structure):
    """dccn_rx_buffers"""
    _fields_ = [("x", c_void_p), ("bits", c_void_p), ("params", c_void_p), ("grads", c_void_p),
                ("adam_m", c_void_p), ("adam_v", c_void_p), ("reg_coef", c_void_p), ("adam", c_void_p),
                ("x_norm", c_void_p), ("fft_out", c_void_p), ("z", c_void_p), ("prob", c_void_p),
                ("dz", c_void_p), ("dfft", c_void_p), ("metrics", c_void_p), ("tx_power", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t),
                ("x_next", c_void_p), ("x_prenormalised", c_int), ("x_norm_next", c_void_p), ("norm_slot", c_int),
                ("keep_dense_grad", c_int), ("reg_uniform_dense", c_int), ("x_next_ready", c_void_p),
                ("gen_next", c_void_p), ("tuning", c_void_p)]


class GenStatic(C.Structure):
    """dccn_gen_static"""
    _fields_ = [("bits_out", c_void_p), ("cell_map", c_void_p), ("const_tab", c_void_p), ("pilot_re", c_float),
                ("pilot_im", c_float), ("idft", c_void_p), ("coeff", c_void_p), ("alpha", c_void_p), ("n_taps", c_int),
                ("L", c_int), ("identity", c_int), ("snr_db", c_void_p), ("y", c_void_p), ("noise", c_void_p),
                ("power_partial", c_void_p), ("noise_partial", c_void_p), ("noise_power_out", c_void_p), ("tx_out", c_void_p),
                ("frames", c_int), ("S", c_int), ("K", c_int), ("CP", c_int), ("D", c_int), ("nbits", c_int),
                ("seed", C.c_ulonglong), ("offset", C.c_uint),
                ("n_profiles", c_int), ("tap_stride", c_int), ("profiles", c_void_p), ("H_out", c_void_p), ("h_rep", c_int)]


class GenProfile(C.Structure):
    """dccn_gen_profile"""
    _fields_ = [("coeff", c_void_p), ("alpha", c_void_p), ("n_taps", c_int), ("L", c_int), ("identity", c_int),
                ("reserved", c_int)]


class EqShape(C.Structure):
    """dccn_eq_shape"""
    _fields_ = [("batch", c_int), ("S", c_int), ("K", c_int), ("CP", c_int), ("cp", c_int), ("F", c_int),
                ("D", c_int), ("nbits", c_int), ("pilot_size", c_int), ("P", c_int)]


class EqBuffers(C.Structure):
    """dccn_eq_buffers"""
    _fields_ = [("x", c_void_p), ("bits", c_void_p), ("eq_params", c_void_p), ("eq_grads", c_void_p),
                ("adam_m", c_void_p), ("adam_v", c_void_p), ("reg_coef", c_void_p), ("adam", c_void_p),
                ("rx_params", c_void_p), ("out_eq", c_void_p), ("chest", c_void_p), ("snr_db", c_void_p),
                ("pilot_carriers", c_void_p), ("prob", c_void_p), ("metrics", c_void_p), ("tx_power", c_void_p),
                ("workspace", c_void_p), ("workspace_bytes", c_size_t), ("reg_uniform", c_int), ("rx_folded", c_void_p),
                ("x_next", c_void_p), ("x_prenormalised", c_int), ("norm_slot", c_int), ("x_next_virtual", c_void_p),
                ("tuning", c_void_p), ("gen_next_rides", c_int), ("monitor", c_void_p)]


class EqMonitor(C.Structure):
    """dccn_eq_monitor"""
    _fields_ = [("chest", c_void_p), ("chan", c_void_p), ("chan_per_symbol", c_int), ("B", c_int), ("S", c_int), ("K", c_int),
                ("metrics", c_void_p), ("tx_power", c_void_p), ("noise_power", c_void_p), ("acc5", c_void_p),
                ("rms_out", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


class ChannelGroup(C.Structure):
    """dccn_channel_group"""
    _fields_ = [("frames", c_void_p), ("n_frames", c_int), ("coeff", c_void_p), ("alpha", c_void_p),
                ("n_taps", c_int), ("L", c_int), ("identity", c_int), ("Fd", c_float)]


METRICS_BYTES = C.sizeof(Metrics)
ADAM_STATE_BYTES = C.sizeof(AdamState)

_vp, _i, _ll, _f, _sz = c_void_p, c_int, c_longlong, c_float, c_size_t

# name -> (restype, argtypes); must list every symbol include/dccn.h declares
SIGNATURES = {
    "dccn_strerror": (c_char_p, [_i]),
    "dccn_version": (_i, []),
    "dccn_build_id": (c_char_p, []),
    "dccn_last_hip_error": (_i, []),
    "dccn_device_info": (_i, [POINTER(c_int), POINTER(c_int), POINTER(c_size_t), c_char_p, _i]),
    "dccn_batch_moment_norm_workspace_size": (_sz, [_i, _i]),
    "dccn_batch_moment_norm_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _sz, _vp]),
    "dccn_clip_power_workspace_size": (_sz, [_ll]),
    "dccn_clip_power": (_i, [_vp, _vp, _vp, _ll, _f, _vp, _sz, _vp]),
    "dccn_cconv_gemm_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "dccn_cconv_gemm_bwd_w_workspace_size": (_sz, [_i, _i, _i]),
    "dccn_cconv_gemm_bwd_w": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "dccn_cconv_gemm_bwd_x": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "dccn_dense_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "dccn_dense_bwd_x": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "dccn_dense_bwd_w_workspace_size": (_sz, [_i, _i, _i]),
    "dccn_dense_bwd_w": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "dccn_dense_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "dccn_dense_bwd_slabs": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _sz, POINTER(c_int), _vp]),
    "dccn_tail_param_count": (_i, [_i]),
    "dccn_demod_tail_workspace_size": (_sz, [_ll, _i]),
    "dccn_demod_tail_loss_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _ll, _i, _vp, _sz, _vp]),
    "dccn_demod_tail_loss_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp, _sz, _vp]),
    "dccn_dense_tail_workspace_size": (_sz, [_i, _i, _i]),
    "dccn_dense_tail_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "dccn_dense_tail_fwd_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "dccn_cconv_im2col": (_i, [_vp, _vp] + [_i] * 14 + [_vp]),
    "dccn_eq_bottleneck_supported": (_i, [_i] * 3),
    "dccn_eq_bottleneck_workspace_size": (C.c_size_t, [_i] * 3),
    "dccn_eq_bottleneck_fwd": (_i, [_vp] * 7 + [_i] * 3 + [_vp]),
    "dccn_eq_bottleneck_bwd": (_i, [_vp] * 11 + [_i] * 3 + [_vp, C.c_size_t, _vp]),
    "dccn_cconv_patch_supported": (_i, [_i] * 9),
    "dccn_cconv_patch_fwd": (_i, [_vp, _vp, _vp, _vp] + [_i] * 15 + [_vp]),
    "dccn_cconv_col2im": (_i, [_vp, _vp] + [_i] * 14 + [_vp]),
    "dccn_rx_gen_next_supported": (_i, [POINTER(RxShape)]),
    "dccn_step_monitor_add": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "dccn_tuning_count": (_i, []),
    "dccn_tuning_snapshot": (_i, [_vp, _i]),
    "dccn_chain_group_max": (_i, []),
    "dccn_eq_group_supported": (_i, [POINTER(EqShape)]),
    "dccn_eq_train_step_grouped": (_i, [_i, _vp, _vp, AdamHParams, _vp]),
    "dccn_gen_static_frames_grouped": (_i, [_i, _vp, _vp]),
    "dccn_gen_static_apply_grouped": (_i, [_i, _vp, _vp, _vp, _vp]),
    "dccn_eq_monitor_accumulate_grouped": (_i, [_i, _vp, _vp]),
    "dccn_cconv_patch_bwd_supported": (_i, [_i] * 11),
    "dccn_cconv1d_bwd_supported": (_i, [_i] * 7),
    "dccn_cconv1d_bwd_workspace_size": (_sz, [_i]),
    "dccn_cconv1d_bwd": (_i, [_vp] * 6 + [_i] * 9 + [_vp, _sz, _vp]),
    "dccn_cconv_patch_bwd_w_workspace_size": (C.c_size_t, [_i] * 7),
    "dccn_cconv_patch_bwd_w": (_i, [_vp, _vp, _vp, _vp] + [_i] * 15 + [_vp, C.c_size_t, _vp]),
    "dccn_cconv_patch_bwd_x_workspace_size": (C.c_size_t, [_i] * 4),
    "dccn_cconv_patch_bwd_x": (_i, [_vp, _vp, _vp] + [_i] * 15 + [_vp, C.c_size_t, _vp]),
    "dccn_metrics_table_add": (_i, [_vp, _vp, _vp]),
    "dccn_metrics_table_set": (_i, [_vp, _vp, _vp]),
    "dccn_ingraph_awgn_workspace_size": (_sz, [_i, _i]),
    "dccn_ingraph_awgn": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, C.c_ulonglong, C.c_uint, _vp, _sz, _vp]),
    "dccn_classical_workspace_size": (_sz, []),
    "dccn_dense_fwd_ld": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "dccn_classical_pilot_ls": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _vp]),
    "dccn_classical_gain": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _vp, _vp, _sz, _vp]),
    "dccn_classical_estimate": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _sz, _vp]),
    "dccn_classical_detect": (_i, [_vp] * 8 + [_i] * 7 + [_vp, _sz, _vp]),
    "dccn_set_tuning": (_i, [_i, _i]),
    "dccn_get_tuning": (_i, [_i]),
    "dccn_dense_tail_supported": (_i, [_i, _i, _i, _i]),
    "dccn_rx_bwd_fused_supported": (_i, [POINTER(RxShape)]),
    "dccn_rx_dense_tail_fused": (_i, [POINTER(RxShape), _i]),
    "dccn_rx_norm_rides_backward": (_i, [POINTER(RxShape)]),
    "dccn_rx_backward_workspace_size": (_sz, [_i, _i, _i, _i, _i]),
    "dccn_rx_backward": (_i, [_vp] * 9 + [_i] * 6 + [_vp, _sz, _vp]),
    "dccn_adam_tf_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, AdamHParams, _ll, _vp]),
    "dccn_rx_param_offsets": (_i, [POINTER(RxShape), POINTER(c_longlong)]),
    "dccn_rx_workspace_size": (_sz, [POINTER(RxShape), _i]),
    "dccn_rx_eval_step": (_i, [POINTER(RxShape), POINTER(RxBuffers), _vp]),
    "dccn_rx_train_step": (_i, [POINTER(RxShape), POINTER(RxBuffers), AdamHParams, _vp]),
    "dccn_rx_normalise": (_i, [POINTER(RxShape), POINTER(RxBuffers), _vp]),
    "dccn_rx_graph_create": (_i, [POINTER(RxShape), POINTER(RxBuffers), _i, AdamHParams, _vp, POINTER(c_void_p)]),
    "dccn_rx_graph_launch": (_i, [_vp, _vp]),
    "dccn_rx_graph_destroy": (_i, [_vp]),
    "dccn_timer_create": (_i, [POINTER(c_void_p)]),
    "dccn_timer_start": (_i, [_vp, _vp]),
    "dccn_timer_stop": (_i, [_vp, _vp]),
    "dccn_timer_elapsed_ms": (_i, [_vp, POINTER(c_float)]),
    "dccn_timer_destroy": (_i, [_vp]),
    "dccn_stream_synchronize": (_i, [_vp]),
    "dccn_eq_rx_folded_floats": (_sz, [POINTER(EqShape)]),
    "dccn_eq_norm_rides": (_i, [POINTER(EqShape)]),
    "dccn_eq_rx_fold": (_i, [POINTER(EqShape), _vp, _vp, _vp]),
    "dccn_eq_monitor_workspace_size": (_sz, [_i, _i, _i]),
    "dccn_eq_monitor_accumulate": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "dccn_step_trace_bytes": (_sz, [_i]),
    "dccn_step_trace_enable": (_i, [_vp, _sz, _i]),
    "dccn_step_trace_steps": (_ll, []),
    "dccn_step_trace_geometry": (None, [POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    # equaliser stage
    "dccn_layer_norm_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "dccn_layer_norm_bwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "dccn_tanh_fwd": (_i, [_vp, _vp, _ll, _vp]),
    "dccn_tanh_bwd": (_i, [_vp, _vp, _vp, _ll, _vp]),
    "dccn_equalize_fwd": (_i, [_vp, _vp, _vp, _vp, _ll, _vp]),
    "dccn_equalize_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp]),
    "dccn_pilot_snr": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "dccn_cconv2d_same_expand": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "dccn_cconv2d_same_reduce": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "dccn_eq_param_offsets": (_i, [POINTER(EqShape), POINTER(c_longlong)]),
    "dccn_eq_workspace_size": (_sz, [POINTER(EqShape), _i]),
    "dccn_gen_static_supported": (_i, [_i, _i, _i]),
    "dccn_gen_static_partials": (_i, [_i]),
    "dccn_gen_static_frames": (_i, [POINTER(GenStatic), _vp]),
    "dccn_gen_static_apply": (_i, [POINTER(GenStatic), _vp, _vp, _vp]),
    "dccn_eq_workspace_tensor": (_i, [POINTER(EqShape), _i, C.c_char_p, POINTER(C.c_size_t), POINTER(C.c_size_t)]),
    "dccn_eq_eval_step": (_i, [POINTER(EqShape), POINTER(EqBuffers), _vp]),
    "dccn_eq_train_step": (_i, [POINTER(EqShape), POINTER(EqBuffers), AdamHParams, _vp]),
    "dccn_eq_graph_create": (_i, [POINTER(EqShape), POINTER(EqBuffers), _i, AdamHParams, _vp, POINTER(c_void_p)]),
    # device-side input generator
    "dccn_philox_fill": (_i, [_vp, _ll, C.c_uint, C.c_uint, C.c_ulonglong, _vp]),
    "dccn_ofdm_tx_frames": (_i, [_vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, C.c_ulonglong,
                                 C.c_uint, _vp]),
    "dccn_channel_awgn_workspace_size": (_sz, [_i, _i, _i]),
    "dccn_channel_doppler_awgn_workspace_size": (_sz, [_i, _i, _i, _i]),
    "dccn_channel_doppler_awgn": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i,
                                       C.c_ulonglong, C.c_uint, _vp, _sz, _vp]),
    "dccn_channel_groups_awgn_workspace_size": (_sz, [_i, _i, _i]),
    "dccn_channel_groups_awgn": (_i, [_vp, POINTER(ChannelGroup), _i, _vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _i, _vp,
                                      _i, C.c_ulonglong, C.c_uint, _vp, _sz, _vp]),
    "dccn_crc32c": (C.c_uint32, [C.c_uint32, _vp, _sz]),
    "dccn_channel_awgn": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, C.c_ulonglong,
                               C.c_uint, _vp, _sz, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """dlopen libdccn.so (built in-tree by ``__graft_entry__.build()`` / ``make -C dl_ofdm_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DccnError(
            "libdccn.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C dl_ofdm_amd/csrc`; this package has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)      # AttributeError if the symbol is missing
        except AttributeError:
            # kernel experiments against an OLDER build of the library (tools/build_rev.sh): tolerate entry points it lacks
            if os.environ.get("DCCN_LIB_PATH") and os.environ.get("DCCN_LIB_ALLOW_MISSING"):
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    # kernel experiments: DCCN_TUNE="key=value,key=value" applies dccn_set_tuning pairs at load time
    for kv in filter(None, os.environ.get("DCCN_TUNE", "").split(",")):
        k, v = kv.split("=")
        if lib.dccn_set_tuning(int(k), int(v)) != 0:
            raise DccnError("DCCN_TUNE: bad tuning pair %r" % kv)
    return lib


def check(status: int, what: str = "dccn call") -> None:
    if status != 0:
        lib = load()
        msg = lib.dccn_strerror(status).decode()
        raise DccnError("%s failed: %s (status %d, hipError %d)" % (what, msg, status, lib.dccn_last_hip_error()))


def device_info():
    """(cu_count, wavefront, hbm_bytes, arch) of the current HIP device; raises without a GPU."""
    lib = load()
    cu, wf, hbm = c_int(0), c_int(0), c_size_t(0)
    arch = C.create_string_buffer(64)
    check(lib.dccn_device_info(C.byref(cu), C.byref(wf), C.byref(hbm), arch, 64), "dccn_device_info")
    return cu.value, wf.value, hbm.value, arch.value.decode()
