"""ctypes binding of libpvq.so (include/pvq.h).  Fails loudly if the HIP library is missing:
there is no CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PVQ_DEV_LIB=1 selects the developer build of the same sources (libpvq_dev.so, -DPVQ_DEV_KNOBS: it alone reads the PVQ_* environment
# knobs that the tile-shape / fallback tests, the A/B scripts and the phase-stamp tools use).  The product library reads no environment.
# PVQ_DEV_LIB=<name> (anything else) selects lib/libpvq_<name>.so: a developer build of OTHER sources kept beside it for a same-box A/B of two
# code versions (scripts/dev_ab_knob.py PVQ_DEV_LIB 1,<name>); such files are never committed or shipped.
_dev = os.environ.get("PVQ_DEV_LIB", "")
LIB_PATH = os.path.join(_HERE, "lib", "libpvq.so" if not _dev else ("libpvq_dev.so" if _dev == "1" else "libpvq_%s.so" % _dev))

# every symbol include/pvq.h declares
EXPORTS = [
    "pvq_status_string", "pvq_last_error", "pvq_abi_version", "pvq_vqt_default_params", "pvq_vqt_create",
    "pvq_vqt_destroy", "pvq_vqt_get_params", "pvq_vqt_n_bins", "pvq_vqt_delay_seconds", "pvq_vqt_window_union",
    "pvq_vqt_n_groups", "pvq_vqt_group_info", "pvq_vqt_group_csr", "pvq_vqt_filter_params",
    "pvq_vqt_calculate_instant_db", "pvq_vqt_calculate_batch_db", "pvq_vqt_calculate_batch_db_device",
    "pvq_vqt_set_algo", "pvq_vqt_last_algo", "pvq_vqt_resolve_algo", "pvq_analysis_default_params", "pvq_analyze_batch_device",
    "pvq_analyze_batch", "pvq_vqt_analyze_batch_device", "pvq_vqt_calculate_batch_db_streams", "pvq_vqt_analyze_batch_streams", "pvq_plan_shard", "pvq_vqt_analyze_batch_multi", "pvq_vqt_set_profiling", "pvq_vqt_last_kernel_ms",
    "pvq_vqt_kernel_name", "pvq_vqt_last_kernel_launches", "pvq_vqt_last_frames_per_launch", "pvq_vqt_set_gemm_precision", "pvq_vqt_set_workspace_limit", "pvq_vqt_blockdft_columns", "pvq_vqt_set_twiddle_fp16",
    "pvq_analysis_full_default_params", "pvq_analysis_state_create", "pvq_analysis_state_destroy",
    "pvq_analysis_batch_create", "pvq_analysis_batch_destroy", "pvq_analysis_batch_update_vqt_smoothing_duration",
    "pvq_analysis_batch_preprocess_device", "pvq_analysis_batch_preprocess_pcm", "pvq_analysis_batch_get_field", "pvq_analysis_batch_get_scalars",
    "pvq_analysis_state_update_vqt_smoothing_duration", "pvq_analysis_state_preprocess",
    "pvq_analysis_state_bin_to_frequency", "pvq_analysis_state_n_buckets", "pvq_analysis_state_get_field",
    "pvq_analysis_state_get_peaks", "pvq_analysis_state_get_peaks_continuous", "pvq_analysis_state_scene_calmness",
    "pvq_analysis_state_tuning_grid_inaccuracy",
    "pvq_mono_agc_create", "pvq_mono_agc_destroy", "pvq_mono_agc_freeze_gain", "pvq_mono_agc_is_gain_frozen",
    "pvq_mono_agc_gain", "pvq_mono_agc_process", "pvq_train_chunk_samples", "pvq_train_condition_stream",
    "pvq_train_frames_db", "pvq_train_rows", "pvq_npy_write_f32", "pvq_stream_create", "pvq_stream_destroy",
    "pvq_stream_push", "pvq_stream_gain", "pvq_stream_chunk_size_ms", "pvq_stream_frame_db", "pvq_stream_read",
    "pvq_calculate_color", "pvq_led_frame", "pvq_host_alloc", "pvq_host_free",
    "pvq_vqt_input_status", "pvq_vqt_last_gemm_flop", "pvq_vqt_last_sclk_mhz",
    "pvq_vqt_bandwidths_3db", "pvq_vqt_warning_count", "pvq_vqt_warning",
]

PVQ_OK = 0
PVQ_ERR_ABOVE_NYQUIST = 1
PVQ_ERR_WINDOW_EXCEEDS_NFFT = 2
PVQ_ERR_BAD_LENGTH = 3
PVQ_ERR_INVALID_ARG = 4
PVQ_ERR_NO_DEVICE = 5
PVQ_ERR_DEVICE = 6
PVQ_ERR_UNSUPPORTED = 7
PVQ_ERR_INTERNAL = 8
PVQ_ERR_NONFINITE_INPUT = 9

ALGO_AUTO, ALGO_FFT, ALGO_BLOCKDFT = 0, 1, 2
GEMM_F32, GEMM_BF16X3 = 0, 1


class CParams(C.Structure):
    _fields_ = [
        ("sr", C.c_float),
        ("n_fft", C.c_uint32),
        ("min_freq", C.c_float),
        ("octaves", C.c_uint32),
        ("buckets_per_octave", C.c_uint32),
        ("sparsity_quantile", C.c_float),
        ("quality", C.c_float),
        ("gamma", C.c_float),
    ]


class CAnalysisParams(C.Structure):
    _fields_ = [
        ("peak_min_prominence", C.c_float),
        ("peak_min_height", C.c_float),
        ("bass_min_prominence", C.c_float),
        ("bass_min_height", C.c_float),
        ("highest_bassnote", C.c_uint32),
        ("harmonic_threshold", C.c_float),
    ]


class CAnalysisBatchOutputs(C.Structure):   # pvq_analysis_batch_outputs (device pointers)
    _fields_ = [(n, C.c_void_p) for n in ("x_vqt_smoothed", "x_vqt_peakfiltered", "x_vqt_afterglow", "calmness", "pitch_accuracy",
                                          "pitch_deviation", "peak_mask", "peak_count", "center", "size")] + \
               [("max_peaks", C.c_uint32), ("scene_calmness", C.c_void_p), ("tuning_grid_inaccuracy", C.c_void_p)]


class CShard(C.Structure):   # pvq_shard
    _fields_ = [("first_frame", C.c_uint64), ("n_frames", C.c_uint64), ("sample_begin", C.c_uint64), ("sample_end", C.c_uint64),
                ("n_lead", C.c_uint64)]


class CAnalysisFullParams(C.Structure):
    _fields_ = [
        ("spectrogram_length", C.c_uint32),
        ("peak_min_prominence", C.c_float), ("peak_min_height", C.c_float),
        ("bass_min_prominence", C.c_float), ("bass_min_height", C.c_float),
        ("highest_bassnote", C.c_uint32),
        ("vqt_smoothing_duration_base_ns", C.c_uint64),
        ("vqt_smoothing_calmness_min", C.c_float), ("vqt_smoothing_calmness_max", C.c_float),
        ("note_calmness_smoothing_duration_ns", C.c_uint64),
        ("scene_calmness_smoothing_duration_ns", C.c_uint64),
        ("tuning_inaccuracy_smoothing_duration_ns", C.c_uint64),
        ("harmonic_threshold", C.c_float),
    ]


_lib = None


def load():
    """Load libpvq.so; raise (never fall back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C pitchvis_amd/csrc` (or "
            "`python -c 'import __graft_entry__ as g; g.build()'`).  pitchvis_amd has no CPU fallback."
        )
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.so.1.  Both HIP
    # runtimes in one process do not coexist (whichever initialises second sees no GPU), and the
    # dynamic loader de-duplicates by SONAME, so: let torch's copy load first whenever torch is
    # installed; libpvq then binds to that same runtime.  Without torch the system ROCm runtime
    # is used.  (torch is plumbing here: device memory, streams, torch.distributed.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    fp, up, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p
    L.pvq_status_string.argtypes = [C.c_int]; L.pvq_status_string.restype = C.c_char_p
    L.pvq_last_error.argtypes = []; L.pvq_last_error.restype = C.c_char_p
    L.pvq_abi_version.argtypes = []; L.pvq_abi_version.restype = C.c_uint32
    L.pvq_vqt_default_params.argtypes = [C.POINTER(CParams)]
    L.pvq_vqt_create.argtypes = [C.POINTER(CParams), C.c_int, C.POINTER(vp), fp]; L.pvq_vqt_create.restype = C.c_int
    L.pvq_vqt_destroy.argtypes = [vp]
    L.pvq_vqt_get_params.argtypes = [vp, C.POINTER(CParams)]
    L.pvq_vqt_n_bins.argtypes = [vp]; L.pvq_vqt_n_bins.restype = C.c_uint32
    L.pvq_vqt_delay_seconds.argtypes = [vp]; L.pvq_vqt_delay_seconds.restype = C.c_double
    L.pvq_vqt_window_union.argtypes = [vp]; L.pvq_vqt_window_union.restype = C.c_uint32
    L.pvq_vqt_bandwidths_3db.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]; L.pvq_vqt_bandwidths_3db.restype = C.c_int
    L.pvq_vqt_warning_count.argtypes = [vp]; L.pvq_vqt_warning_count.restype = C.c_uint32
    L.pvq_vqt_warning.argtypes = [vp, C.c_uint32, C.c_char_p, C.c_size_t]; L.pvq_vqt_warning.restype = C.c_int
    L.pvq_vqt_n_groups.argtypes = [vp]; L.pvq_vqt_n_groups.restype = C.c_uint32
    L.pvq_vqt_group_info.argtypes = [vp, C.c_uint32, up]; L.pvq_vqt_group_info.restype = C.c_int
    L.pvq_vqt_group_csr.argtypes = [vp, C.c_uint32, C.c_int, up, up, fp]; L.pvq_vqt_group_csr.restype = C.c_int
    L.pvq_vqt_filter_params.argtypes = [vp, fp, fp, up, up]; L.pvq_vqt_filter_params.restype = C.c_int
    L.pvq_vqt_calculate_instant_db.argtypes = [vp, fp, C.c_size_t, fp]; L.pvq_vqt_calculate_instant_db.restype = C.c_int
    L.pvq_vqt_calculate_batch_db.argtypes = [vp, fp, C.c_size_t, C.c_size_t, C.c_size_t, fp]
    L.pvq_vqt_calculate_batch_db.restype = C.c_int
    L.pvq_vqt_calculate_batch_db_device.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp, vp, vp]
    L.pvq_vqt_calculate_batch_db_device.restype = C.c_int
    L.pvq_vqt_set_algo.argtypes = [vp, C.c_int]; L.pvq_vqt_set_algo.restype = C.c_int
    L.pvq_vqt_last_algo.argtypes = [vp]; L.pvq_vqt_last_algo.restype = C.c_int
    L.pvq_vqt_resolve_algo.argtypes = [vp, C.c_size_t, C.c_size_t]; L.pvq_vqt_resolve_algo.restype = C.c_int
    L.pvq_vqt_blockdft_columns.argtypes = [vp]; L.pvq_vqt_blockdft_columns.restype = C.c_uint32
    L.pvq_vqt_set_gemm_precision.argtypes = [vp, C.c_int]; L.pvq_vqt_set_gemm_precision.restype = C.c_int
    L.pvq_vqt_set_workspace_limit.argtypes = [vp, C.c_uint64]; L.pvq_vqt_set_workspace_limit.restype = C.c_int
    L.pvq_analysis_default_params.argtypes = [C.POINTER(CAnalysisParams)]
    L.pvq_analyze_batch_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(CAnalysisParams), vp, vp, vp, vp, C.c_uint32, vp]
    L.pvq_analyze_batch_device.restype = C.c_int
    L.pvq_analyze_batch.argtypes = [vp, fp, C.c_size_t, C.POINTER(CAnalysisParams), up, up, fp, fp, C.c_uint32]
    L.pvq_analyze_batch.restype = C.c_int
    L.pvq_vqt_analyze_batch_device.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(CAnalysisParams),
                                               vp, vp, vp, vp, vp, C.c_uint32, vp]
    L.pvq_vqt_analyze_batch_device.restype = C.c_int
    szp = C.POINTER(C.c_size_t)
    L.pvq_vqt_calculate_batch_db_streams.argtypes = [vp, C.POINTER(vp), szp, szp, C.c_uint32, C.c_size_t, vp, C.c_size_t, vp]
    L.pvq_vqt_calculate_batch_db_streams.restype = C.c_int
    L.pvq_vqt_analyze_batch_streams.argtypes = [vp, C.POINTER(vp), szp, szp, C.c_uint32, C.c_size_t, C.POINTER(CAnalysisParams), vp, C.c_size_t,
                                                vp, vp, vp, vp, C.c_uint32, vp]
    L.pvq_vqt_analyze_batch_streams.restype = C.c_int
    L.pvq_plan_shard.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(CShard)]; L.pvq_plan_shard.restype = C.c_int
    L.pvq_vqt_analyze_batch_multi.argtypes = [C.POINTER(vp), C.c_uint32, fp, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(CAnalysisParams),
                                              fp, up, up, fp, fp, C.c_uint32]
    L.pvq_vqt_analyze_batch_multi.restype = C.c_int
    L.pvq_vqt_set_profiling.argtypes = [vp, C.c_int]; L.pvq_vqt_set_profiling.restype = C.c_int
    L.pvq_vqt_last_kernel_ms.argtypes = [vp, fp, C.c_uint32]; L.pvq_vqt_last_kernel_ms.restype = C.c_uint32
    L.pvq_vqt_last_kernel_launches.argtypes = [vp, up, C.c_uint32]; L.pvq_vqt_last_kernel_launches.restype = C.c_uint32
    L.pvq_vqt_last_frames_per_launch.argtypes = [vp]; L.pvq_vqt_last_frames_per_launch.restype = C.c_uint32
    L.pvq_vqt_kernel_name.argtypes = [C.c_uint32]; L.pvq_vqt_kernel_name.restype = C.c_char_p
    afp = C.POINTER(CAnalysisFullParams)
    L.pvq_analysis_full_default_params.argtypes = [afp]
    L.pvq_analysis_state_create.argtypes = [C.c_float, C.c_uint32, C.c_uint32, afp, C.POINTER(vp)]
    L.pvq_analysis_state_create.restype = C.c_int
    L.pvq_analysis_batch_create.argtypes = [C.c_int, C.c_float, C.c_uint32, C.c_uint32, afp, C.c_uint32, C.POINTER(vp)]; L.pvq_analysis_batch_create.restype = C.c_int
    L.pvq_analysis_batch_destroy.argtypes = [vp]
    L.pvq_analysis_batch_update_vqt_smoothing_duration.argtypes = [vp, C.c_int, C.c_uint64]; L.pvq_analysis_batch_update_vqt_smoothing_duration.restype = C.c_int
    L.pvq_analysis_batch_preprocess_device.argtypes = [vp, vp, C.c_size_t, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(CAnalysisBatchOutputs), vp]
    L.pvq_analysis_batch_preprocess_device.restype = C.c_int
    L.pvq_analysis_batch_preprocess_pcm.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t, C.c_uint64, vp,
                                                    C.POINTER(CAnalysisBatchOutputs), vp]
    L.pvq_analysis_batch_preprocess_pcm.restype = C.c_int
    L.pvq_analysis_batch_get_field.argtypes = [vp, C.c_uint32, C.c_int, fp]; L.pvq_analysis_batch_get_field.restype = C.c_int
    L.pvq_analysis_batch_get_scalars.argtypes = [vp, C.c_uint32, fp, fp]; L.pvq_analysis_batch_get_scalars.restype = C.c_int
    L.pvq_analysis_state_destroy.argtypes = [vp]
    L.pvq_analysis_state_update_vqt_smoothing_duration.argtypes = [vp, C.c_int, C.c_uint64]
    L.pvq_analysis_state_update_vqt_smoothing_duration.restype = C.c_int
    L.pvq_analysis_state_preprocess.argtypes = [vp, fp, C.c_size_t, C.c_uint64]; L.pvq_analysis_state_preprocess.restype = C.c_int
    L.pvq_analysis_state_bin_to_frequency.argtypes = [vp, C.c_uint32]; L.pvq_analysis_state_bin_to_frequency.restype = C.c_float
    L.pvq_analysis_state_n_buckets.argtypes = [vp]; L.pvq_analysis_state_n_buckets.restype = C.c_uint32
    L.pvq_analysis_state_get_field.argtypes = [vp, C.c_int, fp]; L.pvq_analysis_state_get_field.restype = C.c_int
    L.pvq_analysis_state_get_peaks.argtypes = [vp, up, C.c_uint32]; L.pvq_analysis_state_get_peaks.restype = C.c_uint32
    L.pvq_analysis_state_get_peaks_continuous.argtypes = [vp, fp, fp, C.c_uint32]
    L.pvq_analysis_state_get_peaks_continuous.restype = C.c_uint32
    L.pvq_analysis_state_scene_calmness.argtypes = [vp]; L.pvq_analysis_state_scene_calmness.restype = C.c_float
    L.pvq_analysis_state_tuning_grid_inaccuracy.argtypes = [vp]; L.pvq_analysis_state_tuning_grid_inaccuracy.restype = C.c_float
    ip, bp = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    L.pvq_vqt_set_twiddle_fp16.argtypes = [vp, C.c_int]; L.pvq_vqt_set_twiddle_fp16.restype = C.c_int
    L.pvq_mono_agc_create.argtypes = [C.c_float, C.c_float, C.POINTER(vp)]; L.pvq_mono_agc_create.restype = C.c_int
    L.pvq_mono_agc_destroy.argtypes = [vp]
    L.pvq_mono_agc_freeze_gain.argtypes = [vp, C.c_int]
    L.pvq_mono_agc_is_gain_frozen.argtypes = [vp]; L.pvq_mono_agc_is_gain_frozen.restype = C.c_int
    L.pvq_mono_agc_gain.argtypes = [vp]; L.pvq_mono_agc_gain.restype = C.c_float
    L.pvq_mono_agc_process.argtypes = [vp, fp, C.c_size_t]
    L.pvq_train_chunk_samples.argtypes = [vp]; L.pvq_train_chunk_samples.restype = C.c_size_t
    L.pvq_train_condition_stream.argtypes = [vp, fp, fp, C.c_size_t, C.c_size_t, fp, fp]
    L.pvq_train_condition_stream.restype = C.c_int
    L.pvq_train_frames_db.argtypes = [vp, fp, C.c_size_t, C.c_size_t, C.c_size_t, fp]; L.pvq_train_frames_db.restype = C.c_int
    L.pvq_train_rows.argtypes = [fp, C.c_size_t, C.c_uint32, up, ip, fp, fp, fp, fp]; L.pvq_train_rows.restype = C.c_int
    L.pvq_npy_write_f32.argtypes = [C.c_char_p, fp, C.c_uint64]; L.pvq_npy_write_f32.restype = C.c_int
    L.pvq_stream_create.argtypes = [vp, C.c_size_t, C.c_int, C.POINTER(vp)]; L.pvq_stream_create.restype = C.c_int
    L.pvq_stream_destroy.argtypes = [vp]
    L.pvq_stream_push.argtypes = [vp, fp, C.c_size_t]; L.pvq_stream_push.restype = C.c_int
    L.pvq_stream_gain.argtypes = [vp]; L.pvq_stream_gain.restype = C.c_float
    L.pvq_stream_chunk_size_ms.argtypes = [vp]; L.pvq_stream_chunk_size_ms.restype = C.c_float
    L.pvq_stream_frame_db.argtypes = [vp, fp]; L.pvq_stream_frame_db.restype = C.c_int
    L.pvq_stream_read.argtypes = [vp, fp, C.c_size_t]; L.pvq_stream_read.restype = C.c_int
    L.pvq_calculate_color.argtypes = [C.c_uint16, C.c_float, fp, C.c_float, C.c_float, fp]
    L.pvq_led_frame.argtypes = [C.c_uint32, C.c_uint16, fp, fp, C.c_uint32, fp, C.c_float, C.c_float, bp]
    L.pvq_led_frame.restype = C.c_size_t
    L.pvq_host_alloc.argtypes = [C.c_size_t]; L.pvq_host_alloc.restype = C.c_void_p
    L.pvq_host_free.argtypes = [C.c_void_p]
    L.pvq_vqt_input_status.argtypes = [vp, vp]; L.pvq_vqt_input_status.restype = C.c_int
    L.pvq_vqt_last_gemm_flop.argtypes = [vp]; L.pvq_vqt_last_gemm_flop.restype = C.c_double
    L.pvq_vqt_last_sclk_mhz.argtypes = [vp]; L.pvq_vqt_last_sclk_mhz.restype = C.c_float
    _lib = L
    return L
