"""ctypes binding of libdcs.so (include/dcs.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device
is visible when a context is requested, this module raises.
"""
import ctypes
import os
import re
from ctypes import (POINTER, byref, c_char_p, c_double, c_float, c_int, c_int64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCS_LIB") or os.path.join(_HERE, "libdcs.so")  # DCS_LIB: experiment builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "dcs.h")

DCS_OK, DCS_EINVAL, DCS_EUNSUPPORTED, DCS_EHIP, DCS_ENOMEM, DCS_ESHAPE = 0, -1, -2, -3, -4, -5

TAGS = dict(stft=0, conv1=1, conv2=2, fc=3, fc1x=4, deconv2=5, final=6, istft=7, ola=8, tile=9, pool=10, unpool=11,
            mask=12, score=13, decoder=14)


class DcsError(RuntimeError):
    pass


_lib = None


def header_symbols():
    """Every function name include/dcs.h declares."""
    with open(HEADER_PATH, "r") as fh:
        text = fh.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dcs_[a-z0-9_]+)\s*\(", text)))


def load():
    """dlopen libdcs.so (no device needed) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            "deepconvsep_amd: %s is missing -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or deepconvsep_amd/csrc/build.sh. "
            "There is no CPU fallback." % LIB_PATH)
    # torch bundles its own libamdhip64.so.7 / libhsa-runtime64 (same sonames as /opt/rocm's).  Whichever
    # copy is mapped first serves the whole process, and a process that mixes the two runtimes loses the
    # device ("no ROCm-capable device is detected").  Import torch first so that libdcs.so binds to the
    # runtime torch uses -- device pointers and streams are then shared by construction.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64, i32, f32, f64 = c_void_p, c_int64, c_int, c_float, c_double

    def proto(name, res, *args):
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = list(args)

    proto("dcs_version", i32)
    proto("dcs_last_error", c_char_p)
    proto("dcs_create", i32, i32, vp, POINTER(vp))
    proto("dcs_destroy", i32, vp)
    proto("dcs_synchronize", i32, vp)
    proto("dcs_frame_count", i64, i64, i32)
    proto("dcs_inverse_length", i64, i64, i32, i32)
    proto("dcs_tile_count", i64, i64, i32, i32, i32)
    proto("dcs_stft_plan", i32, vp, i32, i32, POINTER(c_double), POINTER(vp))
    proto("dcs_stft_plan_destroy", i32, vp)
    proto("dcs_stft_forward_f32", i32, vp, vp, i64, vp, vp, i64, i64)
    proto("dcs_stft_forward_f64", i32, vp, vp, i64, vp, vp, i64, i64)
    proto("dcs_stft_forward_f32_clips", i32, vp, vp, i64, i64, i64, vp, vp, i64, i64)
    proto("dcs_stft_forward_f64_clips", i32, vp, vp, i64, i64, i64, vp, vp, i64, i64)
    proto("dcs_stft_inverse_f32", i32, vp, vp, i64, vp, i64, i64, i32, f32, vp, i64)
    proto("dcs_stft_inverse_f64", i32, vp, vp, i64, vp, i64, i64, i32, f64, vp, i64)
    proto("dcs_tile", i32, vp, vp, i64, i64, i32, i64, i32, i32, i32, i32, f32, vp, i64)
    proto("dcs_overlap_add", i32, vp, vp, i64, i32, i32, i32, i32, POINTER(c_double), vp, i64, i64)
    proto("dcs_model_create", i32, vp, i32, i32, i32, i32, POINTER(vp), POINTER(i64), i32, POINTER(vp))
    proto("dcs_model_destroy", i32, vp)
    proto("dcs_model_num_sources", i32, vp)
    proto("dcs_model_set_conv_precision", i32, vp, i32)
    proto("dcs_model_set_latency_stages", i32, vp, i32)
    proto("dcs_model_set_score_semantics", i32, vp, i32, i32)
    proto("dcs_lat_pack_b_host", i64, vp, i32, i32, i32, i32, i32, vp, i64)
    proto("dcs_lat_pack_deconv2_host", i64, vp, i32, vp, i64)
    proto("dcs_model_out_channels", i32, vp)
    proto("dcs_model_final_kernel", i32, vp, i64, i64, i32)
    proto("dcs_model_forward_masked", i32, vp, vp, i64, i32, i32, vp)
    proto("dcs_model_forward", i32, vp, vp, i64, i32, vp)
    proto("dcs_separate", i32, vp, vp, vp, i64, i32, i32, f32, i32, i32, vp, POINTER(i64), POINTER(i64))
    proto("dcs_pcm_to_int16", i32, vp, vp, i64, vp)
    proto("dcs_pcm16_to_float", i32, vp, vp, i64, i32, i32, i64, i64, vp, i64)
    proto("dcs_gather", i32, vp, vp, vp, i64, vp, i32)
    proto("dcs_wav_pool_create", i32, i32, POINTER(vp))
    proto("dcs_wav_pool_destroy", None, vp)
    proto("dcs_wav_read_pcm16_async", i32, vp, i32, vp, vp, vp, vp, vp, vp, vp, POINTER(vp))
    proto("dcs_wav_write_pcm16_async", i32, vp, i32, vp, vp, vp, vp, vp, vp, POINTER(vp))
    proto("dcs_wav_batch_done", i32, vp)
    proto("dcs_wav_batch_wait", i32, vp)
    proto("dcs_score_masks", i32, vp, vp, i64, i64, i32, vp, i32, i32, i32, i64, i64, vp, vp)
    proto("dcs_score_masks_norm", i32, vp, vp, i64, i64, i32, vp, i32, i32, i32, i64, i64, i32, vp, vp)
    proto("dcs_separate_stereo", i32, vp, vp, vp, i64, i64, i32, i32, f32, vp, vp, i64, POINTER(i64), POINTER(i64))
    proto("dcs_separate_scoreinformed", i32, vp, vp, vp, i64, POINTER(c_double), i32, i32, i32, i32, f32, i32, i32, vp,
          POINTER(i64), POINTER(i64))
    proto("dcs_separate_batch", i32, vp, vp, vp, i64, i64, i64, i32, i32, f32, i32, i32, vp, POINTER(i64), POINTER(i64))
    proto("dcs_separate_ragged", i32, vp, vp, vp, POINTER(i64), i64, i64, i32, i32, f32, i32, i32, vp, i64, POINTER(i64),
          POINTER(i64))
    proto("dcs_separate_spectra", i32, vp, vp, vp, i64, i32, i32, f32, i32, i32, vp, vp, vp, i64)
    proto("dcs_timing_enable", i32, vp, ctypes.c_uint)
    proto("dcs_timing_stride", i32, vp, i32)
    proto("dcs_timing_reset", i32, vp)
    proto("dcs_timing_query", i32, vp, i32, POINTER(c_double), POINTER(i64))
    proto("dcs_debug_check_guards", i32, vp, POINTER(i64))
    _lib = lib
    return lib


def check(rc):
    """Map a dcs_status to the exception the reference's Python would have raised."""
    if rc == DCS_OK:
        return
    msg = load().dcs_last_error().decode("utf-8", "replace")
    if rc in (DCS_EINVAL, DCS_ESHAPE):
        raise ValueError(msg)
    if rc == DCS_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == DCS_ENOMEM:
        raise MemoryError(msg)
    raise DcsError("libdcs error %d: %s" % (rc, msg))


def frame_count(n_samples, hop):
    return int(load().dcs_frame_count(int(n_samples), int(hop)))


def inverse_length(n_frames, hop, frame):
    return int(load().dcs_inverse_length(int(n_frames), int(hop), int(frame)))


def tile_count(n_frames, tc, ov, tiler):
    return int(load().dcs_tile_count(int(n_frames), int(tc), int(ov), int(tiler)))
