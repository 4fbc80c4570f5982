"""ctypes binding of libspmx.so (include/spmx.h).  No compute lives here.

The library is built in-tree by ``sentencepiece_amd/csrc/Makefile`` (hipcc,
gfx950).  If it is missing this module raises: the product has no CPU path.
"""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libspmx.so")

# every symbol include/spmx.h declares: (name, restype, argtypes)
_H = C.c_void_p
_U64 = C.c_uint64
SYMBOLS = [
    ("spmx_create", C.c_int, [C.c_void_p, _U64, C.c_int, C.POINTER(_H)]),
    ("spmx_create_from_file", C.c_int, [C.c_char_p, C.c_int, C.POINTER(_H)]),
    ("spmx_destroy", None, [_H]),
    ("spmx_last_error", C.c_char_p, [_H]),
    ("spmx_set_encode_extra_options", C.c_int, [_H, C.c_char_p]),
    ("spmx_set_decode_extra_options", C.c_int, [_H, C.c_char_p]),
    ("spmx_set_vocabulary", C.c_int, [_H, C.POINTER(C.c_char_p), C.POINTER(_U64), _U64]),
    ("spmx_reset_vocabulary", C.c_int, [_H]),
    ("spmx_piece_size", C.c_int, [_H]),
    ("spmx_piece_to_id", C.c_int, [_H, C.c_char_p, _U64]),
    ("spmx_id_to_piece", C.c_int64, [_H, C.c_int, C.c_char_p, _U64]),
    ("spmx_unk_id", C.c_int, [_H]),
    ("spmx_piece_type", C.c_int, [_H, C.c_int]),
    ("spmx_bos_id", C.c_int, [_H]),
    ("spmx_eos_id", C.c_int, [_H]),
    ("spmx_pad_id", C.c_int, [_H]),
    ("spmx_model_type", C.c_int, [_H]),
    ("spmx_model_flags", C.c_uint32, [_H]),
    ("spmx_encode_batch_device", C.c_int,
     [_H, C.c_void_p, _U64, C.c_void_p, _U64, C.c_void_p, _U64, C.c_void_p, C.c_void_p, C.POINTER(_U64)]),
    ("spmx_encode_batch", C.c_int,
     [_H, C.c_void_p, C.c_void_p, _U64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("spmx_encode_batch_ex", C.c_int,
     [_H, C.c_void_p, C.c_void_p, _U64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
      C.POINTER(_U64)]),
    ("spmx_encode_batch_device_ex", C.c_int,
     [_H, C.c_void_p, _U64, C.c_void_p, _U64, C.c_void_p, _U64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_U64),
      C.POINTER(_U64)]),
    ("spmx_unk_piece", C.c_int64, [_H, C.c_char_p, _U64]),
    ("spmx_encode_batch_views", C.c_int,
     [_H, C.c_void_p, _U64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(_U64)]),
    ("spmx_free", None, [C.c_void_p]),
    ("spmx_encode", C.c_int, [_H, C.c_char_p, _U64, C.c_void_p, _U64, C.POINTER(_U64)]),
    ("spmx_decode_batch_device", C.c_int,
     [_H, C.c_void_p, C.c_void_p, _U64, C.c_void_p, _U64, C.c_void_p, C.c_void_p, C.POINTER(_U64)]),
    ("spmx_decode_batch", C.c_int,
     [_H, C.c_void_p, C.c_void_p, _U64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("spmx_decode", C.c_int, [_H, C.c_void_p, _U64, C.c_void_p, _U64, C.POINTER(_U64)]),
    ("spmx_encode_batch_spans_device", C.c_int,
     [_H, C.c_void_p, _U64, C.c_void_p, _U64, C.c_void_p, _U64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
      C.c_void_p, C.c_void_p, C.POINTER(_U64)]),
    ("spmx_encode_batch_spans", C.c_int,
     [_H, C.c_void_p, C.c_void_p, _U64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
      C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("spmx_encode_batch_spans_ex", C.c_int,
     [_H, C.c_void_p, C.c_void_p, _U64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
      C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(_U64)]),
    ("spmx_status_message", C.c_char_p, [C.c_int]),
    ("spmx_normalize_batch_device", C.c_int,
     [_H, C.c_void_p, C.c_void_p, _U64, C.c_void_p, _U64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_U64)]),
    ("spmx_normalize_batch", C.c_int,
     [_H, C.c_void_p, C.c_void_p, _U64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("spmx_nbest_encode_batch", C.c_int,
     [_H, C.c_void_p, C.c_void_p, _U64, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
      C.POINTER(C.c_void_p)]),
    ("spmx_nbest_encode_batch_spans", C.c_int,
     [_H, C.c_void_p, C.c_void_p, _U64, C.c_int] + [C.POINTER(C.c_void_p)] * 8),
    ("spmx_sample_encode_batch_spans", C.c_int,
     [_H, C.c_void_p, C.c_void_p, _U64, C.c_int, C.c_float, _U64] + [C.POINTER(C.c_void_p)] * 6),
    ("spmx_encode_batch_multi", C.c_int,
     [C.POINTER(_H), C.c_int, C.c_void_p, C.c_void_p, _U64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
      C.POINTER(C.c_void_p), C.POINTER(_U64)]),
    ("spmx_all_gather_ids", C.c_int,
     [C.c_void_p, C.c_int, C.c_int, C.c_void_p, _U64, C.c_void_p, _U64, C.c_void_p, _U64, C.c_void_p, _U64, C.c_void_p,
      C.c_void_p, C.c_void_p, C.c_void_p]),
    ("spmx_gather_scratch_words", _U64, [C.c_int]),
    ("spmx_decode_batch_pieces", C.c_int,
     [_H, C.c_void_p, C.c_void_p, _U64, C.c_void_p, C.c_void_p, _U64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("spmx_decode_unk_option", C.c_int, [_H]),
    ("spmx_piece_score", C.c_int, [_H, C.c_int, C.POINTER(C.c_float)]),
    ("spmx_serialized_model", C.c_int, [_H, C.POINTER(C.c_char_p), C.POINTER(_U64)]),
    ("spmx_rccl_unique_id", C.c_int, [C.c_void_p]),
    ("spmx_rccl_comm_init", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]),
    ("spmx_rccl_comm_destroy", C.c_int, [C.c_void_p]),
    ("spmx_gather_last_error", C.c_char_p, []),
    ("spmx_encode_file", C.c_int, [_H, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(_U64), C.POINTER(_U64)]),
    ("spmx_sample_encode_batch", C.c_int,
     [_H, C.c_void_p, C.c_void_p, _U64, C.c_int, C.c_float, _U64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("spmx_encode_batch_original", C.c_int, [_H, C.c_void_p, C.c_void_p, _U64, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("spmx_split_lines_device", C.c_int,
     [_H, C.c_void_p, _U64, C.c_void_p, _U64, C.c_void_p, _U64, C.c_void_p, C.POINTER(_U64), C.POINTER(_U64)]),
    ("spmx_set_profiling", C.c_int, [_H, C.c_int]),
    ("spmx_last_profile_name", C.c_int, [_H, C.c_int, C.c_char_p, _U64]),
    ("spmx_last_phase_cycles", C.c_int, [_H, C.c_void_p]),
    ("spmx_handle_info", C.c_int, [_H, C.c_void_p, C.c_void_p]),
    ("spmx_last_profile", C.c_int,
     [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
]

_lib = None


def bind(path):
    """ctypes handle of a library exporting include/spmx.h, with every prototype set.  Besides lib() below the only
    caller is the test suite, which binds tests/emu/libspmx_emu.so (the same api.cc over a CPU model of the
    wavefront) explicitly; the product never does."""
    l = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(l, name)   # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    return l


def lib():
    """Loads libspmx.so once.  Raises if it has not been built."""
    global _lib
    if _lib is None:
        path = os.environ.get("SPMX_LIB") or LIB_PATH       # SPMX_LIB: an A/B build of the same library (csrc/Makefile `variants`)
        if not os.path.exists(path):
            raise RuntimeError(
                "%s is missing: build it with `make -C sentencepiece_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback" % path)
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 / HSA runtime under torch/lib, libspmx
        # resolves /opt/rocm's.  If libspmx initialises HIP first, a later torch.cuda init finds "No HIP GPUs"
        # (seen on the MI355X box); with torch's libraries loaded first both share them.  The device-resident forms
        # take torch tensors anyway, so torch is imported first when it is installed.
        if "torch" not in sys.modules and not os.environ.get("SPMX_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        _lib = bind(path)
    return _lib
