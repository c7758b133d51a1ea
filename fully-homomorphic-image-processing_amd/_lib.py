"""ctypes binding of libfhe_hip.so (the C ABI declared in include/fhe_hip.h).

There is no fallback: if the shared library is missing, or a compute entry point is called
without a HIP device, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FHE_HIP_LIB") or os.path.join(_HERE, "libfhe_hip.so")      # FHE_HIP_LIB: another build of the same library (compiler-flag experiments)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "fhe_hip.h")
HEADER_PATHS = [HEADER_PATH, os.path.join(os.path.dirname(_HERE), "include", "fhe_circuits.h"), os.path.join(os.path.dirname(_HERE), "include", "fhe_stream.h")]

FHE_OK = 0
ABI_VERSION = 4      # FHE_ABI_VERSION of the include/fhe_hip.h this table was written against


class FheError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libfhe_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None

_vp, _u32, _u64, _i, _sz, _dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_size_t, C.c_double

# name -> (restype, argtypes); status-returning functions have restype int and are checked
SIGNATURES = {
    "fhe_last_error": (C.c_char_p, []),
    "fhe_abi_version": (_u32, []),
    "fhe_ctx_create": (_i, [_u32, C.POINTER(_u64), _u32, _u64, _i, C.POINTER(_vp)]),
    "fhe_ctx_destroy": (_i, [_vp]),
    "fhe_ctx_has_ctct_tables": (_i, [_vp]),
    "fhe_ctx_device": (_i, [_vp]),
    "fhe_ctx_bind_thread": (_i, [_vp]),
    "fhe_stream_create": (_i, [C.POINTER(_vp)]),
    "fhe_stream_destroy": (_i, [_vp]),
    "fhe_ctx_n": (_u32, [_vp]),
    "fhe_ctx_k": (_u32, [_vp]),
    "fhe_ctx_t": (_u64, [_vp]),
    "fhe_ctx_q": (_u64, [_vp, _u32]),
    "fhe_default_coeff_modulus": (_i, [_u32, _i, C.POINTER(_u64)]),
    "fhe_dev_alloc": (_i, [_sz, C.POINTER(_vp)]),
    "fhe_dev_free": (_i, [_vp]),
    "fhe_host_alloc": (_i, [_sz, C.POINTER(_vp)]),
    "fhe_host_free": (_i, [_vp]),
    "fhe_upload": (_i, [_vp, _vp, _sz, _vp]),
    "fhe_download": (_i, [_vp, _vp, _sz, _vp]),
    "fhe_copy": (_i, [_vp, _vp, _sz, _vp]),
    "fhe_stream_sync": (_i, [_vp]),
    "fhe_gather": (_i, [_vp, _u64, _u64, _vp, _u64, _vp]),
    "fhe_frac_encode": (_i, [_u32, _u64, _dbl, _i, _i, _vp]),
    "fhe_frac_decode": (_dbl, [_u32, _u64, _vp, _i, _i]),
    "fhe_add": (_i, [_vp, _vp, _vp, _vp, _u64, _vp]),
    "fhe_sub": (_i, [_vp, _vp, _vp, _vp, _u64, _vp]),
    "fhe_negate": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "fhe_add_sizes": (_i, [_vp, _vp, _u32, _vp, _u32, _vp, _u64, _i, _vp]),
    "fhe_plain_ntt_words": (_sz, [_vp]),
    "fhe_plain_prepare": (_i, [_vp, _vp, _u32, _vp, _vp]),
    "fhe_plain_ntt_mul": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "fhe_multiply_plain": (_i, [_vp, _vp, _vp, _u64, _vp, _vp]),
    "fhe_multiply_plain_sparse": (_i, [_vp, _vp, _vp, _u64, _vp, _u32, _vp]),
    "fhe_cubic_coeffs": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u64, _vp]),
    "fhe_cubic_combine": (_i, [_vp, _vp, _vp, _vp, _u32, _vp, _u32, _vp, _u64, _vp]),
    "fhe_add_plain": (_i, [_vp, _vp, _u64, _u64, _vp, _u32, _i, _vp]),
    "fhe_ntt_forward": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "fhe_ntt_inverse": (_i, [_vp, _vp, _vp, _u64, _vp]),
    "fhe_dyadic_multiply": (_i, [_vp, _vp, _vp, _vp, _u64, _vp]),
    "fhe_multiply_scratch_bytes": (_sz, [_vp, _u32, _u32, _u64]),
    "fhe_multiply": (_i, [_vp, _vp, _u32, _vp, _u32, _vp, _u64, _vp, _sz, _vp]),
    "fhe_square": (_i, [_vp, _vp, _u32, _vp, _u64, _vp, _sz, _vp]),
    "fhe_multiply_operand_words": (C.c_size_t, [_vp, _u32, _u64]),
    "fhe_multiply_prepare": (_i, [_vp, _vp, _u32, _u64, _vp, _vp]),
    "fhe_multiply_prepared": (_i, [_vp, _vp, _vp, _u32, _vp, _vp, _u32, _vp, _u64, _vp, C.c_size_t, _vp]),
    "fhe_multiply_prepared_shared": (_i, [_vp, _vp, _vp, _u32, _vp, _u32, _u64, _u64, _u64, _vp, _u64, _vp, C.c_size_t, _vp]),
    "fhe_evk_digits": (_u32, [_vp, _u32]),
    "fhe_relinearize": (_i, [_vp, _vp, _u64, _u64, _vp, _u32, _vp, _sz, _vp]),
    "fhe_relinearize_scratch_bytes": (_sz, [_vp, _u32, _u64]),
    "fhe_relinearize_to": (_i, [_vp, _vp, _u64, _vp, _u64, _u64, _vp, _u32, _vp, _sz, _vp]),
    "fhe_relinearize_poly": (_i, [_vp, _vp, _u64, _u32, _vp, _u64, _u64, _vp, _u32, _vp, _sz, _vp]),
    "fhe_relinearize_n": (_i, [_vp, _vp, _u32, _u64, _vp, _u64, _u64, _vp, _u32, _vp, _sz, _vp]),
    "fhe_evk_words": (_sz, [_vp, _u32]),
    "fhe_relinearize_n_scratch_bytes": (_sz, [_vp, _u32, _u32, _u64]),
    "fhe_dct_plan_create": (_i, [_vp, _vp, _i, _i, _vp, C.POINTER(_vp)]),
    "fhe_dct_plan_destroy": (_i, [_vp]),
    "fhe_dct8x8_scratch_bytes": (_sz, [_vp, _u64]),
    "fhe_dct_path": (_i, [_vp]),
    "fhe_arith_path": (_i, [_vp]),
    "fhe_dct8x8_quant": (_i, [_vp, _vp, _vp, _vp, _u64, _vp, _sz, _vp]),
    "fhe_rgb_to_ycc": (_i, [_vp, _vp, _vp, _vp, _u64, _i, _i, _vp]),
    "fhe_rgb_to_ycc_blocks": (_i, [_vp, _vp, _u64, _i, _i, _vp]),
    "fhe_fill_random": (_i, [_vp, _vp, _u64, _u64, _u64, _vp]),
    "fhe_digest": (_i, [_vp, _vp, _u64, _u64, _vp, _vp]),
    "fhe_count_unreduced": (_i, [_vp, _vp, _u64, _vp, _vp]),
    "fhe_noise_cdt": (None, [_vp]),
    "fhe_frac_encode_batch": (_i, [_vp, _vp, _u64, _i, _i, _vp, _vp]),
    "fhe_encrypt_scratch_bytes": (_sz, [_vp, _u64]),
    "fhe_encrypt_batch": (_i, [_vp, _vp, _vp, _u64, C.c_char_p, _u64, _vp, _vp, _sz, _vp]),
    "fhe_encrypt_draws": (_i, [_vp, C.c_char_p, _u64, _u64, _vp, _vp]),
    "fhe_ctx_modulus_bits": (_u32, [_vp]),
    "fhe_decrypt_scratch_bytes": (_sz, [_vp, _u32, _u64]),
    "fhe_decrypt_batch": (_i, [_vp, _vp, _vp, _u32, _u64, _vp, _vp, _vp, _sz, _vp]),
    # include/fhe_circuits.h
    "fhe_circuits_create": (_i, [_vp, _i, _i, C.POINTER(_vp)]),
    "fhe_circuits_destroy": (_i, [_vp]),
    "fhe_circuits_create_relin": (_i, [_vp, _i, _i, _vp, _u32, C.POINTER(_vp)]),
    "fhe_circuits_relin_dbc": (_u32, [_vp]),
    "fhe_circuits_create_relin_at": (_i, [_vp, _i, _i, _vp, _u32, _u32, C.POINTER(_vp)]),
    "fhe_circuits_relin_placement": (_u32, [_vp]),
    "fhe_circuits_out_size": (_u32, [_vp, _i, _u32]),
    "fhe_resize_sample_plan": (_i, [_u32, _u32, _u32, _u32, _i, _vp, _vp, _vp]),
    "fhe_cubic_scratch_bytes": (_sz, [_vp, _u32, _u64]),
    "fhe_cubic": (_i, [_vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _u64, _vp, _sz, _vp]),
    "fhe_linear_scratch_bytes": (_sz, [_vp, _u32, _u64]),
    "fhe_linear": (_i, [_vp, _vp, _vp, _u32, _vp, _vp, _u64, _vp, _sz, _vp]),
    "fhe_sample_bicubic_scratch_bytes": (_sz, [_vp, _u64]),
    "fhe_sample_bicubic": (_i, [_vp, _vp, _u64, _vp, _vp, _vp, _vp, _u64, _vp, _sz, _vp]),
    "fhe_sample_linear_scratch_bytes": (_sz, [_vp, _u64]),
    "fhe_sample_linear": (_i, [_vp, _vp, _u64, _vp, _vp, _vp, _vp, _u64, _vp, _sz, _vp]),
    "fhe_resize_bicubic_shared_scratch_bytes": (_sz, [_vp, _u32, _u32, _u32, _u32, _u32, _u32, _i]),
    "fhe_resize_bicubic_shared": (_i, [_vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _sz, _vp]),
    "fhe_resize_source_rows": (_i, [_u32, _u32, _u32, _u32, _i, C.POINTER(_u32), C.POINTER(_u32)]),
    "fhe_resize_bicubic_shared_rows_scratch_bytes": (_sz, [_vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _i]),
    "fhe_resize_bicubic_shared_rows": (_i, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _sz, _vp]),
    "fhe_homomorphic_sincos_scratch_bytes": (_sz, [_vp, _u64]),
    "fhe_homomorphic_sincos": (_i, [_vp, _i, _vp, _vp, _vp, _u64, _vp, _sz, _vp]),
    "fhe_approximated_step_out_size": (_u32, [_i]),
    "fhe_approximated_step_scratch_bytes": (_sz, [_vp, _i, _u32]),
    "fhe_approximated_step": (_i, [_vp, _vp, _vp, _vp, _i, _i, _dbl, _u32, _u32, _vp, _vp, _vp, _sz, _vp]),
    "fhe_approximated_step_range_scratch_bytes": (_sz, [_vp, _i, _u32, _u32, _u32]),
    "fhe_approximated_step_range": (_i, [_vp, _vp, _vp, _vp, _i, _i, _dbl, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _sz, _vp]),
    "fhe_decode_channel_range_scratch_bytes": (_sz, [_vp, _i, _u32, _u32, _u32, _u32]),
    "fhe_decode_channel_range": (_i, [_vp, _vp, _u32, _vp, _vp, _vp, _i, _i, _dbl, _u32, _u32, _u32, _u32, _vp, _vp, _sz, _vp]),
    "fhe_decode_channel_scratch_bytes": (_sz, [_vp, _i, _u32, _u32]),
    "fhe_decode_channel": (_i, [_vp, _vp, _u32, _vp, _vp, _vp, _i, _i, _dbl, _u32, _u32, _vp, _vp, _sz, _vp]),
    # include/fhe_stream.h
    "fhe_io_record_bytes": (_sz, [_u32, _u32, _u32]),
    "fhe_io_read_records": (_i, [_i, _u64, _u64, _u32, _u32, _u32, _vp, _u32]),
    "fhe_io_write_records": (_i, [_i, _u64, _u64, _u32, _u32, _u32, _vp, _u32]),
    "fhe_io_open": (_i, [C.c_char_p, _i, _u64, C.POINTER(_vp)]),
    "fhe_io_close": (_i, [_vp]),
    "fhe_io_size": (_u64, [_vp]),
    "fhe_io_transfer": (_i, [_vp, _u64, _u64, _u32, _u32, _u32, _vp, _u32]),
}
# entry points whose int return value is a count (>= 0) or an error (< 0)
_COUNT_RETURN = {"fhe_default_coeff_modulus", "fhe_frac_encode", "fhe_dct_path", "fhe_arith_path", "fhe_ctx_device", "fhe_ctx_has_ctct_tables"}


def load():
    """Load libfhe_hip.so and bind every symbol of include/fhe_hip.h.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libfhe_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)" % LIB_PATH
        )
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = L.fhe_abi_version()
    if got != ABI_VERSION:
        raise ImportError("%s reports ABI version %d, this binding was written against %d: rebuild the library (make -C csrc)" % (LIB_PATH, got, ABI_VERSION))
    _lib = L
    return L


def call(name, *args):
    """Call a status-returning entry point; raise FheError on failure."""
    L = load()
    rc = getattr(L, name)(*args)
    if rc < 0:
        raise FheError(rc, L.fhe_last_error().decode("utf-8", "replace"))
    return rc
