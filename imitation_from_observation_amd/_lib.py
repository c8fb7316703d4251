"""ctypes binding of libctxtrans.so (include/ctxtrans.h).  Loads the in-tree library and fails
loudly when it is missing -- there is no Python/CPU fallback for the compute path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libctxtrans.so")

CTX_OK, CTX_E_INVALID, CTX_E_DEVICE, CTX_E_NOMEM, CTX_E_STATE = 0, -1, -2, -3, -4
CTX_VARIANT_SKIPNEW = 0
CTX_VARIANT_REAL = 1
CTX_VARIANT_INCEPTION2 = 2
CTX_PREC_F32 = 0
CTX_PREC_BF16X3 = 1
CTX_DP_UNIQUE_ID_BYTES = 128


class CtxConfig(ctypes.Structure):
    """ctx_config of include/ctxtrans.h (ABI 2).  Trailing fields default to zero = the reference's defaults."""
    _fields_ = [(n, ctypes.c_int32) for n in ("variant", "H", "W", "C", "df_dim", "featsize", "max_batch", "precision")] + \
               [("strides", ctypes.c_int32 * 4), ("kernels", ctypes.c_int32 * 4), ("filters", ctypes.c_int32 * 4),
                ("keep_prob", ctypes.c_float), ("loss_terms", ctypes.c_int32)]


CTX_LOSS_RECON1, CTX_LOSS_RECON2, CTX_LOSS_SIM = 1, 2, 4
# ablations_code/ablations.py:175-182: ablation_type -> terms of `loss`
LOSS_ABLATIONS = {"None": 7, "L2": 3, "L2L3": 1, "L1": 6}


class CtxProfEntry(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 56), ("kernel", ctypes.c_char * 40), ("flops", ctypes.c_double),
                ("ms", ctypes.c_float), ("useful_frac", ctypes.c_float)]


class CtxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libctxtrans error {code}: {msg}")
        self.code = code


_c = ctypes
_P = _c.c_void_p
_F = _c.POINTER(_c.c_float)
_U8 = _c.POINTER(_c.c_uint8)
_CFG = _c.POINTER(CtxConfig)

# name -> (restype, argtypes): every symbol include/ctxtrans.h declares
class CnnBuf(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("h", "w", "c")]


class CnnOp(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("kind", "src", "dst", "dst_ch0", "kh", "kw", "stride", "same", "cout", "lane")] + \
               [("w_off", ctypes.c_int64), ("b_off", ctypes.c_int64)] + \
               [(n, ctypes.c_int32) for n in ("src_ch0", "src_c", "nsplit", "dst2", "dst2_ch0", "reserved")]


CTX_CNN_CONV, CTX_CNN_MAXPOOL, CTX_CNN_AVGPOOL = 0, 1, 2
BUCKET_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64)

SIGNATURES = {
    "ctx_abi_version": (_c.c_int, []),
    "ctx_create": (_c.c_int, [_CFG, _c.c_int, _c.POINTER(_P)]),
    "ctx_create_ex": (_c.c_int, [_CFG, _c.c_int, _P, _P, _c.POINTER(_P)]),
    "ctx_destroy": (None, [_P]),
    "ctx_last_error": (_c.c_char_p, [_P]),
    "ctx_param_total_for": (_c.c_int64, [_CFG]),
    "ctx_arena_bytes": (_c.c_int64, [_CFG]),
    "ctx_param_total": (_c.c_int64, [_P]),
    "ctx_param_count": (_c.c_int, [_P]),
    "ctx_param_info": (_c.c_int, [_P, _c.c_int, _c.POINTER(_c.c_char_p), _c.POINTER(_c.c_int),
                                  _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64)]),
    "ctx_set_params": (_c.c_int, [_P, _F, _c.c_size_t]),
    "ctx_get_params": (_c.c_int, [_P, _F, _c.c_size_t]),
    "ctx_get_grads": (_c.c_int, [_P, _F, _c.c_size_t]),
    "ctx_set_adam_state": (_c.c_int, [_P, _F, _F, _c.c_size_t, _c.c_int64]),
    "ctx_get_adam_state": (_c.c_int, [_P, _F, _F, _c.c_size_t, _c.POINTER(_c.c_int64)]),
    "ctx_init_params": (_c.c_int, [_P, _c.c_uint64]),
    "ctx_translate": (_c.c_int, [_P, _U8, _U8, _c.c_int, _c.c_int, _F, _F]),
    "ctx_encode": (_c.c_int, [_P, _U8, _c.c_int, _F, _F]),
    "ctx_translate_f32": (_c.c_int, [_P, _F, _F, _c.c_int, _c.c_int, _F, _F]),
    "ctx_encode_f32": (_c.c_int, [_P, _F, _c.c_int, _F]),
    "ctx_translate_dev": (_c.c_int, [_P, _P, _P, _c.c_int, _c.c_int, _F, _F]),
    "ctx_encode_dev": (_c.c_int, [_P, _P, _c.c_int, _F]),
    "ctx_reward_set_cache": (_c.c_int, [_P, _c.c_int, _F, _F, _c.c_int]),
    "ctx_reward_costs": (_c.c_int, [_P, _c.c_int, _U8, _c.c_int, _c.c_float, _c.c_int, _F]),
    "ctx_train_step": (_c.c_int, [_P, _F, _F, _F, _c.c_int, _c.c_float, _F]),
    "ctx_set_dropout_seed": (_c.c_int, [_P, _c.c_uint64]),
    "ctx_train_step_u8": (_c.c_int, [_P, _U8, _U8, _U8, _c.c_int, _c.c_float, _F]),
    "ctx_demos_upload": (_c.c_int, [_P, _U8, _c.c_int, _c.c_int]),
    "ctx_train_step_sampled": (_c.c_int, [_P, _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32), _c.c_int, _c.c_float, _F]),
    "ctx_eval_sampled": (_c.c_int, [_P, _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32), _c.c_int, _F, _F, _F]),
    "ctx_last_outputs": (_c.c_int, [_P, _F, _F, _F]),
    "ctx_eval": (_c.c_int, [_P, _F, _F, _F, _c.c_int, _F, _F, _F]),
    "ctx_dev_forward_backward": (_c.c_int, [_P, _P, _P, _P, _c.c_int, _c.c_int]),
    "ctx_dev_frames": (_c.c_int, [_P, _c.c_int, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_void_p)]),
    "ctx_dev_forward": (_c.c_int, [_P, _P, _P, _P, _c.c_int]),
    "ctx_dev_train_step": (_c.c_int, [_P, _P, _P, _P, _c.c_int, _c.c_float]),
    "ctx_set_grad_bucket_callback": (_c.c_int, [_P, _P, _P]),
    "ctx_dev_adam": (_c.c_int, [_P, _c.c_float]),
    "ctx_dev_scalars": (_c.c_int, [_P, _F]),
    "ctx_dev_params": (_P, [_P]),
    "ctx_dev_grads": (_P, [_P]),
    "ctx_dev_scalar_buf": (_P, [_P]),
    "ctx_stream": (_P, [_P]),
    "ctx_sync": (_c.c_int, [_P]),
    "ctx_dev_outputs": (_c.c_int, [_P, _c.POINTER(_P), _c.POINTER(_P), _c.POINTER(_P), _c.POINTER(_P)]),
    "ctx_last_codes": (_c.c_int, [_P, _F, _F, _c.POINTER(_c.c_int)]),
    "ctx_option_count": (_c.c_int, []),
    "ctx_option_name": (_c.c_char_p, [_c.c_int]),
    "ctx_get_option": (_c.c_int, [_P, _c.c_char_p, _c.POINTER(_c.c_int)]),
    "ctx_set_option": (_c.c_int, [_P, _c.c_char_p, _c.c_int]),
    "ctx_dp_unique_id": (_c.c_int, [_U8]),
    "ctx_dp_init": (_c.c_int, [_P, _U8, _c.c_int, _c.c_int]),
    "ctx_dp_world": (_c.c_int, [_P, _c.POINTER(_c.c_int), _c.POINTER(_c.c_int)]),
    "ctx_dp_allreduce_grads": (_c.c_int, [_P]),
    "ctx_dp_train_step": (_c.c_int, [_P, _P, _P, _P, _c.c_int, _c.c_float, _F]),
    "ctx_dp_scalars": (_c.c_int, [_P, _F]),
    "ctx_dp_train_step_sampled": (_c.c_int, [_P, _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32), _c.c_int, _c.c_float, _F]),
    "ctx_dp_eval_sampled": (_c.c_int, [_P, _c.POINTER(_c.c_int32), _c.POINTER(_c.c_int32), _c.c_int, _F, _F, _F]),
    "ctx_dp_allreduce_host_f64": (_c.c_int, [_P, _c.POINTER(_c.c_double), _c.c_size_t]),
    "ctx_profile_step": (_c.c_int, [_P, _P, _P, _P, _c.c_int, _c.c_float, _c.c_int, _c.POINTER(CtxProfEntry), _c.c_int,
                                    _c.POINTER(_c.c_int)]),
    "ctx_debug_read": (_c.c_int, [_P, _c.c_char_p, _F, _c.c_size_t]),
    "ctx_cnn_create": (_c.c_int, [_c.POINTER(CnnBuf), _c.c_int, _c.POINTER(CnnOp), _c.c_int, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _P,
                                  _c.POINTER(_P)]),
    "ctx_cnn_destroy": (None, [_P]),
    "ctx_cnn_last_error": (_c.c_char_p, [_P]),
    "ctx_cnn_set_weights": (_c.c_int, [_P, _F, _c.c_size_t]),
    "ctx_cnn_forward_u8": (_c.c_int, [_P, _U8, _c.c_int, _F]),
    "ctx_cnn_forward_u8_dev": (_c.c_int, [_P, _U8, _c.c_int, _c.POINTER(_P)]),
    "ctx_cnn_forward_dev": (_c.c_int, [_P, _P, _c.c_int, _c.POINTER(_P)]),
    "ctx_cnn_read_buffer": (_c.c_int, [_P, _c.c_int, _c.c_int, _F]),
    "ctx_cnn_profile": (_c.c_int, [_P, _c.c_int, _c.c_int, _F, _c.c_int]),
    "ctx_cnn_stream": (_P, [_P]),
    "ctx_cnn_sync": (_c.c_int, [_P]),
}

_lib = None


def load():
    """Returns the loaded library; raises if libctxtrans.so has not been built
    (python -c 'import __graft_entry__ as g; g.build()' or make -C imitation_from_observation_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (make -C {os.path.join(_HERE, 'csrc')}). "
            "There is no CPU fallback for the translator.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(lib, handle, rc):
    if rc != CTX_OK:
        msg = lib.ctx_last_error(handle)
        raise CtxError(rc, msg.decode() if msg else "")
