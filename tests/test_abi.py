"""The C-ABI boundary without a GPU: libctxtrans.so loads, exports every symbol include/ctxtrans.h
declares, the ctypes table matches the header, and the product path FAILS LOUDLY without a device
(there is no CPU fallback)."""
import ctypes
import os
import re

import pytest

from imitation_from_observation_amd import _lib


def header_functions(repo_root):
    src = open(os.path.join(repo_root, "include", "ctxtrans.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ctx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_header_symbol(built_lib, repo_root):
    names = header_functions(repo_root)
    assert len(names) >= 30
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/ctxtrans.h but not exported"


def test_ctypes_table_matches_header(built_lib, repo_root):
    assert sorted(_lib.SIGNATURES) == header_functions(repo_root)


def test_abi_version(built_lib):
    assert built_lib.ctx_abi_version() == 4


def test_param_total_is_the_references(built_lib):
    from imitation_from_observation_amd import Translator
    assert Translator.param_total(64, 64, 64, 1024) == 47_647_811
    assert Translator.param_total(48, 48, 64, 1024) == 47_647_811 - 2 * (8192 - 4608) * 1024 - (8192 - 4608) * 1025
    assert Translator.arena_floats(64, 64, 64, 1024) % 256 == 0


@pytest.mark.parametrize("kw", [dict(H=60), dict(W=40), dict(df_dim=48), dict(featsize=100), dict(max_batch=0), dict(C=1),
                                dict(variant=7)])
def test_bad_config_is_rejected(built_lib, kw):
    base = dict(variant=0, H=64, W=64, C=3, df_dim=64, featsize=1024, max_batch=4, precision=0)
    base.update(kw)
    cfg = _lib.CtxConfig(**base)
    assert built_lib.ctx_param_total_for(ctypes.byref(cfg)) == _lib.CTX_E_INVALID
    h = ctypes.c_void_p()
    assert built_lib.ctx_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == _lib.CTX_E_INVALID
    assert not h.value
    assert built_lib.ctx_last_error(None)


def test_no_cpu_fallback(built_lib):
    """On a box without a GPU the product path must refuse to run, not compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from imitation_from_observation_amd import CtxError, Translator
    with pytest.raises(CtxError) as ei:
        Translator(32, 32, 32, 128, max_batch=2)
    assert ei.value.code == _lib.CTX_E_DEVICE
    assert "no CPU path" in str(ei.value)


def test_null_handle_calls_return_errors(built_lib):
    assert built_lib.ctx_param_count(None) == _lib.CTX_E_INVALID
    assert built_lib.ctx_sync(None) == _lib.CTX_E_INVALID
    assert built_lib.ctx_dev_adam(None, 1e-4) == _lib.CTX_E_INVALID
    built_lib.ctx_destroy(None)   # no-op


def test_product_package_never_imports_the_oracle(repo_root):
    pkg = os.path.join(repo_root, "imitation_from_observation_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("checker", ""), f"{f} mentions the oracle"


def test_checkpoint_paths_without_extension_resolve_to_one_file(tmp_path):
    """The reference's Saver paths have no extension (train_script.py:181-182); save and load must agree on the file."""
    from imitation_from_observation_amd import Translator
    p = str(tmp_path / "model_5000_12.30_4.00_5.00_0.00")
    assert Translator.checkpoint_file(p) == p + ".npz"
    assert Translator.checkpoint_file(p + ".npz") == p + ".npz"
    assert Translator.checkpoint_file(tmp_path / "ck.npz") == str(tmp_path / "ck.npz")


def test_result_pool_recycles_only_released_arrays():
    """Translator's result arrays above the pool's threshold are recycled once the caller has let go of them (warm pages instead of a
    fresh > 32 MB mapping per call) and never while anything -- the array itself or a view of it -- is still held."""
    import numpy as np
    from imitation_from_observation_amd.translator import _ResultPool
    pool = _ResultPool(min_bytes=1024)
    a = pool.get((64, 64))
    ida = id(a)
    b = pool.get((64, 64))
    assert b is not a                                   # `a` is still held: a second array
    del a
    c = pool.get((64, 64))
    assert id(c) == ida                                 # released: recycled
    view = c[:4]
    del c
    d = pool.get((64, 64))
    assert id(d) != ida and view.base is not None       # a view keeps its base out of circulation
    small = pool.get((4, 4))
    assert pool.get((4, 4)) is not small                # below the threshold: plain np.empty


def test_options_are_enumerable_and_documented(built_lib):
    """Every per-handle switch the library knows (ctx_option_count / ctx_option_name) is documented in include/ctxtrans.h, and no
    other CTX_* environment variable is read by the kernels' sources (only CTX_RCCL_LIB, the dlopen path, and the debugging aid
    CTX_DEBUG_POISON, which fills fresh device buffers with NaN)."""
    import glob
    import re
    names = [built_lib.ctx_option_name(i).decode() for i in range(built_lib.ctx_option_count())]
    assert len(names) == len(set(names)) >= 10 and built_lib.ctx_option_name(len(names)) is None
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(ROOT, "include", "ctxtrans.h")).read()
    for n in names:
        assert re.search(r"^\s\*\s+%s\s+-?\d+\s" % re.escape(n), header, re.M), f"option {n} is not documented in include/ctxtrans.h"
    env = set()
    for f in glob.glob(os.path.join(ROOT, "imitation_from_observation_amd", "csrc", "*")):
        if f.endswith((".hip", ".h", ".cpp", ".inc")):
            env |= set(re.findall(r'getenv\("(CTX_[A-Z0-9_]+)"\)', open(f).read()))
    assert env == {"CTX_RCCL_LIB", "CTX_DEBUG_POISON"}, env
    assert "CTX_DEBUG_POISON" in header
