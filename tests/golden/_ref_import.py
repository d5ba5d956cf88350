"""Import shim that loads the *reference* trainer modules in place from /root/reference.

Only used by tests/golden/make_golden.py, in the build container, to generate fixtures.
Nothing here (or anything it imports) travels to the GPU box or is used by the product.
The shim is ours; it only arranges for deepof/clustering/*.py to be importable without the
reference's heavy optional dependencies.
"""
import sys
import types

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference():
    pkg = types.ModuleType("deepof")
    pkg.__path__ = [REF + "/deepof"]
    sys.modules["deepof"] = pkg
    _stub("h5py")
    _stub("duckdb")
    _stub("optuna", Trial=object, TrialPruned=Exception)
    _stub("IPython")
    _stub("IPython.display", clear_output=lambda *a, **k: None)

    class _SW:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, _):
            return lambda *a, **k: None

    _stub("torch.utils.tensorboard", SummaryWriter=_SW)
    _stub("deepof.utils", validate_parameter=lambda *a, **k: None)
    import deepof.config  # noqa: F401
    import deepof.data_loading  # noqa: F401
    import deepof.clustering.models_new as M
    import deepof.clustering.losses as L
    import deepof.clustering.training as T
    import deepof.clustering.model_utils_new as U
    import deepof.clustering.dataset as D
    import deepof.clustering.censNetConv_pt as C

    return types.SimpleNamespace(M=M, L=L, T=T, U=U, D=D, C=C)


PREPROCESS_NAMES = (
    "GlobalScalerSpec", "infer_column_types", "scale_table", "_pp_make_scaler", "_pp_sanitize_numeric",
    "_pp_load_and_prepare_table", "_pp_filter_low_variance", "_pp_init_empty_output_container", "_section_standardize",
    "_pp_pass1_collect_samples", "_pp_fit_global_scaler", "_pp_apply_global", "_pp_pass2_scale_and_save",
)


def load_reference_preprocessing():
    """The reference's table-preprocessing functions (deepof/utils.py:2342-3027), executed in place.

    deepof/utils.py as a whole needs cv2, numba, sleap_io, segment_anything ... (absent here), so only the
    definitions listed above are compiled -- straight from the file where it lies, nothing is copied -- into a
    module that stands in for ``deepof.utils`` (the functions call each other through that name)."""
    import ast
    import copy
    import os
    from dataclasses import dataclass
    from typing import Any, Dict, List, Optional

    import numpy as np
    import pandas as pd
    from sklearn.preprocessing import MinMaxScaler, RobustScaler, StandardScaler

    if "deepof" not in sys.modules:
        pkg = types.ModuleType("deepof")
        pkg.__path__ = [REF + "/deepof"]
        sys.modules["deepof"] = pkg
        _stub("h5py")
        _stub("duckdb")
    import deepof.config
    import deepof.data_loading as DL

    path = REF + "/deepof/utils.py"
    tree = ast.parse(open(path).read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in PREPROCESS_NAMES]
    assert len(keep) == len(PREPROCESS_NAMES), [n.name for n in keep]
    mod = sys.modules.get("deepof.utils") or _stub("deepof.utils")
    sys.modules["deepof"].utils = mod

    class _NoBar:
        def __init__(self, *a, **k):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def update(self, *a):
            pass

    mod.__dict__.update(np=np, pd=pd, os=os, copy=copy, dataclass=dataclass, Any=Any, Dict=Dict, List=List, Optional=Optional,
                        StandardScaler=StandardScaler, MinMaxScaler=MinMaxScaler, RobustScaler=RobustScaler,
                        get_dt=DL.get_dt, save_dt=DL.save_dt, tqdm=_NoBar, deepof=sys.modules["deepof"],
                        PROGRESS_BAR_FIXED_WIDTH=deepof.config.PROGRESS_BAR_FIXED_WIDTH)
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), mod.__dict__)
    return mod
