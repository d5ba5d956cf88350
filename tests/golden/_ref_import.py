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
