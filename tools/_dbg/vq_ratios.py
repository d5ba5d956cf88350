import sys, ctypes, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from deepof_amd import _capi
lib = _capi.bind(ctypes.CDLL(sys.argv[1]))
src = open('/root/repo/tests/parity_common.py').read()
src = src.replace("    assert err <= atol + rtol * scale, (name, err, scale)\n    return err / max(scale, 1e-30)\n", "    RAT.append((err / (atol + rtol * scale), name, err, scale))\n    return err / max(scale, 1e-30)\n")
ns = {'RAT': [], '__name__': 'pc2', '__file__': '/root/repo/tests/parity_common.py'}
exec(compile(src, 'pc2', 'exec'), ns)
try:
    print(ns['run_vqvae_tcn_ref_check'](lib, 'cuda', '/root/repo/tests/golden'))
except Exception as e:
    print("EXC", repr(e)[:300])
R = sorted(ns['RAT'], reverse=True)
for r in R[:14]: print(r)
print(len(R), np.median([r[0] for r in R]))
