import sys, ctypes, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from deepof_amd import _capi
libpath = sys.argv[1]
lib = _capi.bind(ctypes.CDLL(libpath))
src = open('/root/repo/tests/parity_common.py').read()
src = src.replace("    err = float(np.abs(got - ref).max())\n    assert err <= atol + rtol * scale, (name, err, scale)\n", "    err = float(np.abs(got - ref).max())\n    RAT.append((err / (atol + rtol * scale), name, err, scale))\n")
ns = {'RAT': [], '__name__': 'pc2', '__file__': '/root/repo/tests/parity_common.py'}
exec(compile(src, 'pc2', 'exec'), ns)
try:
    print(ns['run_vade_tcn_b64_check'](lib, 'cuda', '/root/repo/tests/golden'))
except Exception as e:
    print("EXC", repr(e)[:500])
R = sorted(ns['RAT'], reverse=True)
for r in R[:25]: print(r)
print(len(R), np.median([r[0] for r in R]))
