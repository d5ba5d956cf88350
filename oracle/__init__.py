"""CPU oracle for the DeepOF unsupervised-embedding hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain NumPy / PyTorch-CPU fp32 restatement of the reference algorithm
(mlfpm/deepof v0.9.0, ``deepof/clustering``).  It exists to *check* the HIP path:

* only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
  import it -- never ``deepof_amd`` (the product), which fails loudly without its HIP library;
* every function cites the reference ``file:line`` it restates;
* parity status: **pinned** -- ``tests/test_oracle_golden.py`` checks it against fixtures in
  ``tests/golden/*.npz`` that were produced by importing the reference itself in the build
  container (``tests/golden/make_golden.py``; the reference never travels to the GPU box).

The arithmetic that the reference delegates to third-party code is PyTorch's (``nn.GRU``,
``conv1d``, ``layer_norm``, ``linalg.svdvals``; pinned torch==2.8.0 in the reference's
pyproject.toml:42-49, 2.10.0 here -- semantics unchanged) and is used here through the same
PyTorch ops on CPU.
"""
