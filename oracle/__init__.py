"""CPU oracle for the DGR inference hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy / torch-CPU) of the algorithm that
`DeepGlobalRegistration.register()` runs in the reference
(`core/deep_global_registration.py:238-324`).  It is the *checker* for the HIP
path: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
leg may import it.  Nothing under `deepglobalregistration_amd/` imports it and
the product path fails loudly when the HIP library is missing.

Parity status (see DESIGN.md "Oracle"):

* `oracle.knn`, `oracle.registration`: **pinned** -- cross-checked against the
  reference's own importable modules (`core/knn.py`, `core/registration.py`,
  `core/loss.py`, `core/metrics.py`) by `tests/golden/make_golden.py`; the
  resulting vectors are committed under `tests/golden/`.
* `oracle.me_semantics`, `oracle.resunet`: **parity unpinned** -- the
  arithmetic lives in MinkowskiEngine==0.5.4 (`requirements.txt:24`), which is
  neither vendored under /root/reference nor installable offline.  The
  restatement follows ME's published algorithm and the reference's call sites
  (`model/resunet.py:419-649`, `model/residual_block.py:15-134`,
  `model/common.py:11-21`); the conventions that cannot be verified here
  (kernel-offset enumeration order, transposed-map convention, quantize
  ordering) live in exactly one function each in `me_semantics.py`.
  The NETWORK on top of those conventions (topology, op order, state-dict
  layout) is pinned: `tests/golden/make_golden_model.py` runs the reference's
  own `ResUNetBN2C` classes over a stand-in for the ME import
  (`tests/golden/me_stub`) and `tests/test_oracle_model_golden.py` holds
  `oracle.resunet` to the committed outputs.
* `oracle.pipeline`: the learned branch of `register()` is **pinned** by a
  run of the reference's own `DeepGlobalRegistration.register()` over the same
  stand-in (`tests/golden/make_golden_register.py`,
  `tests/test_oracle_register_golden.py`); the safeguard / ICP tail goes
  through `oracle.open3d_reg` (unpinned, below).
* `oracle.open3d_reg`: **parity unpinned** -- Open3D==0.17.0 is not installed
  and not vendored; restates `RegistrationICP` and
  `RegistrationRANSACBasedOnCorrespondence` as the reference calls them
  (`core/deep_global_registration.py:50-64, 302-322`), with a counter-based
  sampler in place of Open3D's mt19937 streams.
"""
