"""CPU oracle for the LLaDA sampling / likelihood hot path.

TEST INFRASTRUCTURE ONLY.  This package is a numpy restatement of the reference
algorithm (taiqihe/LLaDA ``generate.py`` and ``get_log_likelihood.py`` plus the
LLaDA-8B non-causal forward they call).  It exists to check the HIP product path
and to serve as the timed CPU baseline; nothing in ``llada_amd`` may import it.

Pinning status:
  * sampler (``oracle.sampler``) and likelihood (``oracle.likelihood``): pinned by
    golden vectors captured from the reference itself (``tools/make_golden.py``
    imports ``/root/reference/generate.py`` / ``get_log_likelihood.py`` and writes
    ``tests/golden/*.npz``).
  * model forward (``oracle.model``): the reference's model code
    (``modeling_llada.py``, loaded from the HF hub with ``trust_remote_code``) is not
    part of the reference tree, so the forward is **parity unpinned** against the
    reference; it is pinned only against a stand-in (transformers' Llama run
    bidirectionally at tiny shapes, ``tests/golden/forward_*.npz``).
"""
