"""CPU oracle for the joint-heat-map inference path of max-andr/joint-cnn-mrf.

TEST INFRASTRUCTURE ONLY.  Nothing under ``joint-cnn-mrf_amd/`` may import, call or
link this package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / the reported CPU baseline.

PARITY UNPINNED.  The reference delegates all arithmetic on this path to TensorFlow 1.x
(`/root/reference/main.py:4`), which is not vendored, not pinned and not installable in
this image, and the reference ships no tests, golden vectors or fixtures for the path
(SURVEY.md section 4 / 8c).  The oracle is therefore a *restatement* of
`main.py:29-125,212-217` + `evaluation.py:15-24` under the documented TF-1.x op semantics
spelled out in `jcm_oracle.py`.  It is pinned only by
  * closed-form known-answer tests (tests/test_oracle_kat.py),
  * a second, independently written formulation (`jcm_oracle_torch.py`, torch-CPU ops +
    `scipy.signal.convolve2d`) that must agree with the numpy one,
  * the reference's own NumPy-level prior builder run in this container
    (tests/golden/make_golden.py executes `prepare_pairwise_distribution.py` itself).
"""
