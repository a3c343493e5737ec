"""TESTS-ONLY stand-in for AllenNLP 2.4.0 (README.md:25-27 of the reference; not installable here: no network).

Purpose: let the reference's OWN files — MemVul/model_memory.py, MemVul/reader_memory.py, MemVul/custom_metric.py,
MemVul/custom_PTM_embedder.py and predict_memory.py — be imported VERBATIM from /root/reference and executed, so that
golden fixtures come from the reference's code rather than from a restatement (oracle/ref_harness/run_reference.py,
tests/golden/make_ref_golden.py).  Only the AllenNLP surface those files touch on the predict_memory.py path is
provided, restated from AllenNLP 2.4.0's published behaviour (each piece cites the upstream module it follows).
Nothing under memvul_amd/ imports this package; it is not on sys.path unless the harness puts it there.
"""
__version__ = "2.4.0-stub"
