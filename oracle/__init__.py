"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (stock torch fp32 ops) of the reference's hot path:
DeepLabV3+/V2 over ResNet + the categorical memory of network/memory.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package -- and only as the checker / the timed CPU baseline. The product
path (pinthememory_amd/) never imports it and fails loudly without its HIP library.

Pinning: the reference ships no tests/golden vectors, so the
oracle is pinned against the reference ITSELF, imported in the build container
(oracle/import_reference.py, recipe of SURVEY.md Appendix A): tests/test_oracle_vs_reference.py
proves bit-equality of outputs and autograd grads when /root/reference exists, and
oracle/make_golden.py captured the committed fixtures under tests/golden/ from
the imported reference (not from this restatement).
"""
