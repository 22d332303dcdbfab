"""CPU oracle for the GenSeg segmentation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`semantic_segmentation_amd/`) may import this package; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do, and only as
the checker / the reported CPU baseline.
"""
