"""ORACLE: CPU restatement of the reference's MFP hot path (test infrastructure only).

PARITY UNPINNED -- see ``np_ref.py``.  Never imported by the product (``flex-dm_amd/``).
"""
