"""TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT PATH.

CPU restatement (C, `liboracle.so`) of the algorithms the HIP kernels replace, used as the
checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.  See
bp_oracle.c, qary_oracle.c, mc_oracle.c for what each function follows in the reference
and what pins it.
"""
