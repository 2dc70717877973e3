"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the SubGAcc hot path of SUREL+ (sampler, SpG build, SpJoin).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(surel_plus_amd/) never does.  See oracle/subgacc_oracle.c for the reference citations and the
parity status (PINNED against tests/golden/ and, when present, oracle/_ref).
"""
from .oracle import *  # noqa: F401,F403
