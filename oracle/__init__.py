"""CPU oracle for the hamilton equations-of-motion path -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  PARITY UNPINNED: see the header of hamk_oracle.c.
"""
