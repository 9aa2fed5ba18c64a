"""CPU oracle for the rdx hot path. TEST INFRASTRUCTURE ONLY — see oracle/rdx_oracle.c header.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
