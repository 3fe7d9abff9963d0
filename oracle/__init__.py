"""TEST INFRASTRUCTURE: CPU oracle for the fly_bProject hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product (fly_bproject_amd/) never does.  See fly_oracle.h for parity status.
"""
