"""TEST INFRASTRUCTURE ONLY: CPU oracle for the Muskingum-Cunge hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  The product (troute_amd) never does.
"""
