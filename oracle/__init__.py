"""CPU oracle for the sequitr hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package.  sequitr_amd/ never does (the product path fails loudly
when the HIP library is missing instead of falling back to the CPU).
"""
