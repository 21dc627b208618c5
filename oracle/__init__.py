"""CPU oracle for the MixDQ W8A8 operator stack -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package,
and only as the checker.  Nothing under mixdq_amd/ imports it.
"""
