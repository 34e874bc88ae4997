"""CPU oracle for the OccNeRF hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (occnerf_amd/) never does.
"""
