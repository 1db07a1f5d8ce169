"""CPU oracle package -- TEST INFRASTRUCTURE ONLY (see ucnerf_oracle.py header).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
