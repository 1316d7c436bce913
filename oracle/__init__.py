"""CPU oracle of the hot path -- TEST INFRASTRUCTURE, never imported by the product package.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
