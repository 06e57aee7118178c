"""CPU oracle: a restatement of the reference's algorithm for the hot path.  TEST INFRASTRUCTURE ONLY —
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by layoutdetr_amd."""
