"""CPU oracle package -- test infrastructure only (see the header of each module).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
