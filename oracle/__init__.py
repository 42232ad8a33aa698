"""oracle/ -- TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's query path.
Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import it."""
