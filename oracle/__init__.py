"""CPU restatement of the reference algorithm: TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).  The product never imports this package."""
