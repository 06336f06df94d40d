"""CPU oracles for the two WISE hot paths.  TEST INFRASTRUCTURE ONLY: nothing under wise_amd/ may
import this package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do."""
