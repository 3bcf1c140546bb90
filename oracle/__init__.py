"""TEST INFRASTRUCTURE ONLY.  CPU restatements of the reference hot path used as the parity
checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package never imports this."""
