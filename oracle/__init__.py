"""oracle/ -- CPU checkers for the block-sparse SpMM hot path.  TEST INFRASTRUCTURE ONLY.

  oracle.oracle : ctypes binding of liboracle.so, the plain-C restatement (sparta_oracle.c)
  oracle.ref    : ctypes binding of _ref/libsparta_ref.so, the real reference compiled from /root/reference

Only tests/, tests/golden/make_golden.py, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
package; the product package `sparta_amd` never does.
"""
