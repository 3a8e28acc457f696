"""CPU oracle of the physics step -- TEST INFRASTRUCTURE ONLY (see oracle/mjoracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (mjlab_amd/) never does.
"""
