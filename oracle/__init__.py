"""CPU oracle of the rasterize + quantize hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (gscodec_studio_amd/) never does.
"""
