"""Streaming floor at slab size: nf_time_device_copy (the bench's yardstick kernel: whole 16-byte accesses) at working sets from 8 MB to 2 GB.
A 256 x 256 x 32 slab's endpoint pass moves 72 B per cell = 151 MB; what does a plain copy of that many bytes take on the same GPU?
usage: python profiles/tools/r04_copy_floor.py  (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from helpers import make_hip, synthetic_inputs
s = make_hip(synthetic_inputs(8, 8, 8, 2, seed=1))
print("bytes moved (read + write)   GB/s   us per copy")
for mb in (8, 16, 32, 64, 96, 151, 192, 256, 384, 512, 1024, 2048):
    moved = mb * 1e6
    g = s.time_device_copy(int(moved / 2), 50)          # the argument is the size of the copy; read + write are counted
    print(f"{mb:6d} MB   {g:8.1f}   {moved / (g * 1e9) * 1e6:8.2f}", flush=True)
s.close()
