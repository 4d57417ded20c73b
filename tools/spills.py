"""developer: kernels of the built library that spill registers (or whose mangled name contains argv[1]), with their
register counts -- pointwise_amd.build.kernel_resources()."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pointwise_amd import build
pat = sys.argv[1] if len(sys.argv) > 1 else None
for name, v, vs, ss, scr in build.kernel_resources():
    if vs or (pat and pat in name):
        print("%-72s vgprs %3d  vgpr spills %3d  sgpr spills %3d  scratch %4d B" % (name[:72], v, vs, ss, scr))
