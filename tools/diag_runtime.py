"""Which HIP runtime(s) are mapped once torch and libveto_amd.so live in one process?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.zeros(1, device="cuda")
from veto_amd import native
native.load_library()
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "hsa-runtime" in l})
print("\n".join(libs))
