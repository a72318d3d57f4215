import ctypes as C, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
for abl, name in ((0, "full"), (1, "no exp2"), (5, "no softmax VALU"), (3, "no PV mfma"), (4, "no S mfma"), (2, "no barrier/vmcnt")):
    mb.lib.f5_debug_set_attn_ablation(abl)
    print(name, end=": ")
    mb.attn_case(64, 16, 937, 0)
mb.lib.f5_debug_set_attn_ablation(0)

for ver in (2, 3, 4, 3, 4, 2):
    mb.lib.f5_debug_set_attn_version(ver)
    print("version", ver, end=": ")
    mb.attn_case(64, 16, 937, 0)
mb.lib.f5_debug_set_attn_version(2)
