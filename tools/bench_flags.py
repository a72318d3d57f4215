"""Run bench.py in-process after setting debug hooks:  python tools/bench_flags.py order=1 ring=0 attn=3 -- --steps 3"""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E
lib = E.load_library()
args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
for kv in args[:split]:
    k, v = kv.split("=")
    v = int(v)
    if k == "order": lib.f5_debug_set_gemm_order(v)
    elif k == "tile": lib.f5_debug_set_gemm_tile(v)
    elif k == "attn": lib.f5_debug_set_attn_version(v)
    elif k == "big": lib.f5_debug_set_gemm_big_kernel(v, -1)
    elif k == "ring": lib.f5_debug_set_gemm_ring(v)
    elif k == "wide": lib.f5_debug_set_attn_wide(v)
    elif k == "kvsplit": lib.f5_debug_set_attn_kvsplit(v)
    elif k == "cptps": lib.f5_debug_set_convpos_tps(v)
    elif k == "cpxcd": lib.f5_debug_set_convpos_xcd_map(v)
    elif k == "nband": lib.f5_debug_set_gemm_nband(v)
    elif k == "lnfuse": lib.f5_debug_set_ln_fusion(v)
    elif k == "attnvar": lib.f5_debug_set_attn_variant(v)
    elif k == "streamk": lib.f5_debug_set_gemm_streamk(v)
    elif k == "gflags": lib.f5_debug_set_gemm_flags(v)      # 8 = residual update by no-return L2 atomics (experiment)
    elif k == "qkvtile": lib.f5_debug_set_gemm_qkv_tile(v)  # small-M QKV: 0 auto, 12 / 13 = 8-wave 128x256 ring with transposed q / k wave tiles
    elif k == "qkvtr": lib.f5_debug_set_qkv_transposed(v)   # 0 = straight q / k tiles in the 256x256 QKV kernel
    elif k == "qpremul": lib.f5_debug_set_q_premul(v)       # 0 = plain q (attention multiplies by scale * log2 e itself)
    else: raise SystemExit(f"unknown flag {k}")
sys.argv = ["bench.py"] + args[split + 1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
