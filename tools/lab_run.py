"""Times a few GEMM shapes (graph replay) with the library selected by SDLT_KERNEL_LIB.  usage: lab_run.py [shape ...]  shape = M,N,K,tile,stages,splitk,lora"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_probe import bench
shapes = sys.argv[1:] or ["1024,1280,5120,1,0,1,0", "1024,1280,5120,1,0,3,0", "4096,5120,1280,1,0,1,0", "4096,5120,1280,4,0,1,0", "1024,1280,1280,2,0,1,1", "8192,8192,8192,6,0,1,0"]
out = []
for sh in shapes:
    M, N, K, tile, st, sk, lora = [int(v) for v in sh.split(",")]
    us = bench(M, N, K, tile, sk, bool(lora), None, stages=st)
    out.append(f"{sh}:{us:.1f}us/{2*M*N*K/us/1e6:.0f}TF")
print(os.environ.get("SDLT_KERNEL_LIB", "default").split("/")[-1], " ".join(out))
