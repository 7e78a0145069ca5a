#!/usr/bin/env python3
"""L2 -> CU throughput of LDS-DMA vs VGPR loads (tools/microbench/vmem_paths.hip).
build (here): hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/microbench/vmem_paths.hip -o tools/microbench/libvmem_paths.so
run (GPU box): python tools/microbench/vmem_paths.py"""
import ctypes as C
import os

import torch

lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvmem_paths.so"))
lib.vmem_pull.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]
buf = torch.randint(0, 255, (512 << 20,), dtype=torch.uint8, device="cuda")
sink = torch.zeros(4096, device="cuda")
torch.cuda.synchronize()
iters, blocks = 2000, 256
for window_mb in (8, 16, 128, 448):
    for mode, name in ((0, "LDS-DMA (global_load_lds_dwordx4)"), (1, "VGPR (global_load_dwordx4)"), (2, "half DMA / half VGPR"),
                       (3, "LDS-DMA, 8 rows x 128 B per instr.")):
        ms = C.c_float()
        rc = lib.vmem_pull(mode, buf.data_ptr(), window_mb << 20, iters, blocks, sink.data_ptr(), C.byref(ms))
        assert rc == 0, rc
        tot = blocks * iters * 65536
        print(f"window {window_mb:4d} MB  {name:36s} {ms.value:8.3f} ms  {tot / ms.value / 1e9:7.2f} TB/s  "
              f"{tot / ms.value / 1e6 / blocks:6.1f} GB/s per CU")
