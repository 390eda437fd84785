"""TB/s of read / write mixes against the contiguous run per row of a wave instruction (stream_run.hip)."""
import ctypes as C, os, sys
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstream_run.so"))
lib.stream_run.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 603984
rows -= rows % 16
src = torch.randn(5 * rows, 128, device="cuda")
dst = torch.empty(5 * rows, 128, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for nr, nw in ((2, 5), (5, 2), (2, 2), (1, 1)):
    for nt in (0, 2, 3):
        line = []
        for run in (4, 8, 16, 32, 64):
            f = lambda: lib.stream_run(src.data_ptr(), dst.data_ptr(), rows, rows * 32, nr, nw, run, 4096, nt, st)
            for _ in range(3):
                assert f() == 0
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                f()
            b.record()
            torch.cuda.synchronize()
            t = a.elapsed_time(b) / 10 * 1e-3
            line.append(f"{run * 16:>4} B {(nr + nw) * rows * 512 / t / 1e12:.2f}")
        print(f"rows {rows} reads {nr} writes {nw} nt {nt}: " + " | ".join(line), flush=True)
