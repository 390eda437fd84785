"""HBM roofline for read / write mixes (profiles/tools/stream/stream_mix.hip): TB/s per mix, linear and 16-row x 64-B pieces."""
import ctypes as C, os, sys
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstream.so"))
lib.stream_mix.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 603992
rows -= rows % 16
src = torch.randn(5 * rows, 128, device="cuda")
dst = torch.empty(5 * rows, 128, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for nr, nw in ((0, 1), (0, 4), (1, 1), (2, 2), (4, 4), (1, 3), (3, 1), (2, 5), (1, 5), (5, 2)):
    for pieces in (0, 1):
        for nt in (0, 2, 3):
            blocks = 4096
            f = lambda: lib.stream_mix(src.data_ptr(), dst.data_ptr(), rows, rows * 32, nr, nw, pieces, blocks, nt, st)
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
            print(f"rows {rows} reads {nr} writes {nw} {'pieces' if pieces else 'linear'} nt {nt}: {(nr + nw) * rows * 512 / t / 1e12:.2f} TB/s", flush=True)
