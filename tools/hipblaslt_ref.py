"""What the vendor library reaches on the encoder's product shapes on THIS box (random FP16 operands, FP32 accumulate, plain C = A W^T with no
epilogue) -- a yardstick for gemmTiled8 (tools/gemm8_probe.py), never part of the product: python tools/hipblaslt_ref.py"""
import os
import time

import torch

SHAPES = [(168000, 1024, 1024), (168000, 3072, 1024), (168000, 4096, 1024), (168000, 1024, 4096), (168000, 49152, 1024), (16500, 4096, 1024)]


def main():
    torch.manual_seed(0)
    shapes = SHAPES
    if os.environ.get("PROBE_SHAPES"):
        shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ["PROBE_SHAPES"].split(",")]
    for (M, N, K) in shapes:
        a = (torch.rand((M, K), device="cuda", dtype=torch.float16) - 0.5)
        w = (torch.rand((N, K), device="cuda", dtype=torch.float16) - 0.5)
        for _ in range(2):
            c = a @ w.t()
        torch.cuda.synchronize()
        it = 3 if N > 10000 else 10
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(it):
                c = a @ w.t()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / it)
        print("hipBLASLt %6d x %5d x %4d: %7.1f us  %6.0f TFLOP/s (f16 out)" % (M, N, K, best * 1e6, 2.0 * M * N * K / best / 1e12), flush=True)
        del a, w, c


if __name__ == "__main__":
    main()
