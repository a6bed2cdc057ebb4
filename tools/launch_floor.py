"""The floor under the C2 mel loop: a hipGraph with the loop's launch topology - the same NUMBER of dependent launches on one stream, the same grid
/ block geometry per launch - in which every kernel is EMPTY (ss_debug_null_launch). Its replay time is what the launch edges alone cost; the
difference to the real loop is what kernels can still win ("launch structure").
    python tools/launch_floor.py [--B 8] [--T 1500] [--steps 100]
Per network evaluation of the fp32 C2 loop (run_residual_stack, diffusion.hip): input projection, 20 x (gate, residual projection) minus the last
projection, skip GEMM, output projection + sampler update (42 launches); once per loop: q-sample, conditioner projection, 20 addend re-layouts."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stylesinger_amd import lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--T", type=int, default=1500)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    rows = a.B * a.T
    lib = L.load()
    # (grid, block) of each launch of one evaluation at this shape: what the library launches (gate: 16x16x4 tiles of 32 quads -> ceil(rows / 4 / 32)
    # x 4 column tiles; residual projection: 96-row tiles; skip GEMM 64-row tiles x 2 column tiles; the small ones one block per 256 elements)
    gate = (-(-rows // 128) * 4, 256)
    res = (-(-rows // 96), 256)
    skip = (-(-rows // 64) * 2, 256)
    small = (-(-rows * 80 // 256), 256)
    per_eval = [small] + [gate, res] * 19 + [gate] + [skip, small]
    once = [small, (-(-rows // 128) * 80, 256)] + [small] * 20
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        lib.ss_debug_null_launch(1, 64, st.cuda_stream)
        st.synchronize()
        with torch.cuda.graph(g, stream=st):
            for gr, bl in once:
                L.check(lib.ss_debug_null_launch(gr, bl, st.cuda_stream), "null")
            for _ in range(a.steps):
                for gr, bl in per_eval:
                    L.check(lib.ss_debug_null_launch(gr, bl, st.cuda_stream), "null")
        g.replay()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(a.iters):
            g.replay()
        e1.record(st)
        st.synchronize()
    n = len(once) + a.steps * len(per_eval)
    ms = e0.elapsed_time(e1) / a.iters
    print(f"null-kernel graph of the C2 mel loop's topology ({a.B} x {a.T}, {a.steps} evaluations): {n} dependent launches, {ms:.2f} ms per replay = "
          f"{ms * 1e3 / n:.2f} us per launch edge")
    return ms, n


if __name__ == "__main__":
    main()
