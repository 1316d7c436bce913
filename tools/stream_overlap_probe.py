"""Do MFMA-bound Slow-pathway kernels and thin, latency/bandwidth-bound Fast-pathway kernels overlap on MI355X when issued
on two HIP streams -- eagerly, as fork/join branches of ONE captured graph, and as two graphs replayed on two streams?

Prints milliseconds per repetition for: serial, two-stream eager, one graph (serial capture), one graph with a forked
branch, two graphs on two streams.  The answer decides whether running the two pathways of a SlowFast stage concurrently
can pay (they are independent between lateral connections).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slowfast_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N = 32


def conv_case(Ci, T, H, W, Co, k, s, p):
    geom = ops.ConvGeom((N, Ci, T, H, W), Co, k, s, p)
    x = ops.cl_empty(geom.in_shape, dev)
    x.normal_()
    w = torch.randn((Co, Ci) + k, device=dev) * 0.05
    wf, wd = ops.prep_weights(w, geom)
    dy = ops.cl_empty(geom.out_shape, dev)
    dy.normal_()
    dw = torch.zeros_like(w)
    return geom, x, wf, wd, dy, dw


slow = [conv_case(256, 8, 14, 14, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
        conv_case(1024, 8, 14, 14, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
        conv_case(512, 8, 7, 7, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1))]
fast = [conv_case(8, 32, 56, 56, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
        conv_case(8, 32, 56, 56, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
        conv_case(32, 32, 56, 56, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0))]
fast_act = ops.cl_empty((N, 32, 32, 56, 56), dev)
fast_act.normal_()
sc = torch.ones(32, device=dev)
sh = torch.zeros(32, device=dev)
REP = 4


def run_slow():
    for _ in range(REP):
        for geom, x, wf, wd, dy, dw in slow:
            ops.conv_fwd(x, wf, geom, stats=False)
            ops.conv_dgrad(dy, wd, geom)
            ops.conv_wgrad(x, dy, geom, dw)


def run_fast():
    for _ in range(REP):
        for geom, x, wf, wd, dy, dw in fast:
            ops.conv_fwd(x, wf, geom, stats=False)
            ops.conv_wgrad(x, dy, geom, dw)
        ops.bn_act(fast_act, sc, sh, relu=True)


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


side = torch.cuda.Stream()


def two_streams():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        run_fast()
    run_slow()
    main.wait_stream(side)


def serial():
    run_fast()
    run_slow()


run_fast(); run_slow(); two_streams()
torch.cuda.synchronize()
res = {}
res["slow only"] = timeit(run_slow)
res["fast only"] = timeit(run_fast)
res["eager serial"] = timeit(serial)
res["eager two streams"] = timeit(two_streams)


def capture(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


g_serial = capture(serial)
res["graph serial"] = timeit(g_serial.replay)
g_fork = capture(two_streams)
res["graph fork/join"] = timeit(g_fork.replay)
g_slow, g_fast = capture(run_slow), capture(run_fast)


def two_graphs():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        g_fast.replay()
    g_slow.replay()
    main.wait_stream(side)


res["graph slow only"] = timeit(g_slow.replay)
res["graph fast only"] = timeit(g_fast.replay)
res["two graphs on two streams"] = timeit(two_graphs)
for k, v in res.items():
    print(f"{k:28s} {v:8.3f} ms")
