"""PLS_HIP_OPT_GRAPH: a repeated device-memory fit replayed as one hipGraph launch against the same fit enqueued kernel by
kernel.  Shapes where launches matter most: the reference's small examples on the general plan (PLS_HIP_TINY=0: 3 launches
per component) and one eighth of config 3 (12 us of fixed cost beside a 184 us pass)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PLS_HIP_TINY"] = "0"
import numpy as np, torch, pls_amd

side = torch.cuda.Stream()   # the legacy default stream cannot be captured
torch.cuda.set_stream(side)
h = pls_amd.Handle()
out = {}
for name, (N, K, M, A, algo) in (("toy_shape_10x15_m2_A2", (10, 15, 2, 2, pls_amd.ALGO_KERNEL)), ("nir_shape_60x401_A10", (60, 401, 1, 10, pls_amd.ALGO_KERNEL)),
                                 ("mid_4096x512_A20_nipals", (4096, 512, 1, 20, pls_amd.ALGO_NIPALS)),
                                 ("C3eighth_131072x512_A20_nipals", (131072, 512, 1, 20, pls_amd.ALGO_NIPALS))):
    X = h.synth_x(0, N, K, pls_amd.SEED_DEFAULT); Y = h.synth_y(0, N, M, pls_amd.SEED_DEFAULT)
    h.set_option(pls_amd.OPT_ALGO, algo)
    row = {}
    ref = None
    for tag, g in (("eager", 0), ("graph", 1)):
        h.set_option(pls_amd.OPT_GRAPH, g)
        o = h.fit_device(X, Y, A); torch.cuda.synchronize()
        for _ in range(3): h.fit_device(X, Y, A, out=o)          # second call captures, third replays
        torch.cuda.synchronize()
        reps = 200 if N < 100000 else 50
        t0 = time.perf_counter()
        for _ in range(reps): h.fit_device(X, Y, A, out=o)
        torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            h.fit_device(X, Y, A, out=o); torch.cuda.synchronize()
        tl = (time.perf_counter() - t0) / reps
        Bv = o["B"].cpu().numpy().copy()
        if ref is None: ref = Bv
        row[tag] = {"back_to_back_us": round(tb * 1e6, 1), "synchronised_us": round(tl * 1e6, 1), "B_equals_eager_bitwise": bool((Bv == ref).all())}
    h.set_option(pls_amd.OPT_GRAPH, 0)
    out[name] = row
    print(name, row, flush=True)
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout", "w"), indent=1)
