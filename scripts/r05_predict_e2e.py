#!/usr/bin/env python3
"""Round 5: where the time of one host-pointer ccgp_predict_batch call goes (Ground-Vibrations table: 1000 draws x 150 test
sites at n = 50, d = 9, K = 2): the whole call, the kernels alone on resident inputs, and the copies a synchronous call with
pageable host buffers cannot avoid -- the draws host -> device, the two S x m tables device -> host, and the host copy out of
the pinned buffer -- each measured by itself."""
import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ccgp_amd  # noqa
from ccgp_amd import api
import bench


def timed(f, reps, sync):
    for _ in range(max(3, reps // 5)):
        f()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    sync()
    return 1e6 * (time.perf_counter() - t0) / reps


def main():
    h = api.Handle(0)
    sets, P5 = bench.cfg5_inputs()
    dev0 = torch.device("cuda", 0)
    f64 = dict(dtype=torch.float64, device=dev0)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    S = P5.shape[0]
    out = []
    for si in (0, len(sets) - 1):
        Xs, ys, Xts = sets[si]
        n, d = Xs.shape
        m = Xts.shape[0]
        K = 2
        row = {"set": si, "n": n, "d": d, "S": S, "m": m}
        row["us_call_host_pointers"] = timed(lambda: h.predict_batch(Xs, ys, K, P5, Xts, 1.0), 100, torch.cuda.synchronize)
        # the C entry point with result arrays that exist already (no allocation, no page faults, no wrapper conversions)
        from ctypes import c_double as cd
        L = api.lib()
        Xf, yf, Pf, Xtf = (np.asfortranarray(a, dtype=np.float64) for a in (Xs, ys, P5, Xts))
        omean, ovar = np.zeros((S, m), order="F"), np.zeros((S, m), order="F")
        obeta, ost = np.zeros(S), np.zeros(S, dtype=np.int32)
        def c_call():
            rc = L.ccgp_predict_batch(h._h, api._p(Xf), n, d, api._p(yf), K, api._p(Pf), S, api._p(Xtf), m, cd(1.0),
                                      api._p(omean), api._p(ovar), api._p(obeta), api._ipt(ost))
            assert rc >= 0
        row["us_c_call_reused_result_arrays"] = timed(c_call, 200, torch.cuda.synchronize)
        def alloc_touch():
            a, b = np.empty((S, m), order="F"), np.empty((S, m), order="F")
            a.reshape(-1, order="F")[::512] = 0.0
            b.reshape(-1, order="F")[::512] = 0.0
        row["us_allocating_and_touching_fresh_result_arrays"] = timed(alloc_touch, 200, lambda: None)
        dX, dy, dXt = bench.col_major(torch, Xs, **f64), torch.tensor(ys, **f64), bench.col_major(torch, Xts, **f64)
        dP = bench.col_major(torch, P5, **f64)
        mean, var = torch.empty(S * m, **f64), torch.empty(S * m, **f64)
        bt, st = torch.empty(S, **f64), torch.zeros(S, dtype=torch.int32, device=dev0)
        run = lambda: h.predict_batch_dev(dX, n, d, dy, K, dP, S, dXt, m, 1.0, mean, var, bt, st)
        row["us_kernels_back_to_back"] = timed(run, 200, torch.cuda.synchronize)
        row["us_kernels_one_call_then_sync"] = timed(lambda: (run(), torch.cuda.synchronize()), 200, torch.cuda.synchronize)
        # the copies, by themselves
        in_bytes = 8 * (Xs.size + ys.size + P5.size + Xts.size)
        out_bytes = 8 * (2 * S * m + S) + 4 * S
        pin_in = torch.empty(in_bytes, dtype=torch.uint8).pin_memory()
        dev_in = torch.empty(in_bytes, dtype=torch.uint8, device=dev0)
        pin_out = torch.empty(out_bytes, dtype=torch.uint8).pin_memory()
        dev_out = torch.empty(out_bytes, dtype=torch.uint8, device=dev0)
        row["bytes_in"], row["bytes_out"] = in_bytes, out_bytes
        row["us_h2d_pinned"] = timed(lambda: (dev_in.copy_(pin_in, non_blocking=True), torch.cuda.synchronize()), 200, torch.cuda.synchronize)
        row["us_d2h_pinned"] = timed(lambda: (pin_out.copy_(dev_out, non_blocking=True), torch.cuda.synchronize()), 200, torch.cuda.synchronize)
        src = pin_out.numpy()
        dsts = [np.empty(out_bytes, dtype=np.uint8) for _ in range(4)]     # fresh result arrays, as a caller's would be
        k = [0]
        def host_copy():
            np.copyto(dsts[k[0] % 4], src)
            k[0] += 1
        row["us_host_copy_out_of_pinned"] = timed(host_copy, 200, lambda: None)
        hin = np.empty(in_bytes, dtype=np.uint8)
        row["us_host_copy_into_pinned"] = timed(lambda: np.copyto(pin_in.numpy(), hin), 200, lambda: None)
        row["us_sum_of_parts"] = (row["us_kernels_one_call_then_sync"] + row["us_h2d_pinned"] + row["us_d2h_pinned"] +
                                  row["us_host_copy_out_of_pinned"] + row["us_host_copy_into_pinned"])
        row["call_over_kernels"] = row["us_call_host_pointers"] / row["us_kernels_back_to_back"]
        row["c_call_over_kernels"] = row["us_c_call_reused_result_arrays"] / row["us_kernels_back_to_back"]
        row["parts_over_kernels"] = row["us_sum_of_parts"] / row["us_kernels_back_to_back"]
        print(json.dumps(row), flush=True)
        out.append(row)
    h.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r05_predict_e2e.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
