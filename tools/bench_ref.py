"""Times the reference-stream kernels (SMMC_FLAG_STREAM_REF, smmc_ref_kernels.hip) with HIP events on the
engine's stream: the windowed kernel at 360 periods (and the tree kernel forced onto the same shape), the tree
kernel at 700, 1000, 1300 and 1816 periods beside the generic kernel forced onto them, the generic kernel at 2000, and the
default Philox table stream on the same shapes.  One JSON line per case.

usage: bench_ref.py [n_paths_360] [n_paths_1000]   |   bench_ref.py --traj
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import stock_market_monte_carlo_amd as S  # noqa: E402
from tests.conftest import load_table  # noqa: E402


def traj_cases():
    """Reference-stream keepdata (mc_simulations_keepdata's own stream, src/simulations.cpp:175-186): the windowed kernel
    at 4e6 x 361 values and the tree kernel at 1.5e6 x 1001, the Philox table stream's keepdata beside them.  HBM-bound:
    4 (P + 1) bytes per path."""
    import ctypes as C
    import torch
    from stock_market_monte_carlo_amd import _lib
    table = load_table()
    cases = (("ref windowed traj", 4_000_000, 360, "ref", ""), ("ref windowed traj", 4_000_000, 360, "ref", "8"),
             ("ref windowed traj", 4_000_000, 360, "ref", "4"), ("ref windowed traj", 4_000_000, 360, "ref", "2"),
             ("ref windowed traj", 4_000_000, 360, "ref", "1"), ("philox table traj", 4_000_000, 360, 3, ""),
             ("ref tree traj", 1_500_000, 1000, "ref", ""), ("ref tree traj", 1_500_000, 1000, "ref", "8"),
             ("ref tree traj", 1_500_000, 1000, "ref", "4"), ("ref tree traj", 1_500_000, 1000, "ref", "2"),
             ("ref tree traj", 1_500_000, 1000, "ref", "1"), ("philox table traj", 1_500_000, 1000, 3, ""),
             ("ref windowed traj", 16_000_000, 360, "ref", ""))
    for name, n, p, stream, rows in cases:
        if rows:
            os.environ["SMMC_REF_TRAJ_ROWS"] = rows  # read at every call; "" = the library's own choice
        else:
            os.environ.pop("SMMC_REF_TRAJ_ROWS", None)
        eng = S.Engine(0)
        eng.set_table(table)
        sim = S.Engine.make_sim(n, p, S.MODE_TABLE, 1000, stream=stream)
        traj = torch.empty((n, p + 1), dtype=torch.float32, device=eng.tdevice)
        final = torch.empty(n, dtype=torch.float32, device=eng.tdevice)

        def run():
            eng._enter()
            _lib.check(eng._L.smmc_engine_simulate_keepdata(eng._h, C.byref(sim), C.c_void_p(traj.data_ptr()), C.c_void_p(final.data_ptr())))

        for _ in range(3):
            run()
        eng.sync()
        rounds = []
        for _ in range(5):
            eng.timing(True)
            for _ in range(10):
                run()
            ms, k = eng.kernel_ms()
            eng.timing(False)
            rounds.append(ms / 10)
        ms = sorted(rounds)[2]
        b = 4.0 * n * (p + 1)
        print(json.dumps({"case": name, "rows_per_stream": rows or "auto", "n_paths": n, "n_periods": p, "kernel_ms": ms, "bytes": b, "TBps": b / ms / 1e9,
                          "frac_of_8TBps": b / ms / 1e9 / 8.0, "rounds_ms": rounds,
                          "divide": ("fast", "exact", "checked")[eng.divide_kind(sim, keepdata=True)],
                          "last_column_mean": float(traj[:, -1].double().mean()), "final_mean": float(final.double().mean())}), flush=True)
        eng.close()


def main():
    import torch
    if len(sys.argv) > 1 and sys.argv[1] == "--traj":
        return traj_cases()
    n360 = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    n1000 = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
    table = load_table()
    cases = (("ref windowed", n360, 360, "ref", "auto"), ("ref tree (forced)", n360, 360, "ref", "tree"),
             ("philox table", n360, 360, 3, "auto"),
             ("ref tree", n1000, 700, "ref", "auto"), ("ref generic (forced)", n1000, 700, "ref", "generic"),
             ("ref tree", n1000, 1000, "ref", "auto"), ("ref generic (forced)", n1000, 1000, "ref", "generic"),
             ("philox table", n1000, 1000, 3, "auto"),
             ("ref tree", n1000, 1300, "ref", "auto"), ("ref generic (forced)", n1000, 1300, "ref", "generic"),
             ("ref tree", n1000, 1816, "ref", "auto"), ("ref generic (forced)", n1000, 1816, "ref", "generic"),
             ("ref generic", n1000, 2000, "ref", "auto"))
    for name, n, p, stream, knob in cases:
        os.environ["SMMC_REF_KERNEL"] = knob  # read when an engine is created and (tree) at every launch
        eng = S.Engine(0)
        eng.set_table(table)
        sim = S.Engine.make_sim(n, p, S.MODE_TABLE, 1000, stream=stream)
        out = torch.empty(n, dtype=torch.float32, device=eng.tdevice)
        eng.simulate(sim, out=out)
        eng.sync()
        eng.timing(True)
        reps = 5
        for _ in range(reps):
            eng.simulate(sim, out=out)
        ms, k = eng.kernel_ms()
        eng.timing(False)
        ms /= reps  # per simulate call: the windowed / tree launch, its redo launch and (generic) every sub-launch
        print(json.dumps({"case": name, "n_paths": n, "n_periods": p, "kernel_ms": ms, "paths_per_s": n / ms * 1e3,
                          "hbm_GBps_final_values": 4.0 * n / ms / 1e6, "divide": ("fast", "exact", "checked")[eng.divide_kind(sim)],
                          "mean": float(out.double().mean())}), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
