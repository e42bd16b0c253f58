"""Development aid (GPU box): arbitrates serial aggregation cases of tools/serial_sweep.py in which the GPU and the oracle end
further apart than the suite's tolerance.  For each case (regenerated from the sweep's own generator) four trajectories on
the same inputs: the serial GPU walk, the fp32 oracle, the float64 model (tests/f64_model.py) and the fp32 oracle started one
ulp away in ONE table element.  If the GPU is no further from the float64 trajectory than the oracle is, and a one-ulp
perturbation moves the oracle by as much as the two differ, the difference is conditioning, not an indexing error.
usage: python tools/serial_arbiter.py <sweep cases> <sweep seed> <case,case,...>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


from tests.serial_cases import dist, run_f64, run_gpu, run_oracle, sweep_cases  # noqa: E402


if __name__ == "__main__":
    cases, seed = int(sys.argv[1]), int(sys.argv[2])
    want = [int(x) for x in sys.argv[3].split(",")]
    for case, c in sweep_cases(cases, seed):
        if case not in want:
            continue
        g = run_gpu(c)
        o, f, p = run_oracle(c), run_f64(c), run_oracle(c, nudge=True)
        print(f"case {case}: d={c['d']} N={c['N']} U={c['U']} I={c['I']} T={c['T']} agg={c['agg']} {g[3]}  (distances in units of the 3e-4 tolerance)")
        print(f"   gpu-oracle {dist(g[:3], o):9.3g}   gpu-f64 {dist(g[:3], f):9.3g}   oracle-f64 {dist(o, f):9.3g}   oracle-(oracle + 1 ulp) {dist(p, o):9.3g}")
        for steps in (1, 8, 32, 33, 64, c["T"]):
            if steps <= c["T"]:
                gs, os_, fs = run_gpu(c, steps), run_oracle(c, steps), run_f64(c, steps)
                print(f"   after {steps:3d} steps: gpu-oracle {dist(gs[:3], os_):9.3g}  gpu-f64 {dist(gs[:3], fs):9.3g}  oracle-f64 {dist(os_, fs):9.3g}", flush=True)
