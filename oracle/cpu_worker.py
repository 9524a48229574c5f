"""TEST INFRASTRUCTURE -- one worker of bench.py's multi-process CPU baseline.

Runs the oracle (operation-by-operation port of the reference's numpy/scipy path) for one
synthetic baseline with ONE BLAS thread and prints the seconds its iterations took.  bench.py
starts P of these side by side, the reference's own deployment model (one baseline stream per
MPI rank, `jobscript.sh.template:4-5`).  Never imported by the product.

    python -m oracle.cpu_worker N T M flag_frac k0 niter
"""
import os
import sys

for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ[v] = "1"


def main():
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from hydra_pspec_amd import synthetic          # numpy-only input generator (no GPU touched)
    from oracle import pspec_ref
    N, T, M = (int(x) for x in sys.argv[1:4])
    frac, k0, niter = float(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    d = synthetic.make_baselines(N, T, M, k0=k0, nbl=1, flag_frac=frac)
    t0 = time.perf_counter()
    pspec_ref.gibbs_sample_with_fg(d["vis"][0], d["flags"][0], d["S_initial"], d["fgmodes"], d["Ninv"],
                                   d["ps_prior"], Niter=niter, seed=d["seed"])
    print(f"CPU_WORKER_SECONDS {time.perf_counter() - t0:.6f}", flush=True)


if __name__ == "__main__":
    main()
