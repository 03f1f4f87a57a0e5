"""Worker of tests/test_gpu_pshard.py: n ranks of ONE process on one GPU with as many hardware queues as ranks (GPU_MAX_HW_QUEUES is
read when the HIP runtime starts, hence a process of its own): n launches on n streams that wait for each other inside the kernels
must all be in flight together, and streams that share a hardware queue run one after the other."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    levels, ranks = int(sys.argv[1]), [int(a) for a in sys.argv[2:]]
    from treeqp_amd import capi, problems as P
    p = P.linear_chain(2, levels, levels)
    nk = p.nk()
    nx = np.full(p.Nn, p.nx, dtype=np.int32)
    nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
    flat = capi.TreeQp(nx, nu, nk).fill_lti(p).flat()
    g = capi.TqGpu(nk, nx, nu).upload(flat, p.lambda0)
    ref_r, ref = g.solve(), g.solution()
    g.close()
    for n in ranks:
        ms = [capi.TqGpu(nk, nx, nu).upload(flat, p.lambda0).pshard_init(r, n) for r in range(n)]
        for rep in range(2):
            rs = capi.pshard_solve_local(ms)
            assert all((r["status"], r["iter"], r["ls_total"]) == (ref_r["status"], ref_r["iter"], ref_r["ls_total"]) for r in rs), (n, rs)
            for m in ms:
                sol = m.solution()
                for k in ("x", "u", "lam", "mu_x", "mu_u"):
                    assert np.array_equal(sol[k], ref[k]), (n, k)
        for m in ms:
            m.close()
        print(f"{n} ranks: ok ({ref_r['iter']} iterations, bit-identical to the single-device solve)", flush=True)


if __name__ == "__main__":
    main()
