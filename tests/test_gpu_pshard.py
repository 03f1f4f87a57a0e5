"""ONE tree over several ranks INSIDE the persistent launch (tqgpu_pshard_*): the workgroups of the single-device
launch dealt over one launch per rank, talking through tagged words written into every rank's hand-over slab.  On the one GPU of
the test box the ranks are (a) n mirrors of one process -- n concurrent launches on n streams that wait for each other inside
the kernels, no host in the loop -- and (b) two PROCESSES whose slabs are mapped into each other through IPC handles, the
mechanism of a multi-GPU node (there the handles name peer memory across xGMI).  The arithmetic of a workgroup does not depend
on which launch it runs in, so the sharded solution equals the single-device one to the last bit."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from treeqp_amd import problems as P

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu(capi):
    if capi.device_count() < 1:
        pytest.fail("no HIP device visible: the -m gpu tests must run on the MI355X box")
    return capi


def _lti(gpu, p):
    nk = p.nk()
    nx = np.full(p.Nn, p.nx, dtype=np.int32)
    nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
    return nk, nx, nu, gpu.TreeQp(nx, nu, nk).fill_lti(p).flat()


@pytest.mark.parametrize("make", [lambda: P.linear_chain(2, 9, 9), lambda: P.linear_chain(2, 11, 11), lambda: P.linear_chain(2, 6, 6, ubound=0.05),
                                  lambda: P.linear_chain(2, 7, 7, nm=3)], ids=["c2_1023", "c3_4095", "127_tight_bounds", "255_nm3"])
def test_two_ranks_of_one_process_match_the_single_device_solve(gpu, make):
    p = make()
    nk, nx, nu, flat = _lti(gpu, p)
    g = gpu.TqGpu(nk, nx, nu).upload(flat, p.lambda0)
    assert g.path == 2
    ref_r, ref = g.solve(), g.solution()
    g.close()
    n = 2
    ms = [gpu.TqGpu(nk, nx, nu).upload(flat, p.lambda0).pshard_init(r, n) for r in range(n)]
    for rep in range(2):                                   # a second solve: launch numbers advance in step, the slabs are not reset
        rs = gpu.pshard_solve_local(ms)
        assert all((r["status"], r["iter"], r["ls_total"]) == (ref_r["status"], ref_r["iter"], ref_r["ls_total"]) for r in rs), rs
        for m in ms:
            sol = m.solution()
            for k in ("x", "u", "lam", "mu_x", "mu_u"):
                assert np.array_equal(sol[k], ref[k]), k
    for m in ms:
        m.close()


@pytest.mark.parametrize("levels,ranks", [(9, (4, 8)), (11, (4, 8))], ids=["c2_1023", "c3_4095"])
def test_four_and_eight_ranks_of_one_process(gpu, tmp_path, levels, ranks):
    """n launches that wait for each other must all be in flight: the HIP runtime gives a process four hardware queues by default and
    streams that share one run one after the other (the bounded waits then end the solve with TQGPU_ETIMEOUT) -- the worker process
    starts with GPU_MAX_HW_QUEUES=8."""
    for n in ranks:
        env = dict(os.environ)
        env["GPU_MAX_HW_QUEUES"] = str(2 * n)
        out = subprocess.run([sys.executable, str(ROOT / "tests" / "pshard_local_worker.py"), str(levels), str(n)], cwd=tmp_path, env=env,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
        assert f"{n} ranks: ok" in out.stdout


def test_launch_number_wrap_is_refused_until_every_rank_rewinds(gpu):
    """The launch number that tags the hand-over words has 16 bits.  A single device wipes its slab when it wraps; ranks that write
    into each other's slabs cannot do that on their own (a peer's early words would be wiped): after 65535 solves tqgpu_pshard_begin
    refuses, tqgpu_pshard_rewind on every rank (all idle) resets, and the solves continue with identical results."""
    p = P.linear_chain(2, 6, 6)
    nk, nx, nu, flat = _lti(gpu, p)
    g = gpu.TqGpu(nk, nx, nu).upload(flat, p.lambda0)
    ref_r, ref = g.solve(), g.solution()
    g.close()
    ms = [gpu.TqGpu(nk, nx, nu).upload(flat, p.lambda0).pshard_init(r, 2) for r in range(2)]
    gpu.pshard_solve_local(ms)                              # connects; solve 1
    done = 1
    with pytest.raises(RuntimeError, match="65535"):
        while done < 70000:
            ms[0].pshard_begin()
            ms[1].pshard_begin()
            r0, r1 = ms[0].pshard_end(), ms[1].pshard_end()
            assert (r0["status"], r0["iter"]) == (ref_r["status"], ref_r["iter"]) == (r1["status"], r1["iter"])
            done += 1
    assert done == 65535
    for m in ms:
        m.pshard_rewind()
    rs = gpu.pshard_solve_local(ms)
    assert all((r["status"], r["iter"], r["ls_total"]) == (ref_r["status"], ref_r["iter"], ref_r["ls_total"]) for r in rs)
    for m in ms:
        sol = m.solution()
        assert all(np.array_equal(sol[k], ref[k]) for k in ("x", "u", "lam", "mu_x", "mu_u"))
        m.close()


def test_pshard_rejects_what_it_cannot_shard(gpu):
    f = P.irregular_clipping_qp()
    g = gpu.TqGpu(f.nk, f.nx, f.nu)
    with pytest.raises(RuntimeError, match="persistent"):
        g.pshard_init(0, 2)
    g.close()
    p = P.linear_chain(2, 3, 3)
    nk, nx, nu, flat = _lti(gpu, p)
    g = gpu.TqGpu(nk, nx, nu)
    with pytest.raises(RuntimeError, match="too small"):
        g.pshard_init(0, 8)
    g.close()


@pytest.mark.parametrize("nproc,levels", [(2, 9), (4, 11)], ids=["2_processes_c2", "4_processes_c3"])
def test_processes_share_one_tree_through_ipc_slabs(gpu, tmp_path, nproc, levels):
    """The product's sharded solve across REAL processes: 2 / 4 processes on this GPU, each a rank; slabs mapped into each other by
    hipIpcGetMemHandle / hipIpcOpenMemHandle; handles and, afterwards, the solution shares travel through gloo."""
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "pshard_worker.py"), str(r), str(nproc), str(port), str(levels)], cwd=tmp_path, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(nproc)]
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            pr.kill()
            out, _ = pr.communicate()
            out += "\n[timeout]"
        outs.append(out)
    assert all(pr.returncode == 0 for pr in procs), "\n---\n".join(o[-3000:] for o in outs)
    assert all("ok" in o for o in outs)
