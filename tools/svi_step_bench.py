"""Times the pieces of SparseVI's enqueued ADAM step on the c5 shapes (D = 301, S = 256, k points), each piece alone in a loop
of 200 back-to-back launches and the whole step, with HIP events on the launch stream.  python tools/svi_step_bench.py [k]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
import torch
import bayesiancoresets_amd as bc

k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
D, S, N, T = 301, 256, 200_000, 200
rs = np.random.RandomState(0)
Z = torch.randn(N, D + 1, dtype=torch.float64, device="cuda")
mu0, Sig0, sigsq = np.zeros(D), 30.0 * np.eye(D), 0.02
smp = bc.LinregPosteriorSampler(mu0, Sig0, sigsq, seed=1)
prj = bc.DeviceProjector("linreg", smp, S, sigsq=sigsq, colsum="moments")
pts = Z[:k].cpu().numpy()
core = prj._dev(pts)
w = torch.from_numpy(np.abs(rs.randn(k))).cuda()
plan = smp.enqueue_plan(S, pts, T)
theta, mean = plan.draw(w, 0)
prj.use_draws(theta, mean=mean)
prj.colsum_and_core_enqueue(Z, core)
prj.colsum_and_core_enqueue(Z, core)          # (second sight: the moments are formed)
state = torch.zeros(3 * k + 3 * T, dtype=torch.float64, device="cuda")
state[3 * k:] = 1.0
lib = prj._lib


def timed(name, fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(T):
        fn(i)
    e1.record()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%-28s %7.2f us/step on the stream   %6.2f us/step of host time to enqueue" % (name, e0.elapsed_time(e1) * 1e3 / T, th * 1e6 / T))


run, buf, _ = prj.enqueue_step_plan(Z, core, True, *plan.buffers())
adam_fn, adam_args = lib.bcx_sparsevi_adam_step, [prj._stream(), k, S, buf.data_ptr(), 1.0, buf[S:].data_ptr(), S, state.data_ptr(), state[k:].data_ptr(),
                                                  state[2 * k:].data_ptr(), state[3 * k:].data_ptr(), 0, 0.9, 0.999, 1e-8, None, 1]
core_args = prj._common(core) + [buf[S:].data_ptr(), S]


def draw(i):
    plan.draw(w, i)


def colsum(i):
    prj._colsum_from_moments(Z, out=buf[:S], tbar=mean)


def corep(i):
    lib.bcx_project_write_raw(*core_args)


def adam(i):
    adam_args[11] = i
    adam_fn(*adam_args)


def step(i):
    plan.draw(w, i)
    run()
    adam(i)


for name, fn in (("draw", draw), ("colsum (closed form)", colsum), ("coreset projection", corep), ("adam", adam), ("whole step", step)):
    timed(name, fn)
    timed(name, fn)
if os.environ.get("BCX_SVI_DBG") == "9":
    torch.cuda.synchronize()
    plan.fast, plan._args = False, None         # (the per-call MFMA kernel: the one with the time stamps)
    for rep in range(3):
        plan.draw(w, 0)
        torch.cuda.synchronize()
        t = plan.rbar[1, :64].cpu().numpy().reshape(8, 8)
        np.set_printoptions(linewidth=200, precision=2, suppress=True)
        print("workgroups (0..7, 2): start | us since start: loads issued+landed, k x k inputs, factor, chunk 0, chunk 1, stored | clock ticks / us")
        t[:, 0] -= t[:, 0].min()
        print(t)
if os.environ.get("BCX_MQ_DBG"):
    torch.cuda.synchronize()
    for rep in range(3):
        colsum(0)
        torch.cuda.synchronize()
        nct, Spad = (D + 15) // 16, (S + 15) // 16 * 16
        print("quad, last workgroup: us since its start: products, partials stored, counter, closing loads, end:",
              prj._mom_work[nct * Spad + 1: nct * Spad + 7].cpu().numpy()[1:])
