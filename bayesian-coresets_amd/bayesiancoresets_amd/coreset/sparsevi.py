"""Sparse variational-inference coreset (reference: bayesiancoresets/coreset/sparsevi.py:6-79).

Each greedy step re-projects the whole data set at the current weighted posterior, picks the point
most correlated with the residual, then runs ``opt_itrs`` projected-ADAM steps on the weights, each
of which needs the column sums of a fresh projection.  The host loop below is the reference's; the
N-sized work runs on the GPU:

* with a ``DeviceProjector`` nothing of size N x S is ever stored: ``project_colsum`` (gradient,
  sparsevi.py:70-74) and ``project_select`` (correlation arg-max, sparsevi.py:49-56) are fused
  projection kernels (csrc/proj.hip);
* with any other projector (user callbacks produce a host array) the projected vectors are handed to
  the same device engine as the Hilbert coresets: one ingest pass gives the row norms and column sums,
  one correlation scan gives the arg-max (``bcx_argmax_correlation``).
Random subsampling (``n_subsample_select`` / ``n_subsample_opt``) follows sparsevi.py:32-35.
"""
import numpy as np

from .coreset import Coreset
from ..util.opt import nn_opt
from ..projector import DeviceProjector
from .. import _native as nat


class SparseVICoreset(Coreset):
    def __init__(self, data, ll_projector, n_subsample_select=None, n_subsample_opt=None, opt_itrs=100,
                 step_sched=lambda i: 1.0 / (1.0 + i), *, row_offset=0, group=None, **kw):
        """``row_offset`` / ``group`` (keyword-only extension): row-sharded construction, one process per
        GPU.  ``data`` is then this rank's contiguous block of rows starting at global row
        ``row_offset``, the projector must be a ``DeviceProjector`` built with the same ``group`` and
        ``row_offset``, and every rank must seed NumPy identically (the sampler is replicated)."""
        self.row_offset, self.group = int(row_offset), group
        self._sharded = group is not None
        if self._sharded and not isinstance(ll_projector, DeviceProjector):
            raise ValueError("row-sharded SparseVI needs a DeviceProjector")
        self.data = data
        self.ll_projector = ll_projector
        n = data.shape[0]
        if self._sharded:
            # the subsample sizes are capped by the GLOBAL row count (sparsevi.py:12-13), known to every rank
            import torch
            import torch.distributed as dist
            t = torch.tensor([float(n)], dtype=torch.float64, device=ll_projector.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            n = int(t.item())
        self.n_global = n
        self.n_subsample_select = None if n_subsample_select is None else min(n, n_subsample_select)
        self.n_subsample_opt = None if n_subsample_opt is None else min(n, n_subsample_opt)
        self.step_sched = step_sched
        self.opt_itrs = opt_itrs
        self._engine = None
        super().__init__(**kw)
        self.pts = np.zeros((0, data.shape[1]))

    def reset(self):
        super().reset()
        self.pts = np.zeros((0, self.data.shape[1]))

    def _build(self, itrs):
        for _ in range(itrs):
            self._select()
            self._optimize()

    # ---- N-sized pieces ------------------------------------------------------------
    def _subsample(self, n_subsample):
        """(positions-or-None, drawn global rows-or-None, local points, scaling).  Row-sharded: every rank draws the
        same indices (replicated NumPy stream) and keeps the drawn rows it owns, in draw order; ``positions`` are
        their places in the drawn array (the identity that decides arg-max ties, sparsevi.py:55-57)."""
        if n_subsample is None:
            return None, None, self.data, 1.0
        sub = np.random.randint(self.n_global, size=n_subsample)                 # sparsevi.py:33
        scaling = self.n_global / n_subsample
        if not self._sharded:
            return None, sub, self.data[sub], scaling
        lo = self.row_offset
        pos = np.flatnonzero((sub >= lo) & (sub < lo + self.data.shape[0]))
        return pos, sub, self.data[sub[pos] - lo], scaling

    def _engine_for(self, vecs):
        """Hand host-resident projected vectors to the device engine: norms + column sums in one pass."""
        n, s = vecs.shape
        eng = self._engine
        if eng is None or eng.n_local != n or eng.d != s:
            if eng is not None:
                eng.close()
            eng = nat.Engine(nat.ALG_FW, n, s, keep_exact_rows=True)
            self._engine = eng
        eng.use_current_stream()
        eng.load_rows_any(vecs)
        rc = eng.finalize(None)
        if rc not in (nat.OK, nat.ERR_ZERO_ROW):
            raise nat.EngineError(rc, eng.lib.bcx_last_error(eng.h).decode())
        # a zero projected vector: norms and column sums are valid, the correlation scan is not (the row's
        # correlation is 0/0); _select treats it as the reference's arithmetic does
        eng.has_zero_row = rc == nat.ERR_ZERO_ROW
        return eng

    def _core_points_device(self):
        """The coreset points as a device tensor, re-uploaded only when ``self.pts`` changed (it is replaced, never edited
        in place, by _select / reset): every ADAM step projects the same points (sparsevi.py:38-39)."""
        if self.pts.shape[0] == 0:
            return None
        c = getattr(self, "_core_dev", None)
        if c is None or c[0] is not self.pts:
            c = self._core_dev = (self.pts, self.ll_projector._dev(self.pts))
        return c[1]

    def _corevecs(self):
        if self.pts.shape[0] == 0:
            return None
        cv = self.ll_projector.project(self.pts)
        return cv.cpu().numpy() if hasattr(cv, "cpu") else np.asarray(cv)

    def _residual(self, n_subsample, w):
        """(resid, sub_idcs, points, engine-or-None, corevecs) after updating the projector at (w, pts)."""
        self.ll_projector.update(w, self.pts)                                     # sparsevi.py:25
        pos, sub, pts, scaling = self._subsample(n_subsample)
        eng = None
        if isinstance(self.ll_projector, DeviceProjector):
            # both projections of sparsevi.py:35-41 (data: column sums only; coreset points: the vectors), one read-back
            colsum, corevecs = self.ll_projector.colsum_and_core(pts, self._core_points_device(), persistent=pts is self.data)
        else:
            vecs = self.ll_projector.project(pts)
            eng = self._engine_for(np.ascontiguousarray(vecs))
            colsum = eng.vector(0)                                                # vecs.sum(axis=0)
            corevecs = self._corevecs()
        S = colsum.shape[0]
        if corevecs is None:
            corevecs = np.zeros((0, S))
        resid = scaling * colsum - w.dot(corevecs)                                # sparsevi.py:47 / :72
        return resid, (pos, sub), pts, eng, corevecs

    # ---- sparsevi.py:44-67 ------------------------------------------------------------
    def _select(self):
        resid, (pos, sub), pts, eng, corevecs = self._residual(self.n_subsample_select, self.wts)
        S = resid.shape[0]
        if eng is None:
            best, row = self.ll_projector.project_select(pts, resid, row_ids=pos)
        elif eng.has_zero_row:
            # corrs = vecs.dot(resid) / ||vecs|| / S is NaN at a zero row and NumPy's argmax returns the first NaN
            # (sparsevi.py:49-56): that row is the pick, and NaN > x is False, so it only enters an EMPTY coreset
            row, best = int(np.flatnonzero(eng.norms() == 0.0)[0]), float("nan")
        else:
            row, score = eng.argmax_correlation(resid)                            # An . resid
            best = score / S
        if corevecs.shape[0]:
            corecorrs = np.fabs(corevecs.dot(resid) / np.sqrt((corevecs ** 2).sum(axis=1))) / S
        else:
            corecorrs = np.zeros(0)
        if corecorrs.size == 0 or best > corecorrs.max():                         # sparsevi.py:56
            f = int(sub[row]) if sub is not None else int(row)
            if f not in self.idcs:                                                # sparsevi.py:60
                self.wts = np.append(self.wts, 0.0)
                self.idcs = np.append(self.idcs, f).astype(np.int64)
                self.pts = np.vstack((self.pts, self._fetch_row(f)[None, :]))

    @staticmethod
    def _host_row(row):
        return row.detach().cpu().numpy() if hasattr(row, "detach") else np.asarray(row)

    def _fetch_row(self, f):
        """data[f] for a global row index (``data`` may be an ndarray or a device tensor); in sharded mode the
        owning rank supplies it."""
        if not self._sharded:
            return self._host_row(self.data[f])
        import torch
        import torch.distributed as dist
        lo = self.row_offset
        buf = torch.zeros(self.data.shape[1], dtype=torch.float64, device=self.ll_projector.device)
        if lo <= f < lo + self.data.shape[0]:
            buf.copy_(torch.as_tensor(np.ascontiguousarray(self._host_row(self.data[f - lo]), dtype=np.float64)))
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
        return buf.cpu().numpy()

    # ---- sparsevi.py:69-76 ------------------------------------------------------------
    def _optimize(self):
        plan = self._enqueue_plan()
        if plan is not None:
            self.wts = self._optimize_enqueued(plan)
            return

        def grd(w):
            resid, sub, pts, eng, corevecs = self._residual(self.n_subsample_opt, w)
            return -corevecs.dot(resid) / corevecs.shape[1]
        self.wts = nn_opt(self.wts, grd, opt_itrs=self.opt_itrs, step_sched=self.step_sched)

    # ---- the same loop with the weights resident on the device (csrc/svi.hip) -----------------------------------------------
    ENQUEUE = True      # False: always the host loop above (tests compare the two)

    def _enqueue_plan(self):
        """A draw plan when the whole ADAM loop can be enqueued: device projector, the full data set at every step
        (a per-step sub-sample is drawn on the host, sparsevi.py:33), a non-empty coreset, and a sampler that can draw from
        device-resident weights (``enqueue_plan``: ``bc.LinregPosteriorSampler``).  Same decision on every rank."""
        prj = self.ll_projector
        if not (self.ENQUEUE and isinstance(prj, DeviceProjector) and self.n_subsample_opt is None and self.opt_itrs > 0
                and 0 < self.wts.shape[0] <= 4096 and prj.projection_dimension <= 8192):
            return None
        make = getattr(prj.sampler, "enqueue_plan", None)
        return None if make is None else make(prj.projection_dimension, self.pts, self.opt_itrs)

    def _optimize_enqueued(self, plan, b1=0.9, b2=0.999, eps=1e-8):
        """nn_opt (util/opt.py:4-28) with grd = sparsevi.py:69-76, enqueued: per step the sampler's draw kernel at the current
        device weights, the two projections (column sums of the data, the coreset points), and one kernel for
        resid / gradient / ADAM moments / step / clamp.  The host evaluates the schedule up front and reads the weights once."""
        prj = self.ll_projector
        torch = prj._torch
        k, S, T = self.wts.shape[0], prj.projection_dimension, self.opt_itrs
        sched = np.array([(self.step_sched(i), 1.0 - b1 ** (i + 1), 1.0 - b2 ** (i + 1)) for i in range(T)], dtype=np.float64)
        state = torch.from_numpy(np.concatenate((np.asarray(self.wts, dtype=np.float64), np.zeros(2 * k), sched.ravel()))).to(prj.device)
        w, m1, m2, sched_d = state[:k], state[k:2 * k], state[2 * k:3 * k], state[3 * k:]
        theta, mean = plan.buffers()
        run, buf, _ = prj.enqueue_step_plan(self.data, self._core_points_device(), True, theta, mean)    # sparsevi.py:35-41
        # (more than 16 weights: the ADAM step is two launches over slabs of weights and needs scratch, csrc/svi.hip)
        need = int(prj._lib.bcx_sparsevi_adam_scratch_bytes(k, S))
        work = torch.empty(max(need // 8, 1), dtype=torch.float64, device=prj.device)
        adam, args = prj._lib.bcx_sparsevi_adam_step_ws, [prj._stream(), k, S, buf.data_ptr(), 1.0, buf[S:].data_ptr(), S, w.data_ptr(),
                                                          m1.data_ptr(), m2.data_ptr(), sched_d.data_ptr(), 0, b1, b2, eps, None, 1,
                                                          work.data_ptr(), work.numel() * 8]
        for i in range(T):
            plan.draw(w, i)                                                       # sparsevi.py:25
            run()
            args[11] = i
            prj._check(adam(*args))
        out = w.cpu().numpy()
        check = getattr(plan, "check", None)
        if check is not None:
            check()                                                               # (a factorisation that lost its workgroups raises)
        return out

    def error(self):
        return 0.0   # as in the reference (sparsevi.py:78-79: KL estimate not implemented)
