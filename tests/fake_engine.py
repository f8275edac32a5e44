"""CPU stand-in for bayesiancoresets_amd._native.Engine, for exercising the multi-process
orchestration in bayesiancoresets_amd/sharded.py under gloo WITHOUT a GPU.  Test infrastructure
only: the arithmetic comes from the oracle (oracle/snnls_oracle.py).  It implements the engine
protocol at the tensor level: local scan -> record, replicated apply."""
import numpy as np
import torch

from oracle.snnls_oracle import SnnlsOracle

REC_HDR = 4
CHUNK = 1024
ALG_NAME = {0: "giga", 1: "fw", 2: "omp"}


class FakeEngine(object):
    FULL = None  # the whole N x d matrix (every rank may read it for the replicated reweight only)

    def __init__(self, alg, n_local, d, n_global=None, row_offset=0, rank=0, world_size=1, device=0, **kw):
        self.alg, self.n_local, self.d = ALG_NAME[alg], n_local, d
        self.n_global, self.row_offset, self.rank, self.world = n_global, row_offset, rank, world_size
        self.rows = np.zeros((n_local, d))
        self.itrs = self.done = 0
        self._trace = []

    def tensor_device(self):
        return torch.device("cpu")

    def load_rows_any(self, rows, row_begin=0, center=False):
        rows = rows.numpy() if isinstance(rows, torch.Tensor) else np.asarray(rows)
        if center:
            rows = rows - rows.mean(axis=1)[:, None]
        self.rows[row_begin:row_begin + rows.shape[0]] = rows

    def export_chunk_sums_tensor(self, t, cap_chunks):
        out = t.view(cap_chunks, self.d + 1)
        for c in range((self.n_local + CHUNK - 1) // CHUNK):
            blk = self.rows[c * CHUNK:(c + 1) * CHUNK]
            out[c, :self.d] = torch.from_numpy(blk.sum(axis=0))
            out[c, self.d] = float(np.sqrt((blk ** 2).sum(axis=1)).sum())

    def finalize_any(self, b, gathered, n_gathered):
        self.saw_gathered = gathered is not None      # (sum of row norms needs every shard's chunk sums, also with an explicit b)
        if b is None:
            if gathered is not None:
                g = gathered.view(-1, self.d + 1)[:n_gathered].numpy()
            else:
                t = torch.zeros(((self.n_local + CHUNK - 1) // CHUNK) * (self.d + 1), dtype=torch.float64)
                self.export_chunk_sums_tensor(t, (self.n_local + CHUNK - 1) // CHUNK)
                g = t.view(-1, self.d + 1).numpy()
            acc = np.zeros(self.d + 1)
            for row in g:           # fixed global chunk order
                acc = acc + row
            b = acc[:self.d].copy()
        self.b = np.asarray(b, dtype=np.float64)
        if np.any(np.sqrt((self.rows ** 2).sum(axis=1)) == 0):
            return -3
        X = FakeEngine.FULL
        self.oracle = SnnlsOracle(X.T, self.b, alg=self.alg, mode="onepass")
        self.norms = np.sqrt((self.rows ** 2).sum(axis=1))
        self.An = self.rows / self.norms[:, None]
        return 0

    def build_begin(self, itrs, tol):
        self.itrs, self.done, self._trace = itrs, 0, []
        return self.oracle.reached_numeric_limit

    def step_scan_tensor(self, send, exact=False):
        o = self.oracle
        rec = np.zeros(self.d + REC_HDR)
        if self.done >= self.itrs or self.n_local == 0:
            rec[1] = -1
        else:
            xw = o._Aw()
            if self.alg == "giga":
                nw = np.sqrt((xw ** 2).sum()); nw = 1.0 if nw == 0 else nw
                xh = xw / nw
                cdir = o.bn - o.bn.dot(xh) * xh
                cdir /= np.sqrt((cdir ** 2).sum())
                s0, s1 = self.An.dot(cdir), self.An.dot(xh)
                ok = np.logical_and(s1 > -1.0 + 1e-14, 1.0 - s1 ** 2 > 0.0)
                den = np.where(ok, np.sqrt(np.where(ok, 1.0 - s1 ** 2, 1.0)), np.inf)
                scores = s0 / den
            else:
                scores = self.An.dot(o.b - xw)
            i = int(scores.argmax())
            rec[0], rec[1], rec[2], rec[3] = scores[i], self.row_offset + i, self.norms[i], 1.0
            rec[REC_HDR:] = self.rows[i]
        send.copy_(torch.from_numpy(rec))

    def step_apply_tensor(self, recv):
        if self.done >= self.itrs:
            return
        recs = recv.view(-1, self.d + REC_HDR).numpy()
        valid = [r for r in recs if r[3] == 1.0]
        best = valid[0]
        for r in valid[1:]:
            if r[0] > best[0] or (r[0] == best[0] and r[1] < best[1]):
                best = r
        f = int(best[1])
        o = self.oracle
        if self.alg == "omp" and o.size() > 0:
            # negative direction over the replicated active set (orthopursuit.py:27-35)
            resid = o.b - o._Aw()
            act = np.flatnonzero(o.w > 0)
            neg = -(o.An[:, act].T.dot(resid))
            j = int(neg.argmax())
            if not (best[0] >= neg[j]):
                f = int(act[j])
        o._select = lambda: f
        t = o.build(1)
        self._trace.extend(t)
        self.done += 1

    def poll(self):
        return self.done, False, self.oracle.reached_numeric_limit

    def trace(self, cap):
        sel = np.array([t[0] for t in self._trace], dtype=np.int64)
        err = np.array([t[1] for t in self._trace])
        st = np.array([t[2] for t in self._trace], dtype=np.int32)
        return sel, err, st

    def sparse_weights(self):
        idx = np.flatnonzero(self.oracle.w != 0)
        return idx.astype(np.int64), self.oracle.w[idx]

    def error(self):
        return self.oracle.error()

    def optimize(self, tol):
        before = self.oracle.error()
        self.oracle.optimize()
        return not self.oracle.reached_numeric_limit and self.oracle.error() <= before * (1.0 + tol)

    def reset(self):
        self.oracle.reset()
        self._trace = []
        self.done = 0


class FakeMailboxEngine(FakeEngine):
    """Adds the peer-mailbox protocol of the real engine (exchange_export / attach / probe / enqueue): the
    "device-side" exchange is an all-gather inside enqueue(), which is all the host logic can observe."""
    PROBE_RESULT = 1

    def exchange_export(self):
        return bytes([self.rank]) * 64

    def exchange_attach(self, handles, timeout_s=0.0):
        assert len(handles) == self.world and all(len(h) == 64 for h in handles)
        assert [h[0] for h in handles] == list(range(self.world))      # handles arrive in rank order
        self.attached = True

    def exchange_probe(self):
        return self.PROBE_RESULT

    def exchange_disable(self):
        self.attached = False

    def _exchange_step(self, exact=False):
        import torch.distributed as dist
        rec = self.d + REC_HDR
        send = torch.zeros(rec, dtype=torch.float64)
        recv = torch.zeros(self.world * rec, dtype=torch.float64)
        self.step_scan_tensor(send, exact)
        dist.all_gather(list(recv.view(self.world, -1).unbind(0)), send)
        self.step_apply_tensor(recv)

    def enqueue(self, n):
        assert getattr(self, "attached", False), "enqueue on a row shard needs an attached mailbox"
        for _ in range(n):
            self._exchange_step()

    def enqueue_exact(self):
        self._exchange_step(exact=True)
