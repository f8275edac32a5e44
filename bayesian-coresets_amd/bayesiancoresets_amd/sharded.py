"""Row-sharded solver: one process per GPU, contiguous row blocks, one small all-gather per
greedy iteration.

Every rank scans only its own rows and produces a record {exact score, global index, norm,
flags, raw row} (d + 4 doubles).  The records reach every rank (payload world_size * (d + 4) * 8
bytes, i.e. latency-bound) and every rank applies the *same* fp64 reweight to its replicated
O(d) state, so xw, weights and the trace stay bit-identical on all ranks; no residual
all-reduce is needed.  Contiguous shards plus the (score desc, global index asc) winner rule
keep NumPy's lowest-index tie-break.

Two exchange modes:
  * "mailbox" (default when every rank is on this node): the iteration's tail kernel stores its
    record straight into the peers' mailboxes over xGMI (hipIpc-mapped fine-grained memory) and
    waits for theirs, so a whole build() is enqueued without a host round trip or a collective
    launch per iteration (csrc/resolve.hip: mailbox_exchange).  Set up once through
    torch.distributed (handle all-gather) and verified by a collective probe; any failure falls
    back, on all ranks together, to
  * "collective": resolve kernel -> all_gather (RCCL over xGMI via torch.distributed) -> apply
    kernel, driven from the host every iteration.  BCX_EXCHANGE=collective forces it.
Either way build() ends with a cross-rank check that every rank recorded the same trace.

The driver is written against a small engine protocol (``bayesiancoresets_amd._native.Engine``
implements it on the GPU); tests inject a CPU stand-in to exercise this orchestration under
``gloo`` without a GPU.
"""
import os

import numpy as np

from . import _native as nat

CHUNK_ROWS = nat.CHUNK_ROWS


def shard_bounds(n_global, world_size, align=CHUNK_ROWS):
    """Contiguous row ranges, every boundary a multiple of ``align`` rows (the engine's column-sum
    chunk) so that the chunk partition -- hence b and sum(norms) -- is identical for any world size."""
    n_chunks = (n_global + align - 1) // align
    per = (n_chunks + world_size - 1) // world_size
    bounds = []
    for r in range(world_size):
        lo = min(n_global, r * per * align)
        hi = min(n_global, (r + 1) * per * align)
        bounds.append((lo, hi))
    return bounds, per


class ShardedSolver(object):
    """Collective object: construct / call on every rank of ``group`` with the same arguments."""

    def __init__(self, alg, n_global, d, group=None, device=None, engine_factory=None, solo=False, **engine_kw):
        """``solo``: this process alone holds all ``n_global`` rows as ONE shard even though a process group is
        initialised (not a collective object then: no rank but the caller takes part) -- bench.py's one-shard leg
        inside a multi-rank run."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        spans = dist.is_initialized() and not solo
        self.rank = dist.get_rank(group) if spans else 0
        self.world = dist.get_world_size(group) if spans else 1
        self.alg, self.n_global, self.d = alg, int(n_global), int(d)
        bounds, self.chunks_per_rank = shard_bounds(self.n_global, self.world)
        self.row_begin, self.row_end = bounds[self.rank]
        self.n_local = self.row_end - self.row_begin
        self.bounds = bounds
        factory = engine_factory or nat.Engine
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.engine = factory(alg, self.n_local, self.d, n_global=self.n_global,
                              row_offset=self.row_begin if self.n_local > 0 else 0,
                              rank=self.rank, world_size=self.world, device=device, **engine_kw)
        self.tdev = self.engine.tensor_device() if hasattr(self.engine, "tensor_device") else torch.device("cuda", device)
        rec = self.d + nat.REC_HDR
        self.send = torch.zeros(rec, dtype=torch.float64, device=self.tdev)
        self.recv = torch.zeros(self.world * rec, dtype=torch.float64, device=self.tdev)
        self.reached_numeric_limit = False
        self.last_trace = None
        self._flat_gather = True
        self.exchange = "collective"
        self._xt = None
        # how the exchange mode was decided (reported by bench.py as config.exchange_probe)
        self.probe_info = {"attempted": False, "result": None, "reason": "single shard" if self.world == 1 else
                           ("BCX_EXCHANGE=%s" % os.environ.get("BCX_EXCHANGE") if os.environ.get("BCX_EXCHANGE", "mailbox") != "mailbox"
                            else "engine has no peer mailbox")}
        if self.world > 1 and hasattr(self.engine, "exchange_export") \
                and os.environ.get("BCX_EXCHANGE", "mailbox") == "mailbox":
            self._setup_mailbox()

    def _agree(self, ok):
        """True iff `ok` holds on every rank."""
        flag = self.torch.tensor([1.0 if ok else 0.0], dtype=self.torch.float64, device=self.tdev)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item() > 0.5)

    def _setup_mailbox(self):
        """Map every rank's mailbox (one node: hipIpc) and prove the path with one probe exchange;
        all ranks end up in the same mode."""
        import logging
        import socket
        handle, why = None, ""
        self.probe_info = {"attempted": True, "result": None, "reason": ""}
        try:
            handle = self.engine.exchange_export()
        except nat.EngineError as e:
            why = str(e)
        mine = (socket.gethostname(), handle)
        everyone = [None] * self.world
        self.dist.all_gather_object(everyone, mine, group=self.group)
        ok = all(h is not None and host == mine[0] for host, h in everyone)
        if ok:
            try:
                self.engine.exchange_attach([h for _, h in everyone],
                                            float(os.environ.get("BCX_EXCHANGE_TIMEOUT", "20")))
            except nat.EngineError as e:
                ok, why = False, str(e)
        if self._agree(ok):           # (also a barrier: every mailbox is mapped before the first store)
            res = self.engine.exchange_probe()
            if os.environ.get("BCX_TEST_FAIL_PROBE"):      # tests: exercise the fall-back
                res = -2
            ok = res == 1
            why = why or "probe result %d" % res
            self.probe_info["result"] = int(res)
            if self._agree(ok):
                self.exchange = "mailbox"
                self.probe_info["reason"] = "probe exchange delivered every shard's record intact"
                return
        self.engine.exchange_disable()
        self.probe_info["reason"] = why or "a peer failed"
        if self.rank == 0:
            logging.getLogger().warning("sharded build: peer mailbox unavailable (%s); using the all-gather exchange",
                                        why or "a peer failed")

    def fallback_to_collective(self):
        """Leave mailbox mode (COLLECTIVE call: every rank, e.g. after build() raised on all of them)."""
        if self.exchange == "mailbox":
            self.engine.exchange_disable()
            self.exchange = "collective"

    def _all_gather(self, recv, send):
        """all_gather_into_tensor (one RCCL call); backends without it (gloo on device tensors) get the
        list form over views of the same receive buffer."""
        if self._flat_gather:
            try:
                self.dist.all_gather_into_tensor(recv, send, group=self.group)
                return
            except (RuntimeError, NotImplementedError):
                self._flat_gather = False
        views = list(recv.view(self.world, -1).unbind(0))
        self.dist.all_gather(views, send, group=self.group)

    # ---- construction ------------------------------------------------------
    def load_local(self, rows, local_row_begin=0, center=False):
        """rows: torch tensor (device or cpu) or ndarray holding local rows
        [local_row_begin, local_row_begin + len(rows)).  center: the rows are raw log-likelihoods
        (``DeviceProjector.project_uncentred``); the ingest pass subtracts the row means (projector.py:21)."""
        if center:
            self.engine.load_rows_any(rows, local_row_begin, center=True)
        else:
            self.engine.load_rows_any(rows, local_row_begin)

    def finalize(self, b=None):
        torch, dist = self.torch, self.dist
        gathered_ptr, n_gathered, keep = None, 0, None
        if self.world > 1:
            # always: even with a caller-supplied b the sum of row norms (Frank-Wolfe's sigma, frankwolfe.py:25) is
            # a sum over ALL shards' chunk sums -- local sums alone would give every rank a different, too small sigma
            per = self.chunks_per_rank
            mine = torch.zeros(per * (self.d + 1), dtype=torch.float64, device=self.tdev)
            self.engine.export_chunk_sums_tensor(mine, per)
            allc = torch.zeros(self.world * per * (self.d + 1), dtype=torch.float64, device=self.tdev)
            self._all_gather(allc, mine)
            keep = allc
            gathered_ptr = allc
            n_gathered = (self.n_global + CHUNK_ROWS - 1) // CHUNK_ROWS
        rc = self.engine.finalize_any(b, gathered_ptr, n_gathered)
        if self.world > 1:
            # a zero row / zero b on any shard fails the constructor everywhere
            flag = torch.tensor([float(-rc)], dtype=torch.float64, device=self.tdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            worst = -int(flag.item())
            if rc == nat.OK and worst != nat.OK:
                rc = worst
        del keep
        return rc

    # ---- the hot loop --------------------------------------------------------
    def _iterations(self, n, exact=False):
        """n host-driven iterations (scan -> all-gather of the records -> replicated apply).  A rank whose engine fails
        keeps taking part in the all-gathers of the batch -- with an empty record -- so that its peers are never left
        inside a collective; the error is returned (not raised) and settled by every rank together in build()."""
        err = None
        for _ in range(n):
            if err is None:
                try:
                    self.engine.step_scan_tensor(self.send, exact)
                except nat.EngineError as e:
                    err = e
                    self.send.zero_()          # flags 0: "no candidate from this shard"
            if self.world > 1:
                xt = self._xt
                if xt is not None:
                    # the exchange step on the stream's clock (collective mode has no device-side stamps: the all-gather is a
                    # library call): from the record being ready to the gathered records being usable
                    e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
                    e0.record()
                    self._all_gather(self.recv, self.send)
                    e1.record()
                    xt.append((e0, e1))
                else:
                    self._all_gather(self.recv, self.send)
            if err is None:
                try:
                    self.engine.step_apply_tensor(self.recv if self.world > 1 else self.send)
                except nat.EngineError as e:
                    err = e
                    self.send.zero_()
        return err

    def time_exchange(self, on=True):
        """Collective mode: bracket every all-gather of the records with events (``exchange_times_us`` reads them)."""
        self._xt = [] if on else None

    def exchange_times_us(self):
        """Microseconds of every all-gather since ``time_exchange(True)`` (synchronises); the list is emptied."""
        xt, self._xt = (self._xt or []), ([] if self._xt is not None else None)
        if not xt:
            return []
        self.torch.cuda.synchronize()
        return [e0.elapsed_time(e1) * 1e3 for e0, e1 in xt]

    def _settle(self, err):
        """Collective: None if no rank failed; otherwise EVERY rank raises -- the failing ones their own error, the others
        one that names the failing ranks -- so a failure on one rank can neither hang the others in a collective nor let
        them carry on with a state their peer no longer shares."""
        if self.world == 1:
            if err is not None:
                raise err
            return
        if self._agree(err is None):
            return
        msgs = [None] * self.world
        self.dist.all_gather_object(msgs, None if err is None else "%s" % err, group=self.group)
        if err is not None:
            raise err
        bad = ["rank %d: %s" % (r, m) for r, m in enumerate(msgs) if m is not None]
        raise nat.EngineError(nat.ERR_STATE, "sharded build stopped: " + "; ".join(bad))

    def build(self, itrs, tol=1e-12):
        itrs = int(itrs)
        if self.engine.build_begin(itrs, tol):
            return None
        remaining = itrs
        while True:
            on_device = self.exchange == "mailbox" or (
                self.world == 1 and hasattr(self.engine, "enqueue") and not os.environ.get("BCX_SHARDED_GENERIC"))
            err, done, need_exact, limit = None, 0, 0, 0
            try:
                if on_device:
                    self.engine.enqueue(remaining)       # scan + merged resolve / (exchange) / apply launches
                else:
                    err = self._iterations(remaining)
                if err is None:
                    done, need_exact, limit = self.engine.poll()   # replicated state: same answer on every rank
            except nat.EngineError as e:
                err = e
            self._settle(err)                          # (one tiny all-reduce per batch; raises on every rank or on none)
            if need_exact:
                try:
                    if self.exchange == "mailbox":
                        self.engine.enqueue_exact()
                    else:
                        err = self._iterations(1, exact=True)
                    if err is None:
                        done, need_exact, limit = self.engine.poll()
                        if need_exact:
                            err = nat.EngineError(nat.ERR_STATE, "exact scan did not resolve the iteration")
                except nat.EngineError as e:
                    err = e
                self._settle(err)
            if limit or done >= itrs:
                break
            remaining = itrs - done
        self.reached_numeric_limit = bool(limit)
        self.last_trace = self.engine.trace(itrs)
        self._check_replicated(self.last_trace)
        return self.last_trace

    def _check_replicated(self, trace):
        """The design invariant, checked once per build(): every rank recorded the same (selection,
        error, status) sequence.  One tiny all-reduce; a mismatch means a broken exchange."""
        if self.world == 1:
            return
        sel, err, status = trace
        digest = np.zeros(3)
        if len(sel):
            wgt = 1.0 + (np.arange(len(sel)) % 8191)
            digest[:] = (float(np.dot(sel % 65521, wgt)), float(np.dot(status, wgt)), float(err[-1]))
        t = self.torch.tensor(np.concatenate([digest, -digest]), dtype=self.torch.float64, device=self.tdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        t = t.cpu().numpy()
        if not np.array_equal(t[:3], -t[3:]):
            raise nat.EngineError(nat.ERR_STATE, "ranks disagree on the build trace (exchange mode %s)" % self.exchange)

    # ---- read-out --------------------------------------------------------------
    def sparse_weights(self):
        return self.engine.sparse_weights()

    def error(self):
        return self.engine.error()

    def size(self):
        idx, w = self.engine.sparse_weights()
        return int((w > 0).sum())
