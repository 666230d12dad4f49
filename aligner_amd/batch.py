"""Batch driver of the DP path: many independent query x target pairs per launch.

The reference's only batch site is the ten-thread loop of calculate_p_value
(aligner-core/src/statistics/mod.rs:255-286), one perform_alignment per pair per thread.  Here a batch is packed
once (PairBatch), staged in HBM (StagedBatch = aln_batch_create) and filled by persistent waves pulling pairs from
a device work queue; semantics == map of perform_alignment over the pairs.
"""
import ctypes as C

import numpy as np

from . import _ffi
from . import runtime

RESULT_DTYPE = np.dtype([("f", "<f8"), ("score", "<f8"), ("end_y", "<u4"), ("end_x", "<u4"), ("start_y", "<u4"),
                         ("start_x", "<u4"), ("aln_len", "<u4"), ("status", "<i4"), ("passes", "<u4"),
                         ("flags", "<u4")])
assert RESULT_DTYPE.itemsize == 48


class PairBatch:
    """Packed residue codes + offset tables (the layout aln_align_batch takes)."""

    def __init__(self, seqs, q_off, q_len, t_off, t_len):
        self.seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        self.q_off = np.ascontiguousarray(q_off, dtype=np.uint64)
        self.q_len = np.ascontiguousarray(q_len, dtype=np.uint64)
        self.t_off = np.ascontiguousarray(t_off, dtype=np.uint64)
        self.t_len = np.ascontiguousarray(t_len, dtype=np.uint64)

    @classmethod
    def from_pairs(cls, pairs):
        """pairs: iterable of (query_codes, target_codes)."""
        chunks, q_off, q_len, t_off, t_len, pos = [], [], [], [], [], 0
        for q, t in pairs:
            q = np.asarray(q, dtype=np.uint8)
            t = np.asarray(t, dtype=np.uint8)
            q_off.append(pos); q_len.append(len(q)); pos += len(q)
            t_off.append(pos); t_len.append(len(t)); pos += len(t)
            chunks += [q, t]
        seqs = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
        return cls(seqs, q_off, q_len, t_off, t_len)

    def __len__(self):
        return len(self.q_off)

    @property
    def cells(self):
        return int((self.q_len.astype(np.uint64) * self.t_len.astype(np.uint64)).sum())

    def query(self, i):
        return self.seqs[int(self.q_off[i]):int(self.q_off[i] + self.q_len[i])]

    def target(self, i):
        return self.seqs[int(self.t_off[i]):int(self.t_off[i] + self.t_len[i])]

    def select(self, idx):
        """Sub-batch of the given pair indices (re-packed)."""
        return PairBatch.from_pairs((self.query(i), self.target(i)) for i in idx)

    def tb_layout(self):
        cap = 2 * (self.q_len + self.t_len + np.uint64(2))
        off = np.zeros(len(self), dtype=np.uint64)
        if len(self) > 1:
            off[1:] = np.cumsum(cap)[:-1]
        return off, int(cap.sum())


class StagedBatch:
    """A batch resident in HBM (aln_batch handle).  run() is asynchronous on the given HIP stream."""

    def __init__(self, batch, semantics, del_, ext, matrix, device=None, outputs=0, blank=98, **kw):
        self.lib = _ffi.load()
        self.batch = batch
        self.semantics = semantics
        p, self._keep = runtime.make_params(semantics, del_, ext, matrix, outputs=outputs, blank=blank, **kw)
        st = C.c_int(0)
        self.handle = self.lib.aln_batch_create(runtime.context(device), C.byref(p), batch.seqs.ctypes.data,
                                                batch.q_off.ctypes.data, batch.q_len.ctypes.data,
                                                batch.t_off.ctypes.data, batch.t_len.ctypes.data, len(batch),
                                                C.byref(st))
        if not self.handle:
            runtime.raise_for_status(st.value, "aln_batch_create")
            raise RuntimeError("aln_batch_create returned NULL")

    def enable_timing(self, on=True):
        self.lib.aln_batch_enable_timing(self.handle, int(on))

    def run(self, stream=None):
        runtime.raise_for_status(self.lib.aln_batch_run(self.handle, stream), "aln_batch_run")

    def sync(self):
        runtime.raise_for_status(self.lib.aln_batch_sync(self.handle), "aln_batch_sync")

    def timing(self):
        f, t, n = C.c_double(0), C.c_double(0), C.c_uint32(0)
        runtime.raise_for_status(self.lib.aln_batch_timing(self.handle, C.byref(f), C.byref(t), C.byref(n)),
                                 "aln_batch_timing")
        return dict(fill_ms=f.value, traceback_ms=t.value, fill_launches=n.value)

    @property
    def cells(self):
        return int(self.lib.aln_batch_cells(self.handle))

    @property
    def direction_bytes(self):
        return int(self.lib.aln_batch_direction_bytes(self.handle))

    @property
    def results_device_ptr(self):
        return int(self.lib.aln_batch_results_device(self.handle))

    def fetch(self, want_traceback=True):
        n = len(self.batch)
        res = np.zeros(n, dtype=RESULT_DTYPE)
        tb = tb_off = None
        if want_traceback:
            tb_off, total = self.batch.tb_layout()
            tb = np.zeros(max(total, 1), dtype=np.uint8)
        st = self.lib.aln_batch_fetch(self.handle, res.ctypes.data, tb.ctypes.data if want_traceback else None,
                                      tb_off.ctypes.data if want_traceback else None)
        runtime.raise_for_status(st, "aln_batch_fetch")
        return BatchResult(self.batch, res, tb, tb_off)

    def close(self):
        if self.handle:
            self.lib.aln_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchResult:
    def __init__(self, batch, results, tb, tb_off):
        self.batch, self.results, self.tb, self.tb_off = batch, results, tb, tb_off

    def __len__(self):
        return len(self.results)

    def aligned(self, i):
        """(aligned query codes, aligned target codes) of pair i."""
        n = int(self.results["aln_len"][i])
        off = int(self.tb_off[i])
        cap = int(self.batch.q_len[i] + self.batch.t_len[i]) + 2
        return self.tb[off:off + n], self.tb[off + cap:off + cap + n]

    def coords(self, i, semantics):
        r = self.results[i]
        if semantics == _ffi.CORE_GLOBAL:
            return ((1, int(self.batch.q_len[i])), (1, int(self.batch.t_len[i])))
        return ((int(r["start_x"]) + 1, int(r["end_x"]) + 1), (int(r["start_y"]) + 1, int(r["end_y"]) + 1))


def align_batch(batch, semantics, del_, ext, matrix, device=None, want_traceback=True, tb_off=None, outputs=None, blank=98,
                devices=None, out=None, **kw):
    """Blocking batch call through aln_align_batch: host buffers in, host buffers out.  The library cuts the batch into
    chunks and overlaps upload, fill, traceback and download (aln_host.hip).  Returns a BatchResult.

    tb_off: optional caller-chosen offsets of the aligned strings (default: the documented cumulative layout, which the
    library copies back without a per-pair scatter).  devices: a list of GPU ids of this process to shard the chunks over
    (runtime.context_multi); default: the one device of `device`.  out: the BatchResult of an earlier call on a batch of the same
    shape, whose arrays are written again -- a caller in a loop keeps its buffers; fresh ones are 0.44 GB of untouched pages for the
    C5 batch, and faulting them in while the library copies into them costs more than the copies (74 against 50 ms per call)."""
    lib = _ffi.load()
    outs = outputs if outputs is not None else _ffi.OUT_SCORE | (_ffi.OUT_TRACEBACK if want_traceback else 0)
    p, keep = runtime.make_params(semantics, del_, ext, matrix, outputs=outs, blank=blank, **kw)
    n = len(batch)
    res = out.results if out is not None and len(out.results) == n else np.zeros(n, dtype=RESULT_DTYPE)
    tb = None
    if want_traceback:
        if tb_off is None:
            tb_off, total = batch.tb_layout()
        else:
            tb_off = np.ascontiguousarray(tb_off, dtype=np.uint64)
            cap = 2 * (batch.q_len + batch.t_len + np.uint64(2))
            total = int((tb_off + cap).max()) if n else 0
        tb = out.tb if out is not None and out.tb is not None and len(out.tb) == max(total, 1) else np.zeros(max(total, 1), dtype=np.uint8)
    ctx = runtime.context_multi(devices) if devices is not None else runtime.context(device)
    st = lib.aln_align_batch(ctx, C.byref(p), batch.seqs.ctypes.data, batch.q_off.ctypes.data,
                             batch.q_len.ctypes.data, batch.t_off.ctypes.data, batch.t_len.ctypes.data, n,
                             res.ctypes.data, tb.ctypes.data if want_traceback else None,
                             tb_off.ctypes.data if want_traceback else None)
    runtime.raise_for_status(st, "aln_align_batch")
    return BatchResult(batch, res, tb, tb_off if want_traceback else None)


def align_batch_staged(batch, semantics, del_, ext, matrix, device=None, want_traceback=True, **kw):
    """The same through the staged API (aln_batch_create / run / fetch): the whole batch resident in HBM."""
    outs = _ffi.OUT_SCORE | (_ffi.OUT_TRACEBACK if want_traceback else 0)
    sb = StagedBatch(batch, semantics, del_, ext, matrix, device=device, outputs=outs, **kw)
    try:
        sb.run()
        return sb.fetch(want_traceback)
    finally:
        sb.close()
