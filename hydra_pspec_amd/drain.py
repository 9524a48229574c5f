"""Output drain of the batched sampler: device history -> pinned host buffers -> per-baseline ``.npy`` files.

The reference writes from inside the chain loop, synchronously, everything sampled so far at every
``write_Niter`` flush (pspec.py:625-653, utils.py:272-312).  Here the sampler never waits for a file:

* :meth:`ChainDrain.submit` (sampling thread) starts the device -> host copy of the chunk just sampled on a copy
  stream of its own, into one of TWO pinned staging buffers, and returns; chunk k + 1 samples while chunk k drains
  (a third chunk waits for a free buffer: back-pressure instead of unbounded host memory);
* a writer thread waits for the copy's event and appends the new rows to each baseline's files
  (:mod:`hydra_pspec_amd.npy_append`: O(Niter) bytes over a run, every file a valid ``.npy`` of shape
  ``(done, ...)`` at any instant), ``workers`` baselines at a time (``os.pwrite`` releases the GIL);
* ``cov-eor.npy`` (``--outputs all``) is the covariance of the LAST sample only, replaced atomically at each flush
  with the reference's row-slice quirk (rows ``[:done]`` on a periodic write, pspec.py:630 vs :648).

No GPU work is issued from the writer thread.  Without a GPU (``--dry_run``) the staging copy is a host copy.
"""
import os
import queue
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

from .npy_append import NpyAppender, read_header

HISTORY_FILES = {"signal_ps": "dps-eor.npy", "ln_post": "ln-post.npy", "signal_cr": "gcr-eor.npy",
                 "fg_amps": "fg-amps.npy", "chisq": "chisq.npy"}
THINNED = ("signal_cr", "fg_amps", "chisq")


def circulant_cov(ps_over_n2):
    """``Fop^H diag(p) Fop`` (reference pspec.py:313-322) on the host without the two N^3 products: with the centred
    operator of utils.fourier_operator the result is the circulant ``S[j,k] = c[(j-k) mod N]``,
    ``c = N ifft(ifftshift(p))``."""
    p = np.asarray(ps_over_n2, dtype=float)
    N = p.size
    c = np.fft.ifft(np.fft.ifftshift(p)) * N
    j = np.arange(N)
    return c[(j[:, None] - j[None, :]) % N]


def row_bytes(name, T, N, M):
    return {"signal_ps": 8 * N, "ln_post": 8, "signal_cr": 16 * T * N, "fg_amps": 16 * T * M, "chisq": 8 * T * N}[name]


def estimate_bytes(names, nbl, T, N, M, Niter, chunk, thin):
    """(disk bytes of the whole run, bytes of ONE staged chunk) for the histories ``names`` of ``nbl`` baselines."""
    disk = stage = 0
    for k in names:
        rows_all = -(-Niter // thin) if k in THINNED else Niter
        rows_chunk = -(-chunk // thin) if k in THINNED else chunk
        disk += nbl * rows_all * row_bytes(k, T, N, M)
        stage += nbl * rows_chunk * row_bytes(k, T, N, M)
    return disk, stage


def checkpoint_rows(bdir, names, thin):
    """Iterations the checkpoint files of one baseline directory hold consistently (0 if any is missing rows): the
    smallest over the files, the thinned ones counted in iterations; a multiple of ``thin``."""
    done = None
    for k in names:
        shape, _, _ = read_header(Path(bdir) / HISTORY_FILES[k])
        rows = shape[0] * (thin if k in THINNED else 1)
        done = rows if done is None else min(done, rows)
    done = done or 0
    return done - done % thin if thin > 1 else done


class ChainDrain:
    def __init__(self, torch, bdirs, names, T, N, M, chunk, thin=1, all_out=False, use_gpu=True, fsync=False,
                 workers=4):
        self.torch, self.bdirs, self.names = torch, [Path(b) for b in bdirs], tuple(names)
        self.nbl, self.T, self.N, self.M = len(bdirs), T, N, M
        self.chunk, self.thin, self.all_out = int(chunk), int(thin), bool(all_out)
        self.use_gpu, self.fsync, self.workers = bool(use_gpu), bool(fsync), max(1, int(workers))
        self.write_times = [0.0] * self.nbl
        self.shapes = {"signal_ps": (N,), "ln_post": (), "signal_cr": (T, N), "fg_amps": (T, M), "chisq": (T, N)}
        self.dtypes = {k: (np.complex128 if k in ("signal_cr", "fg_amps") else np.float64) for k in HISTORY_FILES}
        self.files = [{k: NpyAppender(b / HISTORY_FILES[k], self.shapes[k], self.dtypes[k], fsync=fsync)
                       for k in self.names} for b in self.bdirs]
        self._free, self._jobs = queue.Queue(), queue.Queue()
        self._bufs = []
        self._err = None
        self._thread = None
        self._copy_stream = None
        self.bytes_written = 0
        self.t_copy_wait = self.t_files = self.t_backpressure = 0.0

    # ------------------------------------------------------------------ set-up
    def start(self, iter0=0):
        """Create (``iter0 == 0``) or continue (cut back to ``iter0`` iterations) every history file; allocate the two
        staging buffers; start the writer."""
        for b, d in enumerate(self.bdirs):
            d.mkdir(parents=True, exist_ok=True)
            for k, f in self.files[b].items():
                f.start(-(-iter0 // self.thin) if k in THINNED else iter0)
        torch = self.torch
        tdt = {np.dtype(np.float64): torch.float64, np.dtype(np.complex128): torch.complex128}
        for _ in range(2):
            buf = {}
            for k in self.names:
                rows = -(-self.chunk // self.thin) if k in THINNED else self.chunk
                buf[k] = torch.empty((self.nbl, rows) + self.shapes[k], dtype=tdt[np.dtype(self.dtypes[k])],
                                     pin_memory=self.use_gpu)
            self._bufs.append(buf)
            self._free.put(len(self._bufs) - 1)
        if self.use_gpu:
            self._copy_stream = torch.cuda.Stream()
        self._thread = threading.Thread(target=self._writer, name="hydra-pspec-drain", daemon=True)
        self._thread.start()
        return self

    # ------------------------------------------------------------------ sampling thread
    def submit(self, out, n, done, periodic):
        """Chunk of ``n`` iterations just sampled (``out``: the dict GibbsBatch.run returned, device or host tensors);
        ``done`` iterations exist after it.  Returns once the copy is in flight."""
        self._raise_if_failed()
        t0 = time.perf_counter()
        slot = self._free.get()
        self.t_backpressure += time.perf_counter() - t0
        if slot is None:            # the writer died while we waited
            self._raise_if_failed()
            raise RuntimeError("output drain stopped")
        buf, torch = self._bufs[slot], self.torch
        views, event = {}, None
        if self.use_gpu:
            self._copy_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._copy_stream):
                for k in self.names:
                    views[k] = buf[k][:, :out[k].shape[1]]
                    views[k].copy_(out[k], non_blocking=True)
                event = torch.cuda.Event()
                event.record(self._copy_stream)
        else:
            for k in self.names:
                views[k] = buf[k][:, :out[k].shape[1]]
                views[k].copy_(out[k])
        # `out` rides along: the device tensors must outlive the copy
        self._jobs.put((slot, views, event, out, int(n), int(done), bool(periodic)))

    def close(self, abort=False):
        """Drain what is queued, stop the writer, surface its error.  ``abort``: the sampler failed -- the queued
        chunks (complete ones) are still written, then the error of the caller propagates."""
        if self._thread is not None:
            self._jobs.put(None)
            self._thread.join()
            self._thread = None
        if not abort:
            self._raise_if_failed()

    def _raise_if_failed(self):
        if self._err is not None:
            err, self._err = self._err, None
            raise err

    # ------------------------------------------------------------------ writer thread
    def _write_baseline(self, b, arrs, done, periodic):
        tw = time.perf_counter()
        nbytes = 0
        for k in self.names:
            a = arrs[k][b]
            self.files[b][k].append(a)
            nbytes += a.nbytes
        if self.all_out:
            S_last = circulant_cov(arrs["signal_ps"][b, -1] / self.N ** 2)
            fn = self.bdirs[b] / "cov-eor.npy"
            tmp = self.bdirs[b] / "cov-eor.npy.tmp.npy"
            np.save(tmp, S_last[:done] if periodic else S_last)
            os.replace(tmp, fn)
        self.write_times[b] += time.perf_counter() - tw
        return nbytes

    def _writer(self):
        pool = ThreadPoolExecutor(self.workers) if self.workers > 1 else None
        try:
            while True:
                job = self._jobs.get()
                if job is None:
                    return
                slot, views, event, out, n, done, periodic = job
                try:
                    if self._err is None:
                        t0 = time.perf_counter()
                        if event is not None:
                            event.synchronize()
                        del out, job
                        t1 = time.perf_counter()
                        arrs = {k: v.numpy() for k, v in views.items()}
                        if pool is None:
                            nb = sum(self._write_baseline(b, arrs, done, periodic) for b in range(self.nbl))
                        else:
                            nb = sum(pool.map(lambda b: self._write_baseline(b, arrs, done, periodic),
                                              range(self.nbl)))
                        self.bytes_written += nb
                        self.t_copy_wait += t1 - t0
                        self.t_files += time.perf_counter() - t1
                except BaseException as e:      # keep consuming so that the sampler never blocks on a dead writer
                    self._err = e
                finally:
                    self._free.put(slot)
        finally:
            if pool is not None:
                pool.shutdown()
            self._free.put(None)
