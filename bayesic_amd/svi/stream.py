"""Mini-batch streaming for SVI (README.md:69-79): host data set -> HBM slots over
PCIe on a copy stream, overlapped with the updates running on the context's stream.
Thin wrapper over the C ABI (bsc_loader_*, include/bayesic_hip.h); the update
kernels only ever see device-resident batches, so the hot path is unchanged.

    loader = MiniBatchLoader(ctx, max_rows, D)
    loader.submit(X0, y0)
    for t in range(steps):
        if t + 1 < steps:
            loader.submit(X[t + 1], y[t + 1])      # crosses PCIe during update t
        dX, dy, rows = loader.acquire()             # ctx stream waits for the copy of batch t
        model.set_batch(dX, dy, rows)
        model.step()
        loader.release()                            # slot reusable once step t has run

Host arrays should be page-locked by the runtime: ``loader.pinned_empty(shape)`` gives a numpy array in
hipHostMalloc memory (bsc_host_alloc; valid until ``close()``), torch tensors made with
``pin_memory=True`` work too.  A page-locked batch must stay valid and unchanged until ``n_slots``
further submits have returned (the loader keeps a reference that long).  Anything else -- an ordinary
numpy array -- is copied by the host, inside ``submit``, into a page-locked bounce buffer of the slot:
a fraction of the PCIe rate, but the device never reads a page that malloc owns, and the source is free
when submit returns.  (There is no ``pin(array)`` any more: registering heap memory in place is how a
heap address became the address of a GPU fault in round 2 -- DESIGN.md section 10.)
"""
import ctypes

import numpy as np

from .. import _ffi


def _host_pointer(a):
    if isinstance(a, np.ndarray):
        if a.dtype != np.float32:
            raise TypeError("host batches must be float32")
        return a.ctypes.data
    if hasattr(a, "data_ptr"):                      # CPU torch tensor
        import torch
        if a.device.type != "cpu" or a.dtype != torch.float32:
            raise TypeError("host batches must be float32 CPU tensors")
        return a.data_ptr()
    raise TypeError("host batch must be a numpy array or a CPU torch tensor")


class MiniBatchLoader(object):
    def __init__(self, ctx, max_rows, D, n_slots=2):
        self.ctx = ctx
        self.max_rows, self.D, self.n_slots = int(max_rows), int(D), int(n_slots)
        h = ctypes.c_void_p()
        _ffi.check(ctx.lib.bsc_loader_create(ctx.handle, self.max_rows, self.D, self.n_slots,
                                             ctypes.byref(h)), "bsc_loader_create")
        self.handle = h
        self._in_flight = []       # host arrays whose copies may still be running
        self._host_allocs = []     # hipHostMalloc blocks handed out by pinned_empty

    @staticmethod
    def _strides_ok(X, y):
        if isinstance(X, np.ndarray):
            return X.ndim == 2 and X.strides[1] == 4 and X.strides[0] % 4 == 0 and \
                y.ndim == 1 and y.strides[0] == 4, X.strides[0] // 4
        return X.dim() == 2 and X.stride(1) == 1 and y.dim() == 1 and y.stride(0) == 1, X.stride(0)

    def pinned_empty(self, shape, dtype=np.float32):
        """Uninitialised numpy array in page-locked memory owned by the HIP runtime (hipHostMalloc):
        the source to stream from at the full PCIe rate.  Valid until ``close()``."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = ctypes.c_void_p()
        _ffi.check(self.ctx.lib.bsc_host_alloc(max(nbytes, 1), ctypes.byref(p)), "bsc_host_alloc")
        self._host_allocs.append(p)
        buf = (ctypes.c_char * max(nbytes, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def submit(self, X, y):
        ok, ldx = self._strides_ok(X, y)
        if not ok or X.shape[0] != y.shape[0] or X.shape[1] != self.D:
            raise ValueError("batch must be X [rows, %d] (unit column stride) and y [rows]" % self.D)
        _ffi.check(self.ctx.lib.bsc_loader_submit(self.handle, _host_pointer(X), int(ldx),
                                                  _host_pointer(y), int(X.shape[0])),
                   "bsc_loader_submit")
        # bsc_loader_submit host-synchronises on the copy that last used the slot it reuses, so once
        # submission k + n_slots has returned the source of submission k has been read: the last
        # n_slots sources are the ones that may still be crossing PCIe
        self._in_flight.append((X, y))
        del self._in_flight[:-self.n_slots]

    def acquire(self):
        """(device pointer of X, device pointer of y, rows) of the oldest submitted batch;
        the context's stream waits for its copy."""
        dX, dy, rows = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
        _ffi.check(self.ctx.lib.bsc_loader_acquire(self.handle, ctypes.byref(dX), ctypes.byref(dy),
                                                   ctypes.byref(rows)), "bsc_loader_acquire")
        return int(dX.value), int(dy.value), int(rows.value)

    def release(self):
        _ffi.check(self.ctx.lib.bsc_loader_release(self.handle), "bsc_loader_release")

    def close(self):
        if self.handle:
            self.ctx.lib.bsc_loader_destroy(self.handle)      # drains the copy stream and the context's stream
            self.handle = None
            self._in_flight = []
            for p in self._host_allocs:
                self.ctx.lib.bsc_host_free(p)
            self._host_allocs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
