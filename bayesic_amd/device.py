"""Device context: one per process / GPU, wrapping a ``bsc_ctx``.

torch supplies device memory (``torch.empty(..., device='cuda')``) and the HIP
stream; every compute call goes through the C ABI.  Creating a ``Context``
without a visible gfx950 GPU raises -- there is no CPU path.
"""
import ctypes
import os
import weakref

import torch

from . import _ffi


def option_names(lib=None):
    """The kernel-selection options of the library (bsc_ctx_option_name), in its order."""
    lib = lib or _ffi.load_library()
    names, i = [], 0
    while True:
        key = ctypes.c_char_p()
        _ffi.check(lib.bsc_ctx_option_name(i, ctypes.byref(key)), "bsc_ctx_option_name")
        if key.value is None:
            return names
        names.append(key.value.decode())
        i += 1


class Context:
    """`options`: {name: int} handed to bsc_ctx_set_option after creation (kernel selection is a property of the
    context; the LIBRARY reads no environment variable).  For A/B runs of whole programs (tools/ab_*.py, bench.py
    under a profiler) this class -- not the library -- also honours BSC_<NAME> environment variables, but only in a
    process that says BSC_PROFILING_BUILDS=1; explicit `options` win over them."""

    def __init__(self, device=None, stream=None, options=None):
        if not torch.cuda.is_available():
            raise _ffi.BayesicHipError(
                "bayesic_amd needs a gfx950 (MI355X) GPU: torch.cuda.is_available() is False "
                "and there is no CPU fallback")
        self.lib = _ffi.load_library()
        if device is None:
            device = torch.cuda.current_device()
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self._stream = stream if stream is not None else torch.cuda.current_stream(self.device)
        handle = ctypes.c_void_p()
        _ffi.check(self.lib.bsc_ctx_create(self.device_index, self._stream.cuda_stream,
                                           ctypes.byref(handle)), "bsc_ctx_create")
        self.handle = handle
        merged = {}
        if os.environ.get("BSC_PROFILING_BUILDS") == "1":
            merged["profiling_builds"] = 1
            for name in option_names(self.lib):
                v = os.environ.get("BSC_" + name.upper())
                if v is not None and name != "profiling_builds":
                    merged[name] = int(v)
        if options:
            first = {k: v for k, v in options.items() if k == "profiling_builds"}
            merged.update(first)
            merged.update({k: v for k, v in options.items() if k != "profiling_builds"})
        try:
            if "profiling_builds" in merged:
                self.set_option("profiling_builds", merged.pop("profiling_builds"))
            for k, v in merged.items():
                self.set_option(k, v)
        except Exception:
            self.lib.bsc_ctx_destroy(self.handle)
            self.handle = None
            raise
        self._record = None     # the call list being recorded (record_begin), else None
        self.has_comm = False   # an RCCL communicator (comm_init), even of one rank
        self._graphs = weakref.WeakSet()   # live Graph objects recorded on this context

    # -- plumbing ----------------------------------------------------------
    def close(self):
        if getattr(self, "handle", None):
            for g in list(self._graphs):          # recorded graphs go before the stream they ran on
                g.destroy()
            self.lib.bsc_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return self._stream

    def set_stream(self, stream):
        """Move the context (kernels, collectives) to `stream` AND make it torch's current stream
        on this device: torch's caching allocator ties a block to the stream that was current when
        it was allocated, so allocating on one stream and launching on another lets a freed
        intermediate be handed out while kernels on the context stream still use it."""
        self._stream = stream
        torch.cuda.set_stream(stream)
        _ffi.check(self.lib.bsc_ctx_set_stream(self.handle, stream.cuda_stream),
                   "bsc_ctx_set_stream")

    def _check_stream(self):
        """Allocations made through this context must come from the context's stream (see
        set_stream); inside `with torch.cuda.stream(other):` that is not the case."""
        # (called for every intermediate an engine's walk allocates: the raw handle, not a Stream object -- 0.3 us
        # instead of 5)
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        cur = raw(self.device_index) if raw is not None else torch.cuda.current_stream(self.device).cuda_stream
        if cur != self._stream.cuda_stream:
            raise _ffi.BayesicHipError(
                "torch's current stream on %s is not the context's stream: allocate and launch on "
                "one stream (Context.set_stream(s) switches both)" % (self.device,))

    def sync(self):
        _ffi.check(self.lib.bsc_ctx_sync(self.handle), "bsc_ctx_sync")

    # -- a launch sequence as one hipGraph ------------------------------------
    @property
    def can_capture(self):
        """Stream capture needs a stream of its own (not the null stream torch starts on)."""
        return self._stream.cuda_stream != 0

    def capture_begin(self):
        _ffi.check(self.lib.bsc_capture_begin(self.handle), "bsc_capture_begin")

    def capture_end(self, keep=()):
        """-> Graph.  Raises (and leaves the stream usable) when a call inside was not capturable."""
        h = ctypes.c_void_p()
        _ffi.check(self.lib.bsc_capture_end(self.handle, ctypes.byref(h)), "bsc_capture_end")
        return Graph(self, h, list(keep))

    def set_option(self, name, value):
        _ffi.check(self.lib.bsc_ctx_set_option(self.handle, name.encode(), int(value)), "bsc_ctx_set_option")

    def get_option(self, name):
        out = ctypes.c_int64()
        _ffi.check(self.lib.bsc_ctx_get_option(self.handle, name.encode(), ctypes.byref(out)), "bsc_ctx_get_option")
        return int(out.value)

    def read_stamps(self):
        """Option blr_stamps: (rows, 8) uint64 array {start tick, end tick, XCD, HW_ID, and -- folded finish -- partial
        written, ticket taken, every row arrived, ticket} per workgroup of the last D = 256, S <= 8 pass (100 MHz
        ticks); syncs."""
        import numpy as np
        buf = np.zeros((2048, 8), np.uint64)
        n = ctypes.c_int32()
        _ffi.check(self.lib.bsc_blr_read_stamps(self.handle, buf.ctypes.data, 2048, ctypes.byref(n)),
                   "bsc_blr_read_stamps")
        return buf[:n.value]

    def reserve(self, nbytes):
        _ffi.check(self.lib.bsc_ctx_reserve(self.handle, nbytes), "bsc_ctx_reserve")

    def profile(self, every=1):
        """Time the dominant kernel of each entry point with hipEvents on the ctx stream;
        `every` = 0/False: off, n: every n-th launch."""
        _ffi.check(self.lib.bsc_ctx_profile(self.handle, int(every)), "bsc_ctx_profile")

    def profile_read(self, slot=0):
        """(total_ms, launches) of timing slot `slot` since the last read; syncs.  Slot 0 = the
        dominant kernel, 1 = the collective (bsc_allreduce_sum), 2 = the finish kernel."""
        ms, n = ctypes.c_double(), ctypes.c_int64()
        _ffi.check(self.lib.bsc_ctx_profile_read_slot(self.handle, int(slot), ctypes.byref(ms),
                                                      ctypes.byref(n)),
                   "bsc_ctx_profile_read_slot")
        return float(ms.value), int(n.value)

    # -- the exchange step: RCCL behind the C ABI ---------------------------
    @staticmethod
    def comm_unique_id():
        """128 opaque bytes from rank 0 (ncclGetUniqueId); ship them to every rank."""
        buf = ctypes.create_string_buffer(_ffi.COMM_ID_BYTES)
        _ffi.check(_ffi.load_library().bsc_comm_unique_id(buf), "bsc_comm_unique_id")
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        """Collective over all ranks: gives this context its RCCL communicator."""
        if len(unique_id) != _ffi.COMM_ID_BYTES:
            raise ValueError("unique_id must be %d bytes" % _ffi.COMM_ID_BYTES)
        buf = ctypes.create_string_buffer(bytes(unique_id), _ffi.COMM_ID_BYTES)
        _ffi.check(self.lib.bsc_comm_init_rank(self.handle, buf, int(rank), int(world)),
                   "bsc_comm_init_rank")
        self.has_comm = True

    def comm_destroy(self):
        _ffi.check(self.lib.bsc_comm_destroy(self.handle), "bsc_comm_destroy")
        self.has_comm = False

    def comm_info(self):
        r, w, v = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        _ffi.check(self.lib.bsc_comm_info(self.handle, ctypes.byref(r), ctypes.byref(w),
                                          ctypes.byref(v)), "bsc_comm_info")
        return {"rank": r.value, "world": w.value, "rccl_version": v.value}

    @property
    def comm_world(self):
        """Ranks of this context's RCCL communicator (1 when it has none)."""
        return self.comm_info()["world"]

    def allreduce_sum(self, tensor):
        """In-place all-reduce(sum) over the context's communicator, on the context stream."""
        _ffi.check(self.lib.bsc_allreduce_sum(self.handle, tensor.data_ptr(), tensor.numel(),
                                              _ffi.dtype_code(tensor.dtype)), "bsc_allreduce_sum")

    def allreduce_sum_begin(self, tensor, slot):
        """The same collective on the context's second stream, behind everything enqueued so far; kernels
        enqueued next run beside it.  ``allreduce_sum_end(slot)`` before the first reader of `tensor`."""
        _ffi.check(self.lib.bsc_allreduce_sum_begin(self.handle, tensor.data_ptr(), tensor.numel(),
                                                    _ffi.dtype_code(tensor.dtype), int(slot)), "bsc_allreduce_sum_begin")

    def allreduce_sum_end(self, slot):
        _ffi.check(self.lib.bsc_allreduce_sum_end(self.handle, int(slot)), "bsc_allreduce_sum_end")

    def allreduce_max(self, tensor):
        _ffi.check(self.lib.bsc_allreduce_max(self.handle, tensor.data_ptr(), tensor.numel(),
                                              _ffi.dtype_code(tensor.dtype)), "bsc_allreduce_max")

    def read_probe(self, tensor, reps=10):
        """Best pure streaming-read rate (GB/s) over `tensor` on this device; syncs."""
        out = ctypes.c_double()
        _ffi.check(self.lib.bsc_hbm_read_probe(self.handle, tensor.data_ptr(),
                                               tensor.numel() * tensor.element_size(), int(reps),
                                               ctypes.byref(out)), "bsc_hbm_read_probe")
        return float(out.value)

    def info(self):
        buf = (ctypes.c_int64 * 8)()
        _ffi.check(self.lib.bsc_device_info(self.handle, buf), "bsc_device_info")
        keys = ["cu_count", "wave_size", "lds_bytes", "clock_khz", "l2_bytes", "gfx", "hbm_mib"]
        return dict(zip(keys, list(buf)))

    def empty(self, shape, dtype=torch.float32):
        self._check_stream()
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype=torch.float32):
        self._check_stream()
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def to_device(self, array, dtype=None):
        t = torch.as_tensor(array)
        if dtype is not None:
            t = t.to(dtype)
        return t.contiguous().to(self.device)

    # -- timing on the ctx stream -----------------------------------------
    def event(self):
        return Event(self)

    def call(self, name, *args):
        """Invoke entry point `name`; torch tensors are passed as device pointers."""
        raw = [a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args]
        fn = getattr(self.lib, name)
        if self._record is not None:
            self._record.append((fn, raw, name))
        _ffi.check(fn(self.handle, *raw), name)

    # -- a launch sequence as a list of C-ABI calls (DeviceBackend.replay_call) -------------------------------------
    def record_begin(self):
        """From now on every ``call`` is also appended -- the bound library function, its arguments as they were
        handed to it (device addresses, extents, ctypes arrays) -- to a list that ``record_end`` returns.  The calls
        EXECUTE as usual (unlike a graph capture, which only records)."""
        if self._record is not None:
            raise _ffi.BayesicHipError("Context.record_begin: a recording is already running")
        self._record = []

    def record_end(self):
        calls, self._record = self._record, None
        return calls

    def replay(self, calls):
        """Re-issue a recorded call list: the same launches on the same addresses, without the Python walk that
        produced them (about 2 us of host time per launch instead of ~28)."""
        h = self.handle
        for fn, raw, name in calls:
            rc = fn(h, *raw)
            if rc:
                _ffi.check(rc, name)


class Graph:
    """A captured launch sequence (bsc_capture_begin / bsc_capture_end).  `keep` holds every buffer the
    recorded launches touch: the graph stores their addresses."""

    def __init__(self, ctx, handle, keep):
        self.ctx, self.handle, self.keep = ctx, handle, keep
        ctx._graphs.add(self)

    def launch(self):
        if not self.handle:
            raise _ffi.BayesicHipError("the graph was destroyed (its context has been closed)")
        _ffi.check(self.ctx.lib.bsc_graph_launch(self.ctx.handle, self.handle), "bsc_graph_launch")

    def destroy(self):
        """Idempotent.  A graph never outlives its context's stream: Context.close() destroys the
        graphs recorded on it first (the runtime touches the stream a graph last ran on when the graph
        is destroyed; left to the garbage collector that happened at some later, unrelated moment)."""
        if self.handle:
            handle, self.handle = self.handle, None
            self.ctx.lib.bsc_graph_destroy(handle)
            self.keep = []

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Event:
    """hipEvent recorded on the context stream through the C ABI."""

    def __init__(self, ctx):
        self.ctx = ctx
        h = ctypes.c_void_p()
        _ffi.check(ctx.lib.bsc_event_create(ctypes.byref(h)), "bsc_event_create")
        self.handle = h

    def record(self):
        _ffi.check(self.ctx.lib.bsc_event_record(self.ctx.handle, self.handle),
                   "bsc_event_record")
        return self

    def elapsed_ms(self, stop):
        ms = ctypes.c_float()
        _ffi.check(self.ctx.lib.bsc_event_elapsed_ms(self.handle, stop.handle, ctypes.byref(ms)),
                   "bsc_event_elapsed_ms")
        return float(ms.value)

    def __del__(self):
        try:
            if self.handle:
                self.ctx.lib.bsc_event_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


_default = None


def default_context():
    global _default
    if _default is None:
        _default = Context()
    return _default
