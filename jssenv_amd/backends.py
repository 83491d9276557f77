"""Where the env tensors live and which library steps them: the two backends behind ``BatchedJssEnv``.

* ``HipBackend`` (default): torch.cuda tensors + ``libjss_hip.so``.  Constructing it without a GPU or without the built
  extension raises -- there is NO silent fallback.
* ``CpuBackend`` (``device="cpu"``, explicit only): NumPy arrays + ``libjss_cpu.so``, the from-scratch C++/OpenMP twin with
  identical symbols (BASELINE config 1, "runs without a GPU"; also bench.py's ``cpu_baseline`` kind "twin").

torch is used for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _abi


class HipBackend:
    """Device memory = torch.cuda tensors; kernels = libjss_hip.so on torch's current stream."""

    name = "hip"
    default_kernel = "auto"

    def __init__(self, device=None):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("jssenv_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no silent "
                               "CPU fallback -- pass device='cpu' to run on the host-core twin (libjss_cpu.so) on purpose")
        path = _abi.library_path()
        if not os.path.isfile(path):
            raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        self.torch = torch
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise ValueError(f"HipBackend needs a cuda device, got {self.device}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = _abi.bind(C.CDLL(path))
        if not self.lib.jss_backend().startswith(b"hip"):
            raise RuntimeError(f"{path} is not the HIP library ({self.lib.jss_backend()!r})")
        self._scalars = {}
        self._stream_arrays = {}
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)

    # -- memory ----------------------------------------------------------------------------
    def zeros(self, shape, dtype):
        return self.torch.zeros(shape, dtype=getattr(self.torch, dtype), device=self.device)

    def from_numpy(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def zeros_pinned(self, shape, dtype):
        """Zero-filled page-locked HOST memory the device reads and writes in place (hipHostMalloc: coherent, uncached on
        the GPU): a tensor on the CPU whose data_ptr() the kernels take like any other -- the B = 1 facade's arena."""
        return self.torch.zeros(shape, dtype=getattr(self.torch, dtype), pin_memory=True)

    def ptr(self, x):
        return 0 if x is None else x.data_ptr()

    def numpy(self, x):
        if isinstance(x, np.ndarray):              # (host-derived attributes are NumPy already)
            return x
        if x.device.type == "cpu":                 # a view of a host arena: the kernels write it in place -- wait, then copy
            self.sync()
            return x.detach().numpy().copy()
        return x.detach().cpu().numpy()

    def as_device(self, x, dtype):
        t = self.torch.as_tensor(x, device=self.device)
        return t.to(getattr(self.torch, dtype)).contiguous()

    def stage(self, buf, x):
        """`x` (device tensor of any integer dtype, NumPy array, list) as a contiguous device tensor of buf's dtype:
        `x` itself when it already is one, otherwise copied into the preallocated `buf` -- nothing is allocated on the
        device (an int64 tensor, what argmax returns, is converted by the copy kernel)."""
        t = self.torch
        if isinstance(x, t.Tensor):
            if tuple(x.shape) != tuple(buf.shape):      # checked first: the kernels read buf.numel() elements
                raise ValueError(f"expected shape {tuple(buf.shape)}, got {tuple(x.shape)}")
            if x.device == buf.device and x.dtype == buf.dtype and x.is_contiguous():
                return x
            buf.copy_(x)
            return buf
        a = np.ascontiguousarray(np.asarray(x), dtype=np.dtype(str(buf.dtype).split(".")[-1]))
        if a.shape != tuple(buf.shape):
            raise ValueError(f"expected shape {tuple(buf.shape)}, got {a.shape}")
        buf.copy_(t.from_numpy(a))
        return buf

    def copy_into(self, dst, src):
        if isinstance(src, np.ndarray):
            src = self.torch.from_numpy(np.ascontiguousarray(src))
        dst.copy_(src)

    def carve(self, arena, off, shape, dtype):
        """View of `arena` (flat uint8 tensor) at byte offset `off` as `shape` / `dtype`."""
        dt = getattr(self.torch, dtype)
        n = int(np.prod(shape)) * dt.itemsize
        return arena[off:off + n].view(dt).view(shape)

    def snapshot(self, arena, host=None):
        """One device -> host copy of a whole arena into a pinned staging buffer (reused); returns the NumPy view."""
        t = self.torch
        if host is None or host.numel() != arena.numel():
            host = t.empty(arena.numel(), dtype=t.uint8, pin_memory=True)
        host.copy_(arena, non_blocking=True)
        t.cuda.current_stream(self.device).synchronize()
        return host, host.numpy()

    def select_into(self, out, flags, a, b):
        """out[...] = where(flags != 0, a (scalar), b): one elementwise kernel, nothing allocated (`flags` is a uint8
        tensor holding 0 / 1, reinterpreted as bool)."""
        self.torch.where(flags.view(self.torch.bool), self.scalar(a, out.dtype), b, out=out)

    def scalar(self, value, dtype):
        """Cached 0-dim device tensor (created outside any stream capture: BatchedJssEnv makes the ones it needs
        when it is constructed)."""
        key = (int(value), str(dtype))
        if key not in self._scalars:
            dt = getattr(self.torch, dtype) if isinstance(dtype, str) else dtype
            self._scalars[key] = self.torch.tensor(int(value), dtype=dt, device=self.device)
            self._scalars[(int(value), str(dt))] = self._scalars[key]
        return self._scalars[key]

    # -- execution -------------------------------------------------------------------------
    def stream(self):
        """Raw handle of torch's current stream on this backend's device (the private accessor where torch has it: 0.3 us
        against 3 us for building a torch.cuda.Stream object and asking for its .cuda_stream -- per launch)."""
        raw = self._raw_stream
        if raw is not None:
            return raw(self.device.index)
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def sync(self):
        self.torch.cuda.current_stream(self.device).synchronize()

    def on_device(self):
        """Context making this backend's device current (kernel launches go to its streams)."""
        t = self.torch
        if t.cuda.current_device() == self.device.index:
            return _NULL_CTX
        return t.cuda.device(self.device)

    def with_streams(self, n, fn, events=None):
        """Call fn(streams) where streams is a (void* * n) array: the current stream plus n-1 side streams that
        are forked from it before the call and joined back into it afterwards (stream-ordered for the caller,
        capturable in a hipGraph).  The side streams are created once per process and device and shared by every
        env: HIP deals streams onto a handful of hardware queues in creation order, and a side stream that lands
        on the caller's queue serialises the sub-batches it was created to overlap (seen in a long-running bench
        process: 2 sub-batches slower than one launch until the streams were pinned like this).  The fork / join
        EVENTS belong to the caller (`events`: a dict the env object keeps), so two envs driven from two host
        threads order their side work against their own streams; the side streams themselves are still shared,
        i.e. such envs' sub-batch work is serialised on them -- one host thread per device is the intended use."""
        t = self.torch
        main = t.cuda.current_stream(self.device)
        pool = self.side_pool(n - 1)
        side = pool["streams"][:n - 1]
        ev = events if events is not None else pool.setdefault("events", {})
        if "fork" not in ev:
            ev["fork"], ev["join"] = t.cuda.Event(), []
        while len(ev["join"]) < n - 1:
            ev["join"].append(t.cuda.Event())
        if side:
            ev["fork"].record(main)
            for st in side:
                st.wait_event(ev["fork"])
        arr = (C.c_void_p * n)(main.cuda_stream, *[st.cuda_stream for st in side])
        rc = fn(arr)
        for i, st in enumerate(side):
            ev["join"][i].record(st)
            main.wait_event(ev["join"][i])
        return rc

    def stream_array(self, n):
        """(void* * n): the current stream + n - 1 of the process-wide side streams, for JSS_ROLLOUT_FORK_JOIN calls."""
        main = self.stream()
        key = (n, main)
        arr = self._stream_arrays.get(key)
        if arr is None:
            side = self.side_pool(n - 1)["streams"][:n - 1]
            arr = self._stream_arrays[key] = (C.c_void_p * n)(main, *[st.cuda_stream for st in side])
        return arr

    def side_pool(self, n):
        """The process-wide side streams of this device (at least n of them)."""
        t = self.torch
        pool = _SIDE_STREAMS.setdefault(self.device.index, {"streams": []})
        while len(pool["streams"]) < n:
            pool["streams"].append(t.cuda.Stream(device=self.device))
        return pool

    def close(self):
        pass


_SIDE_STREAMS = {}    # device index -> side streams / events of HipBackend.with_streams (process-wide)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL_CTX = _NullCtx()


class CpuBackend:
    """Host memory = NumPy arrays; stepping = libjss_cpu.so (C++17 + OpenMP over envs, same C ABI, written from the
    kernels' queue-free state).  Explicit choice only (``device='cpu'``); never a fallback of the HIP path."""

    name = "cpu"
    default_kernel = "auto"
    device = "cpu"

    def __init__(self, threads: int = 0):
        from .build import build_cpu_twin
        try:
            path = build_cpu_twin()               # no-op when the in-tree library is newer than its sources
        except Exception as exc:                  # no compiler on this host: a prebuilt library of this ABI still serves
            path = _abi.library_path("libjss_cpu.so")
            if not os.path.isfile(path):
                raise RuntimeError(f"device='cpu' needs {path}: build it with `g++ -O3 -std=c++17 -fopenmp -fPIC -shared "
                                   f"-Iinclude jssenv_amd/csrc/jss_cpu.cpp -o {path}` (automatic build failed: {exc})") from exc
        self.lib = _abi.bind(C.CDLL(path))
        if not self.lib.jss_backend().startswith(b"cpu"):
            raise RuntimeError(f"{path} is not the CPU twin ({self.lib.jss_backend()!r})")
        self.threads = int(threads)
        self._keep = None

    def zeros(self, shape, dtype):
        return np.zeros(shape, dtype=getattr(np, dtype))

    def from_numpy(self, a):
        return np.ascontiguousarray(a).copy()

    def ptr(self, x):
        if x is None:
            return 0
        assert x.flags["C_CONTIGUOUS"]
        return x.ctypes.data

    def numpy(self, x):
        return np.array(x, copy=True)

    def as_device(self, x, dtype):
        a = np.ascontiguousarray(np.asarray(x).astype(getattr(np, dtype), copy=False))
        self._keep = a
        return a

    def stage(self, buf, x):
        a = np.asarray(x)
        if a.shape != buf.shape:
            raise ValueError(f"expected shape {buf.shape}, got {a.shape}")
        if a.dtype == buf.dtype and a.flags["C_CONTIGUOUS"]:
            self._keep = a
            return a
        buf[...] = a
        return buf

    def copy_into(self, dst, src):
        dst[...] = src

    def carve(self, arena, off, shape, dtype):
        return _carve_numpy(arena, off, shape, dtype)

    def snapshot(self, arena, host=None):
        return arena, arena                 # host memory already: the "copy" is the arena itself

    def select_into(self, out, flags, a, b):
        np.copyto(out, b)
        out[flags != 0] = a

    def stream(self):
        return 0

    def sync(self):
        pass

    def on_device(self):
        return _NULL_CTX

    def with_streams(self, n, fn, events=None):
        return fn((C.c_void_p * n)())

    def close(self):
        pass


def _carve_numpy(arena, off, shape, dtype):
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    return arena[off:off + n].view(dt).reshape(shape)


def make_backend(device=None):
    """``None`` / ``'cuda[:i]'`` -> HipBackend (raises without a GPU); ``'cpu'`` -> CpuBackend."""
    if device is not None and str(device).startswith("cpu"):
        return CpuBackend()
    return HipBackend(device)
