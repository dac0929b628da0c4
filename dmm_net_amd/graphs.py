"""HIP-graph capture that is safe to REPLAY on this runtime: ``SafeGraph`` = ``torch.cuda.CUDAGraph(keep_graph=True)`` whose
memset / device-to-device memcpy nodes are turned into kernel nodes (``dmm_graph_nodes_to_kernels``,
``csrc/dmm_graph.hip``) before the graph is instantiated.

On the ROCm runtime of this image a replayed memset node is not reliably ordered before the kernel node behind it
(``profiles/r04_graph_memset_node.txt``).  The library's own kernels never use one; graphs that also hold MIOpen /
hipBLASLt / torch launches (the trainer's captured step, ``train_encoder.py``) do -- MIOpen's bf16 weight-gradient solvers
clear their fp32 split-K workspace with ``hipMemsetAsync`` -- and replayed non-finite weight gradients until the nodes
were rewritten (``tools/train_encoder_diag.py``).  Nothing in the reference corresponds to this (it launches eagerly).
"""
from __future__ import annotations

import contextlib
import ctypes
import gc

import torch

from . import _lib


class CaptureFailed(RuntimeError):
    """A stream capture was invalidated (by an API call elsewhere in the process, or by a launch the runtime refuses under
    capture): nothing was instantiated, the calling thread's current stream is what it was before."""


class SafeGraph:
    """``with g.capture(pool=...): ...`` then ``g.replay()``.  ``g.rewritten`` = (memset nodes, memcpy nodes) replaced by
    kernel nodes, ``g.left`` = memset / memcpy nodes that stayed (2-D / 3-D copies, host copies).

    The capture is driven by ``capture_begin`` / ``capture_end`` directly, inside a ``torch.cuda.stream`` scope:
    ``torch.cuda.graph.__exit__`` leaves the CAPTURE stream current when ``capture_end`` raises (every later launch of the
    process then fails with "operation not permitted when stream is capturing"), here the previous stream always comes back
    and a failed capture surfaces as ``CaptureFailed``.  Thread-local capture mode: other threads of the process (a DataLoader's
    pin-memory thread, a second ``nn.DataParallel`` replica) may allocate / record events while this one captures."""

    def __init__(self):
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        self.rewritten, self.left = (0, 0), 0

    @contextlib.contextmanager
    def capture(self, pool=None, stream=None):
        if stream is None:
            # torch's own capture stream (what ``torch.cuda.graph`` uses by default): ONE capture stream per process.  With a
            # stream of its own here, a step captured after other captures of the process (bench.py's default line) replayed
            # 15 % slower -- 15.1 against 13.3 ms (A / B in one process; the standalone step was unaffected)
            if torch.cuda.graph.default_capture_stream is None:
                torch.cuda.graph.default_capture_stream = torch.cuda.Stream()
            stream = torch.cuda.graph.default_capture_stream
        # what torch.cuda.graph does on entry: cached blocks go back, so that the graph's private pool gets fresh segments
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()
        stream.wait_stream(torch.cuda.current_stream())
        failed = None
        with torch.cuda.stream(stream):
            try:
                self.graph.capture_begin(pool=pool, capture_error_mode="thread_local")
            except Exception as e:                        # (e.g. the stream is still in a capture that was invalidated)
                raise CaptureFailed(f"capture_begin: {e}") from e
            try:
                yield self
            except BaseException as e:
                failed = e
            try:
                self.graph.capture_end()
            except Exception as e:
                failed = failed or e
        if failed is not None:
            if stream is torch.cuda.graph.default_capture_stream:
                torch.cuda.graph.default_capture_stream = None      # (a stream whose capture was invalidated is not reused)
            if isinstance(failed, Exception) and not isinstance(failed, CaptureFailed):
                raise CaptureFailed(f"{type(failed).__name__}: {failed}") from failed
            raise failed
        torch.cuda.current_stream().wait_stream(stream)
        self._finish()

    def _finish(self):
        n_set, n_cpy, left = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        raw = self.graph.raw_cuda_graph()
        rc = _lib.load().dmm_graph_nodes_to_kernels(ctypes.c_void_p(int(raw)), 3, ctypes.byref(n_set), ctypes.byref(n_cpy),
                                                     ctypes.byref(left))
        _lib.check(rc, "dmm_graph_nodes_to_kernels")
        self.rewritten, self.left = (n_set.value, n_cpy.value), left.value
        self.graph.instantiate()

    def replay(self):
        self.graph.replay()

    def pool(self):
        return self.graph.pool()
