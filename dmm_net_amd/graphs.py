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

import torch

from . import _lib


class SafeGraph:
    """``with g.capture(pool=...): ...`` then ``g.replay()``.  ``g.rewritten`` = (memset nodes, memcpy nodes) replaced by
    kernel nodes, ``g.left`` = memset / memcpy nodes that stayed (2-D / 3-D copies, host copies)."""

    def __init__(self):
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        self.rewritten, self.left = (0, 0), 0

    @contextlib.contextmanager
    def capture(self, pool=None, stream=None):
        kw = {} if stream is None else {"stream": stream}
        # thread_local: other threads of the process (a DataLoader's pin-memory thread, a second nn.DataParallel replica) may
        # allocate / record events while this thread captures; the work of THIS capture is issued by this thread and by autograd's
        # device thread onto the capturing stream
        with torch.cuda.graph(self.graph, pool=pool, capture_error_mode="thread_local", **kw):
            yield self
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
