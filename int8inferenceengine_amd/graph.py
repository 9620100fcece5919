"""Whole-forward replay as one HIP graph.

The reference's forward is a chain of host calls (i8ie/module.py: forward() of the user's Module).  On the GPU a small
batch (the 125-image shard of the 8-GPU configuration) is launch-bound: ~14 dependent launches, each preceded by
Python + pybind dispatch.  `GraphedForward` records the launches of one `forward(x)` into a HIP graph
(include/i8ie_hip.h, i8ie_graph_*) and replays them with one call; results are the same bytes as the eager forward
(the same kernels on the same buffers).

    g = GraphedForward(net, x)       # x: resident i8ie FP32 tensor; runs net(x) once eagerly, then captures it
    y = g()                          # replay; y is always the same device tensor, overwritten by every replay
    g.load(next_batch)               # new input values into the captured input buffer (same shape)

The ctx must run on a stream of its own (the default after `import i8ie`; when borrowing torch's stream with
`use_stream`, make a `torch.cuda.Stream()` current first: the legacy default stream cannot be captured).
Everything the forward needs lazily (workspace, re-packed weights, offset vectors) is created by the eager run;
a forward whose launches depend on host-side values that change from call to call cannot be captured.
"""
import _CXX_i8ie as cx


class GraphedForward:
    def __init__(self, forward, x):
        self.x = x
        self.forward = forward  # the graph replays kernels that read the layers' device weights: keep them alive
        y = forward(x)          # eager: creates every lazily built cache, sizes the workspace
        y.data.prefetch()
        cx.synchronize()
        del y
        cx.graph_begin()
        try:
            self.y = forward(x)
            self.y.data.prefetch()  # a recorded (deferred) last launch is issued here, inside the capture
        finally:
            self.graph = cx.graph_end()
        self.kernel_nodes, self.nodes = self.graph.nodes()

    def __call__(self, x=None):
        if x is not None and x is not self.x:
            raise ValueError("a captured forward reads the tensor it was captured with: use load() to change its values")
        self.graph.launch()
        return self.y

    def load(self, array):
        cx.upload_into(self.x.data, array)
