"""Replay of a whole forward as one HIP graph (no counterpart in the reference, whose TF2 path
wraps the same call in tf.function): for callers that want a forward as a single graph launch,
or that drive the per-layer mirrors from Python, where for one or two samples the host cannot
issue the ~110 launches as fast as the GPU retires them.  Shapes, the city graph and the weights are frozen at capture; the inputs are copied into
static buffers before every replay."""
import torch

from . import plan as _plan


class CapturedForward:
    """cap = CapturedForward(model, x, graph); y = cap(x, edges, glob)

    x [N, F] or [B, N, F], graph = {'edge_index', 'edges', 'node_loc', 'global'} as for
    model.predict().  The returned tensor is the graph's static output buffer (clone it to keep a
    result across replays).  Index tensors are shared, not copied: they must stay alive and
    unchanged."""

    def __init__(self, model, x, graph, warmup=2):
        self.model = model
        self.x = x.clone()
        self.edges = [e.clone() for e in graph['edges']]
        self.glob = graph['global'].clone()
        self.edge_index = list(graph['edge_index'])
        self.node_loc = graph['node_loc']
        self.scratch = _plan.WorkspaceScope()               # the graph's own scratch, alive as long as it is
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with self.scratch:
            with torch.cuda.stream(side):                   # builds every plan and workspace first
                for _ in range(max(1, warmup)):
                    self._run()
            torch.cuda.current_stream(x.device).wait_stream(side)
            torch.cuda.synchronize(x.device)
            self.hip_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.hip_graph):
                self.y = self._run()

    def _run(self):
        g = {'edge_index': list(self.edge_index), 'edges': list(self.edges), 'node_loc': self.node_loc,
             'global': self.glob}
        return self.model.predict(self.x, g)

    def __call__(self, x, edges=None, glob=None):
        self.x.copy_(x)
        if edges is not None:
            for dst, src in zip(self.edges, edges):
                dst.copy_(src)
        if glob is not None:
            self.glob.copy_(glob)
        self.hip_graph.replay()
        return self.y
