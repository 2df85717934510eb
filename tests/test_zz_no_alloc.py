"""gunet_unet_forward_static_f32: the whole-model forward with its intermediates in the library's
static scratch buffer (no device allocation, no caller workspace), called through the C ABI and
compared bit for bit with gunet_unet_forward_f32 over a caller workspace.

Two calls in a row on the same GPU share the buffer; the second uses a smaller graph, so stale
contents of the first call lie behind and beyond everything it touches."""
import ctypes as C

import numpy as np
import pytest
import torch

from gpu_helpers import graph_to_dev, make_case, to_dev

pytestmark = pytest.mark.gpu


def _forward_pair(amd, oracle, W, size, batch, seed):
    from gunet_amd import _lib, plan
    from gunet_amd._lib import call, current_stream, ptr, ptr4
    x, graph, _ = make_case(oracle, *size, graph_seed=seed, sample_seed=seed + 1)
    model = amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                          use_bias=True, use_global=True, output_activation=None).load_weights(W, 'cuda')
    g = graph_to_dev(graph)
    rng = np.random.default_rng(seed)
    xb = to_dev(np.stack([x * rng.random(x.shape, dtype=np.float32) for _ in range(batch)]), torch.float32)
    glob = g['global'].repeat(batch, 1).contiguous()
    feats = [e.unsqueeze(0).repeat(batch, 1, 1).contiguous() for e in g['edges']]
    up = plan.get_unet_plan(g['node_loc'], g['edge_index'], 3)
    ws = model._native_weights(xb.shape[2], feats[0].shape[2], glob.shape[1], xb.device)
    u_last, g_last = model._native_widths
    n = xb.shape[1]
    nbytes = int(_lib.lib.gunet_unet_workspace_bytes(up.handle, C.byref(ws), batch))
    assert 0 < nbytes <= _lib.STATIC_SCRATCH_BYTES
    results = []
    for static in (False, True):
        out = torch.full((batch, n, 96), float('nan'), dtype=torch.float32, device='cuda')
        edges_out = [torch.full((batch, up.n_edges[0][q], u_last), float('nan'), dtype=torch.float32, device='cuda')
                     for q in range(4)]
        global_out = torch.full((batch, g_last), float('nan'), dtype=torch.float32, device='cuda')
        in_ptrs, out_ptrs = ptr4(feats), ptr4(edges_out)
        if static:
            call('gunet_unet_forward_static_f32', up.handle, C.byref(ws), batch, ptr(xb), C.byref(in_ptrs), 0, ptr(glob),
                 1, 1, ptr(out), C.byref(out_ptrs), ptr(global_out), current_stream(xb.device))
        else:
            scratch = torch.empty(nbytes + 256, dtype=torch.uint8, device='cuda')
            base = (scratch.data_ptr() + 255) // 256 * 256
            call('gunet_unet_forward_f32', up.handle, C.byref(ws), batch, ptr(xb), C.byref(in_ptrs), 0, ptr(glob),
                 1, 1, ptr(out), C.byref(out_ptrs), ptr(global_out), C.c_void_p(base), nbytes,
                 current_stream(xb.device))
        torch.cuda.synchronize()
        results.append([out.cpu().numpy(), global_out.cpu().numpy()] + [e.cpu().numpy() for e in edges_out])
    return results, nbytes


def test_static_scratch_forward_matches_workspace_forward(oracle):
    import gunet_amd as amd
    from gunet_amd import _lib
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    sizes = []
    # the second call is on a smaller graph with a smaller batch: it must not depend on what the
    # first left in the buffer
    for size, batch, seed in (((96, 90, 0.5), 2, 3), ((30, 26, 0.55), 1, 5)):
        (want, got), nbytes = _forward_pair(amd, oracle, W, size, batch, seed)
        sizes.append(nbytes)
        for a, b in zip(want, got):
            assert not np.isnan(b).any()
            assert np.array_equal(a, b)           # same kernels in the same order: bit-identical
    assert sizes[1] < sizes[0]


def test_static_scratch_forward_in_bf16_mode(oracle):
    """The same pair of calls in bf16 mode, where the intermediates inside the scratch buffer are bf16
    tensors: still bit-identical to the workspace form, outputs float32 and free of NaN."""
    import gunet_amd as amd
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    amd.set_matmul_mode('bf16')
    try:
        for size, batch, seed in (((96, 90, 0.5), 2, 3), ((30, 26, 0.55), 1, 5)):
            (want, got), _ = _forward_pair(amd, oracle, W, size, batch, seed)
            for a, b in zip(want, got):
                assert not np.isnan(b).any()
                assert np.array_equal(a, b)
    finally:
        amd.set_matmul_mode('f32')


def test_static_scratch_forward_allocates_nothing_and_runs_under_capture(oracle):
    """Device memory in use is the same before and after a call (hipMemGetInfo through torch), and
    the call can be recorded into a HIP graph right after its plan was made -- the plan's creation
    has already looked up the buffer's address -- and replayed with the eager call's bits."""
    import gunet_amd as amd
    from gunet_amd import plan
    from gunet_amd._lib import call, current_stream, ptr, ptr4
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    x, graph, _ = make_case(oracle, 44, 38, 0.5, graph_seed=9, sample_seed=10)
    model = amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                          use_bias=True, use_global=True, output_activation=None).load_weights(W, 'cuda')
    g = graph_to_dev(graph)
    xb = to_dev(x[None], torch.float32)
    feats = [e.unsqueeze(0).contiguous() for e in g['edges']]
    glob = g['global'].contiguous()
    up = plan.get_unet_plan(g['node_loc'], g['edge_index'], 3)
    ws = model._native_weights(xb.shape[2], feats[0].shape[2], glob.shape[1], xb.device)
    u_last, g_last = model._native_widths
    n = xb.shape[1]

    def buffers():
        return (torch.full((1, n, 96), float('nan'), device='cuda'),
                [torch.full((1, up.n_edges[0][q], u_last), float('nan'), device='cuda') for q in range(4)],
                torch.full((1, g_last), float('nan'), device='cuda'))

    def run(out, edges_out, global_out, stream):
        in_ptrs, out_ptrs = ptr4(feats), ptr4(edges_out)
        call('gunet_unet_forward_static_f32', up.handle, C.byref(ws), 1, ptr(xb), C.byref(in_ptrs), 0, ptr(glob),
             1, 1, ptr(out), C.byref(out_ptrs), ptr(global_out), stream)

    # captured first: nothing but the plan's creation has touched the static buffer's symbol so far
    cap_out = buffers()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    hip_graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(hip_graph, stream=side):
        run(*cap_out, current_stream(xb.device))
    hip_graph.replay()
    torch.cuda.synchronize()
    eager_out = buffers()
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    run(*eager_out, current_stream(xb.device))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before       # no device allocation, none freed
    assert torch.isfinite(eager_out[0]).all()
    assert torch.equal(cap_out[0], eager_out[0]) and torch.equal(cap_out[2], eager_out[2])
    for q in range(4):
        assert torch.equal(cap_out[1][q], eager_out[1][q])


def test_static_scratch_refuses_what_does_not_fit(oracle):
    import gunet_amd as amd
    from gunet_amd import _lib, plan
    from gunet_amd._lib import GunetError
    W = oracle.make_weights(oracle.graph_unet_shapes(), 1)
    x, graph, _ = make_case(oracle, 300, 280, 0.5)
    g = graph_to_dev(graph)
    model = amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                          use_bias=True, use_global=True, output_activation=None).load_weights(W, 'cuda')
    model.static_scratch = True
    with pytest.raises(GunetError) as err:
        model(to_dev(x, torch.float32), g)
    assert err.value.code == -3 and 'static scratch' in str(err.value)
    model.static_scratch = False       # the workspace form takes the same graph
    y, _ = model(to_dev(x, torch.float32), graph_to_dev(graph))
    assert torch.isfinite(y).all()
