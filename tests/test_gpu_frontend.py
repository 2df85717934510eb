"""The model's caller (reference models/base_model.py:274-325) on the GPU against the oracle:
street-map CNN at the node pixels, edge features, per-sample inputs, and _call end to end.

Tolerance: copies (edge features, the concatenated x) bit-exact; float results
max|got - want| / max|want| <= 1e-5."""
import numpy as np
import pytest
import torch

from conftest import assert_close, rel_err
from gpu_helpers import to_dev

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(scope='module')
def amd():
    import gunet_amd
    return gunet_amd


def _city(oracle, height, width, density, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    mask = rng.random((height, width)) < density
    node_loc, ei, d, _ = oracle.image_to_graph(oracle.directional_maps_from_mask(mask))
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    return mask, node_loc, quads, oracle.synth_street_map(mask, seed + 1)


def _weights(oracle, seed=1):
    shapes = dict(oracle.graph_unet_shapes())
    shapes.update(oracle.cnn_shapes())
    return oracle.make_weights(shapes, seed)


def _model(amd, W):
    return amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                         use_bias=True, use_global=True).load_weights(W, 'cuda')


def test_street_encoding_and_edge_features(amd, oracle):
    mask, node_loc, quads, smap = _city(oracle, 23, 31, 0.6, 3)
    mask[0, :] = mask[:, 0] = mask[-1, :] = mask[:, -1] = True         # nodes on every border
    node_loc, ei, d, _ = oracle.image_to_graph(oracle.directional_maps_from_mask(mask))
    quads = oracle.sample_quadrant_subgraphs(ei, d)
    smap = oracle.synth_street_map(mask, 4)
    W = _weights(oracle)
    want = oracle.forward_street_encoding(smap, W, node_loc)
    assert want.shape == (node_loc.shape[0], 8) and np.abs(want).max() > 0
    model = _model(amd, W)
    x = {'image': {'street_map': to_dev(smap)}, 'graph': {'node_loc': to_dev(node_loc, torch.int64)}}
    got = model.forward_street_encoding(x)
    assert_close(got.cpu().numpy(), want, TOL)
    for q in quads:
        feats = model.compute_edge_features(got, to_dev(q, torch.int32))
        assert np.array_equal(feats.cpu().numpy(), oracle.compute_edge_features(got.cpu().numpy(), q))
    with pytest.raises(TypeError):
        model.forward_street_encoding({'image': {'street_map': to_dev(smap).float()}, 'graph': x['graph']})


def test_model_call(amd, oracle):
    """_call from the reference's nested input dict to the clamped prediction; single sample and a
    batch of two different samples of one city."""
    mask, node_loc, quads, smap = _city(oracle, 40, 52, 0.45, 0)
    W = _weights(oracle)
    model = _model(amd, W)
    n = node_loc.shape[0]
    graph_dev = {'edge_index': [to_dev(q, torch.int32) for q in quads], 'node_loc': to_dev(node_loc, torch.int64)}
    image_dev = {'street_map': to_dev(smap)}
    wants, seeds, tencs, wds = [], [], [], []
    for b in range(2):
        sn, te, wd = oracle.synth_sample(node_loc, quads, 300 + b, time_idx=5 * b, weekday=3 + b)
        x_np = {'graph': {'seed_nodes': sn, 'edge_index': quads, 'node_loc': node_loc},
                'image': {'street_map': smap}, 'time_enc': te, 'weekday': wd}
        wants.append(oracle.model_call(x_np, W))
        seeds.append(sn); tencs.append(te); wds.append(wd)
        x_dev = {'graph': dict(graph_dev, seed_nodes=to_dev(sn)), 'image': image_dev,
                 'time_enc': to_dev(te), 'weekday': to_dev(wd)}
        got = model._call(x_dev)
        assert tuple(got.shape) == (n, 96) and float(got.min()) >= 0
        assert_close(got.cpu().numpy(), wants[b], TOL, str(b))
    xb = {'graph': dict(graph_dev, seed_nodes=to_dev(np.stack(seeds))), 'image': image_dev,
          'time_enc': to_dev(np.concatenate(tencs)), 'weekday': to_dev(np.concatenate(wds))}
    got = model._call(xb)
    for b in range(2):
        assert_close(got[b].cpu().numpy(), wants[b], TOL, str(b))
    with pytest.raises(NotImplementedError):
        model._call(xb, training=True)


def test_model_inputs(amd, oracle):
    """x = [seed | street encoding] is a copy (bit-exact); the global vector matches the oracle."""
    rng = np.random.Generator(np.random.PCG64(8))
    n, batch = 1000, 3
    seed = rng.random((batch, n, 96), dtype=np.float32)
    enc = rng.random((n, 8), dtype=np.float32)
    tenc = rng.standard_normal((batch, 12, 2)).astype(np.float32)
    wd = np.array([0, 6, 3])
    lib = amd._lib
    xin = torch.empty((batch, n, 104), device='cuda')
    glob = torch.empty((batch, 113), device='cuda')
    ws = torch.empty(int(lib.lib.gunet_model_inputs_workspace_bytes(n, batch, 96, 8)), dtype=torch.uint8, device='cuda')
    t_last = to_dev(np.ascontiguousarray(tenc[:, -1]))
    lib.call('gunet_model_inputs_f32', lib.ptr(to_dev(seed)), 96, lib.ptr(to_dev(enc)), 8, lib.ptr(t_last), 2,
             lib.ptr(to_dev(wd, torch.int32)), n, batch, lib.ptr(xin), lib.ptr(glob), lib.ptr(ws), ws.numel(), None)
    torch.cuda.synchronize()
    for b in range(batch):
        want_x, want_g = oracle.assemble_inputs(seed[b], enc, [np.zeros((0, 2), np.int32)] * 4, None,
                                                tenc[b:b + 1], wd[b:b + 1])
        assert np.array_equal(xin[b].cpu().numpy(), want_x)
        assert rel_err(glob[b:b + 1].cpu().numpy(), want_g['global']) <= 1e-6
    with pytest.raises(amd.GunetError):
        lib.call('gunet_model_inputs_f32', lib.ptr(to_dev(seed)), 96, lib.ptr(to_dev(enc)), 8, lib.ptr(t_last), 2,
                 lib.ptr(to_dev(wd, torch.int32)), n, batch, lib.ptr(xin), lib.ptr(glob), lib.ptr(ws), 16, None)


def test_image_to_graph(amd, oracle):
    """Graph construction from the 8 directional maps: bit-exact with the oracle, incl. maps that
    point out of the image (masked) and the self-loop option."""
    rng = np.random.Generator(np.random.PCG64(11))
    for (h, w, dens) in [(9, 14, 0.6), (33, 27, 0.4), (495, 436, 0.45)]:
        mask = rng.random((h, w)) < dens
        maps = oracle.directional_maps_from_mask(mask).astype(np.uint8)
        if h < 100:
            maps[:, 0, :] |= 1          # edges leaving the image through every border
            maps[:, -1, :] |= 1
            maps[:, :, 0] |= 1
            maps[:, :, -1] |= 1
            maps = maps * np.uint8(255)   # 0 / 255 valued
        for selfloops in (True, False):
            want_loc, want_ei, want_dir, want_mask = oracle.image_to_graph(maps.copy(), add_selfloops=selfloops)
            loc, ei, direction, m = amd.data_utils.image_to_graph(to_dev(maps), return_img_mask=True,
                                                                  add_selfloops=selfloops)
            assert np.array_equal(loc.cpu().numpy(), want_loc)
            assert np.array_equal(ei.cpu().numpy(), want_ei)
            assert np.array_equal(direction.cpu().numpy(), want_dir)
            assert np.array_equal(m.cpu().numpy(), want_mask)
    empty = amd.data_utils.image_to_graph(torch.zeros((8, 5, 6), dtype=torch.uint8, device='cuda'))
    assert empty[0].shape == (0, 2) and empty[1].shape == (0, 2) and empty[2].shape == (0,)


def test_prediction_image_and_metrics(amd, oracle):
    rng = np.random.Generator(np.random.PCG64(12))
    h, w = 21, 17
    mask = rng.random((h, w)) < 0.5
    node_loc = np.stack(np.nonzero(mask), axis=-1).astype(np.int64)
    n = node_loc.shape[0]
    pred = (rng.random((2, n, 96), dtype=np.float32) * np.float32(1.2)).astype(np.float32)    # some above 1 -> clip
    pred[0, :5, :8] = np.array([0.5, 1.5, 2.5, 3.5, 254.5], dtype=np.float32)[:, None] / np.float32(255)   # ties
    got = amd.submission.graph_to_image(to_dev(pred), to_dev(node_loc, torch.int64), h, w)
    assert tuple(got.shape) == (2, 6, h, w, 8) and got.dtype == torch.uint8
    for b in range(2):
        assert np.array_equal(got[b].cpu().numpy(), oracle.predict_sample_image(pred[b], mask))
    one = amd.submission.graph_to_image(to_dev(pred[1]), to_dev(node_loc, torch.int64), h, w)
    assert torch.equal(one, got[1])

    target = rng.random((n, 96), dtype=np.float32)
    seed = rng.random((n, 96), dtype=np.float32)
    want = oracle.compute_metrics(target, pred[0])
    have = amd.evaluation.compute_metrics(to_dev(target), to_dev(pred[0]))
    assert set(have) == set(want) and len(want) == 13
    for k in want:
        assert abs(have[k] - want[k]) <= 1e-5 * abs(want[k]), k
    so = amd.evaluation.compute_metrics(to_dev(target), to_dev(seed), score_only=True)
    assert list(so) == ['mse_scaled']
    model = amd.GraphUNet(32, out_units=96, depth=3, activation='relu', layer_type='geo_quadrant_gcn',
                          use_bias=True, use_global=True)
    data = {'graph': {'seed_nodes': to_dev(seed), 'target_nodes': to_dev(target)},
            'image': {'street_map': torch.zeros((h, w), dtype=torch.uint8, device='cuda')}}
    full = model.compute_metrics(data, to_dev(pred[0]))
    want_full = oracle.model_compute_metrics(seed, target, pred[0], h, w)
    for k in ('score', 'rel_score', 'mse', 'mae_scaled_speed', 'mse_scaled_60_min'):
        assert abs(full[k] - want_full[k]) <= 1e-5 * abs(want_full[k]), k
    # batched sums = per-sample sums
    sums = amd.evaluation.error_sums(to_dev(np.stack((target, seed))), to_dev(pred))
    assert torch.equal(sums[0], amd.evaluation.error_sums(to_dev(target), to_dev(pred[0])))
    with pytest.raises(NotImplementedError):
        amd.evaluation.compute_metrics(to_dev(target), to_dev(seed), mask=torch.ones(1))
