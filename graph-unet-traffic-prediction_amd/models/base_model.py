"""Caller of the Graph U-Net (reference models/base_model.py:259-331, inference part): the
street-map CNN, the edge features gathered from it, the per-sample inputs and the final clamp."""
import ctypes as C
from collections import OrderedDict

import torch

from .. import _lib, plan as _plan
from .._lib import call, ptr, current_stream


_CITY_LIMIT = 8      # (city, batch) entries of street encodings / edge features kept per model


class GraphBaseModel:
    """`_call(x)` takes the reference's nested input
        x = {'graph': {'seed_nodes': [N, S] (or [B, N, S]), 'edge_index': 4 x int32 [E_q, 2],
                       'node_loc': int64 [N, 2]},
             'image': {'street_map': uint8 [H, W]}, 'time_enc': [B, T, 2], 'weekday': int [B]}
    and returns max(model(traffic_input, graph), 0) (reference :287-325).  Everything that depends
    on the city only (street encoding, edge features) is computed once per city and cached."""

    def __init__(self):
        self.cnn1_kernel = None      # [3, 3, 1, 32]  (reference :263)
        self.cnn2_kernel = None      # [3, 3, 32, 8]  (reference :264)
        self._city = OrderedDict()   # least recently used first; at most _CITY_LIMIT entries

    def load_frontend_weights(self, flat, device='cuda'):
        device = torch.device(device)
        self.cnn1_kernel = torch.as_tensor(flat['cnn1/kernel'], dtype=torch.float32).to(device).contiguous()
        self.cnn2_kernel = torch.as_tensor(flat['cnn2/kernel'], dtype=torch.float32).to(device).contiguous()
        if tuple(self.cnn1_kernel.shape) != (3, 3, 1, 32) or tuple(self.cnn2_kernel.shape) != (3, 3, 32, 8):
            raise ValueError('cnn1/kernel must be [3,3,1,32] and cnn2/kernel [3,3,32,8]')
        self._city.clear()
        return self

    def forward_street_encoding(self, x):
        """reference :274-280 -> float32 [N, 8]"""
        if self.cnn1_kernel is None:
            raise RuntimeError('street-map CNN weights are not loaded (load_frontend_weights)')
        img = x['image']['street_map']
        node_loc = x['graph']['node_loc']
        if img.dtype != torch.uint8 or img.dim() != 2 or not img.is_cuda:
            raise TypeError('street_map must be a uint8 [H, W] CUDA tensor')
        if node_loc.dtype != torch.int64 or node_loc.dim() != 2 or node_loc.shape[1] != 2:
            raise TypeError('node_loc must be an int64 [N, 2] tensor')
        img, node_loc = img.contiguous(), node_loc.contiguous()
        out = torch.empty((node_loc.shape[0], 8), dtype=torch.float32, device=img.device)
        call('gunet_street_encoding_f32', ptr(img), img.shape[0], img.shape[1], ptr(self.cnn1_kernel),
             ptr(self.cnn2_kernel), ptr(node_loc), node_loc.shape[0], ptr(out), current_stream(img.device))
        return out

    def compute_edge_features(self, street_encoding, edge_index):
        """reference :282-285 -> float32 [E, 2 * width]"""
        edge_index = edge_index.contiguous()
        if edge_index.dtype != torch.int32:
            raise TypeError('edge_index must be int32')
        n, w = street_encoding.shape
        out = torch.empty((edge_index.shape[0], 2 * w), dtype=torch.float32, device=street_encoding.device)
        call('gunet_edge_features_f32', ptr(street_encoding), n, w, ptr(edge_index), edge_index.shape[0], ptr(out),
             current_stream(out.device))
        return out

    def get_seeds(self, x):
        return x['graph']['seed_nodes']

    def get_targets(self, x):
        return x['graph']['target_nodes']

    def compute_metrics(self, traffic_data, pred):
        """reference :333-351: the metric dictionary plus the leaderboard-style 'score' (the scaled MSE
        spread over all pixels of the image) and 'rel_score' (the same for the untouched input)."""
        from .. import evaluation
        seed, target = self.get_seeds(traffic_data), self.get_targets(traffic_data)
        metrics = evaluation.compute_metrics(target, pred, mask=None)
        seed_metrics = evaluation.compute_metrics(target, seed, mask=None, score_only=True)
        h, w = traffic_data['image']['street_map'].shape[:2]
        n_nodes = float(pred.shape[0])
        n_zeros = float(h * w) - n_nodes
        metrics['score'] = metrics['mse_scaled'] * n_nodes / (n_nodes + n_zeros)
        metrics['rel_score'] = seed_metrics['mse_scaled'] * n_nodes / (n_nodes + n_zeros) / metrics['score']
        return metrics

    def _city_features(self, x):
        g = x['graph']
        tk = _plan._tensor_key
        key = (tk(x['image']['street_map']), tk(g['node_loc']), tuple(tk(e) for e in g['edge_index']),
               tk(self.cnn1_kernel), tk(self.cnn2_kernel))
        hit = self._city.get(key)
        if hit is not None:
            self._city.move_to_end(key)
        else:
            enc = self.forward_street_encoding(x)
            # :294-298; [E_q, 16] each, shared by every sample of the city (never expanded per sample)
            edges = [self.compute_edge_features(enc, e) for e in g['edge_index']]
            # keep the keyed tensors alive so their addresses cannot be reused by another city
            hit = (enc, edges, (x['image']['street_map'], g['node_loc'], list(g['edge_index'])))
            self._city[key] = hit
            while len(self._city) > _CITY_LIMIT:
                self._city.popitem(last=False)
        return hit[0], hit[1]

    def _call(self, x, training=False):
        if training:
            raise NotImplementedError('only the inference path (training=False) is implemented')
        if not getattr(self, 'use_global', True):
            raise NotImplementedError('use_global=False is not implemented')
        seeds = self.get_seeds(x)
        if seeds.dtype != torch.float32:
            raise TypeError('seed_nodes must be float32')
        single = seeds.dim() == 2
        sb = (seeds.unsqueeze(0) if single else seeds).contiguous()
        batch, n, s_width = sb.shape
        enc, edges = self._city_features(x)
        dev = sb.device
        t_enc = x['time_enc'][:, -1].to(torch.float32).contiguous()                       # :309
        weekday = torch.as_tensor(x['weekday']).reshape(-1).to(device=dev, dtype=torch.int32).contiguous()
        if t_enc.shape[0] != batch or weekday.shape[0] != batch:
            raise ValueError(f'time_enc / weekday need one row per sample ({batch})')
        c_out = s_width + enc.shape[1]
        xin = torch.empty((batch, n, c_out), dtype=torch.float32, device=dev)
        glob = torch.empty((batch, c_out + t_enc.shape[1] + 7), dtype=torch.float32, device=dev)
        nbytes = int(_lib.lib.gunet_model_inputs_workspace_bytes(n, batch, s_width, enc.shape[1]))
        scratch = _plan.workspace(nbytes, dev)
        call('gunet_model_inputs_f32', ptr(sb), s_width, ptr(enc), enc.shape[1], ptr(t_enc), t_enc.shape[1],
             ptr(weekday), n, batch, ptr(xin), ptr(glob), ptr(scratch), scratch.numel(), current_stream(dev))
        graph = {'edge_index': list(x['graph']['edge_index']), 'edges': list(edges),
                 'node_loc': x['graph']['node_loc'], 'global': glob}                     # :302-312
        return self.predict(xin[0] if single else xin, graph)                            # :314,325
