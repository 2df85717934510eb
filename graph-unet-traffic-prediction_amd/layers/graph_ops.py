"""Graph primitives on the Graph U-Net path (reference layers/graph_ops.py:5-17)."""
import torch

from .._lib import call, current_stream, ptr


def get_sender_receiver(node_features, edge_index):
    """Features of the sending and of the receiving node of every edge:
    (node_features[edge_index[:, 0]], node_features[edge_index[:, 1]])."""
    x = node_features.contiguous()
    if x.dtype != torch.float32 or x.dim() != 2:
        raise TypeError('node_features must be float32 [N, C]')
    e = edge_index.to(torch.int32).contiguous()
    senders = torch.empty((e.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    receivers = torch.empty_like(senders)
    call('gunet_get_sender_receiver_f32', ptr(x), x.shape[0], x.shape[1], ptr(e), e.shape[0],
         ptr(senders), ptr(receivers), current_stream(x.device))
    return senders, receivers
