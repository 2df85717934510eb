"""Prediction -> uint8 image (reference submission.py:69-94)."""
import torch

from ._lib import call, ptr, current_stream

HORIZONS = (0, 1, 2, 5, 8, 11)   # 5, 10, 15, 30, 45 and 60 minutes (reference :88-91)


def graph_to_image(pred, node_loc, height, width, data_scale=255, horizons=HORIZONS, channels=8):
    """pred float32 [N, 12 * channels] (or [B, N, ...]) at the node pixels node_loc int64 [N, 2] ->
    uint8 [len(horizons), H, W, channels] (or [B, ...]): scatter to the image, keep the requested
    time steps, round(pred * data_scale), clip to 0..255; zero where there is no node
    (reference graph_to_image :69-77 followed by predict_sample :87-93)."""
    if pred.dtype != torch.float32 or not pred.is_cuda:
        raise TypeError('pred must be a float32 CUDA tensor')
    single = pred.dim() == 2
    pb = (pred.unsqueeze(0) if single else pred).contiguous()
    batch, n, pw = pb.shape
    if pw % channels or max(horizons) >= pw // channels:
        raise ValueError(f'pred has {pw} columns: not {channels} channels x more than {max(horizons)} time steps')
    node_loc = node_loc.contiguous()
    if node_loc.dtype != torch.int64 or tuple(node_loc.shape) != (n, 2):
        raise TypeError(f'node_loc must be int64 [{n}, 2]')
    hz = torch.tensor(list(horizons), dtype=torch.int32, device=pb.device)
    out = torch.empty((batch, len(horizons), height, width, channels), dtype=torch.uint8, device=pb.device)
    call('gunet_prediction_image_u8', ptr(pb), n, pw, batch, ptr(node_loc), height, width, ptr(hz), len(horizons),
         channels, float(data_scale), ptr(out), current_stream(pb.device))
    return out[0] if single else out


def predict_sample(model, sample, data_scale=255):
    """model._call(sample) as a uint8 image (reference predict_sample :79-94); the image size is
    that of sample['image']['street_map']."""
    pred = model._call(sample)
    h, w = sample['image']['street_map'].shape
    return graph_to_image(pred, sample['graph']['node_loc'], int(h), int(w), data_scale)
