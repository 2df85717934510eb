"""Metrics of a prediction in graph format (reference evaluation.py:173-244)."""
import math

import torch

from . import _lib, plan as _plan
from ._lib import call, ptr, current_stream

VOL_CHANNEL_SLICE = slice(0, 8, 2)     # reference data/__init__.py:6
SPEED_CHANNEL_SLICE = slice(1, 8, 2)   # reference data/__init__.py:7


def error_sums(y_true, y_pred):
    """float64 [B, 2, C] (or [2, C]): per column the sum over the rows of (t - p)^2 and of |t - p|.
    Every metric below is a ratio of these; a multi-GPU run all-reduces them (sharding.sum_metrics)."""
    if y_true.shape != y_pred.shape or y_true.dtype != torch.float32 or y_pred.dtype != torch.float32:
        raise TypeError('y_true and y_pred must be float32 tensors of one shape')
    single = y_true.dim() == 2
    t = (y_true.unsqueeze(0) if single else y_true).contiguous()
    p = (y_pred.unsqueeze(0) if single else y_pred).contiguous()
    batch, n, c = t.shape
    sums = torch.empty((batch, 2, c), dtype=torch.float64, device=t.device)
    scratch = _plan.workspace(int(_lib.lib.gunet_error_sums_workspace_bytes(n, batch, c)), t.device)
    call('gunet_error_sums_f32', ptr(t), ptr(p), n, c, batch, ptr(sums), ptr(scratch), scratch.numel(),
         current_stream(t.device))
    return sums[0] if single else sums


def metrics_from_sums(sums, n_rows, data_scale=255, score_only=False):
    """The reference's metric dictionary from error sums [2, C] over n_rows rows (host floats)."""
    sq, ab = [list(map(float, row)) for row in sums.tolist()]
    c = len(sq)
    count = float(n_rows) * c
    mse = sum(sq) / count
    mse_scaled = mse * data_scale * data_scale
    if score_only:
        return {'mse_scaled': mse_scaled}
    mae = sum(ab) / count
    # the reference slices the LAST axis of the [N, features * time] tensors with slice(1, 8, 2) /
    # slice(0, 8, 2) (:214-219): columns 1,3,5,7 / 0,2,4,6, i.e. the first time step only
    speed_cols, vol_cols = list(range(c))[SPEED_CHANNEL_SLICE], list(range(c))[VOL_CHANNEL_SLICE]
    out = {
        'mse': mse,
        'mae': mae,
        'mse_scaled': mse_scaled,
        'mae_scaled': mae * data_scale,
        'mae_scaled_speed': sum(ab[i] for i in speed_cols) / (float(n_rows) * len(speed_cols)) * data_scale,
        'mae_scaled_volume': sum(ab[i] for i in vol_cols) / (float(n_rows) * len(vol_cols)) * data_scale,
        'rmse_scaled': math.sqrt(mse_scaled),
    }
    width = c // 12
    for i in (0, 1, 2, 5, 8, 11):
        cols = range(width * i, width * (i + 1))
        out[f'mse_scaled_{(i + 1) * 5}_min'] = sum(sq[j] for j in cols) / (float(n_rows) * width) * data_scale ** 2
    return out


def compute_metrics(y_true, y_pred, data_scale=255, mask=None, score_only=False):
    """reference compute_metrics for graph-format tensors [N, features * time]; image format
    (mask is not None) is not implemented."""
    if mask is not None:
        raise NotImplementedError('image-format metrics (masked_mse) are not implemented')
    if y_true.dim() != 2:
        raise ValueError('graph-format metrics take [N, features * time] tensors')
    return metrics_from_sums(error_sums(y_true, y_pred), y_true.shape[0], data_scale, score_only)
