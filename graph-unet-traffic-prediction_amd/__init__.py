"""MI355X-native Graph U-Net forward path: HIP kernels behind the reference's layer API.

Layout mirrors the reference modules that lie on the path:
    layers.hybrid_gnn   GeoQuadrantGCN, GraphDownsamplingBlock, GraphUpsamplingBlock,
                        GraphUpsampling2x2, graph_pooling_2x2, remove_duplicate_edges
    layers.graph_ops    get_sender_receiver
    data.data_utils     sample_quadrant_subgraphs, DIR_TO_EDGE, QUADRANT_IDS
    models.graph_unet   GraphUNet
"""
from . import _lib                                   # raises if libgunet_hip.so is missing
from ._lib import GunetError, LIB_PATH, set_matmul_mode, get_matmul_mode, set_kernel_choice
from . import plan
from . import layers
from .layers import hybrid_gnn, graph_ops
from .data import data_utils
from .models import graph_unet, base_model
from . import evaluation, submission
from .capture import CapturedForward
from .models.graph_unet import GraphUNet

__all__ = ['GunetError', 'LIB_PATH', 'set_matmul_mode', 'get_matmul_mode', 'plan', 'layers', 'hybrid_gnn', 'graph_ops', 'data_utils',
           'graph_unet', 'GraphUNet']
