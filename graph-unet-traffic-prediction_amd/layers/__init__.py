"""Layer registry (reference layers/__init__.py:6-19).  Only the layer on the Graph U-Net path
is provided; the reference's 'gcn', 'gat' and 'mpnn' baselines have a different call signature
and cannot be used inside the U-Net blocks."""
from . import hybrid_gnn

# list of all possible layers
LAYERS = {
    'geo_quadrant_gcn': hybrid_gnn.GeoQuadrantGCN,
}


def get(layer_name, default=None):
    layer = LAYERS.get(layer_name.lower(), default)
    if not layer:
        raise NotImplementedError(f'Layer with name "{layer_name}" is not implemented.')
    return layer
