from .graph_unet import GraphUNet

MODELS = {'GraphUNet': GraphUNet}
