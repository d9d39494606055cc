"""models/vgg_large.lua:3-25 -- the reference's large model factory."""
from .model_utilities import create_model


def vgg_large(cfg):
    layers = [
        dict(filters=64, kW=3, kH=3, padW=1, padH=1, dropout=0.0, conv_steps=2),
        dict(filters=128, kW=3, kH=3, padW=1, padH=1, dropout=0.4, conv_steps=2),
        dict(filters=256, kW=3, kH=3, padW=1, padH=1, dropout=0.4, conv_steps=3),
        dict(filters=512, kW=3, kH=3, padW=1, padH=1, dropout=0.4, conv_steps=3),
    ]
    anchor_nets = [dict(kW=3, n=256, input=3), dict(kW=3, n=256, input=4), dict(kW=5, n=256, input=4),
                   dict(kW=7, n=256, input=4)]
    class_layers = [dict(n=1024, dropout=0.5, batch_norm=True), dict(n=512, dropout=0.5)]
    return create_model(cfg, layers, anchor_nets, class_layers)
