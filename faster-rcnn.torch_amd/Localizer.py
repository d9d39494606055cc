"""Localizer -- maps rects between input-image space and feature-map space by walking the
kernel/stride/padding list of every convolution and pooling layer on the path to an output.
Host-side mirror of Localizer.lua.  The reference discovers that list by walking nngraph nodes
(Localizer.lua:8-36); here the node handed to the constructor carries it directly (`node.layers`,
produced by the native model runtime, frcnn_model_localizer_layers)."""
import math

from .Rect import Rect


def _lua_mod(a, b):  # Lua 5.1 / LuaJIT: a - floor(a/b)*b
    return a - math.floor(a / b) * b


class Localizer(object):
    def __init__(self, outnode):
        layers = outnode.layers if hasattr(outnode, "layers") else outnode
        self.layers = [dict(kW=int(l[0]), kH=int(l[1]), dW=int(l[2]), dH=int(l[3]), padW=int(l[4]), padH=int(l[5]))
                       for l in layers]

    new = None

    def inputToFeatureRect(self, rect, layer_index=None):  # Localizer.lua:41-67
        n = layer_index or len(self.layers)
        minX, minY, maxX, maxY = rect.minX, rect.minY, rect.maxX, rect.maxY
        for l in self.layers[:n]:
            kW, kH, dW, dH = l["kW"], l["kH"], l["dW"], l["dH"]
            if dW < kW:  # :45-47
                minX -= kW - dW; minY -= kH - dH; maxX += kW - dW; maxY += kH - dH
            minX += l["padW"]; minY += l["padH"]; maxX += l["padW"]; maxY += l["padH"]  # :49
            minX = minX / dH  # :52 (dH for both axes, as in the reference)
            minY = minY / dH  # :53
            if _lua_mod(maxX - kW, dW) == 0:
                maxX = max((maxX - kW) / dW + 1, minX + 1)
            else:
                maxX = max(math.ceil((maxX - kW) / dW) + 1, minX + 1)
            if _lua_mod(maxY - kH, dH) == 0:
                maxY = max((maxY - kH) / dW + 1, minY + 1)  # :60 (dW, as in the reference)
            else:
                maxY = max(math.ceil((maxY - kH) / dH) + 1, minY + 1)
        return Rect(minX, minY, maxX, maxY).snapToInt()  # :66

    def inputToFeatureRectBatch(self, rects):
        """inputToFeatureRect for an (n, 4) float64 array of rects at once -- the same double-precision
        operations in the same order, elementwise (used by the per-step example assembly)."""
        import numpy as np
        r = np.array(rects, dtype=np.float64).reshape(-1, 4)
        minX, minY, maxX, maxY = r[:, 0].copy(), r[:, 1].copy(), r[:, 2].copy(), r[:, 3].copy()
        for l in self.layers:
            kW, kH, dW, dH = l["kW"], l["kH"], l["dW"], l["dH"]
            if dW < kW:
                minX -= kW - dW; minY -= kH - dH; maxX += kW - dW; maxY += kH - dH
            minX += l["padW"]; minY += l["padH"]; maxX += l["padW"]; maxY += l["padH"]
            minX = minX / dH
            minY = minY / dH
            ax = maxX - kW
            exact = (ax - np.floor(ax / dW) * dW) == 0
            maxX = np.maximum(np.where(exact, ax / dW + 1, np.ceil(ax / dW) + 1), minX + 1)
            ay = maxY - kH
            exact = (ay - np.floor(ay / dH) * dH) == 0
            maxY = np.maximum(np.where(exact, ay / dW + 1, np.ceil(ay / dH) + 1), minY + 1)
        return np.stack([np.floor(minX), np.floor(minY), np.ceil(maxX), np.ceil(maxY)], 1)

    def featureToInputRect(self, minX, minY, maxX, maxY, layer_index=None):  # Localizer.lua:69-79
        n = layer_index or len(self.layers)
        for l in reversed(self.layers[:n]):
            minX = minX * l["dW"] - l["padW"]
            minY = minY * l["dH"] - l["padW"]
            maxX = maxX * l["dW"] - l["padH"] + l["kW"] - l["dW"]
            maxY = maxY * l["dH"] - l["padH"] + l["kH"] - l["dH"]
        return Rect(minX, minY, maxX, maxY)


Localizer.new = Localizer
