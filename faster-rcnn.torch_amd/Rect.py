"""Rect -- half-open axis-aligned rectangle [minX,maxX) x [minY,maxY) in Lua-number (double)
coordinates.  Host-side mirror of the reference's Rect class (Rect.lua:12-171): same method
names, same arithmetic.  The broken table constructor / clone() of the reference (Rect.lua:13-17,
165-167) are not reproduced; clone() here simply copies."""
import math


class Rect(object):
    __slots__ = ("minX", "minY", "maxX", "maxY", "layer", "aspect", "index")

    def __init__(self, minX, minY, maxX, maxY):
        self.minX = minX; self.minY = minY; self.maxX = maxX; self.maxY = maxY
        self.layer = None; self.aspect = None; self.index = None

    new = None  # set below: Rect.new(...) like torch.class

    @staticmethod
    def empty():  # Rect.lua:26-28
        return Rect(0, 0, 0, 0)

    @staticmethod
    def fromXYWidthHeight(x, y, width, height):  # Rect.lua:30-32
        return Rect(x, y, x + width, y + height)

    @staticmethod
    def fromCenterWidthHeight(centerX, centerY, width, height):  # Rect.lua:34-36
        return Rect.fromXYWidthHeight(centerX - width * 0.5, centerY - height * 0.5, width, height)

    def scale(self, factorX, factorY=None):  # Rect.lua:38-43
        if factorY is None:
            factorY = factorX
        return Rect(self.minX * factorX, self.minY * factorY, self.maxX * factorX, self.maxY * factorY)

    def inflate(self, x, y):  # Rect.lua:45-47
        return Rect(self.minX - x, self.minY - y, self.maxX + x, self.maxY + y)

    def size(self):
        return self.width(), self.height()

    def width(self):
        return self.maxX - self.minX

    def height(self):
        return self.maxY - self.minY

    def area(self):
        return self.width() * self.height()

    def center(self):  # Rect.lua:65-67
        return (self.minX + self.maxX) / 2, (self.minY + self.maxY) / 2

    def isEmpty(self):
        return self.minX == self.maxX and self.minY == self.maxY

    def clip(self, c):  # Rect.lua:73-80
        return Rect(min(max(self.minX, c.minX), c.maxX), min(max(self.minY, c.minY), c.maxY),
                    max(min(self.maxX, c.maxX), c.minX), max(min(self.maxY, c.maxY), c.minY))

    def containsPt(self, x, y):
        return self.minX <= x and x < self.maxX and self.minY <= y and y < self.maxY

    def contains(self, o):
        return self.containsPt(o.minX, o.minY) and self.containsPt(o.maxX, o.maxY)

    def overlaps(self, o):  # Rect.lua:90-93 (strict)
        return self.minX < o.maxX and self.maxX > o.minX and self.minY < o.maxY and self.maxY > o.minY

    def normalize(self):
        l, r = (self.minX, self.maxX) if self.minX <= self.maxX else (self.maxX, self.minX)
        t, b = (self.minY, self.maxY) if self.minY <= self.maxY else (self.maxY, self.minY)
        return Rect(l, t, r, b)

    def unpack(self):
        return self.minX, self.minY, self.maxX, self.maxY

    @staticmethod
    def union(a, b):
        return Rect(min(a.minX, b.minX), min(a.minY, b.minY), max(a.maxX, b.maxX), max(a.maxY, b.maxY))

    @staticmethod
    def intersect(a, b):  # Rect.lua:126-136
        minx = max(a.minX, b.minX); miny = max(a.minY, b.minY)
        maxx = min(a.maxX, b.maxX); maxy = min(a.maxY, b.maxY)
        if maxx >= minx and maxy >= miny:
            return Rect(minx, miny, maxx, maxy)
        return Rect.empty()

    @staticmethod
    def IoU(a, b):  # Rect.lua:138-141 (no +1 convention, unlike nms)
        i = Rect.intersect(a, b).area()
        return i / (a.area() + b.area() - i)

    def totensor(self):  # FloatTensor under main.lua:51
        import numpy as np
        return np.array([self.minX, self.minY, self.maxX, self.maxY], dtype=np.float32)

    def snapToInt(self):  # Rect.lua:147-149
        return Rect(math.floor(self.minX), math.floor(self.minY), math.ceil(self.maxX), math.ceil(self.maxY))

    def offset(self, x, y):
        return Rect(self.minX + x, self.minY + y, self.maxX + x, self.maxY + y)

    def vertices(self):
        return [(self.minX, self.minY), (self.maxX, self.minY), (self.maxX, self.maxY), (self.minX, self.maxY)]

    def clone(self):
        r = Rect(self.minX, self.minY, self.maxX, self.maxY)
        r.layer = self.layer; r.aspect = self.aspect; r.index = self.index
        return r

    def __repr__(self):
        return "{ min: (%.2f, %.2f), max: (%.2f, %.2f), size: (%.2f x %.2f) }" % (
            self.minX, self.minY, self.maxX, self.maxY, self.width(), self.height())


Rect.new = Rect
