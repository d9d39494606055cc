-- nms.lua -- main.lua:7 / Detector.lua:3 `require 'nms'`: the global nms(boxes, overlap, scores) of the reference's nms.lua
-- (same dispatch on `scores`, same 1-based LongTensor result) served by frcnn_nms_host.  Found in place of the reference's
-- file when bindings/ precedes it on package.path.
local hip = require 'frcnn_hip'
nms = hip.nms
return nms
