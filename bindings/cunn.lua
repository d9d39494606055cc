-- cunn.lua -- found in place of the `cunn` package when bindings/ precedes it on package.path (LUA_PATH="<repo>/bindings/?.lua;;").
-- The reference's main.lua:6, objective.lua:1 and Detector.lua:1 say `require 'cunn'` and stay byte-identical: what they get
-- is the C-ABI binding (device tensors, create_model, nms, cutorch.*, optim.rmsprop: frcnn_hip.lua) plus the stand-alone
-- nn modules of SURVEY 8b (frcnn_nn.lua) that the reference's OWN objective.lua / Detector.lua construct with :cuda().
-- Not executed in this image (no Lua runtime); checked statically by tests/test_abi.py.
local hip = require 'frcnn_hip'
require 'frcnn_nn'
return hip
