-- tools/make_torch7_fixtures.lua -- the one route from "parity unpinned" to "parity pinned by the reference".
--
-- Run ONCE on a machine that has the reference's own Torch7 stack (torch, nn, nngraph, optim; no GPU needed), from the
-- root of a checkout of andreaskoepf/faster-rcnn.torch:
--
--     cd faster-rcnn.torch && th /path/to/tools/make_torch7_fixtures.lua /path/to/repo/tests/golden/torch7_fixtures.t7
--
-- It writes ONE torch object file (ASCII, the reference's own save_obj format, utilities.lua:113-118) holding, for every row
-- of oracle/ASSUMPTIONS.md, a small input and what the reference's packages compute from it, plus the flat parameter order
-- (utilities.lua:136-147) and one save_model snapshot (utilities.lua:126-134).  tests/test_torch7_fixtures.py consumes the
-- file when it is present (faster-rcnn.torch_amd/t7.py reads it) and compares every entry with oracle/ -- and skips, saying
-- so, while it is absent.  NOTHING here has ever been executed: this image has no Lua.  The file is data (inputs and
-- expected outputs), not source.
--
-- Every tensor is a FloatTensor unless said otherwise; seeds are fixed; sizes are tiny (the whole file is < 1 MB).

require 'torch'
require 'nn'
require 'nngraph'
require 'optim'
require 'image'

torch.setdefaulttensortype('torch.FloatTensor')
local out_path = arg and arg[1] or 'torch7_fixtures.t7'
local F = { version = 1, rows = {} }

local function seeded(n) torch.manualSeed(1000 + n) end
local function rnd(...) return torch.randn(...):float() end

-- row 1: nn.SpatialConvolution -- forward, updateGradInput, accGradParameters (accumulating, scale 1)
do
  seeded(1)
  local m = nn.SpatialConvolution(2, 3, 3, 3, 1, 1, 1, 1)
  local x, gy = rnd(2, 5, 6), rnd(3, 5, 6)
  local y = m:forward(x):clone()
  m:zeroGradParameters()
  local gx = m:backward(x, gy):clone()
  local gw1, gb1 = m.gradWeight:clone(), m.gradBias:clone()
  m:backward(x, gy)                                 -- a second call must ADD
  F.rows[1] = { weight = m.weight:clone(), bias = m.bias:clone(), x = x, gy = gy, y = y, gx = gx, gw = gw1, gb = gb1,
                gw_twice = m.gradWeight:clone() }
  local v = nn.SpatialConvolution(2, 3, 5, 5)       -- valid convolution (anchor nets, model_utilities.lua:31)
  local xv = rnd(2, 7, 8)
  F.rows[1].valid = { weight = v.weight:clone(), bias = v.bias:clone(), x = xv, y = v:forward(xv):clone() }
end

-- row 2: nn.PReLU() -- one shared slope, initial value, forward / backward / slope gradient
do
  seeded(2)
  local m = nn.PReLU()
  local x, gy = rnd(3, 4, 5), rnd(3, 4, 5)
  local init = m.weight:clone()
  local y = m:forward(x):clone()
  m:zeroGradParameters()
  local gx = m:backward(x, gy):clone()
  F.rows[2] = { init = init, x = x, gy = gy, y = y, gx = gx, gslope = m.gradWeight:clone() }
end

-- row 3: nn.SpatialDropout(p) as installed -- training output (is there a rescale?), evaluate output, backward
do
  seeded(3)
  local m = nn.SpatialDropout(0.4)
  local x, gy = torch.ones(16, 3, 4):float(), torch.ones(16, 3, 4):float()
  m:training()
  local y = m:forward(x):clone()
  local gx = m:backward(x, gy):clone()
  m:evaluate()
  local ye = m:forward(x):clone()
  F.rows[3] = { p = 0.4, x = x, y_train = y, gx_train = gx, y_eval = ye }
end

-- row 4: nn.Dropout(p)
do
  seeded(4)
  local m = nn.Dropout(0.5)
  local x = torch.ones(8, 32):float()
  m:training()
  local y = m:forward(x):clone()
  local gx = m:backward(x, x):clone()
  m:evaluate()
  F.rows[4] = { p = 0.5, x = x, y_train = y, gx_train = gx, y_eval = m:forward(x):clone() }
end

-- row 5: nn.SpatialMaxPooling(2,2,2,2):ceil() -- odd sizes, ties (first maximum wins?)
do
  seeded(5)
  local m = nn.SpatialMaxPooling(2, 2, 2, 2):ceil()
  local x = rnd(2, 5, 7)
  x[1][1][1] = 3; x[1][1][2] = 3; x[1][2][1] = 3; x[1][2][2] = 3     -- a four-way tie in the first window
  local y = m:forward(x):clone()
  local gy = rnd(2, 3, 4)
  local gx = m:backward(x, gy):clone()
  F.rows[5] = { x = x, y = y, gy = gy, gx = gx }
end

-- row 6: nn.SpatialAdaptiveMaxPooling(kw, kh) -- cell bounds, indices, a strided (narrowed) input view, scatter-add backward
do
  seeded(6)
  local m = nn.SpatialAdaptiveMaxPooling(3, 2)          -- kw = 3, kh = 2
  local full = rnd(2, 9, 11)
  local view = full:narrow(2, 2, 7):narrow(3, 3, 5)     -- rows 2..8, columns 3..7: what extract_roi_pooling_input returns
  local y = m:forward(view):clone()
  local gy = rnd(2, 2, 3)
  local gx = m:backward(view, gy):clone()
  F.rows[6] = { kw = 3, kh = 2, full = full, row0 = 2, rows = 7, col0 = 3, cols = 5, y = y, indices = m.indices:clone():float(), gy = gy, gx = gx }
  local s = nn.SpatialAdaptiveMaxPooling(6, 6)          -- a window smaller than the grid (cells overlap)
  local xs = rnd(1, 4, 5)
  F.rows[6].small = { x = xs, y = s:forward(xs):clone(), indices = s.indices:clone():float() }
end

-- row 7: nn.Linear
do
  seeded(7)
  local m = nn.Linear(6, 4)
  local x, gy = rnd(3, 6), rnd(3, 4)
  local y = m:forward(x):clone()
  m:zeroGradParameters()
  local gx = m:backward(x, gy):clone()
  F.rows[7] = { weight = m.weight:clone(), bias = m.bias:clone(), x = x, gy = gy, y = y, gx = gx, gw = m.gradWeight:clone(), gb = m.gradBias:clone() }
end

-- row 8: nn.BatchNormalization(n) -- eps, momentum, biased / unbiased running variance, evaluate mode
do
  seeded(8)
  local m = nn.BatchNormalization(5)
  local w0, b0 = m.weight:clone(), m.bias:clone()
  local x, gy = rnd(4, 5), rnd(4, 5)
  m:training()
  local y = m:forward(x):clone()
  m:zeroGradParameters()
  local gx = m:backward(x, gy):clone()
  local rm, rv = m.running_mean:clone(), (m.running_var or m.running_std):clone()
  m:evaluate()
  local ye = m:forward(x):clone()
  F.rows[8] = { eps = m.eps, momentum = m.momentum, weight = w0, bias = b0, x = x, gy = gy, y_train = y, gx = gx,
                gw = m.gradWeight:clone(), gb = m.gradBias:clone(), running_mean = rm, running_var_or_std = rv,
                has_running_var = m.running_var ~= nil, y_eval = ye }
  local one = nn.BatchNormalization(5)                  -- a batch of ONE row (R = 1)
  one:training()
  local x1 = rnd(1, 5)
  local ok, y1 = pcall(function() return one:forward(x1):clone() end)
  F.rows[8].single_row = { x = x1, ok = ok, y = ok and y1 or nil, running_var_or_std = (one.running_var or one.running_std):clone() }
end

-- rows 9-12: LogSoftMax and the three criteria as objective.lua configures them (:24-27)
do
  seeded(9)
  local x = rnd(4, 7)
  local lsm = nn.LogSoftMax()
  local y = lsm:forward(x):clone()
  local gy = rnd(4, 7)
  F.rows[9] = { x = x, y = y, gy = gy, gx = lsm:backward(x, gy):clone() }
  local nll = nn.ClassNLLCriterion()
  local t = torch.Tensor({ 1, 7, 3, 3 })
  F.rows[10] = { x = y, target = t:float(), loss = nll:forward(y, t), gx = nll:backward(y, t):clone(), sizeAverage = nll.sizeAverage }
  local sl1 = nn.SmoothL1Criterion(); sl1.sizeAverage = false
  local a, b = rnd(5, 4) * 2, rnd(5, 4)
  F.rows[11] = { x = a, target = b, loss = sl1:forward(a, b), gx = sl1:backward(a, b):clone() }
  local ce = nn.CrossEntropyCriterion()
  local v = rnd(2)
  F.rows[12] = { x = v, target = 2, loss = ce:forward(v, 2), gx = ce:backward(v, 2):clone() }
end

-- row 13: optim.rmsprop with the state main.lua:122 passes (learningRate, alpha), two steps
do
  seeded(13)
  local x = rnd(10)
  local g1, g2 = rnd(10), rnd(10)
  local x0 = x:clone()
  local state = { learningRate = 1e-3, alpha = 0.9 }
  local k = 0
  local feval = function(w) k = k + 1; return 0, (k == 1) and g1 or g2 end
  optim.rmsprop(feval, x, state)
  local x1 = x:clone()
  optim.rmsprop(feval, x, state)
  F.rows[13] = { x0 = x0, g1 = g1, g2 = g2, x1 = x1, x2 = x:clone(), epsilon = state.epsilon, m = state.m and state.m:clone() or nil }
end

-- row 14: Tensor:sort with ties -- the permutation TH returns (nms.lua:45 sorts the keys ascending)
do
  local v = torch.Tensor({ 5, 1, 5, 3, 1, 5, 3, 2, 5, 1, 4, 4 }):float()
  local s, i = v:sort()
  F.rows[14] = { v = v, sorted = s, index = i:float() }
  seeded(14)
  local big = torch.floor(torch.rand(300) * 20):float()   -- many ties, above TH's insertion-sort threshold
  local sb, ib = big:sort()
  F.rows[14].big = { v = big, index = ib:float() }
end

-- row 15: torch.random() after manualSeed (Anchors.lua:207-209 sampleNegative, utilities shuffle)
do
  torch.manualSeed(7)
  local d = {}
  for i = 1, 8 do d[i] = torch.random() end
  torch.manualSeed(7)
  local r = {}
  for i = 1, 8 do r[i] = torch.random(1, 100) end
  F.rows[15] = { seed = 7, draws = torch.DoubleTensor(d), range_1_100 = torch.DoubleTensor(r) }
end

-- rows 16, 17: byte-mask indexing keeps order; the reference's own nms() on boxes with a tensor of scores and with 'area'
do
  local I = torch.LongTensor({ 9, 4, 7, 1, 3 })
  local mask = torch.ByteTensor({ 1, 0, 1, 1, 0 })
  F.rows[16] = { I = I:float(), mask = mask:float(), picked = I[mask]:float() }
  local ok = pcall(function() require 'nms' end)
  if ok and nms then
    seeded(17)
    local n = 60
    local x1, y1 = torch.rand(n) * 300, torch.rand(n) * 200
    local b = torch.FloatTensor(n, 4)
    b[{ {}, 1 }] = x1; b[{ {}, 2 }] = y1
    b[{ {}, 3 }] = x1 + torch.rand(n) * 120 + 8; b[{ {}, 4 }] = y1 + torch.rand(n) * 120 + 8
    local scores = torch.rand(n):float()
    F.rows[17] = { boxes = b, scores = scores, overlap = 0.25,
                   pick_default = nms(b, 0.25):float(), pick_scores_tensor = nms(b, 0.25, scores):float(),
                   pick_area = nms(b, 0.25, 'area'):float() }
  end
end

-- row 18: the flat parameter order of the reference's own model (utilities.lua:136-147) and a save_model snapshot
do
  local ok, err = pcall(function()
    require 'utilities'
    local factory = dofile('models/vgg_small.lua')
    local cfg = dofile('config/duplo.lua')
    torch.manualSeed(42)
    local model = factory(cfg)
    local function sizes(net)
      local w = net:parameters()
      local t = {}
      for i = 1, #w do t[i] = torch.LongTensor(w[i]:size():totable()):float() end
      return t
    end
    local weights, gradient = combine_and_flatten_parameters(model.pnet, model.cnet)
    F.rows[18] = { pnet_sizes = sizes(model.pnet), cnet_sizes = sizes(model.cnet), total = weights:nElement() }
    -- the first and last 16 values of every parameter tensor, so that the order can also be recognised by VALUE in the snapshot
    local heads = {}
    local off = 0
    for _, net in ipairs({ model.pnet, model.cnet }) do
      local w = net:parameters()
      for i = 1, #w do
        local n = w[i]:nElement()
        heads[#heads + 1] = { offset = off, count = n, first = weights:narrow(1, off + 1, math.min(16, n)):clone() }
        off = off + n
      end
    end
    F.rows[18].layout = heads
    local snap = out_path .. '.snapshot.t7'
    save_model(snap, weights:narrow(1, 1, 4096):clone(), { lr = 1e-4, name = 'fixture' }, { pcls = { 0.5, 0.25 }, preg = { 1.5 } })
    F.rows[18].snapshot_file = snap
  end)
  if not ok then F.rows[18] = { error = tostring(err) } end
end

-- row 19: pairs() order over the class table of Detector.lua:125 (informational: Lua leaves it unspecified)
do
  local yclass = {}
  for _, c in ipairs({ 5, 2, 9, 2, 1, 5 }) do yclass[c] = (yclass[c] or 0) + 1 end
  local order = {}
  for c, _ in pairs(yclass) do order[#order + 1] = c end
  F.rows[19] = { inserted = torch.Tensor({ 5, 2, 9, 2, 1, 5 }):float(), pairs_order = torch.Tensor(order):float() }
end

-- row 20: the colour spaces load_image offers beside yuv (utilities.lua:210-216)
do
  local rgb = torch.rand(3, 9, 11):float()
  rgb[{ {}, 1, 1 }] = torch.Tensor({ 1, 0, 0 }):float(); rgb[{ {}, 1, 2 }] = torch.Tensor({ 0, 1, 0 }):float()
  rgb[{ {}, 1, 3 }] = torch.Tensor({ 0, 0, 1 }):float(); rgb[{ {}, 1, 4 }] = torch.Tensor({ 1, 1, 1 }):float()
  rgb[{ {}, 1, 5 }] = torch.Tensor({ 0, 0, 0 }):float(); rgb[{ {}, 1, 6 }] = torch.Tensor({ .5, .5, .5 }):float()
  rgb[{ {}, 1, 7 }] = torch.Tensor({ .02, .03, .01 }):float()
  F.rows[20] = { rgb = rgb, yuv = image.rgb2yuv(rgb), hsv = image.rgb2hsv(rgb), lab = image.rgb2lab(rgb) }
end

local f = torch.DiskFile(out_path, 'w')   -- ASCII, as the reference's save_obj writes it
f:writeObject(F)
f:close()
print('wrote ' .. out_path)
