-- frcnn_nn.lua -- the stand-alone nn modules of SURVEY 8b over libfrcnn_hip.so, so that the reference's OWN objective.lua
-- (:24-30, :91-186) and Detector.lua (:13-14, :52, :97) run over the library as they are -- one example at a time, with a
-- blocking scalar read per loss: the SLOW path, there to cross-check the batched drop-ins (objective_hip.lua,
-- Detector_hip.lua) the day a LuaJIT host exists.  Not executed in this image; checked statically (tests/test_abi.py).
--
--   nn.SpatialAdaptiveMaxPooling(kw, kh):cuda()   :forward(window view) / .indices / :backward(input, gradOutput)
--   nn.LogSoftMax():cuda()                        :forward(1-D view)
--   nn.CrossEntropyCriterion():cuda()             :forward(input, target) / :backward(input, target)
--   nn.ClassNLLCriterion():cuda()                 the same on R x n log-probabilities, honours .sizeAverage (default true)
--   nn.SmoothL1Criterion():cuda()                 the same, honours .sizeAverage (objective.lua:27 sets it to false)
--
-- The CPU nn classes stay what they are (BatchIterator.lua uses them on FloatTensors); only their :cuda() is replaced: it
-- returns the device-side object.  Strided views (out[{{lo,hi}, y, x}], fmap[{{}, {y0,y1}, {x0,x1}}]) are added to the
-- binding's tensor type here: small ones go through the host element by element, the ROI window is handled in place by
-- frcnn_roi_pool_forward / _backward.
local ffi = require 'ffi'
local hip = require 'frcnn_hip'
local C, check = hip.C, hip.check

-- ------------------------------------------------------------------------------------------ strided views
-- StridedView: { root = DevTensor (contiguous), off = element offset, sizes, strides }.  What the reference does with one:
-- scalar reads t[i] (Anchors.lua:245-252), sub-ranges t[{{a,b}}], :add(x) (objective.lua:106,114,134,184), :zero().
local Strided = {}
local smethods = {}

local function strides_of(sizes)
  local st, s = {}, 1
  for d = #sizes, 1, -1 do st[d] = s; s = s * sizes[d] end
  return st
end

local function sview(root, off, sizes, strides)
  local n = 1
  for _, s in ipairs(sizes) do n = n * s end
  return setmetatable({ root = root, off = off, sizes = sizes, strides = strides, n = n }, Strided)
end

local function is_strided(t) return getmetatable(t) == Strided end

-- e = i | {lo, hi} | {}: the general form of hip's index_table, keeping strides instead of refusing
local function sindex(root, off, sizes, strides, idx)
  local out_s, out_st = {}, {}
  for d = 1, #sizes do
    local e = idx[d]
    local lo, hi, sel
    if e == nil or (type(e) == 'table' and #e == 0) then lo, hi = 1, sizes[d]
    elseif type(e) == 'number' then lo, hi, sel = e, e, true
    else lo, hi = e[1], e[2] or e[1] end
    if lo < 1 or hi > sizes[d] or lo > hi then
      error(string.format('index out of range: dimension %d is %d, got {%d,%d}', d, sizes[d], lo, hi), 3)
    end
    off = off + (lo - 1) * strides[d]
    if not sel then out_s[#out_s + 1] = hi - lo + 1; out_st[#out_st + 1] = strides[d] end
  end
  if #out_s == 0 then                                   -- a scalar: blocking read, like t[i] on a CudaTensor
    local v = ffi.new('float[1]')
    check(C.frcnn_memcpy_d2h(v, root.ptr + off, 4, nil)); check(C.frcnn_stream_sync(nil))
    return tonumber(v[0])
  end
  return sview(root, off, out_s, out_st)
end

-- every element offset of a view, in row-major order of its own shape
local function offsets(t)
  local offs, idx = {}, {}
  for d = 1, #t.sizes do idx[d] = 0 end
  for _ = 1, t.n do
    local o = t.off
    for d = 1, #t.sizes do o = o + idx[d] * t.strides[d] end
    offs[#offs + 1] = o
    local d = #t.sizes
    while d >= 1 do
      idx[d] = idx[d] + 1
      if idx[d] < t.sizes[d] then break end
      idx[d] = 0; d = d - 1
    end
  end
  return offs
end

local function gather(t)                                -- small views only (the 2 / 4 / 6 values of one anchor)
  local h = ffi.new('float[?]', t.n)
  for i, o in ipairs(offsets(t)) do check(C.frcnn_memcpy_d2h(h + (i - 1), t.root.ptr + o, 4, nil)) end
  check(C.frcnn_stream_sync(nil))
  return h
end

local function scatter(t, h)
  for i, o in ipairs(offsets(t)) do check(C.frcnn_memcpy_h2d(t.root.ptr + o, h + (i - 1), 4, nil)) end
  check(C.frcnn_stream_sync(nil))
end

local function host_values(x, n)                        -- a DevTensor, a strided view or a torch tensor as float[n]
  if is_strided(x) then assert(x.n == n); return gather(x) end
  local f = hip.is_tensor(x) and x:float() or x:float():contiguous()
  assert(f:nElement() == n, 'element counts differ')
  local h = ffi.new('float[?]', n)
  ffi.copy(h, f:data(), n * 4)
  return h
end

Strided.__index = function(t, k)
  if type(k) == 'number' then return sindex(t.root, t.off, t.sizes, t.strides, { k }) end
  if type(k) == 'table' then return sindex(t.root, t.off, t.sizes, t.strides, k) end
  return smethods[k]
end
function smethods:size(i) if i then return self.sizes[i] end; return self.sizes end
function smethods:dim() return #self.sizes end
function smethods:nElement() return self.n end
function smethods:cuda() return self end
function smethods:float()
  local f = torch.FloatTensor(unpack(self.sizes))
  ffi.copy(f:data(), gather(self), self.n * 4)
  return f
end
function smethods:zero()
  scatter(self, ffi.new('float[?]', self.n))
  return self
end
-- d:add(x).  The ROI-pooling backward hands back a window view of a scratch map that is zero outside the window
-- (objective.lua:184): adding the WHOLE scratch map to the whole gradient map is the same sum and one call.
function smethods:add(x)
  if is_strided(x) and x.full_map_is_zero_elsewhere and x.root.n == self.root.n then
    check(C.frcnn_add(self.root.ptr, x.root.ptr, self.root.n, nil))
    return self
  end
  local a, b = gather(self), host_values(x, self.n)
  for i = 0, self.n - 1 do a[i] = a[i] + b[i] end
  scatter(self, a)
  return self
end

-- the binding's contiguous tensor learns to hand out strided views where it used to refuse
local dev_mt = getmetatable(hip.tensor({ 1 }))
local dev_index = dev_mt.__index
dev_mt.__index = function(t, k)
  if type(k) == 'table' then
    local ok, res = pcall(dev_index, t, k)
    if ok then return res end
    return sindex(t, 0, t.sizes, strides_of(t.sizes), k)
  end
  return dev_index(t, k)
end

-- FloatTensor:cuda() (objective.lua:110: Anchors.inputToAnchor(...):cuda()) -> a device tensor of the binding
torch.FloatTensor.cuda = function(self) return hip.to_device(self) end

-- ------------------------------------------------------------------------------------------ nn.SpatialAdaptiveMaxPooling
local ibuf = hip.scratch()
local function amp_new(kw, kh)
  local amp = { kw = kw, kh = kh }
  function amp:cuda() return self end
  -- input: the window view fmap[{{}, {y0,y1}, {x0,x1}}] of a C x H x W map (extract_roi_pooling_input, objective.lua:5-13)
  function amp:forward(input)
    local root = input.root or input
    local Cn, H, W = root.sizes[1], root.sizes[2], root.sizes[3]
    local off = input.off or 0
    local y0, x0 = math.floor(off / W) % H, off % W
    local h, w = input.sizes[2], input.sizes[3]
    local win = ffi.new('int[4]', y0 + 1, y0 + h, x0 + 1, x0 + w)      -- 1-based inclusive rows / columns
    local wd = ibuf('win', 16)
    check(C.frcnn_memcpy_h2d(wd.ptr, win, 16, nil)); check(C.frcnn_stream_sync(nil))
    local n = Cn * self.kh * self.kw
    local out = hip.tensor({ Cn, self.kh, self.kw })
    local idx = hip.buffer(n * 4)
    check(C.frcnn_roi_pool_forward(root.ptr, Cn, H, W, ffi.cast('const int*', wd.ptr), 1, self.kh, self.kw, out.ptr,
                                   ffi.cast('int*', idx.ptr), nil))
    -- .indices: flat positions inside the full map (what the backward scatter needs); :clone() as objective.lua:119 does
    self.indices = { buf = idx, n = n, clone = function(ix)
      local b = hip.buffer(ix.n * 4)
      check(C.frcnn_memcpy_d2d(b.ptr, ix.buf.ptr, ix.n * 4, nil))
      return { buf = b, n = ix.n, clone = ix.clone }
    end }
    self.output = out
    return out
  end
  -- gradInput has the window's shape: a window view of a scratch map that is zero elsewhere (see Strided:add)
  function amp:backward(input, gradOutput)
    local root = input.root or input
    local Cn, H, W = root.sizes[1], root.sizes[2], root.sizes[3]
    local g = hip.is_tensor(gradOutput) and gradOutput or hip.to_device(gradOutput)
    local map = hip.tensor({ Cn, H, W }):zero()
    check(C.frcnn_roi_pool_backward(map.ptr, Cn, H, W, g.ptr, ffi.cast('const int*', self.indices.buf.ptr), 1, self.kh, self.kw, nil))
    local v = sview(map, input.off or 0, { Cn, input.sizes[2], input.sizes[3] }, { H * W, W, 1 })
    v.full_map_is_zero_elsewhere = true
    self.gradInput = v
    return v
  end
  return amp
end

-- ------------------------------------------------------------------------------------------ LogSoftMax and the criteria
-- All of them see a handful of numbers per call (2 class logits, 4 box values, R x n log-probabilities): host arithmetic
-- in double on values read back, results as Lua numbers / small device tensors -- the same values the batched kernels
-- (frcnn_rpn_loss, frcnn_cnet_losses) produce for every example at once.
local function lsm_host(h, n)
  local m = -math.huge
  for i = 0, n - 1 do m = math.max(m, h[i]) end
  local s = 0
  for i = 0, n - 1 do s = s + math.exp(h[i] - m) end
  local out = {}
  for i = 0, n - 1 do out[i + 1] = h[i] - m - math.log(s) end
  return out
end

local function dev_from(values, sizes)
  local n = #values
  local h = ffi.new('float[?]', n)
  for i = 1, n do h[i - 1] = values[i] end
  local t = hip.tensor(sizes)
  check(C.frcnn_memcpy_h2d(t.ptr, h, n * 4, nil)); check(C.frcnn_stream_sync(nil))
  return t
end

local function lsm_new()
  local m = {}
  function m:cuda() return self end
  function m:forward(input)                              -- Detector.lua:52: c = lsm:forward(cls_out); c[1], c[2]
    local n = input:nElement()
    self.output = torch.FloatTensor(lsm_host(host_values(input, n), n))
    return self.output
  end
  return m
end

local function cross_entropy_new()                       -- objective.lua:24,104-106,132-134: 1-D input, target 1 | 2
  local m = { sizeAverage = true }
  function m:cuda() return self end
  function m:forward(input, target)
    local n = input:nElement()
    self.lsm = lsm_host(host_values(input, n), n)
    self.output = -self.lsm[target]
    return self.output
  end
  function m:backward(input, target)
    local g = {}
    for i = 1, #self.lsm do g[i] = math.exp(self.lsm[i]) - (i == target and 1 or 0) end
    self.gradInput = dev_from(g, { #g })
    return self.gradInput
  end
  return m
end

local function class_nll_new()                           -- objective.lua:25,174-177: R x n log-probabilities, R targets
  local m = { sizeAverage = true }
  function m:cuda() return self end
  function m:forward(input, target)
    local R, n = input:size(1), input:size(2)
    local h, t = host_values(input, R * n), host_values(target, R)
    local s = 0
    for r = 0, R - 1 do s = s - h[r * n + (t[r] - 1)] end
    self.output = self.sizeAverage and s / R or s
    return self.output
  end
  function m:backward(input, target)
    local R, n = input:size(1), input:size(2)
    local t = host_values(target, R)
    local g = {}
    for i = 1, R * n do g[i] = 0 end
    for r = 0, R - 1 do g[r * n + t[r]] = self.sizeAverage and -1 / R or -1 end
    self.gradInput = dev_from(g, { R, n })
    return self.gradInput
  end
  return m
end

local function smooth_l1_new()                           -- objective.lua:26-27,112-113,170-172
  local m = { sizeAverage = true }
  function m:cuda() return self end
  local function diff(input, target)
    local n = input:nElement()
    local a, b = host_values(input, n), host_values(target, n)
    local d = {}
    for i = 0, n - 1 do d[i + 1] = a[i] - b[i] end
    return d, n
  end
  function m:forward(input, target)
    local d, n = diff(input, target)
    local s = 0
    for i = 1, n do
      local z = math.abs(d[i])
      s = s + (z < 1 and 0.5 * z * z or z - 0.5)
    end
    self.output = self.sizeAverage and s / n or s
    return self.output
  end
  function m:backward(input, target)
    local d, n = diff(input, target)
    local g = {}
    for i = 1, n do
      local z = math.max(-1, math.min(1, d[i]))
      g[i] = self.sizeAverage and z / n or z
    end
    local sizes = {}
    for i, s in ipairs(input:size()) do sizes[i] = s end
    self.gradInput = dev_from(g, sizes)
    return self.gradInput
  end
  return m
end

-- ------------------------------------------------------------------------------------------ install: only :cuda() changes
local function device_twin(class, make)
  if not class then return end
  class.cuda = function(self)
    local twin = make(self)
    if self.sizeAverage ~= nil then twin.sizeAverage = self.sizeAverage end
    return twin
  end
end
if nn then
  device_twin(nn.SpatialAdaptiveMaxPooling, function(self) return amp_new(self.W or self.kW, self.H or self.kH) end)
  device_twin(nn.LogSoftMax, lsm_new)
  device_twin(nn.CrossEntropyCriterion, cross_entropy_new)
  device_twin(nn.ClassNLLCriterion, class_nll_new)
  device_twin(nn.SmoothL1Criterion, smooth_l1_new)
end

return { adaptive_max_pooling = amp_new, log_softmax = lsm_new, cross_entropy = cross_entropy_new, class_nll = class_nll_new,
         smooth_l1 = smooth_l1_new }
