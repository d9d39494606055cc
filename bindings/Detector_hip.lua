-- Detector_hip.lua -- drop-in for the reference's Detector.lua on libfrcnn_hip.so: the same class (`Detector(model)`,
-- `:detect(input) -> winners`, each winner a table { p, a, r, l, r2, class, confidence }), the same pipeline and
-- thresholds (p > 0.95, NMS 0.25, class ~= background and p > 0.2, per-class NMS 0.1) -- but the 26 544-iteration
-- Lua loop of Detector.lua:39-66 is one scan + compaction kernel (frcnn_rpn_scan), the per-candidate pooling loop
-- (:94-98) one batched kernel, and the first NMS runs on the device.  Both NMS calls of the reference pass a tensor as
-- `scores`, which nms.lua:37-43 ignores: boxes are processed by descending max-y.  That behaviour is reproduced.
-- 1:1 with the tested Python host mirror (faster-rcnn.torch_amd/Detector.py); checked statically, not executed here.
--
--   main.lua:  require 'Detector'  ->  require 'Detector_hip'
local ffi = require 'ffi'
local hip = require 'frcnn_hip'
require 'Anchors'            -- the reference's own files, unchanged
require 'Localizer'
require 'objective_hip'      -- extract_roi_pooling_input
local C, check = hip.C, hip.check

local ASPECTS = 3   -- anchors per map position (Anchors.lua:108-109)

local Detector = torch.class('Detector')

function Detector:__init(model)                                         -- Detector.lua:8-15
  local cfg = model.cfg
  self.model = model
  self.anchors = Anchors.new(model.pnet, cfg.scales)
  self.nheads = #model.anchor_nets
  self.localizer = Localizer.new(model.pnet.outnode.children[self.nheads + 1])
  -- the fp32 tables of Anchors.lua:18-19, resident on the device for the scan
  self.aw = hip.to_device(self.anchors.w)
  self.ah = hip.to_device(self.anchors.h)
  self.scratch = hip.scratch()
end

function Detector:detect(input)                                         -- Detector.lua:17-141
  local model = self.model
  local cfg = model.cfg
  local pnet = model.pnet
  local cnet = model.cnet
  local kh, kw = cfg.roi_pooling.kh, cfg.roi_pooling.kw
  local bgclass = cfg.class_count + 1
  local ncls = cfg.class_count + 1
  local cnet_input_planes = model.layers[#model.layers].filters
  local D = kh * kw * cnet_input_planes
  local scratch = self.scratch

  local input_size = input:size()
  pnet:evaluate()                                                       -- :31
  input = hip.to_device(input)                                          -- :32
  local outputs = pnet:forward(input)                                   -- :33

  -- ---- :39-66 on the device: log-softmax of every anchor's two logits, p > 0.95, decode, overlap test, compaction
  local Hs, Ws, maps = ffi.new('int[4]'), ffi.new('int[4]'), ffi.new('const float*[4]')
  for i = 1, 4 do
    local s = outputs[i]:size()
    Hs[i - 1], Ws[i - 1], maps[i - 1] = s[2], s[3], outputs[i].ptr
  end
  local cap = 0   -- every anchor of the four maps may pass: the buffers hold them all (vgg_large 1000x600: 45 015)
  for i = 0, 3 do cap = cap + ASPECTS * Hs[i] * Ws[i] end
  local wsb = tonumber(C.frcnn_rpn_scan_workspace_bytes(Hs, Ws))
  local ws = scratch('scan_ws', wsb)
  local mp = ffi.cast('float*', scratch('match_p', 4 * cap).ptr)
  local mi = ffi.cast('int*', scratch('match_idx', 16 * cap).ptr)
  local mr = ffi.cast('double*', scratch('match_rect', 32 * cap).ptr)
  local mb = ffi.cast('float*', scratch('match_box', 16 * cap).ptr)
  -- counts (device int[4]): matches, NMS candidates, candidates that pass the class test, winners
  local cnt = ffi.cast('int*', scratch('counts', 16).ptr)
  check(C.frcnn_rpn_scan(maps, Hs, Ws, self.aw.ptr, self.ah.ptr, input_size[3], input_size[2], 0.95, cap, mp, mi, mr, mb,
                         cnt, ws.ptr, wsb, nil))
  -- NON-MAXIMUM SUPPRESSION (:74-85) on the device, the match count read from DEVICE memory (no round trip between scan
  -- and NMS); the score tensor is ignored by nms.lua -> key = max-y
  -- (launch and workspace sized for a bound on the matches, not for every anchor of the maps; a frame with more matches
  -- repeats the pass sized by the count just read)
  local ncap = math.min(cap, 16384)
  local nwsb = tonumber(C.frcnn_nms_workspace_bytes(ncap))
  local nws = scratch('nms_ws', nwsb)
  local dpick = ffi.cast('long long*', scratch('pick', 8 * cap).ptr)
  check(C.frcnn_nms_device_n(mb, ncap, cnt, 4, 0.25, 0, 0, nil, dpick, cnt + 1, nws.ptr, nwsb, nil))
  local count = ffi.new('int[2]')
  check(C.frcnn_memcpy_d2h(count, cnt, 8, nil))                          -- ---- read-back 1 of 2: two counts
  check(C.frcnn_stream_sync(nil))
  if count[0] > cap then
    error(string.format('Detector: %d anchors pass p > 0.95, more than the %d the maps hold', count[0], cap))
  end
  if count[0] > ncap then
    nwsb = tonumber(C.frcnn_nms_workspace_bytes(count[0]))
    nws = scratch('nms_ws_full', nwsb)
    check(C.frcnn_nms_device(mb, count[0], 4, 0.25, 0, 0, dpick, cnt + 1, nws.ptr, nwsb, nil))
    check(C.frcnn_memcpy_d2h(count + 1, cnt + 1, 4, nil))
    check(C.frcnn_stream_sync(nil))
  end
  local nm, R = count[0], count[1]

  local winners = {}
  if nm > 0 then                                                        -- :71
    print(string.format('candidates: %d', R))                           -- :87
    -- REGION CLASSIFICATION (:90-101): every candidate's window (objective.lua:5-13 for all of them in one kernel), one
    -- pooling launch (no indices: there is no backward pass), one cnet pass
    cnet:evaluate()
    local fm = outputs[self.nheads + 1]
    local fs = fm:size()
    local nl = #self.localizer.layers
    local layers = ffi.new('int[?]', 6 * nl)
    for i, l in ipairs(self.localizer.layers) do
      local o = 6 * (i - 1)
      layers[o], layers[o + 1], layers[o + 2], layers[o + 3], layers[o + 4], layers[o + 5] = l.kW, l.kH, l.dW, l.dH, l.padW, l.padH
    end
    local dwins = ffi.cast('int*', scratch('wins', 16 * R).ptr)
    check(C.frcnn_roi_windows(mr, dpick, R, layers, nl, fs[2], fs[3], dwins, nil))
    local cinput = hip.view(scratch('cinput', 4 * R * D).ptr, { R, D })
    check(C.frcnn_roi_pool_forward(fm.ptr, fs[1], fs[2], fs[3], dwins, R, kh, kw, cinput.ptr, nil, nil))
    local coutputs = cnet:forward(cinput)                               -- :101
    local bbox_out, cls_out = coutputs[1], coutputs[2]
    local dcls = ffi.cast('int*', scratch('cls', 4 * R).ptr)
    local dconf = ffi.cast('float*', scratch('conf', 4 * R).ptr)
    check(C.frcnn_cnet_decode(cls_out.ptr, R, ncls, dcls, dconf, nil))  -- :110-113 (arg-max of the log-probs)
    -- :106-122 on the device: class test, r2 = Anchors.anchorToInput(r, bbox) in double, survivors compacted in order
    local dbb = ffi.cast('float*', scratch('bb5', 20 * R).ptr)
    local dkc = ffi.cast('int*', scratch('bbcls', 4 * R).ptr)
    local dkeep = ffi.cast('int*', scratch('keep_row', 4 * R).ptr)
    local dr2 = ffi.cast('double*', scratch('r2', 32 * R).ptr)
    check(C.frcnn_detect_post(dcls, dconf, bbox_out.ptr, mr, dpick, R, bgclass, 0.2, dbb, dkc, dkeep, dr2, cnt + 2, nil))
    -- per-class NMS (:125-136), every class in ONE device pass: rows only suppress rows of their own class; a stable
    -- partition of the picks by class is, per class, exactly nms(bb, 0.1, bb[{{}, 5}]) -- the score tensor is ignored by
    -- nms.lua:42, the key is max-y.  The survivor count is read from device memory.
    local cwsb = tonumber(C.frcnn_nms_workspace_bytes(R))
    local cws = scratch('nms_ws2', cwsb)
    local cpick = ffi.cast('long long*', scratch('wpick', 8 * R).ptr)
    check(C.frcnn_nms_device_n(dbb, R, cnt + 2, 5, 0.1, 0, 0, dkc, cpick, cnt + 3, cws.ptr, cwsb, nil))
    -- one record of 16 doubles per winner behind a 128-byte header that carries the four counts
    local out = ffi.cast('double*', scratch('winners', 128 * (R + 1)).ptr)
    check(C.frcnn_memcpy_d2d(out, cnt, 16, nil))
    check(C.frcnn_detect_gather(cpick, cnt + 3, R, dkeep, dkc, dbb, dr2, dpick, mp, mr, mi, out + 16, nil))
    local h = ffi.new('double[?]', 16 * (R + 1))
    check(C.frcnn_memcpy_d2h(h, out, 128 * (R + 1), nil))                -- ---- read-back 2 of 2: the winner table
    check(C.frcnn_stream_sync(nil))
    local nwin = ffi.cast('int*', h)[3]
    -- classes in ascending order (the reference iterates with pairs(): unspecified), pick order within a class
    local byclass, classes = {}, {}
    for q = 1, nwin do
      local v = h + 16 * q
      local l, a, y, x = tonumber(v[12]), tonumber(v[13]), tonumber(v[14]), tonumber(v[15])
      local det = { p = v[3], a = self.anchors:get(l, a, y, x), l = l, r = Rect.new(v[4], v[5], v[6], v[7]),
                    r2 = Rect.new(v[8], v[9], v[10], v[11]), class = tonumber(v[0]), confidence = v[2] }
      if not byclass[det.class] then
        byclass[det.class] = {}
        classes[#classes + 1] = det.class
      end
      table.insert(byclass[det.class], det)
    end
    table.sort(classes)
    for _, ci in ipairs(classes) do
      for _, x in ipairs(byclass[ci]) do table.insert(winners, x) end
    end
  end

  return winners
end
