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
  local cnt = ffi.cast('int*', scratch('count', 16).ptr)
  check(C.frcnn_rpn_scan(maps, Hs, Ws, self.aw.ptr, self.ah.ptr, input_size[3], input_size[2], 0.95, cap, mp, mi, mr, mb,
                         cnt, ws.ptr, wsb, nil))
  local count = ffi.new('int[1]')
  check(C.frcnn_memcpy_d2h(count, cnt, 4, nil))
  check(C.frcnn_stream_sync(nil))
  if count[0] > cap then
    error(string.format('Detector: %d anchors pass p > 0.95, more than the %d the maps hold', count[0], cap))
  end
  local nm = count[0]

  local winners = {}
  if nm > 0 then                                                        -- :71
    -- NON-MAXIMUM SUPPRESSION (:74-85) on the device; the score tensor is ignored by nms.lua -> key = max-y
    local nwsb = tonumber(C.frcnn_nms_workspace_bytes(nm))
    local nws = scratch('nms_ws', nwsb)
    local dpick = ffi.cast('long long*', scratch('pick', 8 * nm).ptr)
    check(C.frcnn_nms_device(mb, nm, 4, 0.25, 0, 0, dpick, cnt, nws.ptr, nwsb, nil))
    local h_p, h_idx, h_rect = ffi.new('float[?]', nm), ffi.new('int[?]', 4 * nm), ffi.new('double[?]', 4 * nm)
    local h_pick = ffi.new('long long[?]', nm)
    check(C.frcnn_memcpy_d2h(count, cnt, 4, nil))
    check(C.frcnn_memcpy_d2h(h_pick, dpick, 8 * nm, nil))
    check(C.frcnn_memcpy_d2h(h_p, mp, 4 * nm, nil))
    check(C.frcnn_memcpy_d2h(h_idx, mi, 16 * nm, nil))
    check(C.frcnn_memcpy_d2h(h_rect, mr, 32 * nm, nil))
    check(C.frcnn_stream_sync(nil))
    local R = count[0]
    local candidates = {}
    for k = 0, R - 1 do
      local m = tonumber(h_pick[k]) - 1                                 -- 1-based match id -> 0-based row
      local l, a, y, x = h_idx[4 * m], h_idx[4 * m + 1], h_idx[4 * m + 2], h_idx[4 * m + 3]
      candidates[k + 1] = { p = h_p[m], a = self.anchors:get(l, a, y, x), l = l,
                            r = Rect.new(h_rect[4 * m], h_rect[4 * m + 1], h_rect[4 * m + 2], h_rect[4 * m + 3]) }
    end
    print(string.format('candidates: %d', #candidates))                 -- :87

    -- REGION CLASSIFICATION (:90-101): every candidate's window, one pooling launch, one cnet pass
    cnet:evaluate()
    local fm = outputs[self.nheads + 1]
    local fs = fm:size()
    local wins = ffi.new('int[?]', 4 * R)
    for i, v in ipairs(candidates) do
      local _, idx = extract_roi_pooling_input(v.r, self.localizer, fm)
      local o = 4 * (i - 1)
      wins[o], wins[o + 1], wins[o + 2], wins[o + 3] = idx[2][1], idx[2][2], idx[3][1], idx[3][2]
    end
    local dwins = ffi.cast('int*', scratch('wins', 16 * R).ptr)
    check(C.frcnn_memcpy_h2d(dwins, wins, 16 * R, nil))
    local cinput = hip.view(scratch('cinput', 4 * R * D).ptr, { R, D })
    local pidx = ffi.cast('int*', scratch('pidx', 4 * R * D).ptr)
    check(C.frcnn_roi_pool_forward(fm.ptr, fs[1], fs[2], fs[3], dwins, R, kh, kw, cinput.ptr, pidx, nil))
    local coutputs = cnet:forward(cinput)                               -- :101
    local bbox_out, cls_out = coutputs[1], coutputs[2]
    local dcls = ffi.cast('int*', scratch('cls', 4 * R).ptr)
    local dconf = ffi.cast('float*', scratch('conf', 4 * R).ptr)
    check(C.frcnn_cnet_decode(cls_out.ptr, R, ncls, dcls, dconf, nil))  -- :110-113 (arg-max of the log-probs)
    local h_bbox, h_cls, h_conf = ffi.new('float[?]', 4 * R), ffi.new('int[?]', R), ffi.new('float[?]', R)
    check(C.frcnn_memcpy_d2h(h_bbox, bbox_out.ptr, 16 * R, nil))
    check(C.frcnn_memcpy_d2h(h_cls, dcls, 4 * R, nil))
    check(C.frcnn_memcpy_d2h(h_conf, dconf, 4 * R, nil))
    check(C.frcnn_stream_sync(nil))

    local kept = {}
    for i, x in ipairs(candidates) do                                   -- :106-122
      local t = torch.FloatTensor(4)
      for k = 1, 4 do t[k] = h_bbox[4 * (i - 1) + k - 1] end
      x.r2 = Anchors.anchorToInput(x.r, t)                              -- :107
      x.class = h_cls[i - 1]
      x.confidence = h_conf[i - 1]
      if x.class ~= bgclass and math.exp(x.confidence) > 0.2 then       -- :115
        table.insert(kept, x)
      end
    end

    -- per-class NMS (:125-136), every class in ONE device pass (frcnn_nms_device_classes): rows only suppress rows of
    -- their own class; a stable partition of the picks by class is, per class, exactly nms(bb, 0.1, bb[{{}, 5}]) -- the
    -- score tensor is ignored by nms.lua:42, the key is max-y.  One launch sequence and one read-back instead of one per
    -- class (up to 200 with config/imagenet.lua).
    local K = #kept
    if K > 0 then
      local bb = ffi.new('float[?]', 5 * K)
      local kc = ffi.new('int[?]', K)
      for j, r in ipairs(kept) do
        local tt = r.r2:totensor()
        for k = 1, 4 do bb[5 * (j - 1) + k - 1] = tt[k] end
        bb[5 * (j - 1) + 4] = r.confidence
        kc[j - 1] = r.class
      end
      local dbb = ffi.cast('float*', scratch('bb5', 20 * K).ptr)
      local dkc = ffi.cast('int*', scratch('bbcls', 4 * K).ptr)
      check(C.frcnn_memcpy_h2d(dbb, bb, 20 * K, nil))
      check(C.frcnn_memcpy_h2d(dkc, kc, 4 * K, nil))
      local cwsb = tonumber(C.frcnn_nms_workspace_bytes(K))
      local cws = scratch('nms_ws', cwsb)
      local cpick = ffi.cast('long long*', scratch('pick', 8 * K).ptr)
      check(C.frcnn_nms_device_classes(dbb, K, 5, 0.1, 0, 0, dkc, cpick, cnt, cws.ptr, cwsb, nil))
      local h_cp = ffi.new('long long[?]', K)
      check(C.frcnn_memcpy_d2h(count, cnt, 4, nil))
      check(C.frcnn_memcpy_d2h(h_cp, cpick, 8 * K, nil))
      check(C.frcnn_stream_sync(nil))
      -- classes in ascending order (the reference iterates with pairs(): unspecified), pick order within a class
      local byclass, classes = {}, {}
      for q = 0, count[0] - 1 do
        local x = kept[tonumber(h_cp[q])]
        if not byclass[x.class] then
          byclass[x.class] = {}
          classes[#classes + 1] = x.class
        end
        table.insert(byclass[x.class], x)
      end
      table.sort(classes)
      for _, ci in ipairs(classes) do
        for _, x in ipairs(byclass[ci]) do table.insert(winners, x) end
      end
    end
  end

  return winners
end
