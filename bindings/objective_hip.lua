-- objective_hip.lua -- drop-in for the reference's objective.lua on libfrcnn_hip.so: the same two globals
-- (extract_roi_pooling_input, create_objective) with the same arguments and results, same per-image structure as
-- objective.lua:45-218 (pnet forward, sparse RPN loss on the sampled anchors, ROI adaptive max-pooling, cnet
-- forward/backward, ROI-pool backward, pnet backward; normalise; log) -- but every per-example Lua loop that touched
-- the device one scalar at a time is ONE batched call (frcnn_pnet_anchor_loss_begin, frcnn_roi_pool_forward/backward,
-- frcnn_cnet_losses), and the statistics of objective.lua:52-58 stay in a device-side fp64 vector until the single
-- read-back at the end.  1:1 with the tested Python host mirror (faster-rcnn.torch_amd/objective.py); this image has no
-- Lua runtime, so this file is checked statically (tests/test_abi.py) and has not been executed.
--
--   main.lua:  require 'objective'  ->  require 'objective_hip'
local ffi = require 'ffi'
local hip = require 'frcnn_hip'
require 'Localizer'          -- the reference's own files, unchanged
require 'Anchors'
local C, check = hip.C, hip.check

-- objective.lua:5-13.  The strided view `feature_layer_output[idx]` is never materialised: the batched ROI-pooling
-- kernel reads the window in place, so the first result is the index table as well.
function extract_roi_pooling_input(input_rect, localizer, feature_layer_output)
  local r = localizer:inputToFeatureRect(input_rect)
  local s = feature_layer_output:size()
  r = r:clip(Rect.new(0, 0, s[3], s[2]))
  local idx = { {}, { math.min(r.minY + 1, r.maxY), r.maxY }, { math.min(r.minX + 1, r.maxX), r.maxX } }
  return idx, idx
end

function create_objective(model, weights, gradient, batch_iterator, stats)
  local cfg = model.cfg
  local pnet = model.pnet
  local cnet = model.cnet
  local native = model.native

  local bgclass = cfg.class_count + 1                                   -- :20
  local nheads = #model.anchor_nets
  local localizer = Localizer.new(pnet.outnode.children[nheads + 1])   -- :22 (children[5])
  local kh, kw = cfg.roi_pooling.kh, cfg.roi_pooling.kw
  local cnet_input_planes = model.layers[#model.layers].filters
  local D = kh * kw * cnet_input_planes
  local ncls = cfg.class_count + 1
  local acc = hip.buffer(8 * 8)        -- {cls_loss, reg_loss, -, -, creg_loss, ccls_loss, -, -} as device doubles
  local acc_host = ffi.new('double[8]')
  local scratch = hip.scratch()
  local keep                            -- host staging of the last example tables (must outlive the queued copy)

  local function cleanAnchors(examples, outputs)                        -- :32-43
    local i = 1
    while i <= #examples do
      local anchor = examples[i][1]
      local fmSize = outputs[anchor.layer]:size()
      if anchor.index[2] > fmSize[2] or anchor.index[3] > fmSize[3] then
        table.remove(examples, i)
      else
        i = i + 1
      end
    end
  end

  local function lossAndGradient(w)
    if w ~= weights then weights:copy(w) end                            -- :46-48
    gradient:zero()                                                     -- :49
    check(C.frcnn_zero(acc.ptr, 64, nil))
    local cls_count, reg_count = 0, 0                                   -- :52-58 (the four losses live in `acc`)
    local creg_count, ccls_count = 0, 0
    pnet:training()                                                     -- :61-62
    cnet:training()

    local batch = batch_iterator:nextTraining()                         -- :64
    for _, x in ipairs(batch) do
      local img = hip.to_device(x.img)                                  -- :66
      local p, n = x.positive, x.negative
      local outputs = pnet:forward(img, true)                           -- :71 (anchor nets stay in flight)
      cleanAnchors(p, outputs)                                          -- :74-75
      cleanAnchors(n, outputs)
      local delta_outputs = pnet:delta_outputs()                        -- :78-84
      local npos, nneg = #p, #n
      local E = npos + nneg
      local fm = outputs[nheads + 1]
      local fs = fm:size()

      if E > 0 then
        -- ---- host: the example tables, one upload ----------------------------------------------
        local np1 = math.max(npos, 1)
        local ex_anchor = ffi.new('double[?]', 4 * E)
        local ex_roi = ffi.new('double[?]', 4 * np1)
        local ex_idx = ffi.new('int[?]', 4 * E)
        local ex_class = ffi.new('int[?]', np1)
        local wins = ffi.new('int[?]', 4 * E)
        local seen, sp = {}, {}
        for l = 1, nheads do seen[l] = {}; sp[l] = {} end
        for i = 1, E do
          local a = (i <= npos) and p[i][1] or n[i - npos][1]
          local o = 4 * (i - 1)
          ex_anchor[o], ex_anchor[o + 1], ex_anchor[o + 2], ex_anchor[o + 3] = a.minX, a.minY, a.maxX, a.maxY
          ex_idx[o], ex_idx[o + 1], ex_idx[o + 2], ex_idx[o + 3] = a.layer, a.aspect, a.index[2], a.index[3]
          local pooled = a                                              -- negatives pool the anchor rect itself (:137)
          if i <= npos then
            local roi = p[i][2]
            if roi.class_index < 1 or roi.class_index > cfg.class_count then
              error(string.format('roi.class_index %d outside 1..%d', roi.class_index, cfg.class_count))
            end
            ex_roi[o], ex_roi[o + 1], ex_roi[o + 2], ex_roi[o + 3] = roi.rect.minX, roi.rect.minY, roi.rect.maxX, roi.rect.maxY
            ex_class[i - 1] = roi.class_index
            pooled = roi.rect                                           -- positives pool the ground-truth rect (:117)
          end
          local _, idx = extract_roi_pooling_input(pooled, localizer, fm)
          wins[o], wins[o + 1], wins[o + 2], wins[o + 3] = idx[2][1], idx[2][2], idx[3][1], idx[3][2]
          -- where delta_outputs[l] will be non-zero (hint for the sparse anchor-net backward)
          local pos = (a.index[2] - 1) * outputs[a.layer]:size(3) + (a.index[3] - 1)
          if not seen[a.layer][pos] then
            seen[a.layer][pos] = true
            table.insert(sp[a.layer], pos)
          end
        end
        local nsp = 0
        for l = 1, nheads do table.sort(sp[l]); nsp = nsp + #sp[l] end
        local sp_all = ffi.new('int[?]', math.max(nsp, 1))
        local k = 0
        for l = 1, nheads do
          for _, v in ipairs(sp[l]) do sp_all[k] = v; k = k + 1 end
        end
        local b_anchor, b_roi, b_idx, b_class, b_wins, b_sp = 32 * E, 32 * np1, 16 * E, 4 * np1, 16 * E, 4 * nsp
        local total = b_anchor + b_roi + b_idx + b_class + b_wins + b_sp
        local blob = ffi.new('uint8_t[?]', total)
        local o = 0
        ffi.copy(blob + o, ex_anchor, b_anchor); local o_anchor = o; o = o + b_anchor
        ffi.copy(blob + o, ex_roi, b_roi); local o_roi = o; o = o + b_roi
        ffi.copy(blob + o, ex_idx, b_idx); local o_idx = o; o = o + b_idx
        ffi.copy(blob + o, ex_class, b_class); local o_class = o; o = o + b_class
        ffi.copy(blob + o, wins, b_wins); local o_wins = o; o = o + b_wins
        if nsp > 0 then ffi.copy(blob + o, sp_all, b_sp) end
        local o_sp = o
        keep = blob
        local dblob = scratch('blob', total).ptr
        check(C.frcnn_memcpy_h2d(dblob, blob, total, nil))
        for l = 1, nheads do
          check(C.frcnn_pnet_set_sparse_deltas(native, l, ffi.cast('const int*', dblob + o_sp), #sp[l]))
          o_sp = o_sp + 4 * #sp[l]
        end

        -- ---- RPN loss on the sampled anchors (:91-140), queued behind the anchor nets on the library's side
        -- stream and followed by their backward pass; this stream goes on with the last map -------------------
        local ex_loss = ffi.cast('double*', scratch('ex_loss', 16 * E).ptr)
        local crtarget = ffi.cast('float*', scratch('crtarget', 16 * E).ptr)
        local cctarget = ffi.cast('float*', scratch('cctarget', 4 * E).ptr)
        check(C.frcnn_pnet_anchor_loss_begin(native, weights.ptr, gradient.ptr, ffi.cast('const int*', dblob + o_idx),
                                             ffi.cast('const double*', dblob + o_anchor), ffi.cast('const double*', dblob + o_roi),
                                             ffi.cast('const int*', dblob + o_class), npos, nneg, bgclass, ex_loss, crtarget,
                                             cctarget, ffi.cast('double*', acc.ptr), nil))
        -- ---- ROI pooling of every example in one launch (:117-119, :137-139) ---------------------------------
        local cinput = hip.view(scratch('cinput', 4 * E * D).ptr, { E, D })
        local pidx = ffi.cast('int*', scratch('pidx', 4 * E * D).ptr)
        check(C.frcnn_roi_pool_forward(fm.ptr, fs[1], fs[2], fs[3], ffi.cast('const int*', dblob + o_wins), E, kh, kw,
                                       cinput.ptr, pidx, nil))
        -- ---- fine-tuning stage (:146-186) --------------------------------------------------------------------
        local coutputs = cnet:forward(cinput)                           -- :164
        local crout, ccout = coutputs[1], coutputs[2]
        local crdelta = hip.view(scratch('crdelta', 16 * E).ptr, { E, 4 })
        local ccdelta = hip.view(scratch('ccdelta', 4 * E * ncls).ptr, { E, ncls })
        check(C.frcnn_pnet_anchor_loss_wait(native, nil))               -- crtarget is relative to the proposals (:156)
        check(C.frcnn_cnet_losses(crout.ptr, crtarget, ccout.ptr, cctarget, E, npos, ncls, crdelta.ptr, ccdelta.ptr,
                                  ffi.cast('double*', acc.ptr) + 4, nil))               -- :170-177
        local post_roi_delta = cnet:backward(cinput, { crdelta, ccdelta })              -- :179
        check(C.frcnn_roi_pool_backward(delta_outputs[nheads + 1].ptr, fs[1], fs[2], fs[3], post_roi_delta.ptr, pidx, E,
                                        kh, kw, nil))                                   -- :182-185
      else
        for l = 1, nheads do check(C.frcnn_pnet_set_sparse_deltas(native, l, nil, 0)) end
      end

      pnet:backward(img, delta_outputs)                                 -- :189
      reg_count = reg_count + npos                                      -- :194-198
      cls_count = cls_count + npos + nneg
      creg_count = creg_count + npos
      ccls_count = ccls_count + 1
    end

    -- ---- data parallel (SURVEY 8e): sum the flat gradient, the loss accumulators and the counts over the ranks,
    -- between the last pnet:backward and gradient:div (:197-200) ------------------------------------------------
    if hip.comm then
      local counts = ffi.new('double[4]', cls_count, reg_count, creg_count, ccls_count)
      check(C.frcnn_memcpy_h2d(acc.ptr + 16, counts, 16, nil))          -- slots 2, 3 ...
      check(C.frcnn_memcpy_h2d(acc.ptr + 48, counts + 2, 16, nil))      -- ... and 6, 7 of the 8 accumulators
      check(C.frcnn_allreduce_f32(hip.comm, gradient.ptr, gradient.n, nil))
      check(C.frcnn_allreduce_f64(hip.comm, ffi.cast('double*', acc.ptr), 8, nil))
    end
    check(C.frcnn_memcpy_d2h(acc_host, acc.ptr, 64, nil))
    check(C.frcnn_stream_sync(nil))
    if hip.comm then
      cls_count, reg_count, creg_count, ccls_count = acc_host[2], acc_host[3], acc_host[6], acc_host[7]
    end
    local cls_loss, reg_loss, creg_loss, ccls_loss = acc_host[0], acc_host[1], acc_host[4], acc_host[5]

    gradient:div(cls_count)                                             -- :200

    local pcls = cls_loss / cls_count                                   -- :202-205
    local preg = reg_loss / reg_count
    local dcls = ccls_loss / ccls_count
    local dreg = creg_loss / creg_count
    print(string.format('prop: cls: %f (%d), reg: %f (%d); det: cls: %f, reg: %f',
      pcls, cls_count, preg, reg_count, dcls, dreg))                    -- :207-209
    table.insert(stats.pcls, pcls)                                      -- :211-214
    table.insert(stats.preg, preg)
    table.insert(stats.dcls, dcls)
    table.insert(stats.dreg, dreg)

    local loss = pcls + preg                                            -- :216-217
    return loss, gradient
  end

  return lossAndGradient
end
