-- Detector.lua -- main.lua:14 `require 'Detector'`: the batched drop-in (Detector_hip.lua: class Detector(model),
-- :detect(input) -> winners).  See bindings/objective.lua for how to run the reference's own file over the library.
return require 'Detector_hip'
