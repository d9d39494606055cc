-- objective.lua -- main.lua:13 `require 'objective'`: the batched drop-in (objective_hip.lua: same two globals,
-- extract_roi_pooling_input and create_objective, same arguments and results).  With bindings/ in front on package.path
-- main.lua needs no edit.  To run the REFERENCE's own objective.lua over the library instead (per-example slow path, for
-- cross-checking this file), put the reference directory first for this one name: its `require 'cunn'` still resolves to
-- bindings/cunn.lua, whose nn shims (frcnn_nn.lua) serve objective.lua:24-30.
return require 'objective_hip'
