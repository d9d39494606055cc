import cProfile, pstats, sys, os, time
sys.path.insert(0, os.getcwd())
import torch, frcnn_amd as F
cfg=dict(F.duplo_cfg); model=F.vgg_small(cfg)
w,g=F.combine_and_flatten_parameters(model["pnet"],model["cnet"])
it=F.SyntheticBatchIterator(model,pool=4)
stats=dict(pcls=[],preg=[],dcls=[],dreg=[])
f=F.create_objective(model,w,g,it,stats); st=dict(learningRate=1e-4,alpha=0.9)
for _ in range(3): F.rmsprop(f,w,st)
torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable()
t0=time.perf_counter()
for _ in range(20): F.rmsprop(f,w,st)
torch.cuda.synchronize(); dt=time.perf_counter()-t0
pr.disable()
print("ms/step", dt/20*1e3)
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
