import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cone_amd import inference as inf, synth
from cone_amd.config import make_opt
from cone_amd.model import build_model
torch.set_num_threads(1)
opt = make_opt("ego4d", nms_thd=0.5, eval_split_name="test", topk_window=20)
model, _ = build_model(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, 0).items()})
ann, vf, qf = synth.make_dataset(opt, 1, 1, seed=0, ctx_range=(900, 901))
store = inf.FeatureStore(opt, ann, vf, qf)
for _ in range(50):
    inf.predict_split(model, store, opt)
torch.cuda.synchronize()
t=time.perf_counter()
for _ in range(200):
    inf.predict_split(model, store, opt)
torch.cuda.synchronize()
print("ms per split", (time.perf_counter()-t)/200*1e3)
# host-only time: enqueue without waiting
t=time.perf_counter()
hs=[]
for _ in range(200):
    hs.append(inf.predict_split_async(model, store, opt))
t1=time.perf_counter()
for h in hs: h.result()
torch.cuda.synchronize()
print("host enqueue ms per split", (t1-t)/200*1e3, "total", (time.perf_counter()-t)/200*1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    inf.predict_split(model, store, opt)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40)
print(s.getvalue()[:7000])
