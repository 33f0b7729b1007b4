import cProfile, pstats, sys, os, io, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from cone_amd import inference as inf, synth
from cone_amd.config import make_opt
from cone_amd.model import build_model
torch.set_num_threads(1)
opt = make_opt("ego4d", nms_thd=0.5, eval_split_name="test", topk_window=20, eval_bsz=32, window_batch=32768)
model, _ = build_model(opt)
model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, 0).items()})
ann, vf, qf = synth.make_dataset(opt, 1000, 50, seed=0)
store = inf.FeatureStore(opt, ann, vf, qf)
for _ in range(3):
    inf.predict_split(model, store, opt)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    inf.predict_split(model, store, opt)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
