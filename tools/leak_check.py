"""Creates and destroys many contexts in one process and reports the device memory and the resident set before / after: a leak check of
psgsdf_create ... psgsdf_destroy (run on the GPU box: python tools/leak_check.py [n])."""
import os, sys, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from psgradientsdf_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sc = synth.make_scene(N=64, F=8, W=160, H=120, model="SH1")
st = capi.default_settings(sc.model_id)
def one():
    eng = capi.load_engine(sc, sc.K, st, 0); eng.load_scene(sc); eng.init_albedo(); eng.normalize_weights(); eng.optimize(capi.ALL); eng.download_volume(); eng.close()
for _ in range(5):
    one()
torch.cuda.synchronize(); free0, total = torch.cuda.mem_get_info(); rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
for i in range(n):
    one()
torch.cuda.synchronize(); free1, _ = torch.cuda.mem_get_info(); rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print({"contexts": n, "device_bytes_lost": int(free0 - free1), "device_bytes_lost_per_context": (free0 - free1) / n, "max_rss_growth_kb": rss1 - rss0})
