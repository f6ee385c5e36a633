"""How many host threads does the CPU oracle want on this box?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import veto_oracle as vo
from veto_amd import synth
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("no cgroup info", e)
sd = synth.predictor_state_dict(0, layers=4)
b = synth.synthetic_batch(7, 1, 36)
cfg = vo.OracleConfig(layers=4, heads=8)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    vo.forward(sd, cfg, b)
    t0 = time.perf_counter(); vo.forward(sd, cfg, b); dt = time.perf_counter() - t0
    print("threads %3d: %.2f s -> %.0f pairs/s" % (th, dt, 1260 / dt), flush=True)
