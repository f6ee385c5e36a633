"""Times one training step (veto_forward_train + losses + veto_backward through autograd) of VETOPredictor on the
BASELINE cfg-2 batch shape (12 images x 36 objects = 15 120 pairs, 4 layers, 8 heads), dropout at the reference's rates.
usage: tools/train_bench.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from veto_amd import synth, testing
from veto_amd.pairs import prepare_test_pairs

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
cfg = testing.make_config(4, 8)
model = testing.make_predictor(cfg, synth.predictor_state_dict(0, layers=4), dev).train()
batch = synth.synthetic_batch(7, 12, 36)
props = testing.make_proposals(batch, "predcls", dev)
pairs = prepare_test_pairs(dev, props)
n = sum(int(p.shape[0]) for p in pairs)
labels = torch.from_numpy(synth.integers(5, "bench.labels", (n,), 0, 51)).to(dev)
rel_labels = list(labels.split([int(p.shape[0]) for p in pairs]))
kw = dict(roi_features=torch.from_numpy(batch["roi_features"]).to(dev), roi_depth_features=torch.from_numpy(batch["roi_depth_features"]).to(dev))
opt = torch.optim.SGD(model.parameters(), lr=1e-4)


def step():
    opt.zero_grad(set_to_none=True)
    loss = model(props, pairs, rel_labels, None, **kw)[2]["rel_loss"]
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
t_f = t_b = 0.0
t0 = time.perf_counter()
for _ in range(steps):
    opt.zero_grad(set_to_none=True)
    ev[0].record()
    loss = model(props, pairs, rel_labels, None, **kw)[2]["rel_loss"]
    ev[1].record()
    loss.backward()
    ev[2].record()
    opt.step()
    torch.cuda.synchronize()
    t_f += ev[0].elapsed_time(ev[1]); t_b += ev[1].elapsed_time(ev[2])
dt = (time.perf_counter() - t0) / steps
print("training step: %.1f ms (forward+loss %.1f ms, backward %.1f ms, rest = optimizer) -> %.0f pairs/s; loss %.4f; peak memory %.1f GB" %
      (dt * 1e3, t_f / steps, t_b / steps, n / dt, float(loss.detach()), torch.cuda.max_memory_allocated() / 2 ** 30))
