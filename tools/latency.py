"""End-to-end latency of one predictor call (host + device) for small batches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from veto_amd import synth, testing
from veto_amd.pairs import prepare_test_pairs
dev = torch.device("cuda:0")
sd = synth.predictor_state_dict(0, layers=4)
model = testing.make_predictor(testing.make_config(4, 8), sd, dev)
for imgs in (1, 2, 12):
    batch = synth.synthetic_batch(7, imgs, 36)
    props = testing.make_proposals(batch, "predcls", dev)
    pairs = prepare_test_pairs(dev, props)
    rgb = torch.from_numpy(batch["roi_features"]).to(dev); dep = torch.from_numpy(batch["roi_depth_features"]).to(dev)
    def call():
        with torch.no_grad():
            return model(props, pairs, None, None, roi_features=rgb, roi_depth_features=dep)
    for _ in range(3): call()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 20
    for _ in range(n): call()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    t0 = time.perf_counter(); call(); torch.cuda.synchronize(); t_one = time.perf_counter() - t0
    print("%2d img (%5d pairs): host enqueue %.2f ms/call, throughput %.2f ms/call, single-call latency %.2f ms" %
          (imgs, imgs * 1260, t_host * 1e3, t_all * 1e3, t_one * 1e3))
