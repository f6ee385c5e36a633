"""Times the device relation evaluators (veto_sgg_eval) on a benchmark-shaped batch: 12 images x 36 objects
(1260 ranked pairs each).  (The host-side comparison number in DESIGN.md comes from the oracle timed inside
tests/test_sgg_eval.py; tools/ never import oracle/.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from veto_amd import synth
from veto_amd.evaluation import SGGEvaluator

dev = torch.device("cuda:0")
images, zeroshot = synth.synthetic_eval_images(77, [36] * 12, "predcls")
dimg = [{k: torch.as_tensor(v).to(dev) for k, v in im.items()} for im in images]
ev = SGGEvaluator("predcls", 51, zeroshot, device=dev)
for _ in range(3):
    res = ev.evaluate(dimg)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 20
for _ in range(reps):
    res = ev.evaluate(dimg)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
print("veto_sgg_eval: %d images, %.3f ms per batch end to end (host packing + 2 kernels + read-back)" % (len(images), ms))
print(ev.generate_print_string(res), end="")
