"""Device-resident test-time pipeline of the relation head on synthetic data, end to end:
FPN / depth maps -> ROI pooling (veto_roi_pool) -> pair enumeration -> VETOPredictor (veto_forward) ->
PostProcessor (veto_postprocess) -> relation evaluators (veto_sgg_eval).  Prints per-stage device time and the
end-to-end images/s at the BASELINE cfg-2 batch shape (12 images x 36 objects).  usage: tools/eval_pipeline.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from veto_amd import synth, testing
from veto_amd.evaluation import SGGEvaluator
from veto_amd.relation_head import VETORelationHead
from veto_amd.structures import BoxList

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
n_img, n_obj, W, H = 12, 36, 1024, 608
cfg = testing.make_config(4, 8)
head = VETORelationHead(cfg)
head.predictor = testing.make_predictor(cfg, synth.predictor_state_dict(0, layers=4), dev)
head.eval()
batch = synth.synthetic_batch(7, n_img, n_obj)
feats = [torch.randn(n_img, 256, H >> (2 + l), W >> (2 + l), device=dev) for l in range(4)]
depth = torch.randn(n_img, 256, H >> 4, W >> 4, device=dev)
eval_imgs, zeroshot = synth.synthetic_eval_images(5, [n_obj] * n_img, "predcls")
gts = []
for im in eval_imgs:
    g = BoxList(torch.from_numpy(im["gt_boxes"]), (W, H)).to(dev)
    g.add_field("relation_tuple", torch.from_numpy(im["gt_rels"]).to(dev))
    g.add_field("labels", torch.from_numpy(im["gt_classes"]).to(dev))
    gts.append(g)
evaluator = SGGEvaluator("predcls", 51, zeroshot, device=dev)


def one_batch(timers=None):
    props = testing.make_proposals(batch, "predcls", dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    roi, d2d, _, _ = head.box_feature_extractor(feats, props, depth_features=depth)
    ev[1].record()
    _, result, _ = head.forward_pooled(props, roi, d2d)
    ev[2].record()
    res = evaluator.evaluate_boxlists(gts, result)
    ev[3].record()
    torch.cuda.synchronize()
    if timers is not None:
        for i, name in enumerate(("roi_pool", "pairs+predictor+postprocess", "evaluators (incl. read-back)")):
            timers.setdefault(name, []).append(ev[i].elapsed_time(ev[i + 1]))
    return res


for _ in range(2):
    one_batch()
timers = {}
for _ in range(3):
    res = one_batch(timers)
print("per stage, one batch at a time (device time between events, host enqueue gaps included):")
for k, v in timers.items():
    print("  %-32s %.3f ms" % (k, float(np.mean(v))))

# Throughput as an eval loop would run it: no per-batch synchronisation on the compute stream; the evaluators of
# batch i run on a second stream (after an event recorded behind batch i's PostProcessor) while the predictor of
# batch i+1 already executes, so their read-back only waits for their own kernels.
evaluator.reset()
main, side = torch.cuda.current_stream(dev), torch.cuda.Stream(dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
pending = None
for i in range(steps + 1):
    cur = None
    if i < steps:
        props = testing.make_proposals(batch, "predcls", dev)
        roi, d2d, _, _ = head.box_feature_extractor(feats, props, depth_features=depth)
        _, result, _ = head.forward_pooled(props, roi, d2d)
        done = torch.cuda.Event()
        done.record(main)
        cur = (result, done)
    if pending is not None:
        with torch.cuda.stream(side):
            side.wait_event(pending[1])
            evaluator.update_boxlists(gts, pending[0])      # per-relation match ranks accumulate over the split
    pending = cur
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("pipelined: %.2f ms per %d-image batch -> %.0f images/s, %.0f pairs/s end to end" %
      (dt * 1e3, n_img, n_img / dt, n_img * n_obj * (n_obj - 1) / dt))
res = evaluator.finalize()                                   # ONE fold over every image seen, as the reference's evaluators do
print("dataset-level metrics over %d images:" % res["images_evaluated"])
print(evaluator.generate_print_string(res), end="")
