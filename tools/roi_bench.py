"""Times the fused ROI pooling launch (veto_roi_pool: 4 FPN levels + depth map -> two [N, 256, 8, 8] tensors)
at the BASELINE cfg-2 batch shape: 12 images (608 x 1024 after padding) x 36 boxes.  Prints ms per launch and
the write-side bandwidth (the 2 x N x 256 x 64 x 4 output bytes are the algorithmic minimum; the reads are the
ROIs' footprints of the maps, at most that much again for 8x8 outputs with 2x2 samples)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from veto_amd import synth, testing
from veto_amd.poolers import make_roi_box_feature_extractor
from veto_amd.structures import BoxList

dev = torch.device("cuda:0")
n_img, n_obj, W, H = 12, 36, 1024, 608
feats = [torch.randn(n_img, 256, H >> (2 + l), W >> (2 + l), device=dev) for l in range(4)]
depth = torch.randn(n_img, 256, H >> 4, W >> 4, device=dev)
batch = synth.synthetic_batch(7, n_img, n_obj)
boxes = torch.from_numpy(batch["boxes"]).to(dev)
boxes[:, 2:] = torch.minimum(boxes[:, 2:] * 1.6, torch.tensor([W - 1.0, H - 1.0], device=dev))
props = [BoxList(boxes[i * n_obj:(i + 1) * n_obj], (W, H)) for i in range(n_img)]
ext = make_roi_box_feature_extractor(testing.make_config(4, 8), 256, for_relation=True)
for _ in range(3):
    x2d, d2d, _, _ = ext(feats, props, depth_features=depth)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    x2d, d2d, _, _ = ext(feats, props, depth_features=depth)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
out_bytes = 2 * x2d.numel() * 4
print("roi_pool: %d ROIs x 2 maps, %.4f ms per call (host enqueue included), output %.1f MB -> %.0f GB/s written" %
      (n_img * n_obj, ms, out_bytes / 1e6, out_bytes / (ms * 1e-3) / 1e9))
