"""Does replaying veto_forward from a HIP graph shorten a small-batch call?  (1 and 2 images: ~55 short kernels per call.)"""
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
    rgb = torch.from_numpy(batch["roi_features"]).to(dev)
    dep = torch.from_numpy(batch["roi_depth_features"]).to(dev)
    with torch.no_grad():
        ref = torch.cat(list(model(props, pairs, None, None, roi_features=rgb, roi_depth_features=dep)[1]))
    labels = torch.cat([p.get_field("labels") for p in props])       # predcls: hard labels, no logits
    inp, keep, n_objs, n_pairs, device, eng = model._prepare_inputs(props, pairs, rgb, dep, labels, None)
    ws = torch.empty(eng.workspace_bytes(inp.n_obj, inp.n_pair), dtype=torch.uint8, device=dev)
    out = torch.empty((inp.n_pair, model._num_out), dtype=torch.float32, device=dev)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(2):
            eng.forward(side.cuda_stream, inp, ws.data_ptr(), ws.numel(), out.data_ptr(), None)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        eng.forward(side.cuda_stream, inp, ws.data_ptr(), ws.numel(), out.data_ptr(), None)
    g.replay()
    torch.cuda.synchronize()
    err = (out - ref).abs().max().item()
    def timeit(fn, n=50):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    t_direct = timeit(lambda: eng.forward(torch.cuda.current_stream(dev).cuda_stream, inp, ws.data_ptr(), ws.numel(), out.data_ptr(), None))
    t_graph = timeit(g.replay)
    print("%2d img: direct launches %.3f ms/call, graph replay %.3f ms/call (max |diff| vs module call %.1e)" % (imgs, t_direct, t_graph, err))
