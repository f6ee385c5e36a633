"""MI355X-native `VETOPredictor` / `VETOPredictor_MEET` behind the reference's plugin API.

Mirrors pysgg/modeling/roi_heads/relation_head/roi_relation_predictors.py:3876-4139:
  * registered under the same names in a registry of the same protocol (veto_amd.registry),
  * same constructor `Predictor(config, in_channels)` and the same `forward` signature,
  * same 6-tuple return (SURVEY.md section 8b "Return"),
  * same state-dict key names, so reference checkpoints load unchanged.

The modules below are PARAMETER CONTAINERS: the arithmetic of the eval forward runs in
libveto_amd.so (hand-written HIP, include/veto_amd.h) on the tensors' own device and stream.
There is no PyTorch or CPU fallback; if the library is missing, or the inputs are not on a HIP
device, forward raises.
"""
import ctypes
import weakref
import warnings

import torch
from torch import nn

from . import meet_tables, native
from .registry import ROI_RELATION_PREDICTOR

_STATISTICS_PROVIDER = None
_EMBEDDING_PROVIDER = None


def set_statistics_provider(fn):
    """fn(config) -> {'obj_classes': [...], 'rel_classes': [...]} (replaces the reference's
    pysgg.data.get_dataset_statistics, data/build.py:27-77, when pysgg is not importable)."""
    global _STATISTICS_PROVIDER
    _STATISTICS_PROVIDER = fn


def set_embedding_provider(fn):
    """fn(obj_classes, wv_dir, wv_dim) -> FloatTensor [len(obj_classes), wv_dim] (replaces
    utils_motifs.obj_edge_vectors, utils_motifs.py:151-171)."""
    global _EMBEDDING_PROVIDER
    _EMBEDDING_PROVIDER = fn


def _dataset_statistics(config):
    if _STATISTICS_PROVIDER is not None:
        return _STATISTICS_PROVIDER(config)
    try:
        from pysgg.data import get_dataset_statistics  # type: ignore
        return get_dataset_statistics(config)
    except ImportError:
        n_obj, n_rel = meet_tables.NUM_CLASSES[config.GLOBAL_SETTING.DATASET_CHOICE]
        return {"obj_classes": ["obj_%d" % i for i in range(n_obj)],
                "rel_classes": ["rel_%d" % i for i in range(n_rel)]}


def _embedding_vectors(obj_classes, wv_dir, wv_dim):
    if _EMBEDDING_PROVIDER is not None:
        return _EMBEDDING_PROVIDER(obj_classes, wv_dir, wv_dim)
    try:
        from pysgg.modeling.roi_heads.relation_head.utils_motifs import obj_edge_vectors  # type: ignore
        return obj_edge_vectors(obj_classes, wv_dir=wv_dir, wv_dim=wv_dim)
    except ImportError:
        warnings.warn("veto_amd: GloVe loader unavailable; obj_embed initialised N(0,1) "
                      "(load a checkpoint or call set_embedding_provider)")
        return torch.randn(len(obj_classes), wv_dim)


def _mode(config):
    rh = config.MODEL.ROI_RELATION_HEAD
    if rh.USE_GT_BOX:
        return "predcls" if rh.USE_GT_OBJECT_LABEL else "sgcls"
    return "sgdet"


def _precision(config):
    ext = getattr(config, "VETO_AMD", None)
    name = getattr(ext, "PRECISION", "mixed") if ext is not None else "mixed"
    modes = {"precise": native.VETO_PRECISE, "fast": native.VETO_FAST, "mixed": native.VETO_MIXED}
    if name not in modes:
        raise ValueError("VETO_AMD.PRECISION must be one of %s, got %r" % (sorted(modes), name))
    return modes[name]


def _max_chunk(config):
    ext = getattr(config, "VETO_AMD", None)
    return int(getattr(ext, "MAX_CHUNK_PAIRS", 0)) if ext is not None else 0


# ---------------------------------------------------------------------------------------------------
# Parameter containers with the reference's attribute names (=> identical state-dict keys).
# ---------------------------------------------------------------------------------------------------

class _Holder(nn.Module):
    """A module that only owns sub-modules/parameters; the compute lives in the HIP library."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the forward pass runs in libveto_amd.so")


def _attention_params(dim, dropout):
    m = _Holder()
    m.to_qkv = nn.Linear(dim, 3 * dim, bias=False)
    m.to_out = nn.Sequential(nn.Linear(dim, dim), nn.Dropout(dropout))
    return m


def _feedforward_params(dim, hidden):
    m = _Holder()
    m.net = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(0.0), nn.Linear(hidden, dim),
                          nn.Dropout(0.0))
    return m


def _prenorm(dim, fn):
    m = _Holder()
    m.norm = nn.LayerNorm(dim)
    m.fn = fn
    return m


def _transformer_params(config, in_channels):
    vt = config.MODEL.ROI_RELATION_HEAD.VETOTRANSFORMER
    dim, patch = vt.T_INPUT_DIM, vt.PATCH_SIZE
    if dim != 576:
        # same constraint as the reference: proj_d (512) + proj_v (64) are hard-wired, so any other
        # T_INPUT_DIM fails at the token concat (model_veto.py:105-113)
        raise ValueError("VETOTRANSFORMER.T_INPUT_DIM must be 576, got %d" % dim)
    inner = _Holder()
    inner.patch_embed = _Holder()
    pdim = in_channels * 2 * patch * patch
    inner.patch_embed.proj_d = nn.Linear(pdim, 512)
    inner.patch_embed.proj_v = nn.Linear(pdim, 64)
    inner.cls_token = nn.Parameter(torch.randn(1, 1, dim))
    inner.pos_embedding = nn.Parameter(torch.randn(1, 1, dim))
    inner.pos_drop = nn.Dropout(vt.EMB_DROPOUT)
    inner.layers = nn.ModuleList(
        nn.ModuleList([_prenorm(dim, _attention_params(dim, vt.T_DROPOUT)),
                       _prenorm(dim, _feedforward_params(dim, 2 * dim))])
        for _ in range(vt.ENC_LAYERS))
    outer = _Holder()
    outer.transformer = inner
    outer.to_cls_token = nn.Identity()
    return outer


def _xavier_linear(in_f, out_f):
    lin = nn.Linear(in_f, out_f, bias=True)
    nn.init.xavier_normal_(lin.weight)
    return lin


def _build_trunk(m, config, num_obj_cls, with_embed2):
    """Adds the modules shared by VETOPredictor (:4014-4053) and Ensemble (:3668-3704) to `m`."""
    dim = config.MODEL.ROI_RELATION_HEAD.VETOTRANSFORMER.T_INPUT_DIM
    embed_dim = 200
    if with_embed2:
        m.obj_embed2 = nn.Embedding(num_obj_cls, embed_dim)
    m.obj_embed = nn.Embedding(num_obj_cls, embed_dim)
    m.class_projection = nn.Sequential(nn.Linear(2 * embed_dim, dim), nn.ReLU(inplace=True))
    m.bbox_embed = nn.Sequential(nn.Linear(9, 32), nn.ReLU(inplace=True), nn.Dropout(0.1),
                                 nn.Linear(32, 128), nn.ReLU(inplace=True), nn.Dropout(0.1))
    m.pos_embed = nn.Sequential(nn.BatchNorm1d(4, momentum=0.001), nn.Linear(4, 128), nn.ReLU(inplace=True),
                                nn.Dropout(0.1))
    m.location_projection = nn.Sequential(nn.Linear(256, dim), nn.ReLU(inplace=True))
    m.fusion_transformer = _transformer_params(config, in_channels=256)


_TRUNK_KEYS = None


_OFFSET_CACHE = {}


def cached_offsets(n_objs, n_pairs, device):
    """Per-image exclusive prefix sums (objects, pairs) as int32 device tensors.  torch.tensor(list, device=...) is a
    synchronous pageable H2D copy that stalls the host behind all queued GPU work, so batch shapes seen before (the
    common case in an eval loop) re-use their tensors.  Shared by the predictor, the PostProcessor and the evaluator."""
    key = (tuple(n_objs), tuple(n_pairs), str(device))
    hit = _OFFSET_CACHE.get(key)
    if hit is None:
        if len(_OFFSET_CACHE) >= 256:
            _OFFSET_CACHE.clear()
        hit = _OFFSET_CACHE[key] = _offset_tensors(tuple(n_objs), tuple(n_pairs), device)
    return hit


class _NativeForward:
    """Shared device-side plumbing: engine lifetime, weight upload, workspace, one C-ABI forward."""

    def _native_init(self, config, trunk, num_obj_cls, head_modules):
        vt = config.MODEL.ROI_RELATION_HEAD.VETOTRANSFORMER
        self._layers, self._heads = int(vt.ENC_LAYERS), int(vt.NHEADS)
        self._num_obj_cls = num_obj_cls
        self._head_modules = head_modules
        self._num_out = sum(h.out_features for h in head_modules)
        self._precision = _precision(config)
        self._max_chunk = _max_chunk(config)
        # VETO_AMD.COUNT_SATURATION: eval forwards go through veto_forward_saturation (slower: launch per stage, one stream
        # synchronisation) and leave the per-layer counts of clamped mixed-row elements in `self.last_saturation`
        self._count_saturation = bool(getattr(getattr(config, "VETO_AMD", None), "COUNT_SATURATION", False))
        self.last_saturation = None
        object.__setattr__(self, "_trunk", trunk)  # not a sub-module registration
        self._engine = None
        self._engine_device = None
        self._uploaded = None
        self._workspace = None
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.refresh_weights())

    def _weight_tensors(self):
        t = self._trunk
        sd = {k: v for k, v in t.state_dict().items()}
        out = {k: sd[k] for k in sd if k.startswith(("obj_embed.", "class_projection.0.", "pos_embed.",
                                                     "location_projection.0.", "fusion_transformer."))}
        out.pop("pos_embed.0.num_batches_tracked", None)
        heads = self._head_modules
        out["rel_out.weight"] = torch.cat([h.weight for h in heads], 0) if len(heads) > 1 else heads[0].weight
        out["rel_out.bias"] = torch.cat([h.bias for h in heads], 0) if len(heads) > 1 else heads[0].bias
        return out

    def _weights_version(self):
        return tuple((p.data_ptr(), p._version) for p in list(self._trunk.parameters()) + list(self._trunk.buffers()))

    def refresh_weights(self):
        """Forces the next call to re-upload every weight (and rebuild the engine's derived operands: split / mixed rows,
        Wqkv diag(gamma), Mcat, Ncat).  Weight changes are detected through (data_ptr, Tensor._version), which writes
        through `.data` (`p.data.copy_()`, EMA updates, hand-written optimisers, some checkpoint loaders) do NOT bump:
        call this after such a write in eval mode.  In training mode and after load_state_dict it happens by itself."""
        self._uploaded = None

    invalidate = refresh_weights

    def _ensure_engine(self, device):
        if device.type != "cuda":
            raise RuntimeError("veto_amd: the predictor runs only on a HIP device (got %s); there is no CPU path" % device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._engine is None or self._engine_device != idx:
            if self._engine is not None:
                self._engine.close()
            self._engine = native.Engine(self._layers, self._heads, self._num_obj_cls, self._num_out,
                                         precision=self._precision, device=idx, max_chunk_pairs=self._max_chunk)
            self._engine_device = idx
            self._uploaded = None
        ver = self._weights_version()
        if self._trunk.training:
            self._uploaded = None      # a training step changes the weights: never trust the version stamp there
        if self._uploaded != ver:
            tensors = self._weight_tensors()
            stream = torch.cuda.current_stream(device).cuda_stream
            keep = []
            for name, numel in self._engine.weight_specs():
                t = tensors[name].detach().to(device=device, dtype=torch.float32).contiguous()
                if t.numel() != numel:
                    raise RuntimeError("veto_amd: weight %s has %d elements, expected %d" % (name, t.numel(), numel))
                keep.append(t)
                self._engine.load_weight(name, t.data_ptr(), numel, stream)
            torch.cuda.current_stream(device).synchronize()  # `keep` may be freed after this
            self._uploaded = ver
        return self._engine

    def _offsets(self, n_objs, n_pairs, device):
        return cached_offsets(n_objs, n_pairs, device)

    def _train_opts(self):
        """Dropout probabilities of the three sites the training path implements, read from the mirror modules (so that
        `module.p = 0` behaves as in torch), plus a fresh seed drawn from torch's generator."""
        t = self._trunk
        tr = t.fusion_transformer.transformer
        o = native.VetoTrainOpts()
        o.struct_size = ctypes.sizeof(native.VetoTrainOpts)
        o.p_pos = float(t.pos_embed[3].p)
        o.p_emb = float(tr.pos_drop.p)
        ps = {float(layer[0].fn.to_out[1].p) for layer in tr.layers}
        if len(ps) != 1:
            raise NotImplementedError("veto_amd: per-layer attention dropout rates must be equal")
        o.p_attn = ps.pop()
        others = [n for n, m in t.named_modules() if isinstance(m, nn.Dropout) and m.p > 0 and
                  not (n == "pos_embed.3" or n.endswith("pos_drop") or n.endswith("fn.to_out.1"))]
        others = [n for n in others if not n.startswith("bbox_embed")]        # bbox_embed is not on the VETO path
        if others:
            raise NotImplementedError("veto_amd: dropout is built for pos_embed, pos_drop and the attention output only; "
                                      "these modules also have p > 0: %s" % ", ".join(others[:6]))
        o.seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        return o

    def _run_native_train(self, proposals, rel_pair_idxs, roi_features, roi_depth_features, labels, logits):
        """Forward in training mode: BatchNorm1d(4) of pos_embed on batch statistics (and its running statistics
        updated the way nn.BatchNorm1d(momentum=0.001) does, roi_relation_predictors.py:4042-4047)."""
        device = roi_features.device
        stats = torch.empty(12, dtype=torch.float32, device=device)
        if any(isinstance(m, nn.Dropout) and m.p > 0 for n, m in self._trunk.named_modules() if not n.startswith("bbox_embed")):
            raise NotImplementedError("veto_amd: the forward-only training pass (VETO_AMD.TRAIN_FORWARD_ONLY) runs without dropout")
        out = self._run_native(proposals, rel_pair_idxs, roi_features, roi_depth_features, labels, logits, bn_batch_stats=stats)
        bn = self._trunk.pos_embed[0]
        with torch.no_grad():
            m = bn.momentum
            bn.running_mean.mul_(1 - m).add_(stats[0:4].to(bn.running_mean.device), alpha=m)
            bn.running_var.mul_(1 - m).add_(stats[8:12].to(bn.running_var.device), alpha=m)
            bn.num_batches_tracked += 1
        return out

    def _prepare_inputs(self, proposals, rel_pair_idxs, roi_features, roi_depth_features, labels, logits, bn_batch_stats=None):
        """Validates the call and flattens it into a veto_inputs_t.  Returns (inp, keep, n_objs, n_pairs, device, eng);
        `keep` are the tensors the struct points into."""
        device = roi_features.device
        eng = self._ensure_engine(device)
        n_objs = [len(p) for p in proposals]
        n_pairs = [int(p.shape[0]) for p in rel_pair_idxs]
        n_obj, n_pair = sum(n_objs), sum(n_pairs)
        if tuple(roi_features.shape[1:]) != (256, 8, 8) or tuple(roi_depth_features.shape[1:]) != (256, 8, 8):
            raise ValueError("roi features must be [N, 256, 8, 8], got %s / %s"
                             % (tuple(roi_features.shape), tuple(roi_depth_features.shape)))
        if roi_features.shape[0] != n_obj or roi_depth_features.shape[0] != n_obj:
            raise ValueError("roi feature rows (%d) != total proposals (%d)" % (roi_features.shape[0], n_obj))
        f32 = dict(device=device, dtype=torch.float32)
        rgb = roi_features.detach().to(**f32).contiguous()
        dep = roi_depth_features.detach().to(**f32).contiguous()
        boxes = torch.cat([p.bbox for p in proposals], 0).to(**f32).contiguous()
        mode = proposals[0].mode
        pairs = torch.cat([p.reshape(-1, 2) for p in rel_pair_idxs], 0).to(device=device, dtype=torch.int64).contiguous()
        obj_off, pair_off = self._offsets(tuple(n_objs), tuple(n_pairs), device)
        lab = labels.to(device=device, dtype=torch.int64).contiguous() if labels is not None else None
        lg = logits.detach().to(**f32).contiguous() if logits is not None else None
        inp = native.VetoInputs()
        inp.struct_size = ctypes.sizeof(native.VetoInputs)
        inp.n_obj, inp.n_pair, inp.n_img = n_obj, n_pair, len(proposals)
        inp.roi_rgb, inp.roi_depth, inp.boxes = rgb.data_ptr(), dep.data_ptr(), boxes.data_ptr()
        inp.box_mode = 0 if mode == "xyxy" else 1
        inp.obj_labels = lab.data_ptr() if lab is not None else None
        inp.obj_logits = lg.data_ptr() if lg is not None else None
        inp.rel_pairs = pairs.data_ptr()
        inp.img_obj_offset, inp.img_pair_offset = obj_off.data_ptr(), pair_off.data_ptr()
        inp.bn_batch_stats = bn_batch_stats.data_ptr() if bn_batch_stats is not None else None
        keep = [t for t in (rgb, dep, boxes, pairs, lab, lg, obj_off, pair_off, bn_batch_stats) if t is not None]
        return inp, keep, n_objs, n_pairs, device, eng

    def _run_native(self, proposals, rel_pair_idxs, roi_features, roi_depth_features, labels, logits,
                    debug=False, bn_batch_stats=None):
        inp, keep, n_objs, n_pairs, device, eng = self._prepare_inputs(proposals, rel_pair_idxs, roi_features, roi_depth_features,
                                                                       labels, logits, bn_batch_stats)
        n_obj, n_pair = inp.n_obj, inp.n_pair
        f32 = dict(device=device, dtype=torch.float32)
        need = eng.workspace_bytes(n_obj, n_pair)
        if self._workspace is None or self._workspace.numel() < need or self._workspace.device != device:
            self._workspace = None
            self._workspace = torch.empty(need, dtype=torch.uint8, device=device)
        out = torch.empty((n_pair, self._num_out), **f32)
        dbg, extras = None, None
        if debug:
            extras = {"subj_inds": torch.empty(n_pair, dtype=torch.int64, device=device),
                      "obj_inds": torch.empty(n_pair, dtype=torch.int64, device=device),
                      "tokens": torch.empty((n_pair, 19, 576), **f32),
                      "cls": torch.empty((n_pair, 576), **f32)}
            dbg = native.VetoDebugOutputs()
            dbg.struct_size = ctypes.sizeof(native.VetoDebugOutputs)
            dbg.subj_inds, dbg.obj_inds = extras["subj_inds"].data_ptr(), extras["obj_inds"].data_ptr()
            dbg.tokens, dbg.cls = extras["tokens"].data_ptr(), extras["cls"].data_ptr()
        stream = torch.cuda.current_stream(device).cuda_stream
        # (the audit is of the VETO_MIXED operands: the ABI refuses it on a handle that computes in another mode)
        if self._count_saturation and self._precision == native.VETO_MIXED and bn_batch_stats is None and not debug:
            self.last_saturation = eng.forward_saturation(stream, inp, self._workspace.data_ptr(), self._workspace.numel(), out.data_ptr())
        else:
            eng.forward(stream, inp, self._workspace.data_ptr(), self._workspace.numel(), out.data_ptr(), dbg)
        # the inputs above are referenced by enqueued kernels: keep them alive on this stream
        for t in keep:
            t.record_stream(torch.cuda.current_stream(device))
        self.last_debug = extras
        return out, n_objs, n_pairs

    # ---- training path with gradients (veto_forward_train / veto_backward) ---------------------------------------
    def _train_param_spec(self):
        """[(Parameter, engine weight name, first row, row count or None)] for every tensor the backward fills."""
        trunk = dict(self._trunk.named_parameters())
        spec = []
        for name, prm in trunk.items():
            if name.startswith(("obj_embed.", "class_projection.0.", "pos_embed.", "location_projection.0.", "fusion_transformer.")):
                spec.append((prm, name, 0, None))
        row = 0
        for head in self._head_modules:
            spec.append((head.weight, "rel_out.weight", row, head.out_features))
            spec.append((head.bias, "rel_out.bias", row, head.out_features))
            row += head.out_features
        return spec

    def _run_native_train_grad(self, proposals, rel_pair_idxs, roi_features, roi_depth_features, labels, logits=None):
        """Training-mode forward whose result carries an autograd graph: logits [sum P, num_out]."""
        spec = self._train_param_spec()
        # roi_features / roi_depth_features are differentiable inputs: veto_backward returns their gradients (the reference trains
        # its depth backbone through roi_depth_features, tools/relation_train_net.py:166-170)
        return _TrainFn.apply(self, (proposals, rel_pair_idxs, labels, logits), roi_features, roi_depth_features, *[s[0] for s in spec])


class _WsToken:
    """Ownership of the cached training workspace by one (forward, backward) pair; see _TrainFn.forward."""
    __slots__ = ("done", "__weakref__")

    def __init__(self):
        self.done = False


class _TrainFn(torch.autograd.Function):
    """logits = VETO trunk + heads in training mode; backward fills the gradients of every parameter through
    veto_backward.  The parameters are inputs only so that autograd routes their gradients."""

    @staticmethod
    def forward(ctx, owner, call, roi_features, roi_depth_features, *params):
        proposals, rel_pair_idxs, labels, obj_logits = call
        device = roi_features.device
        stats = torch.empty(12, dtype=torch.float32, device=device)
        inp, keep, n_objs, n_pairs, device, eng = owner._prepare_inputs(proposals, rel_pair_idxs, roi_features, roi_depth_features,
                                                                        labels, obj_logits, stats)
        lib = native.load_library()
        opts = owner._train_opts()
        need = lib.veto_train_workspace_bytes(eng.handle, inp.n_obj, inp.n_pair)
        # tens of GB: keep one workspace on the module and hand it to the next step once its backward has run (an
        # allocation of this size goes to the driver every time, and releasing it synchronises the device)
        # The cached workspace belongs to at most one forward whose backward has not run yet.  Ownership is a token held by
        # that forward's autograd context and seen here through a weak reference: a forward that never gets a backward
        # (torch.no_grad(), a loss-only validation pass, an exception in between) drops its context, the token dies with it
        # and the workspace is free again -- no sticky flag.
        ws = owner.__dict__.get("_train_ws")
        holder = owner.__dict__.get("_train_ws_owner")
        busy = holder is not None and holder() is not None and not holder().done
        if ws is None or ws.numel() < need or ws.device != device or busy:
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            if not busy:
                owner.__dict__["_train_ws"] = ws
        ctx.ws_token = None
        if ws is owner.__dict__.get("_train_ws"):
            ctx.ws_token = _WsToken()
            owner.__dict__["_train_ws_owner"] = weakref.ref(ctx.ws_token)
        out = torch.empty((inp.n_pair, owner._num_out), dtype=torch.float32, device=device)
        stream = torch.cuda.current_stream(device)
        native.check(lib.veto_forward_train(eng.handle, ctypes.c_void_p(stream.cuda_stream), ctypes.byref(inp), ctypes.byref(opts),
                                            ctypes.c_void_p(ws.data_ptr()), ws.numel(), ctypes.c_void_p(out.data_ptr())))
        ctx.opts = opts
        bn = owner._trunk.pos_embed[0]
        with torch.no_grad():    # nn.BatchNorm1d(momentum=0.001) running statistics
            m = bn.momentum
            bn.running_mean.mul_(1 - m).add_(stats[0:4].to(bn.running_mean.device), alpha=m)
            bn.running_var.mul_(1 - m).add_(stats[8:12].to(bn.running_var.device), alpha=m)
            bn.num_batches_tracked += 1
        ctx.owner, ctx.inp, ctx.keep, ctx.ws, ctx.eng = owner, inp, keep, ws, eng
        ctx.spec = owner._train_param_spec()
        return out

    @staticmethod
    def backward(ctx, dlogits):
        owner, eng, lib = ctx.owner, ctx.eng, native.load_library()
        device = dlogits.device
        dlogits = dlogits.detach().to(torch.float32).contiguous()
        n_floats = lib.veto_grad_floats(eng.handle)
        flat = torch.empty(n_floats, dtype=torch.float32, device=device)
        stream = torch.cuda.current_stream(device)
        d_rgb = d_dep = None
        n_obj = ctx.inp.n_obj
        if ctx.needs_input_grad[2]:
            d_rgb = torch.empty((n_obj, 256, 8, 8), dtype=torch.float32, device=device)
        if ctx.needs_input_grad[3]:
            d_dep = torch.empty((n_obj, 256, 8, 8), dtype=torch.float32, device=device)
        ctx.opts.d_roi_rgb = d_rgb.data_ptr() if d_rgb is not None else None
        ctx.opts.d_roi_depth = d_dep.data_ptr() if d_dep is not None else None
        native.check(lib.veto_backward(eng.handle, ctypes.c_void_p(stream.cuda_stream), ctypes.byref(ctx.inp), ctypes.byref(ctx.opts),
                                       ctypes.c_void_p(ctx.ws.data_ptr()), ctx.ws.numel(), ctypes.c_void_p(dlogits.data_ptr()),
                                       ctypes.c_void_p(flat.data_ptr())))
        if ctx.ws_token is not None:
            ctx.ws_token.done = True
        offsets = eng.weight_offsets()
        grads = []
        for prm, name, row0, rows in ctx.spec:
            off, numel = offsets[name]
            if rows is None:
                g = flat[off:off + numel]
            else:
                per_row = prm.numel() // rows
                g = flat[off + row0 * per_row: off + (row0 + rows) * per_row]
            grads.append(g.view_as(prm).to(prm.device))
        return (None, None, d_rgb, d_dep) + tuple(grads)


def _offset_tensors(n_objs, n_pairs, device):
    obj_off = torch.tensor([0] + list(_cumsum(n_objs)), dtype=torch.int32, device=device)
    pair_off = torch.tensor([0] + list(_cumsum(n_pairs)), dtype=torch.int32, device=device)
    return obj_off, pair_off


def _cumsum(xs):
    s = 0
    for x in xs:
        s += x
        yield s


def _cat_field(proposals, name):
    return torch.cat([p.get_field(name) for p in proposals], 0)




@ROI_RELATION_PREDICTOR.register("VETOPredictor")
class VETOPredictor(nn.Module, _NativeForward):
    def __init__(self, config, in_channels):
        super().__init__()
        self.mode = _mode(config)
        statistics = _dataset_statistics(config)
        self.obj_classes, self.rel_classes = statistics["obj_classes"], statistics["rel_classes"]
        self.num_obj_cls, self.num_rel_cls = len(self.obj_classes), len(self.rel_classes)
        self.embed_dim = 200
        _build_trunk(self, config, self.num_obj_cls, with_embed2=True)
        vecs = _embedding_vectors(self.obj_classes, getattr(config, "GLOVE_DIR", ""), self.embed_dim)
        with torch.no_grad():
            self.obj_embed.weight.copy_(vecs)
        dim = config.MODEL.ROI_RELATION_HEAD.VETOTRANSFORMER.T_INPUT_DIM
        self.rel_out = _xavier_linear(dim, self.num_rel_cls)
        self.beta_loss = bool(config.GLOBAL_SETTING.BETA_LOSS)
        weights = torch.ones(self.num_rel_cls)
        if self.beta_loss:
            weights = class_balanced_weights(getattr(config.GLOBAL_SETTING, "REL_COUNTS", None), self.num_rel_cls)
        self.criterion_loss_rel = nn.CrossEntropyLoss(weight=weights)
        self.criterion_loss = nn.CrossEntropyLoss()
        self._native_init(config, self, self.num_obj_cls, [self.rel_out])
        self._train_forward_only = bool(getattr(getattr(config, "VETO_AMD", None), "TRAIN_FORWARD_ONLY", False))

    def forward(self, proposals, rel_pair_idxs, rel_labels, logger, roi_features=None,
                roi_depth_features=None, rel_binarys=None):
        if self.mode == "predcls":
            labels = _cat_field(proposals, "labels").long()
            logits = None
            obj_label_for_dist = labels
        else:
            logits = _cat_field(proposals, "predict_logits").detach()
            obj_label_for_dist = _cat_field(proposals, "pred_labels").detach().long()
            labels = None
        if self.training:   # :4127-4136: losses only
            from .losses import ce_loss, relation_ce_loss
            target = torch.cat(list(rel_labels), 0)
            w = self.criterion_loss_rel.weight
            add_losses = {}
            if self._train_forward_only:
                rel, _, _ = self._run_native_train(proposals, rel_pair_idxs, roi_features, roi_depth_features, labels, logits)
                add_losses["rel_loss"] = relation_ce_loss(rel, target, weight=w)[0][0]
            else:
                rel = self._run_native_train_grad(proposals, rel_pair_idxs, roi_features, roi_depth_features, labels, logits)
                add_losses["rel_loss"] = ce_loss(rel, target, weight=w)
            if self.mode != "predcls":   # :4129-4132: CE over obj_dists, which is the ONE-HOT of pred_labels here
                fg = _cat_field(proposals, "labels").long()
                onehot = nn.functional.one_hot(obj_label_for_dist.to(rel.device), self.num_obj_cls).float()
                add_losses["obj_loss"] = relation_ce_loss(onehot, fg)[0][0]
            return None, None, add_losses, None, None, None
        rel, n_objs, n_pairs = self._run_native(proposals, rel_pair_idxs, roi_features, roi_depth_features,
                                                labels, logits, debug=getattr(self, "debug_outputs", False))
        obj_dists = nn.functional.one_hot(obj_label_for_dist.to(rel.device), self.num_obj_cls).float()
        return obj_dists.split(n_objs, dim=0), rel.split(n_pairs, dim=0), {}, None, None, None


def class_balanced_weights(counts, num_cls, beta=0.999):
    """BETA_LOSS class weights (roi_relation_predictors.py:4058-4066). The reference reads a
    hard-coded absolute pickle path; here the 51 counts come from GLOBAL_SETTING.REL_COUNTS."""
    if counts is None:
        raise ValueError("GLOBAL_SETTING.BETA_LOSS needs GLOBAL_SETTING.REL_COUNTS (the predicate counts of "
                         "the reference's pred_counts.pkl)")
    c = torch.as_tensor(counts, dtype=torch.float64)
    if c.numel() != num_cls:
        raise ValueError("REL_COUNTS must have %d entries" % num_cls)
    c = torch.sort(c, descending=True)[0]
    w = (1.0 - beta) / (1.0 - beta ** c)
    return (w * (num_cls / w.sum())).float()


class _EnsembleParams(_Holder):
    """Parameter tree of the reference's `Ensemble` (roi_relation_predictors.py:3661-3744)."""

    def __init__(self, config, num_obj_cls, sizes, obj_classes, experts_per_group=1, expert_group=False):
        super().__init__()
        _build_trunk(self, config, num_obj_cls, with_embed2=False)
        vecs = _embedding_vectors(obj_classes, getattr(config, "GLOVE_DIR", ""), 200)
        with torch.no_grad():
            self.obj_embed.weight.copy_(vecs)
        dim = config.MODEL.ROI_RELATION_HEAD.VETOTRANSFORMER.T_INPUT_DIM
        self.rel_out_group = nn.ModuleList([])
        if expert_group:
            # :3717-3723: experts_per_group lists of K heads; `rel_out` stays bound to the LAST list, so the
            # state dict holds it twice (rel_out.{k} and rel_out_group.{E-1}.{k} are the same tensors)
            for _ in range(experts_per_group):
                self.rel_out = nn.ModuleList(_xavier_linear(dim, g + 2) for g in sizes)
                self.rel_out_group.append(self.rel_out)
        else:
            self.rel_out = nn.ModuleList(_xavier_linear(dim, g + 2) for g in sizes)
        self.CE_loss = nn.CrossEntropyLoss()
        self.criterion_loss = nn.CrossEntropyLoss()


@ROI_RELATION_PREDICTOR.register("VETOPredictor_MEET")
class VETOPredictor_MEET(nn.Module, _NativeForward):
    def __init__(self, config, in_channels):
        super().__init__()
        self.mode = _mode(config)
        statistics = _dataset_statistics(config)
        self.params = {"statistics": statistics, "obj_classes": statistics["obj_classes"],
                       "rel_classes": statistics["rel_classes"]}
        dataset = config.GLOBAL_SETTING.DATASET_CHOICE
        self.group_split_mode = config.GCL_SETTING.GROUP_SPLIT_MODE
        self.max_group_element_number_list = meet_tables.group_sizes(dataset, self.group_split_mode)
        self.incre_idx_list = meet_tables.incre_idx_list(self.max_group_element_number_list)
        self.num_groups = len(self.max_group_element_number_list)
        # :3901-3903: three experts per group unless ENSEMBLE_LEARNING.EXPERT_GROUP is off
        self.expert_group = bool(config.ENSEMBLE_LEARNING.EXPERT_GROUP)
        self.experts_per_group = 3 if self.expert_group else 1
        self.num_obj_cls = len(self.params["obj_classes"])
        self.model = _EnsembleParams(config, self.num_obj_cls, self.max_group_element_number_list,
                                     self.params["obj_classes"], self.experts_per_group, self.expert_group)
        # head columns of the one fused GEMV: expert-major, then group (all heads read the same CLS row)
        if self.expert_group:
            heads = [h for lst in self.model.rel_out_group for h in lst]
        else:
            heads = list(self.model.rel_out)
        self._native_init(config, self.model, self.num_obj_cls, heads)
        self._train_forward_only = bool(getattr(getattr(config, "VETO_AMD", None), "TRAIN_FORWARD_ONLY", False))
        self._dataset, self._sampler = dataset, None
        self._zero_label_padding_mode = str(config.GCL_SETTING.ZERO_LABEL_PADDING_MODE)

    def forward(self, proposals, rel_pair_idxs, rel_labels, logger, roi_features=None,
                roi_depth_features=None, rel_binarys=None):
        if self.mode == "predcls":
            labels = _cat_field(proposals, "labels").long()
            dist_labels = labels
        else:
            if self.mode == "sgdet":
                raise NotImplementedError("veto_amd: MEET sgdet decoding (per-class NMS, "
                                          "roi_relation_predictors.py:3855-3874) is outside the hot path")
            dist_labels = _cat_field(proposals, "pred_labels").detach().long()
            # obj_dists[:, 1:].max(1)[1] + 1 over a one-hot (:3776-3784): the label itself, or 1 for label 0
            labels = torch.where(dist_labels > 0, dist_labels, torch.ones_like(dist_labels))
        if self.training:
            # :3930-3969 expert sampling, :3806-3846 group label remap + per-group CE
            from .losses import MeetTrainingSampler, ce_loss, relation_ce_loss
            if self._train_forward_only:
                rel, _, _ = self._run_native_train(proposals, rel_pair_idxs, roi_features, roi_depth_features, labels, None)
            else:
                rel = self._run_native_train_grad(proposals, rel_pair_idxs, roi_features, roi_depth_features, labels)
            if self._sampler is None or self._sampler.device != rel.device:
                self._sampler = MeetTrainingSampler(self._dataset, self.max_group_element_number_list, device=rel.device,
                                                    zero_label_padding_mode=self._zero_label_padding_mode)
            chosen, group_labels = self._sampler.sample(torch.cat(list(rel_labels), 0))
            add_losses, col = {}, 0
            for e in range(self.experts_per_group):      # :3833-3846: every expert of a group sees the same rows and labels
                for k, g in enumerate(self.max_group_element_number_list):
                    sl = rel[:, col:col + g + 2]
                    key = "group_%d%d_CE_loss" % (k, e + 1) if self.expert_group else "group_%d_CE_loss" % k
                    add_losses[key] = relation_ce_loss(sl, group_labels[k], rows=chosen[k])[0][0] \
                        if self._train_forward_only else ce_loss(sl, group_labels[k], rows=chosen[k])
                    col += g + 2
            if self.mode != "predcls":   # :3823-3827
                obj_logits = _cat_field(proposals, "predict_logits").detach()
                add_losses["obj_loss"] = relation_ce_loss(obj_logits.to(rel.device), _cat_field(proposals, "labels").long())[0][0]
            return None, None, add_losses, self.incre_idx_list, [chosen], None
        rel, n_objs, n_pairs = self._run_native(proposals, rel_pair_idxs, roi_features, roi_depth_features,
                                                labels, None, debug=getattr(self, "debug_outputs", False))
        rel_dists, col = {}, 0
        for e in range(self.experts_per_group):
            for k, g in enumerate(self.max_group_element_number_list):
                # :3833-3841: 'group_<k><expert 1..3>' with EXPERT_GROUP, 'group_<k>' without
                key = "group_%d%d" % (k, e + 1) if self.expert_group else "group_%d" % k
                rel_dists[key] = rel[:, col:col + g + 2]
                col += g + 2
        obj_dists = nn.functional.one_hot(dist_labels.to(rel.device), self.num_obj_cls).float().split(n_objs, dim=0)
        return obj_dists, rel_dists, {}, self.incre_idx_list, None, {}
