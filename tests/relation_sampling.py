"""TEST INFRASTRUCTURE (not part of the product package: SURVEY.md section 2 marks the relation head's pair sampling out of
scope).  A stand-in for the reference's RelationSampling, used by the tests that drive VETORelationHead in training mode; a
drop-in deployment passes the reference's own sampler to VETORelationHead(samp_processor=...).

Relation pair sampling of the relation head (host-side index bookkeeping, mirrors the reference interface).

  RelationSampling.prepare_test_pairs   sampling.py:31-52   -> veto_amd.pairs.prepare_test_pairs (HIP enumeration)
  RelationSampling.gtbox_relsample      sampling.py:54-107  training with GT boxes: per image up to
        int(batch_size_per_image * positive_fraction) foreground pairs (a random subset if there are more), the rest of the
        batch_size_per_image budget from a random permutation of the non-GT ordered pairs, labels foreground-first.
The random draws are torch.randperm calls at the same two points per image, on the proposals' device, in the same order as the
reference, so a run seeded like the reference selects the same pairs.  detect_relsample (sgdet) is not built."""
import torch

from veto_amd.pairs import prepare_test_pairs


class RelationSampling(object):
    def __init__(self, fg_thres, require_overlap, num_sample_per_gt_rel, batch_size_per_image, positive_fraction, max_proposal_pairs,
                 use_gt_box, test_overlap):
        self.fg_thres, self.require_overlap = fg_thres, require_overlap
        self.num_sample_per_gt_rel = num_sample_per_gt_rel
        self.batch_size_per_image, self.positive_fraction = batch_size_per_image, positive_fraction
        self.max_proposal_pairs, self.use_gt_box, self.test_overlap = max_proposal_pairs, use_gt_box, test_overlap

    def prepare_test_pairs(self, device, proposals):
        if not self.use_gt_box:
            raise NotImplementedError("veto_amd: pair preparation for detected boxes (sgdet, overlap filter) is not built")
        return prepare_test_pairs(device, proposals, self.max_proposal_pairs)

    def gtbox_relsample(self, proposals, targets):
        assert self.use_gt_box
        max_fg = int(self.batch_size_per_image * self.positive_fraction)
        pairs_out, labels_out, binaries = [], [], []
        for proposal, target in zip(proposals, targets):
            device = proposal.bbox.device
            n = proposal.bbox.shape[0]
            assert n == target.bbox.shape[0]
            rel = target.get_field("relation")                       # [n, n] predicate matrix, 0 = none
            fg = torch.nonzero(rel > 0)
            head, tail = fg[:, 0].contiguous().view(-1), fg[:, 1].contiguous().view(-1)
            fg_labels = rel[head, tail].contiguous().view(-1)
            proposal.add_field("locating_match", torch.ones(len(proposal), device=device))   # GT boxes: every box is matched
            sym = torch.zeros((n, n), device=device).long()
            sym[head, tail] = 1
            sym[tail, head] = 1
            binaries.append(sym)
            candidate = torch.ones((n, n), device=device).long() - torch.eye(n, device=device).long()
            candidate[head, tail] = 0
            bg = torch.nonzero(candidate > 0)
            if fg.shape[0] > max_fg:                                  # random subset of the foreground pairs
                keep = torch.randperm(fg.shape[0], device=device)[:max_fg]
                fg, fg_labels = fg[keep], fg_labels[keep]
            n_fg = min(fg.shape[0], max_fg)
            keep = torch.randperm(bg.shape[0], device=device)[:self.batch_size_per_image - n_fg]
            bg = bg[keep]
            pairs_out.append(torch.cat((fg, bg), dim=0))
            labels_out.append(torch.cat((fg_labels.long(), torch.zeros(bg.shape[0], device=device).long()), dim=0).contiguous().view(-1))
        return proposals, labels_out, pairs_out, binaries

    def detect_relsample(self, proposals, targets):
        raise NotImplementedError("veto_amd: relation sampling for detected boxes (sgdet) is not built")


def make_roi_relation_samp_processor(cfg):
    """sampling.py:312-324."""
    rh = cfg.MODEL.ROI_RELATION_HEAD
    return RelationSampling(cfg.MODEL.ROI_HEADS.FG_IOU_THRESHOLD, rh.REQUIRE_BOX_OVERLAP, rh.NUM_SAMPLE_PER_GT_REL,
                            rh.BATCH_SIZE_PER_IMAGE, rh.POSITIVE_FRACTION, rh.MAX_PROPOSAL_PAIR, rh.USE_GT_BOX,
                            cfg.TEST.RELATION.REQUIRE_OVERLAP)
