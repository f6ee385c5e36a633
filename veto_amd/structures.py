"""A minimal stand-in for the reference's BoxList (pysgg/structures/bounding_box.py:9-78) carrying
exactly what the VETO predictor reads: `.bbox`, `.size`, `.mode`, `len()`, `get_field`,
`add_field`, `convert`.  Used by tests and bench.py; inside pysgg the real BoxList is passed."""
import torch


class BoxList:
    def __init__(self, bbox, image_size, mode="xyxy"):
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        bbox = torch.as_tensor(bbox, dtype=torch.float32)
        if bbox.ndim != 2 or bbox.size(-1) != 4:
            raise ValueError("bbox should be [N, 4], got %s" % (tuple(bbox.shape),))
        self.bbox, self.size, self.mode = bbox, image_size, mode
        self.extra_fields = {}

    def add_field(self, name, value):
        self.extra_fields[name] = value

    def get_field(self, name):
        return self.extra_fields[name]

    def has_field(self, name):
        return name in self.extra_fields

    def fields(self):
        return list(self.extra_fields)

    def convert(self, mode):
        if mode not in ("xyxy", "xywh"):
            raise ValueError("mode should be 'xyxy' or 'xywh'")
        if mode == self.mode:
            return self
        a, b, c, d = self.bbox.unbind(-1)
        if mode == "xywh":   # +1 pixel convention
            out = torch.stack([a, b, c - a + 1, d - b + 1], dim=-1)
        else:
            out = torch.stack([a, b, a + (c - 1).clamp(min=0), b + (d - 1).clamp(min=0)], dim=-1)
        new = BoxList(out, self.size, mode)
        new.extra_fields = dict(self.extra_fields)
        return new

    def to(self, device):
        new = BoxList(self.bbox.to(device), self.size, self.mode)
        for k, v in self.extra_fields.items():
            new.extra_fields[k] = v.to(device) if hasattr(v, "to") else v
        return new

    def area(self):
        """bounding_box.py:249-259: xyxy boxes use the +1 pixel convention."""
        b = self.bbox
        if self.mode == "xyxy":
            return (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
        return b[:, 2] * b[:, 3]

    def __len__(self):
        return self.bbox.shape[0]
