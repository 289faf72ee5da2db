"""Host wrappers of the loss kernels shared by the nets/* mirrors: the softmax cross-entropy family
(OHNM of nets/model.py:204-261, `ohem_loss` of nets/model_vgg_16.py:243-282,
`PixelLinkNet.build_loss` of nets/pixellink.py:88-263, focal links)."""
import numpy as np
import torch

from . import ops
from ._lib import SoftmaxLossDesc
from .graph import F32, get_default_graph


class Scalar:
    """Device-resident loss: buffer[0] = total, the rest its terms.  `.item()` synchronises."""

    def __init__(self, buf, terms=slice(1, 10)):
        self.data = buf
        self._terms = terms

    def item(self):
        return float(self.data[0].item())

    def terms(self):
        return self.data[self._terms].detach().cpu().numpy()


def to_device(g, arr, dtype=F32):
    if isinstance(arr, torch.Tensor):
        return arr.to(device=g.device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(g.device)


def softmax_loss(g, pixel_logits, link_logits, pixel_labels, link_labels, *, pixel_rule, label_rule,
                 link_gate, focal=None, neg_ratio=3.0):
    """pixel_logits / link_logits: head handles (`.data` f32 [n,h,w,2] / [n,h,w,16], `.grad`).
    Returns Scalar over loss10 = [total, pixel, link_0..7] and records the backward seed."""
    n, h, w, _ = pixel_logits.data.shape
    pl = to_device(g, pixel_labels)
    ll = to_device(g, link_labels)
    if pl.numel() != n * h * w or ll.numel() != n * h * w * 8:
        raise ValueError("label maps must be [n,h,w(,1)] and [n,h,w,8]")
    alpha, gamma = focal if focal is not None else (0.25, 2.0)
    d = SoftmaxLossDesc(n, h * w, pixel_rule, label_rule, int(link_gate), int(focal is not None),
                        float(neg_ratio), float(alpha), float(gamma))
    thr, sums, out = g.empty((n,), F32), g.empty((34,), F32), g.empty((10,), F32)
    ops.softmax_loss_fwd(d, pixel_logits.data, link_logits.data, pl, ll, thr, sums, out, g.workspace())

    def backward():
        pixel_logits.grad = g.empty(pixel_logits.data.shape, F32)
        link_logits.grad = g.empty(link_logits.data.shape, F32)
        ops.softmax_loss_bwd(d, pixel_logits.data, link_logits.data, pl, ll, thr, sums, g.seed_scale(),
                             pixel_logits.grad, link_logits.grad)
    g.record(backward)
    res = Scalar(out)
    res.ohnm_threshold = thr

    def selected_mask():
        """uint8 [n,h,w]: pixels whose CE counts — positives and mined negatives (OHNM_batch's `selected`)."""
        m = g.empty((n, h, w), torch.uint8)
        ops.softmax_loss_selected(d, pixel_logits.data, pl, thr, m)
        return m
    res.selected_mask = selected_mask
    g.collections["losses"].append(res)
    return res
