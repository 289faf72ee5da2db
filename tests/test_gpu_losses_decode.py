"""GPU parity (through the C ABI) of the softmax / OHNM / focal losses and the decode kernels
against the CPU oracle on seeded inputs.  Integer / index outputs (selected masks, detect mask,
component labels) must be bit-exact; float losses to 1e-5 relative, gradients to 1e-5 abs."""
import numpy as np
import pytest
import torch

from oracle import ocr_oracle as O

pytestmark = pytest.mark.gpu


class H:   # minimal head handle
    def __init__(self, t):
        self.data, self.grad = t, None


def _inputs(n=3, q=24, seed=0, all_neg_image=True):
    rng = np.random.default_rng(seed)
    _, pixel, link, _ = O.synthetic_batch(rng, n, q * 4, rects=4)
    if all_neg_image:
        pixel[-1] = 0          # an image without positives: nothing may be mined from it
        link[-1] = 0
    pl = rng.standard_normal((n, q, q, 2)).astype(np.float32) * 2
    ll = rng.standard_normal((n, q, q, 16)).astype(np.float32) * 2
    return pixel, link, pl, ll


def _device_loss(device, fn_name, pixel, link, pl, ll, **kw):
    from tensorflow_ocr_amd import losses
    from tensorflow_ocr_amd.graph import Graph
    g = Graph(device, loss_scale=1.0)
    hp, hl = H(torch.from_numpy(pl).to(device)), H(torch.from_numpy(ll).to(device))
    s = losses.softmax_loss(g, hp, hl, pixel, link, **kw)
    g.backward()
    torch.cuda.synchronize()
    return s, hp.grad.cpu().numpy(), hl.grad.cpu().numpy()


def test_ohnm_loss_matches_oracle(device):
    pixel, link, pl, ll = _inputs()
    s, gp, gl = _device_loss(device, "loss", pixel, link, pl, ll, pixel_rule=0, label_rule=0, link_gate=True)
    tp, tl = torch.from_numpy(pl).requires_grad_(True), torch.from_numpy(ll).requires_grad_(True)
    total, cls, links, sel = O.model_loss_ohnm(torch.from_numpy(pixel), tp, torch.from_numpy(link), tl)
    total.backward()
    out = s.data.cpu().numpy()
    assert abs(out[0] - float(total)) < 1e-4 * max(1, abs(float(total)))
    assert abs(out[1] - float(cls)) < 1e-5
    assert np.allclose(out[2:10], [float(v) for v in links], rtol=1e-5, atol=1e-6)
    # index work is exact: the mining score is one fixed sequence of IEEE f32 operations on both
    # sides (loss_softmax.hip det_exp / O.det_exp_f32), so the k-th-smallest threshold and the mined
    # mask — the kernel's own W map — equal the oracle's bit for bit
    thr = s.ohnm_threshold.cpu().numpy()
    n = pl.shape[0]
    dev_sel = s.selected_mask().cpu().numpy().reshape(n, -1)
    assert np.array_equal(dev_sel, sel.numpy().astype(np.uint8))
    sc = O.neg_score_f32(pl[..., 0], pl[..., 1]).reshape(n, -1)
    lab = pixel.reshape(n, -1)
    for b in range(n - 1):
        negs = np.sort(sc[b][lab[b] == 0])
        k = int(min(3 * (lab[b] == 1).sum(), len(negs)))
        assert thr[b] == negs[k - 1]                 # exactly the k-th smallest P(neg), same float
        assert dev_sel[b].sum() >= (lab[b] == 1).sum() + k          # tie-inclusive
    assert thr[-1] == -1.0                           # no positives -> nothing mined
    assert np.abs(gp - tp.grad.numpy()).max() < 1e-6
    assert np.abs(gl - tl.grad.numpy()).max() < 1e-6


def test_ohem_loss_matches_oracle(device):
    pixel, link, pl, ll = _inputs(all_neg_image=False)
    s, gp, gl = _device_loss(device, "ohem", pixel, link, pl, ll, pixel_rule=1, label_rule=0, link_gate=True)
    tp, tl = torch.from_numpy(pl).requires_grad_(True), torch.from_numpy(ll).requires_grad_(True)
    total, lpix, links = O.ohem_loss(torch.from_numpy(pixel), tp, torch.from_numpy(link), tl)
    total.backward()
    assert abs(s.item() - float(total)) < 1e-4
    assert np.abs(gp - tp.grad.numpy()).max() < 1e-6 and np.abs(gl - tl.grad.numpy()).max() < 1e-6


@pytest.mark.parametrize("focal", [None, (0.25, 2.0)])
def test_pixellink_build_loss_matches_oracle(device, focal):
    pixel, link, pl, ll = _inputs(all_neg_image=False)
    link[..., 3] = 0                                  # a direction with no positive link: zero guard
    s, gp, gl = _device_loss(device, "build", pixel[..., 0], link, pl, ll, pixel_rule=2, label_rule=1,
                             link_gate=False, focal=focal)
    tp, tl = torch.from_numpy(pl).requires_grad_(True), torch.from_numpy(ll).requires_grad_(True)
    p2, ltot, links = O.pixellink_build_loss(tp, tl, torch.from_numpy(pixel[..., 0]), torch.from_numpy(link), focal)
    (p2 + ltot).backward()
    out = s.data.cpu().numpy()
    assert abs(2 * out[1] - float(p2)) < 1e-5 and abs(out[2:10].sum() - float(ltot)) < 1e-4
    assert np.abs(gp - tp.grad.numpy()).max() < 1e-6 and np.abs(gl - tl.grad.numpy()).max() < 2e-6


def test_ohnm_no_negatives_nan_like_reference(device):
    """positives but a direction without negative links -> 0/0 = NaN in the reference formula."""
    pixel, link, pl, ll = _inputs(n=1, all_neg_image=False)
    link[..., 0] = 1
    s, _, _ = _device_loss(device, "loss", pixel, link, pl, ll, pixel_rule=0, label_rule=0, link_gate=True)
    assert np.isnan(s.item())


def test_pixel_detect_and_link_cc_bit_exact(device):
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.tool import pixellink_fn as PF
    g = Graph(device)
    rng = np.random.default_rng(3)
    n, q = 3, 48
    pl, ll = O.synthetic_decode_maps(rng, n, q, strength=2.0)
    ps = PF.pixel_scores(torch.from_numpy(pl), graph=g)                 # [n,q,q,2]
    ls = PF.link_scores(torch.from_numpy(ll), graph=g)                  # [8,n,q,q,2]
    o_ps = torch.softmax(torch.from_numpy(pl), -1).numpy()
    o_ls = np.stack([torch.softmax(torch.from_numpy(ll[..., 2 * i:2 * i + 2]), -1).numpy() for i in range(8)])
    assert np.abs(ps.cpu().numpy() - o_ps).max() < 1e-6 and np.abs(ls.cpu().numpy() - o_ls).max() < 1e-6
    # from here on use the DEVICE scores on both sides so thresholds see identical floats
    ps_np, ls_np = ps.cpu().numpy(), ls.cpu().numpy()
    mask = PF.tf_pixel_detect(ps[..., 1:2].contiguous(), ls, 0.8, 0.8, graph=g).cpu().numpy()
    assert np.array_equal(mask, O.pixel_detect(ps_np[..., 1:2], ls_np, 0.8, 0.8))
    labels, ncomp, comps = PF.link_cc_decode(ps[..., 1].contiguous(), ls, 0.8, 0.9, min_size=10, graph=g)
    labels, ncomp, comps = labels.cpu().numpy(), ncomp.cpu().numpy(), comps.cpu().numpy()
    for b in range(n):
        ol, oc = O.link_cc_union(ps_np[b, :, :, 1], ls_np[:, b, :, :, 1], 0.8, 0.9, 10)
        assert np.array_equal(labels[b], ol)
        assert ncomp[b] == len(oc) and ncomp[b] > 0
        assert [tuple(c) for c in comps[b, :ncomp[b]]] == oc


def test_link_cc_equals_reference_dfs_on_symmetric_links(device):
    """With mutually consistent (symmetric) link predictions the order-dependent directed DFS of
    test_pixellink_fast.py and the order-independent union-find give the same partition."""
    from tensorflow_ocr_amd.graph import Graph
    from tensorflow_ocr_amd.tool import pixellink_fn as PF
    g = Graph(device)
    rng = np.random.default_rng(5)
    q = 40
    _, pixel, link, _ = O.synthetic_batch(rng, 1, q * 4, rects=5)
    ps = (pixel[0, :, :, 0] * 0.98).astype(np.float32)
    ls = (link[0].transpose(2, 0, 1) * 0.99).astype(np.float32)          # [8,q,q], symmetric by construction
    labels, ncomp, _ = PF.link_cc_decode(torch.from_numpy(ps[None]), torch.from_numpy(ls[:, None].copy()),
                                         0.8, 0.9, min_size=10, graph=g)
    ref = O.link_cc_reference_dfs(ps, ls, 0.8, 0.9, 10)
    lab = labels.cpu().numpy()[0]
    # same partition: a bijection between the two label sets
    pairs = set(zip(lab.ravel().tolist(), ref.ravel().tolist()))
    assert len(pairs) == len(set(a for a, _ in pairs)) == len(set(b for _, b in pairs))
    assert int(ncomp.cpu()[0]) == ref.max() and ref.max() > 0
