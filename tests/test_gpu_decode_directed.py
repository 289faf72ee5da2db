"""GPU: `link_cc_decode(mode="reference_dfs")` — the reference's own directed-DFS grouping
(test_pixellink_fast.py:153-178) — bit-exact against the literal script restatement
`O.link_cc_reference_dfs`, on the maps where it differs from the default union labelling."""
import numpy as np
import pytest
import torch

from oracle import ocr_oracle as O

pytestmark = pytest.mark.gpu


def _sm(l):
    e = np.exp(l - l.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def _maps(seed, n, q4, strength):
    rng = np.random.default_rng(seed)
    pl, ll = O.synthetic_decode_maps(rng, n, q4, strength)
    ps = _sm(pl)[..., 1].astype(np.float32)                                                 # [n,q,q]
    ls = np.stack([_sm(ll[..., 2 * d:2 * d + 2])[..., 1] for d in range(8)]).astype(np.float32)   # [8,n,q,q]
    return ps, ls


def _decode(g, ps, ls, pt, lt, ms, mode, max_comps=4096, key_order="py27"):
    from tensorflow_ocr_amd.tool import pixellink_fn as PF
    lab, nc, comps = PF.link_cc_decode(torch.from_numpy(ps), torch.from_numpy(ls), pt, lt, min_size=ms,
                                       max_comps=max_comps, graph=g, mode=mode, key_order=key_order)
    return lab.cpu().numpy(), nc.cpu().numpy(), comps.cpu().numpy()


@pytest.mark.parametrize("seed,n,q4,strength,pt,lt,ms", [
    (0, 2, 64, 3.0, 0.8, 0.9, 10),      # the survey's decode maps: directed == union here
    (1, 2, 64, 1.5, 0.8, 0.9, 10),      # weak links: single boundary pixels reachable one way only
    (2, 3, 48, 0.8, 0.6, 0.7, 3),       # near noise, small size filter: many failing seeds, re-collected later
    (3, 2, 40, 0.4, 0.5, 0.6, 2),
    (4, 1, 96, 2.0, 0.8, 0.9, 10),
    (5, 2, 33, 1.0, 0.7, 0.8, 5),       # odd size: tiles of the union pass are ragged
])
@pytest.mark.parametrize("key_order", ["py27", "ascending"])
def test_reference_dfs_mode_bit_exact(device, seed, n, q4, strength, pt, lt, ms, key_order):
    """key_order="py27": seeds in the iteration order of the script's Python-2 dict — the default;
    "ascending": the order rounds 1-3 used."""
    from tensorflow_ocr_amd.graph import Graph
    g = Graph(device)
    ps, ls = _maps(seed, n, q4, strength)
    lab, nc, comps = _decode(g, ps, ls, pt, lt, ms, "reference_dfs", key_order=key_order)
    for b in range(n):
        ref = O.link_cc_reference_dfs(ps[b], [ls[d, b] for d in range(8)], pt, lt, ms, key_order=key_order)
        assert np.array_equal(lab[b], ref), (b, int((lab[b] != ref).sum()))
        k = int(ref.max())
        assert nc[b] == k
        for gid in range(1, min(k, comps.shape[1]) + 1):        # comps = (seed of the group's search, size)
            members = np.nonzero(ref.ravel() == gid)[0]
            assert comps[b, gid - 1, 1] == len(members) and comps[b, gid - 1, 0] in members


def test_reference_dfs_dict_order_differs_from_ascending_on_a_one_way_chain(device):
    """148 keys 962..1109 of a "right"-links-only chain sit in a 512-slot dict: 1024..1109 are met first (slots 0..85),
    so x >= 64 becomes group 1 and x < 64 group 2; in ascending order the head collects the whole chain."""
    from tensorflow_ocr_amd.graph import Graph
    g = Graph(device)
    ps = np.zeros((1, 24, 160), np.float32)
    ls = np.zeros((8, 1, 24, 160), np.float32)
    ps[0, 6, 2:150] = 0.9
    ls[3, 0, 6, 2:150] = 0.95
    for ko, want_groups in (("py27", 2), ("ascending", 1)):
        lab, nc, comps = _decode(g, ps, ls, 0.8, 0.9, 10, "reference_dfs", key_order=ko)
        ref = O.link_cc_reference_dfs(ps[0], [ls[d, 0] for d in range(8)], 0.8, 0.9, 10, key_order=ko)
        assert np.array_equal(lab[0], ref) and nc[0] == want_groups == ref.max()
    lab, _, comps = _decode(g, ps, ls, 0.8, 0.9, 10, "reference_dfs")
    assert (lab[0, 6, 64:150] == 1).all() and (lab[0, 6, 2:64] == 2).all()
    assert comps[0, 0].tolist() == [6 * 160 + 64, 86] and comps[0, 1].tolist() == [6 * 160 + 2, 62]


def test_reference_dfs_differs_from_union_only_where_the_oracles_do(device):
    """Same maps through both modes: the device's two labellings differ exactly where the two oracles differ
    (the a15 deviation of the default mode, now selectable away)."""
    from tensorflow_ocr_amd.graph import Graph
    g = Graph(device)
    ps, ls = _maps(1, 4, 64, 1.5)
    d_lab, _, _ = _decode(g, ps, ls, 0.8, 0.9, 10, "reference_dfs")
    u_lab, _, _ = _decode(g, ps, ls, 0.8, 0.9, 10, "union")
    n_diff = 0
    for b in range(4):
        A = O.link_cc_reference_dfs(ps[b], [ls[d, b] for d in range(8)], 0.8, 0.9, 10)
        U, _ = O.link_cc_union(ps[b], [ls[d, b] for d in range(8)], 0.8, 0.9, 10)
        assert np.array_equal(d_lab[b], A) and np.array_equal(u_lab[b], U)
        n_diff += int(((A > 0) != (U > 0)).sum())
    assert n_diff > 0


def test_reference_dfs_mode_edge_cases(device):
    from tensorflow_ocr_amd.graph import Graph
    g = Graph(device)
    q = 24
    ps = np.zeros((4, q, q), np.float32)
    ls = np.zeros((8, 4, q, q), np.float32)
    ps[1] = 0.95; ls[:, 1] = 0.99                      # everything linked both ways: one group incl. the frame pixels
    ps[2] = 0.95                                       # no links at all: nothing survives a size filter
    # image 3: a one-way chain along a row — only "right" links (direction 3): from the leftmost key everything to its
    # right is reachable; from any other key only its suffix
    ps[3, 5, 2:20] = 0.9; ls[3, 3, 5, 2:20] = 0.95
    for ms in (0, 4, 30):
        lab, nc, _ = _decode(g, ps, ls, 0.8, 0.9, ms, "reference_dfs")
        for b in range(4):
            ref = O.link_cc_reference_dfs(ps[b], [ls[d, b] for d in range(8)], 0.8, 0.9, ms)
            assert np.array_equal(lab[b], ref), (b, ms)
            assert nc[b] == ref.max()
    with pytest.raises(ValueError):
        _decode(g, ps, ls, 0.8, 0.9, 0, "dfs")


def test_reference_dfs_mode_at_baseline_map_size(device):
    """configs[4]'s map size: 256 x 256 (1024^2 input at 1/4), weak asymmetric links, four images — bit-exact against
    the literal script restatement, and the device time of both modes printed (the directed refinement is one workgroup
    per image walking rounds of frontier expansions: an exactness mode, not the throughput path)."""
    import time
    from tensorflow_ocr_amd.graph import Graph
    g = Graph(device)
    ps, ls = _maps(11, 4, 256, 1.5)
    lab, nc, comps = _decode(g, ps, ls, 0.8, 0.9, 10, "reference_dfs")
    n_diff = 0
    for b in range(4):
        ref = O.link_cc_reference_dfs(ps[b], [ls[d, b] for d in range(8)], 0.8, 0.9, 10)
        assert np.array_equal(lab[b], ref), (b, int((lab[b] != ref).sum()))
        assert nc[b] == ref.max()
        U, _ = O.link_cc_union_fast(ps[b], [ls[d, b] for d in range(8)], 0.8, 0.9, 10)
        n_diff += int(((ref > 0) != (U > 0)).sum())
    times = {}
    for mode in ("union", "reference_dfs"):
        _decode(g, ps, ls, 0.8, 0.9, 10, mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            _decode(g, ps, ls, 0.8, 0.9, 10, mode)
        torch.cuda.synchronize()
        times[mode] = (time.perf_counter() - t0) / 5 * 1e3
    print("4 x 256^2, %d components, %d pixels differ between the two groupings | union %.3f ms, reference_dfs %.3f ms (incl. H2D of the maps)" % (
        int(nc.sum()), n_diff, times["union"], times["reference_dfs"]))
