"""State-dict interop with TF-slim variable names (SURVEY.md §3.5-9, §5 checkpoint row).

Internally the heads that read one feature map are merged into a single parameter whose scope
joins the reference scopes with '+', e.g. `feature_fusion/Conv+Conv_5/weights` [cin, 2+16] holds
`feature_fusion/Conv/weights` [1,1,cin,2] and `feature_fusion/Conv_5/weights` [1,1,cin,16] side by
side.  These helpers split / join along the last axis so state dicts use the reference names.
"""
import numpy as np


def _split_scope(name):
    parts = name.split("/")
    for i, p in enumerate(parts):
        if "+" in p:
            return parts[:i], p.split("+"), parts[i + 1:]
    return None


def internal_to_tf(internal_sd, widths=None):
    """{internal name: array} -> {reference TF name: array}."""
    out = {}
    for name, arr in internal_sd.items():
        sp = _split_scope(name)
        if sp is None:
            out[name] = np.asarray(arr)
            continue
        pre, scopes, post = sp
        arr = np.asarray(arr)
        total = arr.shape[-1]
        ws = (widths or {}).get(name) or _guess_widths(total, len(scopes))
        o = 0
        for s, c in zip(scopes, ws):
            piece = arr[..., o:o + c]
            if post[-1] == "weights":
                piece = piece.reshape((1, 1) + piece.shape[-2:])
            out["/".join(pre + [s] + post)] = piece.copy()
            o += c
    return out


def _guess_widths(total, k):
    if k == 2 and total == 18:
        return (2, 16)
    if total % k:
        raise ValueError("cannot split %d channels over %d scopes" % (total, k))
    return (total // k,) * k


def tf_to_internal(internal_names, tf_sd):
    """Build {internal name: array} for `internal_names` out of a reference-named state dict."""
    out = {}
    for name in internal_names:
        sp = _split_scope(name)
        if sp is None:
            if name in tf_sd:
                out[name] = np.asarray(tf_sd[name])
            continue
        pre, scopes, post = sp
        pieces = []
        for s in scopes:
            key = "/".join(pre + [s] + post)
            if key not in tf_sd:
                pieces = None
                break
            a = np.asarray(tf_sd[key])
            if post[-1] == "weights":
                a = a.reshape(a.shape[-2:])
            pieces.append(a)
        if pieces is not None:
            out[name] = np.concatenate(pieces, axis=-1)
    return out
