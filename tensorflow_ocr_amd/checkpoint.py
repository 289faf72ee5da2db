"""State-dict interop with TF-slim variable names (SURVEY.md §3.5-9, §5 checkpoint row).

Internally the heads that read one feature map are merged into a single parameter whose scope
joins the reference scopes with '+', e.g. `feature_fusion/Conv+Conv_5/weights` [cin, 2+16] holds
`feature_fusion/Conv/weights` [1,1,cin,2] and `feature_fusion/Conv_5/weights` [1,1,cin,16] side by
side.  These helpers split / join along the last axis so state dicts use the reference names.
"""
import os

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
    if k == 2 and total == 9:              # EAST's F_score (1) + geo_map (8) heads on the merge branch's output
        return (1, 8)
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


# ----------------------------------------------------------------------------- TF checkpoint files
EMA_SUFFIX = "/ExponentialMovingAverage"


def save_tf_checkpoint(checkpoint_dir, global_step, tf_state_dict, ema_state_dict=None, basename="model.ckpt",
                       slots=None, scalars=None):
    """`saver.save(sess, checkpoint_path + 'model.ckpt', global_step=global_step)`
    (multigpu_train.py:188-189): writes `<dir>/model.ckpt-<step>.{index,data-00000-of-00001}` as a
    TensorFlow V2 bundle with the reference's variable names, the EMA shadows under
    `<name>/ExponentialMovingAverage` (what `variable_averages.variables_to_restore()` asks for,
    test.py:149-150), `global_step`, and updates the directory's `checkpoint` file.
    `slots` = {slot name: {variable name: array}} are the optimiser's slot variables (`<name>/Adam`,
    `<name>/Adam_1`, `<name>/Momentum`), `scalars` its non-slot ones (`beta1_power`, `beta2_power`):
    `Saver(tf.global_variables())` (multigpu_train.py:144) writes them all."""
    from . import tf_bundle
    tensors = {k: np.asarray(v) for k, v in tf_state_dict.items()}
    for k, v in (ema_state_dict or {}).items():
        tensors[k + EMA_SUFFIX] = np.asarray(v)
    for slot, sd in (slots or {}).items():
        for k, v in sd.items():
            tensors[k + "/" + slot] = np.asarray(v)
    for k, v in (scalars or {}).items():
        tensors[k] = np.asarray(v)
    tensors["global_step"] = np.asarray(int(global_step), np.int64)
    prefix = os.path.join(checkpoint_dir, "%s-%d" % (basename, int(global_step)))
    tf_bundle.write_bundle(prefix, tensors)
    tf_bundle.update_checkpoint_state(checkpoint_dir, prefix)
    return prefix


SLOT_SUFFIXES = ("/Adam", "/Adam_1", "/Momentum")


def _read_raw(path):
    from . import tf_bundle
    prefix = path
    if os.path.isdir(path):
        prefix = tf_bundle.get_checkpoint_state(path)
        if prefix is None:
            raise FileNotFoundError("no `checkpoint` state file in %s" % path)
    if prefix.endswith(".index"):
        prefix = prefix[:-len(".index")]
    if tf_bundle.is_v1_checkpoint(prefix):                # slim model-zoo files (resnet_v1_50.ckpt, train.sh:3)
        return tf_bundle.read_v1_checkpoint(prefix)
    return tf_bundle.read_bundle(prefix)


def save_training_state(checkpoint_dir, graph, opt, basename="model.ckpt"):
    """The whole `Saver(tf.global_variables())` set of one tower, numbered by the optimiser's
    global step: variables (incl. BN moving statistics), EMA shadows, optimiser slots and scalars."""
    ema = internal_to_tf(opt.shadow_state_dict()) if opt.ema is not None else None
    slots = {k: internal_to_tf(v) for k, v in opt.slot_state_dict().items()}
    return save_tf_checkpoint(checkpoint_dir, opt.global_step, internal_to_tf(graph.store.state_dict()), ema,
                              basename, slots=slots, scalars=opt.scalar_state_dict())


def restore_training_state(path, graph, opt, strict=False):
    """`saver.restore(sess, latest_checkpoint)` (multigpu_train.py:153-158) for a tower whose variables
    and optimiser exist (train.TrainStep.build): variables, EMA shadows, optimiser slots, global step.
    Returns the restored global step (None when the file has none, e.g. an ImageNet backbone)."""
    raw = _read_raw(path)
    step = int(raw.pop("global_step")) if "global_step" in raw else None
    names = graph.store.order
    plain, ema, slots = {}, {}, {sfx[1:]: {} for sfx in SLOT_SUFFIXES}
    for k, v in raw.items():
        if k.endswith(EMA_SUFFIX):
            ema[k[:-len(EMA_SUFFIX)]] = v
            continue
        for sfx in SLOT_SUFFIXES:
            if k.endswith(sfx):
                slots[sfx[1:]][k[:-len(sfx)]] = v
                break
        else:
            plain[k] = v
    internal = tf_to_internal(names, plain)
    graph.store.load_state_dict(internal, strict=strict)
    opt.restored_variables = sum(1 for k in internal if k in graph.store.vars)
    opt.load_state(global_step=step, slots={k: tf_to_internal(names, v) for k, v in slots.items() if v},
                   ema=tf_to_internal(names, ema) if ema else None)
    return step


def load_tf_checkpoint(path, use_moving_averages=False):
    """`tf.train.get_checkpoint_state(dir)` + `saver.restore` (test.py:146-150), or a checkpoint
    prefix directly (`slim.assign_from_checkpoint_fn(pretrained_model_path, ...)`,
    multigpu_train.py:149-151).  Returns ({reference variable name: array}, global_step or None);
    with use_moving_averages the EMA shadows replace the raw variables where they exist."""
    raw = _read_raw(path)
    step = int(raw.pop("global_step")) if "global_step" in raw else None
    sd = {k: v for k, v in raw.items() if not k.endswith(EMA_SUFFIX) and not k.endswith(SLOT_SUFFIXES)
          and k not in ("beta1_power", "beta2_power")}
    if use_moving_averages:
        for k, v in raw.items():
            if k.endswith(EMA_SUFFIX):
                sd[k[:-len(EMA_SUFFIX)]] = v
    return sd, step
