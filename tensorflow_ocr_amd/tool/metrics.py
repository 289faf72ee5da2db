"""Mirror of the reference's tool/metrics.py (precision / recall / F bookkeeping over a test set):
`streaming_tp_fp_arrays` :31-64 (running totals in TF local variables -> a small accumulator here),
`precision_recall` :67-79, `fmean` :81-85.  Scalar host work."""
import numpy as np


def safe_divide(numerator, denominator):
    """tf_extended.math.safe_divide: 0 where the denominator is not positive."""
    return float(numerator) / float(denominator) if denominator > 0 else 0.0


class streaming_tp_fp_arrays:
    """tool/metrics.py:31-64: accumulates (num_gbboxes, tp, fp) over images; `.value` is the
    reference's `val` tuple, `.update(...)` its `update_op`."""

    def __init__(self):
        self.v_num_gbboxes = 0
        self.v_tp = np.zeros((0,), bool)
        self.v_fp = np.zeros((0,), bool)

    def update(self, num_gbboxes, tp, fp):
        self.v_num_gbboxes += int(np.sum(num_gbboxes))
        self.v_tp = np.concatenate([self.v_tp, np.asarray(tp, bool).reshape(-1)])
        self.v_fp = np.concatenate([self.v_fp, np.asarray(fp, bool).reshape(-1)])
        return self.value

    @property
    def value(self):
        return self.v_num_gbboxes, self.v_tp, self.v_fp


def precision_recall(num_gbboxes, tp, fp, scope=None):
    """tool/metrics.py:67-79 -> (precision, recall)."""
    tp = float(np.sum(np.asarray(tp, np.float32)))
    fp = float(np.sum(np.asarray(fp, np.float32)))
    recall = safe_divide(tp, float(num_gbboxes))
    precision = safe_divide(tp, tp + fp)
    return precision, recall


def fmean(pre, rec):
    """tool/metrics.py:81-85."""
    return 2 * pre * rec / (pre + rec)
