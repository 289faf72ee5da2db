#!/usr/bin/env python3
"""Round 6, VERDICT r5 item 1 (b): what a fused Winograd F(2x2, 3x3) convolution with 16-bit operands and f32 accumulation
does to the per-layer parity bar (tests/test_gpu_conv_abi.py: max|y - y_ref| <= 1e-3 * max|y_ref| on exact-f16 operands),
emulated on the host.  Emulation of the device form: input transform V = B^T d B by additions of f16 values, each
result rounded to f16 (v_pk_add_f16: 2 adds deep); weight transform U = G g G^T in f32, rounded to f16 once (the per-step
pack); 16 GEMMs over cin accumulated in f32 (MFMA); output transform A^T M A in f32; one rounding to f16.  The direct form
(what conv3x3_w4_kernel computes) = exact products of the f16 operands, f32 accumulation, one rounding to f16.
Reference: float64 on the same f16-exact operands.  Data = the parity tests' own distribution (x ~ N(0,1),
w ~ N(0, 2/(9 cin)))."""
import json
import sys

import numpy as np

BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float32)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)


def h(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def direct64(x, w):
    H, W, C = x.shape
    xp = np.zeros((H + 2, W + 2, C), np.float64)
    xp[1:-1, 1:-1] = x
    y = np.zeros((H, W, w.shape[3]), np.float64)
    for ky in range(3):
        for kx in range(3):
            y += xp[ky:ky + H, kx:kx + W].reshape(-1, C) .dot(w[ky, kx].astype(np.float64)).reshape(H, W, -1)
    return y


def direct_f32acc(x, w):
    H, W, C = x.shape
    xp = np.zeros((H + 2, W + 2, C), np.float32)
    xp[1:-1, 1:-1] = x
    y = np.zeros((H, W, w.shape[3]), np.float32)
    for ky in range(3):
        for kx in range(3):
            y += xp[ky:ky + H, kx:kx + W].reshape(-1, C).dot(w[ky, kx]).reshape(H, W, -1)
    return h(y)


def winograd(x, w, round_v=True, round_u=True, two_stage=True):
    H, W, C = x.shape
    K = w.shape[3]
    xp = np.zeros((H + 2, W + 2, C), np.float32)
    xp[1:-1, 1:-1] = x
    th, tw = H // 2, W // 2
    # d[i][j]: [th, tw, C]
    d = [[xp[i:i + H:2, j:j + W:2] for j in range(4)] for i in range(4)]
    # rows: t[xi][j] = sum_i BT[xi][i] d[i][j], rounded to f16 after the one add it takes
    t = [[None] * 4 for _ in range(4)]
    for xi in range(4):
        for j in range(4):
            s = sum(BT[xi, i] * d[i][j] for i in range(4) if BT[xi, i] != 0)
            t[xi][j] = h(s) if (round_v and two_stage) else s
    V = [[None] * 4 for _ in range(4)]
    for xi in range(4):
        for nu in range(4):
            s = sum(BT[nu, j] * t[xi][j] for j in range(4) if BT[nu, j] != 0)
            V[xi][nu] = h(s) if round_v else s
    U = np.einsum("ai,ijck,bj->abck", G, w, G).astype(np.float32)
    if round_u:
        U = h(U)
    M = [[V[a][b].reshape(-1, C).dot(U[a, b]).reshape(th, tw, K) for b in range(4)] for a in range(4)]
    y = np.zeros((H, W, K), np.float32)
    for p in range(2):
        for q in range(2):
            y[p::2, q::2] = sum(AT[p, a] * AT[q, b] * M[a][b] for a in range(4) for b in range(4) if AT[p, a] * AT[q, b] != 0)
    return h(y)


def main():
    rng = np.random.default_rng(6)
    rows = []
    for (hw, cin, cout) in ((32, 256, 64), (32, 512, 64), (64, 512, 32)):
        x = h(rng.standard_normal((hw, hw, cin)))
        w = h(rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin)))
        ref = direct64(x, w)
        mx = np.abs(ref).max()
        r = {"hw": hw, "cin": cin, "cout": cout}
        for name, y in (("direct_f16_out", direct_f32acc(x, w)),
                        ("winograd_f16_V_f16_U", winograd(x, w)),
                        ("winograd_f16_V_f32_U", winograd(x, w, round_u=False)),
                        ("winograd_f32_V_f16_U", winograd(x, w, round_v=False)),
                        ("winograd_f32_V_f32_U", winograd(x, w, round_v=False, round_u=False))):
            e = np.abs(y - ref)
            r[name] = {"max_over_max": float(e.max() / mx), "rms_over_max": float(np.sqrt((e ** 2).mean()) / mx)}
        rows.append(r)
        print(json.dumps(r), file=sys.stderr)
    # ReLU-activation-like input (non-negative, what the layers really see): the DC component makes V's +/+ sums larger
    x = h(np.maximum(rng.standard_normal((32, 32, 512)), 0))
    w = h(rng.standard_normal((3, 3, 512, 64)) * np.sqrt(2.0 / (9 * 512)))
    ref = direct64(x, w)
    mx = np.abs(ref).max()
    r = {"hw": 32, "cin": 512, "cout": 64, "input": "relu(N(0,1))"}
    for name, y in (("direct_f16_out", direct_f32acc(x, w)), ("winograd_f16_V_f16_U", winograd(x, w))):
        e = np.abs(y - ref)
        r[name] = {"max_over_max": float(e.max() / mx), "rms_over_max": float(np.sqrt((e ** 2).mean()) / mx)}
    rows.append(r)
    json.dump({"bar": 1e-3, "rows": rows}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
